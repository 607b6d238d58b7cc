"""-m gpu: the hand-written fp32 MFMA GEMM (all operand layouts, ragged sizes, split-K) vs torch.matmul in fp64."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _run(dev, a_t, b_t, M, N, K, bias=False, accumulate=False):
    from dp_gsat_amd._lib import call, load, ptr, stream
    g = torch.Generator().manual_seed(M * 7 + N * 3 + K)
    A = torch.randn((K, M) if a_t else (M, K), generator=g)
    B = torch.randn((N, K) if b_t else (K, N), generator=g)
    C0 = torch.randn(M, N, generator=g)
    bv = torch.randn(N, generator=g) if bias else None
    ref = (A.double().t() if a_t else A.double()) @ (B.double().t() if b_t else B.double())
    if bias:
        ref = ref + bv.double()
    if accumulate:
        ref = ref + C0.double()
    Ad, Bd, Cd = A.to(dev), B.to(dev), C0.to(dev).clone()
    bd = bv.to(dev) if bias else None
    wsf = int(load().gsat_gemm_workspace_floats(int(a_t), M, N, K))
    ws = torch.empty(max(wsf, 1), device=dev)
    call("gsat_gemm_f32", int(a_t), int(b_t), M, N, K, ptr(Ad), Ad.shape[1], ptr(Bd), Bd.shape[1], ptr(Cd), N, ptr(bd),
         int(accumulate), ptr(ws), wsf, stream())
    err = (Cd.cpu().double() - ref).abs().max().item()
    assert err <= 2e-6 * K ** 0.5 * max(1.0, ref.abs().max().item()), (err, a_t, b_t, M, N, K)


@pytest.mark.parametrize("a_t,b_t", [(False, True), (False, False), (True, False), (True, True)])
@pytest.mark.parametrize("M,N,K", [(128, 128, 32), (1000, 256, 128), (77, 20, 36), (4, 4, 4), (513, 132, 260), (3001, 64, 64)])
def test_gemm_layouts(dev, a_t, b_t, M, N, K):
    if a_t and M % 4:
        M += 4 - M % 4
    if not b_t and N % 4:
        N += 4 - N % 4
    _run(dev, a_t, b_t, M, N, K)


def test_gemm_bias_accumulate_and_splitk(dev):
    _run(dev, False, True, 300, 96, 64, bias=True)
    _run(dev, False, False, 300, 96, 64, accumulate=True)
    _run(dev, True, False, 128, 256, 50000)            # weight-gradient shape: split-K slabs + ordered reduce
    _run(dev, True, False, 512, 128, 20000, accumulate=True)


def test_gemm_deterministic(dev):
    from dp_gsat_amd._lib import call, load, ptr, stream
    A = torch.randn(30000, 128, device=dev); B = torch.randn(30000, 256, device=dev)
    wsf = int(load().gsat_gemm_workspace_floats(1, 128, 256, 30000))
    outs = []
    for _ in range(2):
        C = torch.empty(128, 256, device=dev); ws = torch.empty(wsf, device=dev)
        call("gsat_gemm_f32", 1, 0, 128, 256, 30000, ptr(A), 128, ptr(B), 256, ptr(C), 256, None, 0, ptr(ws), wsf, stream())
        outs.append(C)
    assert torch.equal(outs[0], outs[1])


@pytest.fixture
def bf16x3():
    import os
    os.environ["GSAT_GEMM_PRECISION"] = "bf16x3"
    yield
    os.environ.pop("GSAT_GEMM_PRECISION", None)


def _run_split(dev, a_t, b_t, M, N, K, bias=False, accumulate=False):
    """split-bf16 path: error budget is the parity band itself, 1e-4 of the output scale (measured ~1e-5)."""
    from dp_gsat_amd._lib import call, load, ptr, stream
    g = torch.Generator().manual_seed(M + 3 * N + 7 * K)
    A = torch.randn((K, M) if a_t else (M, K), generator=g)
    B = torch.randn((N, K) if b_t else (K, N), generator=g)
    C0 = torch.randn(M, N, generator=g)
    bv = torch.randn(N, generator=g) if bias else None
    ref = (A.double().t() if a_t else A.double()) @ (B.double().t() if b_t else B.double())
    if bias:
        ref = ref + bv.double()
    if accumulate:
        ref = ref + C0.double()
    Ad, Bd, Cd = A.to(dev), B.to(dev), C0.to(dev).clone()
    wsf = int(load().gsat_gemm_workspace_floats(int(a_t), M, N, K))
    ws = torch.empty(max(wsf, 1), device=dev)
    call("gsat_gemm_f32", int(a_t), int(b_t), M, N, K, ptr(Ad), Ad.shape[1], ptr(Bd), Bd.shape[1], ptr(Cd), N,
         ptr(bv.to(dev)) if bias else None, int(accumulate), ptr(ws), wsf, stream())
    err = (Cd.cpu().double() - ref).abs().max().item()
    scale = max(1.0, ref.abs().max().item())
    assert err <= 1e-4 * scale, (err, scale, a_t, b_t, M, N, K)
    return err / scale


@pytest.mark.parametrize("a_t,b_t", [(False, True), (False, False), (True, False), (True, True)])
@pytest.mark.parametrize("M,N,K", [(128, 128, 32), (1000, 256, 128), (76, 20, 36), (4, 4, 4), (516, 132, 260), (3004, 64, 1024)])
def test_gemm_bf16x3_layouts(dev, bf16x3, a_t, b_t, M, N, K):
    rel = _run_split(dev, a_t, b_t, M, N, K)
    assert rel < 5e-5


def test_gemm_bf16x3_bias_accumulate_splitk(dev, bf16x3):
    _run_split(dev, False, True, 300, 96, 64, bias=True)
    _run_split(dev, False, False, 300, 96, 64, accumulate=True)
    _run_split(dev, True, False, 128, 256, 50000)
    _run_split(dev, True, False, 512, 128, 20000, accumulate=True)


def test_linear_fn_at_backbone_size(dev):
    """PNA post_nn at C3 size ([51639, 1024] x [128, 1024]^T): forward / dx / dW of LinearFn (split-bf16 above 2 GFLOP)
    against an fp64 evaluation -- all within 1e-4 of the tensor's scale."""
    from dp_gsat_amd.ops import linear
    g = torch.Generator().manual_seed(0)
    M, K, N = 51639, 1024, 128
    x = torch.randn(M, K, generator=g)
    w = torch.randn(N, K, generator=g) * (1.0 / K ** 0.5)
    b = torch.randn(N, generator=g)
    go = torch.randn(M, N, generator=g)
    xd, wd, bd = (t.to(dev).requires_grad_(True) for t in (x, w, b))
    y = linear(xd, wd, bd)
    y.backward(go.to(dev))
    x64, w64, b64, g64 = (t.double().to(dev) for t in (x, w, b, go))
    y64 = x64 @ w64.t() + b64
    for name, got, ref in (("y", y, y64), ("dx", xd.grad, g64 @ w64), ("dw", wd.grad, g64.t() @ x64), ("db", bd.grad, g64.sum(0))):
        err = (got.double() - ref).abs().max().item()
        scale = max(1.0, ref.abs().max().item())
        assert err <= 1e-4 * scale, (name, err, scale)


@pytest.mark.parametrize("b_t", [True, False])
@pytest.mark.parametrize("M,N,K", [
    (51639, 256, 128), (51639, 128, 256),      # C3 node-mode extractor: P = emb W1^T / da1 ; h2 = a1 W2^T / demb  (KS = 1 and 2)
    (12801, 256, 64), (12801, 64, 256),        # C2 edge mode (H = 64): 8 column blocks x KR 32 ; 2 column blocks x 4 k-splits
    (9001, 128, 64), (9001, 64, 128),          # H = 64 node mode
    (20000, 512, 128), (20000, 128, 512),      # C4 edge mode (H = 128): two 256-column chunks ; KR 128
    (8193, 32, 512),                           # one column block, 8 k-splits
])
def test_gemm_weight_stationary(dev, b_t, M, N, K):
    """The persistent weight-stationary kernel (tall-skinny forward / backward-data products, M >= 8192): every wave geometry,
    ragged last tile, bias, bitwise determinism."""
    _run(dev, False, b_t, M, N, K)
    _run(dev, False, b_t, M, N, K, bias=True)
    from dp_gsat_amd._lib import call, ptr, stream
    A = torch.randn(M, K, device=dev); B = torch.randn((N, K) if b_t else (K, N), device=dev)
    outs = []
    for _ in range(2):
        C = torch.empty(M, N, device=dev)
        call("gsat_gemm_f32", 0, int(b_t), M, N, K, ptr(A), K, ptr(B), B.shape[1], ptr(C), N, None, 0, None, 0, stream())
        outs.append(C)
    assert torch.equal(outs[0], outs[1])


@pytest.mark.parametrize("b_t", [True, False])
@pytest.mark.parametrize("M,N,K", [
    (51639, 256, 128), (51639, 128, 256),      # C3 extractor backward: da1 = dh2 W2 (8 column blocks) ; demb = dh1 W1 (4 column blocks x 2 k-splits)
    (12801, 256, 64), (12801, 64, 256),        # H = 64 edge mode
    (9001, 128, 64), (9001, 64, 128),
    (20000, 256, 512), (8193, 32, 256),        # 8 fragment steps of 16 k at K = 512 ; one column block, 8 k-splits
    (20001, 1024, 128), (9000, 512, 64),       # wide outputs: 256-column chunks (the backbone's dx = dy W at C3: 51 639 x 1024 x 128)
])
def test_gemm_weight_stationary_split_bf16(dev, bf16x3, b_t, M, N, K):
    """The persistent weight-stationary kernel on the split-bf16 path (backward-data products of the extractor, M >= 8192): every wave
    geometry, ragged last tile, accumulate, bitwise determinism; error inside the split-bf16 band (~1e-5 of the output scale)."""
    rel = _run_split(dev, False, b_t, M, N, K)
    assert rel < 5e-5
    _run_split(dev, False, b_t, M, N, K, accumulate=True)
    from dp_gsat_amd._lib import call, ptr, stream
    A = torch.randn(M, K, device=dev); B = torch.randn((N, K) if b_t else (K, N), device=dev)
    outs = []
    for _ in range(2):
        C = torch.empty(M, N, device=dev)
        call("gsat_gemm_bf16x3", 0, int(b_t), M, N, K, ptr(A), K, ptr(B), B.shape[1], ptr(C), N, None, 0, None, 0, stream())
        outs.append(C)
    assert torch.equal(outs[0], outs[1])
