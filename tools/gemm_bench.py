"""Micro-benchmark of gsat_gemm_f32 at the extractor's shapes (HIP events around back-to-back launches)."""
import sys, torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dp_gsat_amd._lib import call, load, ptr, stream
dev = torch.device("cuda:0")
shapes = [  # (a_t, b_t, M, N, K, label)
    (0, 1, 51639, 256, 128, "C3 P=emb W1^T"), (0, 1, 51639, 128, 256, "C3 h2=a1 W2^T"), (0, 0, 51639, 256, 128, "C3 da1=dh2 W2"),
    (0, 0, 51639, 128, 256, "C3 demb=dh1 W1"), (1, 0, 128, 256, 51639, "C3 dW2 (split-K)"), (1, 0, 256, 128, 51639, "C3 dW1 (split-K)"),
    (0, 1, 51639, 128, 1024, "C3 post_nn fwd"), (0, 0, 51639, 1024, 128, "C3 post_nn dx"), (1, 0, 128, 1024, 51639, "C3 post_nn dW"),
    (0, 1, 377532, 256, 1024, "c5s h2"), (0, 0, 377532, 1024, 256, "c5s da1"), (1, 0, 256, 1024, 377532, "c5s dW2"),
]
for a_t, b_t, M, N, K, label in shapes:
    A = torch.randn((K, M) if a_t else (M, K), device=dev)
    B = torch.randn((N, K) if b_t else (K, N), device=dev)
    C = torch.empty(M, N, device=dev)
    wsf = int(load().gsat_gemm_workspace_floats(a_t, M, N, K))
    ws = torch.empty(max(wsf, 1), device=dev)
    f = lambda: call("gsat_gemm_f32", a_t, b_t, M, N, K, ptr(A), A.shape[1], ptr(B), B.shape[1], ptr(C), N, None, 0, ptr(ws), wsf, stream())
    for _ in range(3): f()
    torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda._sleep(2_000_000)
        s.record()
        for _ in range(10): f()
        e.record(); torch.cuda.synchronize()
        ts.append(s.elapsed_time(e) / 10 * 1e3)
    t = sorted(ts)[1]
    print(f"{label:22s} {M:7d}x{N:5d}x{K:7d}  {t:9.1f} us  {2.0 * M * N * K / t / 1e6:7.1f} TFLOP/s")
