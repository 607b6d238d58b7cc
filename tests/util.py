"""Comparison helpers for the parity tests."""
import torch

TOL = 1e-4   # north_star: within 1e-4 on attention / embedding tensors (relative to the tensor's scale)


def close(actual, ref, tol=TOL, ref64=None, what=""):
    """|actual - ref| <= tol * max(1, |ref|_inf)  (+ slack for the fp32 oracle's own rounding error when an
    fp64 evaluation of the oracle is supplied: an ill-conditioned case -- e.g. the 1/sigma of PNA's std
    backward next to a tiny variance -- must not be asked to match the fp32 oracle closer than the fp32
    oracle matches exact arithmetic)."""
    a = actual.detach().cpu().double()
    r = ref.detach().cpu().double()
    assert a.shape == r.shape, f"{what}: shape {tuple(a.shape)} != {tuple(r.shape)}"
    if a.numel() == 0:
        return
    scale = max(1.0, r.abs().max().item())
    allowed = tol * scale
    target = r
    if ref64 is not None:
        r64 = ref64.detach().cpu().double()
        allowed += 4.0 * (r - r64).abs().max().item()
        target = r64
    err = (a - target).abs().max().item()
    assert err <= allowed, f"{what}: max abs err {err:.3e} > allowed {allowed:.3e} (scale {scale:.3e})"


def capture_with_dump(fn):
    """Capture one call of ``fn`` into a torch.cuda.CUDAGraph with debug mode on; returns (graph, dot text or None).  The dot text is
    hipGraphDebugDotPrint's dump of the captured graph (None when this ROCm / torch build cannot produce it)."""
    import os
    import tempfile
    graph = torch.cuda.CUDAGraph()
    try:
        graph.enable_debug_mode()
    except Exception:
        pass
    with torch.cuda.graph(graph):
        fn()
    text = None
    try:
        path = os.path.join(tempfile.mkdtemp(), "graph.dot")
        graph.debug_dump(path)
        if os.path.exists(path) and os.path.getsize(path) > 0:
            text = open(path, errors="replace").read()
    except Exception:
        text = None
    return graph, text


def assert_no_memset_nodes(dot_text, what=""):
    """Round 2 found hipMemsetAsync nodes of a captured graph running out of order with the neighbouring replay's kernels (ROCm 7.2): the
    library zeroes with kernels only, and no library call in a captured step may bring a memset node back."""
    if dot_text is None:
        return False
    low = dot_text.lower()
    assert "memset" not in low, f"{what}: the captured graph contains a memset node"
    return True
