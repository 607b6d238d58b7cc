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
