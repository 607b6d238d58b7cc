"""PNA row kernels on a batch with hub rows: one lane group per row (gsat_pna_fwd / _bwd) vs the chunked long-row path (gsat_pna_*_long)."""
import ctypes, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dp_gsat_amd._lib import call, ptr, stream
from dp_gsat_amd.graph_index import BatchIndex

dev = torch.device("cuda:0")
rng = np.random.default_rng(0)
N, H, hubs, hub_deg, extra = 200_000, 128, 4, 50_000, 400_000
src = np.concatenate([rng.integers(0, N, hubs * hub_deg), rng.integers(0, N, extra)])
dst = np.concatenate([np.repeat(np.arange(hubs), hub_deg), rng.integers(0, N, extra)])
ei = torch.from_numpy(np.stack([src, dst])).to(dev)
E = ei.shape[1]
ix = BatchIndex(ei, N)
assert ix.long_rows[0] is not None
x, att = torch.randn(N, H, device=dev), torch.rand(E, device=dev)
aggr, scal = (ctypes.c_int32 * 4)(1, 2, 3, 5), (ctypes.c_int32 * 1)(0)
out = torch.empty(N, 8 * H, device=dev)
dout = torch.randn(N, 8 * H, device=dev)
dxs, dmsg, datt = torch.empty(N, H, device=dev), torch.empty(E, H, device=dev), torch.empty(E, device=dev)
part = ix.pna_partial(H, False)
common = (ptr(x), ptr(att), None)
idx = (ptr(ix.rowptr_dst), ptr(ix.src_by_dst), ptr(ix.eid_by_dst))
cfg = (aggr, 4, scal, 1, 1.0, 1.0)
runs = {
    "fwd, one lane group per row": lambda: call("gsat_pna_fwd", *common, *idx, N, H, *cfg, ptr(out), stream()),
    "fwd, hub rows chunked": lambda: call("gsat_pna_fwd_long", *common, *idx, N, E, H, *cfg, ptr(out), ptr(ix.long_rows[0]), ptr(part), stream()),
    "bwd (dst pass), one lane group per row": lambda: call("gsat_pna_bwd", *common, ptr(dout), *idx, N, H, *cfg, ptr(dxs), ptr(dmsg), ptr(datt), None, stream()),
    "bwd (dst pass), hub rows chunked": lambda: call("gsat_pna_bwd_long", *common, ptr(dout), *idx, N, E, H, *cfg, ptr(dxs), ptr(dmsg), ptr(datt), None,
                                                      ptr(ix.long_rows[0]), ptr(part), stream()),
}
print(f"N={N} E={E} H={H}: {hubs} rows of {hub_deg} in-edges")
for name, f in runs.items():
    for _ in range(2):
        f()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(5):
        f()
    e.record()
    torch.cuda.synchronize()
    print(f"{name:42s} {s.elapsed_time(e) / 5 * 1e3:10.1f} us")
