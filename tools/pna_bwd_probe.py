#!/usr/bin/env python3
"""Times the PNA backward at the C3 shape: two-pass (k_pna_bwd_dst + per-source sum) vs the tiled one-launch kernel, for several
LDS budgets, with graph-aligned or fixed windows.  HIP events around back-to-back launches on torch's stream."""
import ctypes, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import dp_gsat_amd as G
from dp_gsat_amd._lib import call, ptr, stream

wlname = sys.argv[1] if len(sys.argv) > 1 else "c3"
H = int(sys.argv[2]) if len(sys.argv) > 2 else bench.WORKLOADS[wlname]["H"]
dev = torch.device("cuda:0")
data = bench.make_batch(wlname, bench.WORKLOADS[wlname]["graphs"], 0)[0].to(dev)
N, E = data.num_nodes, data.num_edges
x = torch.randn(N, H, device=dev); att = torch.rand(E, device=dev)
A = 4
a_arr, s_arr = (ctypes.c_int32 * A)(1, 2, 3, 5), (ctypes.c_int32 * 1)(0)
dout = torch.randn(N, A * 2 * H, device=dev)
dx_self, dmsg, datt, dx = torch.empty(N, H, device=dev), torch.empty(E, H, device=dev), torch.empty(E, device=dev), torch.empty(N, H, device=dev)
spilled = torch.empty(E, dtype=torch.uint8, device=dev)
compulsory = 4 * A * 2 * N * H + 4 * N * H + 4 * N * H + 16 * E + 4 * N


def timeit(fn, reps=20, rounds=5):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(rounds):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda._sleep(2_000_000)
        s.record()
        for _ in range(reps):
            fn()
        e.record()
        torch.cuda.synchronize()
        ts.append(s.elapsed_time(e) * 1e3 / reps)
    return float(np.median(ts))


ix = G.BatchIndex(data.edge_index, N)
def two_pass():
    call("gsat_pna_bwd", ptr(x), ptr(att), None, ptr(dout), ptr(ix.rowptr_dst), ptr(ix.src_by_dst), ptr(ix.eid_by_dst), N, H,
         a_arr, A, s_arr, 1, 1.0, 1.0, ptr(dx_self), ptr(dmsg), ptr(datt), None, stream())
    call("gsat_aggr_sum_fwd", ptr(dmsg), ptr(dx_self), None, None, ptr(ix.rowptr_src), ptr(ix.slot_dst_of_srcslot), None, N, E, H, 1.0,
         ptr(dx), None, None, stream())
t = timeit(two_pass)
print(f"{wlname} N={N} E={E} H={H} compulsory={compulsory/1e6:.1f} MB")
print(f"two-pass           {t:7.2f} us  {compulsory / t / 1e6:7.1f} GB/s  frac {compulsory / t / 1e6 / 8000:.3f}")
ref_dx, ref_da = dx.clone(), datt.clone()
for aligned, budget, dbgv in [(True, b, 0) for b in (32768, 40960, 49152, 65536, 81920, 98304)] + [(False, 65536, 0)]:
    if True:
        os.environ["GSAT_PNA_TILE_LDS"] = str(budget)
        ix2 = G.BatchIndex(data.edge_index, N)
        if aligned:
            ix2.graphs(data.batch, data.num_graphs)
        tile_ptr, T, rows_nominal, rows_cap, edges_cap, spill = ix2.pna_tiles(H)
        def tiled():
            call("gsat_pna_bwd_tiled", ptr(x), ptr(att), ptr(dout), ptr(ix2.rowptr_dst), ptr(ix2.src_by_dst), ptr(ix2.eid_by_dst),
                 ptr(tile_ptr), T, rows_nominal, rows_cap, edges_cap, ptr(ix2.rowptr_src), ptr(ix2.slot_dst_of_srcslot), N, E, H, a_arr, A, s_arr, 1,
                 ptr(spill[1:]), ptr(spill[:1]), ptr(dx), ptr(dmsg), ptr(datt), None, stream())
        t = timeit(tiled)
        err = (dx - ref_dx).abs().max().item()
        print(f"dbg={dbgv:2d} tiled {'aligned' if aligned else 'fixed  '} lds={budget:6d} rows<={rows_cap:3d} edges<={edges_cap:3d} tiles={T:5d} "
              f"{t:7.2f} us  {compulsory / t / 1e6:7.1f} GB/s  frac {compulsory / t / 1e6 / 8000:.3f}  max|dx-two_pass|={err:.2e}")
