#!/usr/bin/env python3
"""One tiled + one two-pass PNA backward at the C3 shape (for rocprofv3 --pmc runs)."""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import dp_gsat_amd as G
from dp_gsat_amd._lib import call, ptr, stream
wlname = sys.argv[1] if len(sys.argv) > 1 else "c3"
H = bench.WORKLOADS[wlname]["H"]
dev = torch.device("cuda:0")
data = bench.make_batch(wlname, bench.WORKLOADS[wlname]["graphs"], 0)[0].to(dev)
N, E = data.num_nodes, data.num_edges
x = torch.randn(N, H, device=dev); att = torch.rand(E, device=dev)
A = 4
a_arr, s_arr = (ctypes.c_int32 * A)(1, 2, 3, 5), (ctypes.c_int32 * 1)(0)
dout = torch.randn(N, A * 2 * H, device=dev)
dx_self, dmsg, datt, dx = torch.empty(N, H, device=dev), torch.empty(E, H, device=dev), torch.empty(E, device=dev), torch.empty(N, H, device=dev)
ix = G.BatchIndex(data.edge_index, N)
ix.graphs(data.batch, data.num_graphs)
tile_ptr, T, rows_nominal, rows_cap, edges_cap, spill = ix.pna_tiles(H)
for _ in range(3):
    call("gsat_pna_bwd_tiled", ptr(x), ptr(att), ptr(dout), ptr(ix.rowptr_dst), ptr(ix.src_by_dst), ptr(ix.eid_by_dst),
         ptr(tile_ptr), T, rows_nominal, rows_cap, edges_cap, ptr(ix.rowptr_src), ptr(ix.slot_dst_of_srcslot), N, E, H, a_arr, A, s_arr, 1,
         ptr(spill[1:]), ptr(spill[:1]), ptr(dx), ptr(dmsg), ptr(datt), None, stream())
    call("gsat_pna_bwd", ptr(x), ptr(att), None, ptr(dout), ptr(ix.rowptr_dst), ptr(ix.src_by_dst), ptr(ix.eid_by_dst), N, H,
         a_arr, A, s_arr, 1, 1.0, 1.0, ptr(dx_self), ptr(dmsg), ptr(datt), None, stream())
    call("gsat_aggr_sum_fwd", ptr(dmsg), ptr(dx_self), None, None, ptr(ix.rowptr_src), ptr(ix.slot_dst_of_srcslot), None, N, E, H, 1.0,
         ptr(dx), None, None, stream())
torch.cuda.synchronize()
