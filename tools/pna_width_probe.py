"""k_pna_fwd / k_pna_bwd_dst bandwidth as a function of the hidden width (C3 topology): how much do the idle lanes of a
non-power-of-two width (the reference's PNA configs use hidden_size 80) cost?"""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import dp_gsat_amd as G
from dp_gsat_amd._lib import call, ptr, stream
dev = torch.device("cuda:0")
wl = dict(bench.WORKLOADS["c3"], key="c3")
b, _, _ = bench.local_shard("c3", wl["graphs"], 0, 1, 0)
data = b.to(dev)
N, E = data.num_nodes, data.num_edges
ix = G.get_index(data.edge_index, N)
A = 4
a_arr, s_arr = (ctypes.c_int32 * A)(1, 2, 3, 5), (ctypes.c_int32 * 1)(0)


def timeit(f):
    for _ in range(10): f()
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda._sleep(4_000_000)
        s.record()
        for _ in range(30): f()
        e.record(); torch.cuda.synchronize()
        ts.append(s.elapsed_time(e) / 30 * 1e3)
    return sorted(ts)[2]


for H in (32, 64, 80, 96, 128, 160, 256):
    K = A * 2 * H
    x, att = torch.randn(N, H, device=dev), torch.rand(E, device=dev)
    y, dout = torch.empty(N, K, device=dev), torch.randn(N, K, device=dev)
    dxs, dmsg, datt = torch.empty(N, H, device=dev), torch.empty(E, H, device=dev), torch.empty(E, device=dev)
    f = lambda: call("gsat_pna_fwd", ptr(x), ptr(att), None, ptr(ix.rowptr_dst), ptr(ix.src_by_dst), ptr(ix.eid_by_dst), N, H,
                     a_arr, A, s_arr, 1, 1.0, 1.0, ptr(y), stream())
    g = lambda: call("gsat_pna_bwd", ptr(x), ptr(att), None, ptr(dout), ptr(ix.rowptr_dst), ptr(ix.src_by_dst), ptr(ix.eid_by_dst), N, H,
                     a_arr, A, s_arr, 1, 1.0, 1.0, ptr(dxs), ptr(dmsg), ptr(datt), None, stream())
    ix.graphs(data.batch, data.num_graphs)
    tiles = ix.pna_tiles(H)
    dx = torch.empty(N, H, device=dev)
    if tiles:
        td, T, rn, rc, ec, spill = tiles
        h = lambda: call("gsat_pna_bwd_tiled", ptr(x), ptr(att), ptr(dout), ptr(ix.rowptr_dst), ptr(ix.src_by_dst), ptr(ix.eid_by_dst), ptr(td), T, rn, rc, ec,
                         ptr(ix.rowptr_src), ptr(ix.slot_dst_of_srcslot), N, E, H, a_arr, A, s_arr, 1, ptr(spill[1:]), ptr(spill[:1]), ptr(dx), ptr(dmsg), ptr(datt), None, stream())
    fb = 4 * N * H + 8 * A * N * H + 8 * E + 4 * N
    bb = 4 * A * 2 * N * H + 8 * N * H + 4 * E * H + 16 * E + 4 * N
    tf, tb = timeit(f), timeit(g)
    cb = 4 * A * 2 * N * H + 8 * N * H + 16 * E + 4 * N          # compulsory bytes of the whole backward
    tt = timeit(h) if tiles else float("nan")
    print(f"H={H:4d}  fwd {tf:7.2f} us {fb / tf / 1e3:7.0f} GB/s   bwd_dst (1st of 2 passes) {tb:7.2f} us {bb / tb / 1e3:7.0f} GB/s   tiled bwd (whole) {tt:7.2f} us {cb / tt / 1e3:7.0f} GB/s")
