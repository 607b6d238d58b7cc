"""Does k_pna_fwd's time depend on where its 211 MB output starts?  (C3 shape; output placed at different offsets of one big buffer)"""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import dp_gsat_amd as G
from dp_gsat_amd._lib import call, ptr, stream
dev = torch.device("cuda:0")
wl = dict(bench.WORKLOADS["c3"], key="c3")
b, _, _ = bench.local_shard("c3", wl["graphs"], 0, 1, 0)
data = b.to(dev)
N, E, H = data.num_nodes, data.num_edges, 128
ix = G.get_index(data.edge_index, N)
A = 4
a_arr, s_arr = (ctypes.c_int32 * A)(1, 2, 3, 5), (ctypes.c_int32 * 1)(0)
K = A * 2 * H


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


for trial in range(3):
    x = torch.randn(N, H, device=dev)
    att = torch.rand(E, device=dev)
    big = torch.empty(N * K + (64 << 20) // 4, device=dev)
    print(f"trial {trial}: x @ {x.data_ptr():#x}  big @ {big.data_ptr():#x}")
    for off_bytes in [0, 512, 4096, 65536, 1 << 20, (2 << 20) + 4096, 16 << 20, 48 << 20]:
        y = big[off_bytes // 4: off_bytes // 4 + N * K].view(N, K)
        f = lambda: call("gsat_pna_fwd", ptr(x), ptr(att), None, ptr(ix.rowptr_dst), ptr(ix.src_by_dst), ptr(ix.eid_by_dst), N, H,
                         a_arr, A, s_arr, 1, 1.0, 1.0, ptr(y), stream())
        print(f"   out offset {off_bytes:>10d} B: {timeit(f):6.2f} us")
    del big, x, att
    junk = [torch.empty(37 << 20, device=dev) for _ in range(trial + 1)]      # perturb the allocator for the next trial
