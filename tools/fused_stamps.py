#!/usr/bin/env python3
"""Where a tile of the fused extractor forward spends its cycles (diagnostic build only):
    make -C dp_gsat_amd/csrc EXTRA=-DGSAT_FUSED_STAMPS && python tools/fused_stamps.py [c3|c2|c4|c1]
Thread 0 of every workgroup stamps s_memtime after each barrier-delimited phase; sums over workgroups are printed per phase."""
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main(wl="c3", training=1, save_a1=1):
    training, save_a1 = int(training), int(save_a1)
    import bench
    from dp_gsat_amd import _lib
    from dp_gsat_amd._lib import call, ptr, stream
    from dp_gsat_amd.graph_index import BatchIndex
    from dp_gsat_amd.ops import _attn_args
    dev = torch.device("cuda:0")
    cfg = bench.WORKLOADS[wl]
    data, _, _ = bench.make_batch(wl, cfg["graphs"], 0)
    data = data.to(dev)
    H, edge = cfg["H"], cfg["edge_att"]
    N = data.num_nodes
    index = BatchIndex(data.edge_index, N)
    seg = index.graphs(data.batch, data.num_graphs)
    C0, C1, C2 = (2 * H, 4 * H, H) if edge else (H, 2 * H, H)
    g = torch.Generator().manual_seed(0)
    mk = lambda *s: (torch.randn(*s, generator=g) / (s[-1] ** 0.5)).to(dev)
    params = (mk(C1, C0), mk(C1), mk(C2, C1), mk(C2), mk(1, C2), mk(1))
    emb = torch.randn(N, H, generator=g).to(dev)
    M = index.E if edge else N
    f32 = torch.float32
    bufs = (torch.empty(N, C1, device=dev), torch.empty(N, C1, device=dev) if edge else None, torch.empty(M, C1, device=dev) if save_a1 else None,
            torch.empty(M, C2, device=dev), torch.empty(seg.G * (2 * C1 + 2 * C2), device=dev), torch.empty(M, 1, device=dev), torch.empty(M, 1, device=dev))
    args = _attn_args(emb, params, index, seg, edge, training, 0.5, 1234, None, None, None, bufs, None, True)
    n = int(_lib.load().gsat_attn_fwd_workspace_bytes(ctypes.byref(args)))
    ws = torch.zeros(max(n, 256), dtype=torch.uint8, device=dev)
    args.fwd_workspace, args.fwd_workspace_bytes = ptr(ws), n
    for _ in range(3):
        call("gsat_attn_fwd", ctypes.byref(args), stream())
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        call("gsat_attn_fwd", ctypes.byref(args), stream())
    e1.record()
    torch.cuda.synchronize()
    print(f"{wl} training={training} save_a1={save_a1}: fwd call {e0.elapsed_time(e1) / 20 * 1e3:.1f} us (prep + fused), M={M} G={seg.G}")
    cnt = ws[:16].view(torch.int32).cpu().tolist()
    st = ws[64:64 + 120].view(torch.int64).cpu().tolist()
    print("tiles", cnt[0], "big", cnt[2], "workgroups", st[14])
    names = ["0 fetch+meta+loadX", "1 gemm1", "2 edge gather", "3 stats1", "4 apply1", "5 gemm2", "6 H2->lds+h2 store", "7 stats2", "8 head", "9 exit", "10 save_pq (in 3)", "11 stats loop (in 3)", "12", "13"]
    tot = sum(st[:14])
    for nme, v in zip(names, st[:14]):
        print(f"  {nme:22s} {v / max(st[14], 1):10.0f} cycles/wg (~{v / max(st[14], 1) / 2100:6.2f} us at 2.1 GHz)  {100 * v / max(tot, 1):5.1f} %")     # s_memtime counts shader cycles


if __name__ == "__main__":
    main(*(sys.argv[1:] or ["c3"]))
