#!/usr/bin/env python3
"""Phase breakdown of the fused extractor backward (diagnostic build):
    make -C dp_gsat_amd/csrc EXTRA=-DGSAT_FUSED_STAMPS -B attn_fused_bwd.o && make -C dp_gsat_amd/csrc && python tools/fused_bwd_stamps.py [c3|c2|c4]"""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main(wl="c3"):
    os.environ.setdefault("GSAT_ATTN_BWD_FUSED", "1")
    import bench
    import dp_gsat_amd as G
    from dp_gsat_amd import _lib
    dev = torch.device("cuda:0")
    cfg = bench.WORKLOADS[wl]
    data, _, _ = bench.make_batch(wl, cfg["graphs"], 0)
    data = data.to(dev)
    H, edge = cfg["H"], cfg["edge_att"]
    ext = G.ExtractorMLP(H, edge).to(dev).train()
    emb = torch.randn(data.num_nodes, H, device=dev, requires_grad=True)
    for it in range(4):
        emb.grad = None
        z, a = ext.attend(emb, data.edge_index, data.batch, noise="philox")
        torch.autograd.backward([z, a], [torch.ones_like(z), torch.ones_like(a)])
    lib = _lib.load()
    if os.environ.get("GSAT_ATTN_BWD_FUSED") != "1":
        lib.gsat_debug_dual_stamps.restype = ctypes.POINTER(ctypes.c_ulonglong * 16)
        d = list(lib.gsat_debug_dual_stamps().contents)
        for mode in (0, 1):
            v = d[mode * 8:mode * 8 + 8]
            n = max(v[7], 1)
            print(f"dual MODE {mode + 1}: workgroup-runs {v[7]}")
            for nme, x in zip(["0 prologue fill", "1 chunk: prefetch issue + MFMAs + OUT stores", "2 chunk: barrier + LDS stores + barrier", "3 -", "4 partials", "5 -"], v[:6]):
                print(f"   {nme:26s} {x / n:10.0f} cycles per workgroup-run")
        return
    lib.gsat_debug_bwd_counters.restype = ctypes.POINTER(ctypes.c_int * 64)
    c = lib.gsat_debug_bwd_counters().contents
    vals = list(c)
    st = [vals[16 + 2 * i] | (vals[17 + 2 * i] << 32) for i in range(16)]
    print(f"{wl}: tiles {vals[0]} big {vals[2]} workgroups {st[14]}")
    names = ["0 meta + dz + keep bits L2", "1 L2 column pass", "2 da1 MFMA + keep bits L1", "3 L1 column pass", "4 dW2 MFMA", "5 partials"]
    tot = sum(st[:6])
    for n, v in zip(names, st[:6]):
        print(f"  {n:30s} {v / max(st[14], 1):10.0f} cycles/wg (~{v / max(st[14], 1) / 2100:7.2f} us)  {100 * v / max(tot, 1):5.1f} %")


if __name__ == "__main__":
    main(*(sys.argv[1:2] or ["c3"]))
