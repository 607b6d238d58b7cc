#!/usr/bin/env python3
"""torch ops of the scope-A hot-path step that launch their own kernels (torch.profiler, eager): python tools/hotpath_ops.py [workload]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main(wl="c3"):
    import bench
    import dp_gsat_amd as G
    dev = torch.device("cuda:0")
    cfg = dict(bench.WORKLOADS[wl], key=wl)
    data, x_dim, e_dim = bench.make_batch(wl, cfg["graphs"], 0)
    data = data.to(dev)
    G.set_sync_free(True)
    hot = bench.HotPath(cfg, data, dev, seed=0)
    for _ in range(3):
        hot.step()
    torch.cuda.synchronize()
    from torch.profiler import ProfilerActivity, profile
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
        hot.step()
        torch.cuda.synchronize()
    rows = []
    for e in prof.key_averages(group_by_input_shape=True):
        t = getattr(e, "self_device_time_total", None)
        if t is None:
            t = getattr(e, "self_cuda_time_total", 0.0)
        if t > 0:
            rows.append((t, e.count, e.key, str(e.input_shapes)[:90]))
    for t, n, k, shp in sorted(rows, reverse=True)[:40]:
        print(f"{n:4d} x {k:38s} {t:9.1f} us  {shp}")


if __name__ == "__main__":
    main(*sys.argv[1:])
