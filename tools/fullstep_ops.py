#!/usr/bin/env python3
"""Which torch ops of the whole C3 training step still launch their own kernels (torch.profiler, one eager step):
    python tools/fullstep_ops.py [workload]"""
import os
import sys
from collections import Counter

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main(wl="c3"):
    import bench
    dev = torch.device("cuda:0")
    cfg = dict(bench.WORKLOADS[wl], key=wl)
    data, x_dim, e_dim = bench.make_batch(wl, cfg["graphs"], 0)
    data = data.to(dev)
    fs = bench.FullStep(cfg, data, x_dim, e_dim, dev)
    for _ in range(3):
        fs.step()
    torch.cuda.synchronize()
    from torch.profiler import ProfilerActivity, profile
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
        fs.step()
        torch.cuda.synchronize()
    ka = prof.key_averages(group_by_input_shape=True)
    rows = []
    for e in ka:
        t = getattr(e, "self_device_time_total", None)
        if t is None:
            t = getattr(e, "self_cuda_time_total", 0.0)
        if t > 0 and (e.key.startswith("aten::") or e.key.startswith("Optimizer") or "Backward" in e.key):
            rows.append((t, e.count, e.key, str(e.input_shapes)[:80]))
    for t, n, k, shp in sorted(rows, reverse=True)[:45]:
        print(f"{n:4d} x {k:34s} {t:9.1f} us  {shp}")
    # where the small in-place adds come from (python frames of the op)
    from collections import Counter
    where = Counter()
    for ev in prof.events():
        if ev.name in ("aten::add_", "aten::add", "aten::copy_", "aten::fill_", "aten::zero_") and ev.stack:
            frames = [f for f in ev.stack if "dp_gsat_amd" in f or "bench.py" in f or "optim" in f or "autograd" in f]
            where[(ev.name, str(ev.input_shapes)[:40], frames[0] if frames else ev.stack[0])] += 1
    print("---- small ops by first relevant frame")
    for (name, shp, fr), n in where.most_common(30):
        print(f"{n:4d} x {name:12s} {shp:42s} {fr[-110:]}")


if __name__ == "__main__":
    main(*sys.argv[1:])
