"""Profile driver: the whole GSAT training step on one workload (run under rocprofv3)."""
import sys, torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import WORKLOADS, FullStep, local_shard, timed
name = sys.argv[1] if len(sys.argv) > 1 else "c3"
wl = dict(WORKLOADS[name], key=name)
dev = torch.device("cuda:0")
b, x_dim, e_dim = local_shard(name, wl["graphs"], 0, 1, 0)
fs = FullStep(wl, b.to(dev), x_dim, e_dim, dev)
dt = timed(fs.step, 10, 3, dev, False)
print("full step ms", dt / 10 * 1e3)
