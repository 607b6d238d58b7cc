"""Host (CPU) time of the eager scope-A step: time 200 steps WITHOUT waiting for the GPU between them, on a workload scaled down so the
GPU is never the limit (same launch sequence, ~same Python path): python tools/host_time_c3.py [workload] [graphs]"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import WORKLOADS, HotPath, local_shard
name = sys.argv[1] if len(sys.argv) > 1 else "c3"
graphs = int(sys.argv[2]) if len(sys.argv) > 2 else 64
wl = dict(WORKLOADS[name], key=name)
dev = torch.device("cuda:0")
b, x_dim, e_dim = local_shard(name, graphs, 0, 1, 0)
hot = HotPath(wl, b.to(dev), dev)
for _ in range(20):
    hot.step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(200):
    hot.step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"{name} with {graphs} graphs: host {1e3 * (t1 - t0) / 200:.3f} ms/step to issue, {1e3 * (t2 - t0) / 200:.3f} ms/step until the GPU is done")
