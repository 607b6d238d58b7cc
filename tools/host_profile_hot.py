"""cProfile of the eager scope-A step on a small workload (host-bound): where does the Python time go?"""
import cProfile, os, pstats, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import WORKLOADS, HotPath, local_shard
name = sys.argv[1] if len(sys.argv) > 1 else "c1"
wl = dict(WORKLOADS[name], key=name)
dev = torch.device("cuda:0")
b, x_dim, e_dim = local_shard(name, wl["graphs"], 0, 1, 0)
hot = HotPath(wl, b.to(dev), dev)
for _ in range(5):
    hot.step()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(100):
    hot.step()
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(32)
