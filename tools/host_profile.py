"""cProfile of the eager training step on a small workload (where the step is host-bound): where does the Python time go?"""
import cProfile, os, pstats, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import WORKLOADS, FullStep, HotPath, local_shard
name = sys.argv[1] if len(sys.argv) > 1 else "c1"
wl = dict(WORKLOADS[name], key=name)
dev = torch.device("cuda:0")
b, x_dim, e_dim = local_shard(name, wl["graphs"], 0, 1, 0)
fs = FullStep(wl, b.to(dev), x_dim, e_dim, dev)
for _ in range(5):
    fs.step()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(50):
    fs.step()
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)
