"""Where does a data-parallel step spend its time when the compute part replays as a hipGraph?  (launch with torch.distributed.run;
two ranks may share one GPU over gloo for a rehearsal)  python -m torch.distributed.run --nproc-per-node 2 tools/dist_graph_probe.py [backend] [workload]"""
import os, sys, time
import torch
import torch.distributed as dist
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import WORKLOADS, HotPath, local_shard

backend = sys.argv[1] if len(sys.argv) > 1 else "gloo"
name = sys.argv[2] if len(sys.argv) > 2 else "c3"
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")) % torch.cuda.device_count())
torch.cuda.set_device(dev)
dist.init_process_group(backend, rank=rank, world_size=world)
import dp_gsat_amd as G
wl = dict(WORKLOADS[name], key=name)
b, _, _ = local_shard(name, wl["graphs"], rank, world, 0)
hot = HotPath(wl, b.to(dev), dev, seed=rank)
hot.attach_dp()
G.set_sync_free(True)
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(3):
        hot.compute()
torch.cuda.current_stream().wait_stream(side)
graph = torch.cuda.CUDAGraph()
with torch.cuda.graph(graph):
    hot.compute()
torch.cuda.synchronize()
dist.barrier()
for it in range(6):
    t0 = time.perf_counter(); graph.replay(); torch.cuda.synchronize(); t1 = time.perf_counter()
    hot.flat.all_reduce(average=True, async_op=True); t2 = time.perf_counter()
    hot.flat.wait(); t3 = time.perf_counter(); torch.cuda.synchronize(); t4 = time.perf_counter()
    if rank == 0:
        print(f"step {it}: replay+sync {1e3 * (t1 - t0):8.3f} ms | issue all_reduce {1e3 * (t2 - t1):8.3f} | wait {1e3 * (t3 - t2):8.3f} | sync {1e3 * (t4 - t3):8.3f}", flush=True)
torch.cuda.synchronize(); dist.barrier()
t0 = time.perf_counter()
for it in range(20):
    graph.replay()
    hot.reduce()
torch.cuda.synchronize(); dist.barrier()
if rank == 0:
    print(f"step loop without syncs: {1e3 * (time.perf_counter() - t0) / 20:.3f} ms/step", flush=True)
t0 = time.perf_counter()
for it in range(20):
    hot.step()
torch.cuda.synchronize(); dist.barrier()
if rank == 0:
    print(f"step eager: {1e3 * (time.perf_counter() - t0) / 20:.3f} ms/step", flush=True)
dist.barrier()
dist.destroy_process_group()
