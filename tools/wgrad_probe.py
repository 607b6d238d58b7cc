import sys, os, torch
sys.path.insert(0, "/root/repo")
from dp_gsat_amd._lib import call, load, ptr, stream
dev = torch.device("cuda:0")
for (M,N,K) in ((128,256,51639),(256,128,51639)):
    A = torch.randn(K, M, device=dev); B = torch.randn(K, N, device=dev); C = torch.empty(M, N, device=dev)
    wsf = int(load().gsat_gemm_workspace_floats(1, M, N, K)); ws = torch.empty(max(wsf,1), device=dev)
    f = lambda: call("gsat_gemm_bf16x3", 1, 0, M, N, K, ptr(A), M, ptr(B), N, ptr(C), N, None, 0, ptr(ws), wsf, stream())
    for _ in range(3): f()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda._sleep(2_000_000); s.record()
    for _ in range(20): f()
    e.record(); torch.cuda.synchronize()
    print(os.environ.get("GSAT_GEMM_SPLITK_BLOCKS"), os.environ.get("GSAT_GEMM_SPLITK_SLABS"), M, N, K, "splits", wsf // (M*N), f"{s.elapsed_time(e)/20*1e3:.1f} us")
