"""A few launches of gsat_gemm_f32 at one shape (for rocprofv3 --pmc): python tools/gemm_once.py a_t b_t M N K"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dp_gsat_amd._lib import call, load, ptr, stream
a_t, b_t, M, N, K = (int(v) for v in sys.argv[1:6])
dev = torch.device("cuda:0")
A = torch.randn((K, M) if a_t else (M, K), device=dev); B = torch.randn((N, K) if b_t else (K, N), device=dev)
C = torch.empty(M, N, device=dev)
wsf = int(load().gsat_gemm_workspace_floats(a_t, M, N, K)); ws = torch.empty(max(wsf, 1), device=dev)
for _ in range(5):
    call("gsat_gemm_f32", a_t, b_t, M, N, K, ptr(A), A.shape[1], ptr(B), B.shape[1], ptr(C), N, None, 0, ptr(ws), wsf, stream())
torch.cuda.synchronize()
