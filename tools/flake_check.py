import sys, torch, traceback
sys.path.insert(0, "/root/repo")
from tests.test_gpu_attention import test_extractor_fwd_bwd
dev = torch.device("cuda:0")
bad = 0
for i in range(12):
    for H in (128, 80):
        try:
            test_extractor_fwd_bwd(dev, H, True, True)
        except AssertionError as e:
            bad += 1
            print("FAIL", i, H, str(e)[:200])
print("failures", bad)
