"""-m gpu: bench.py end to end on one GPU, as the driver runs it (every leg but the CPU baseline), on the default workload and one GIN
workload: the JSON line must carry the contract's fields, the roofline object of the kernels the step really runs and a whole-step figure."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("workload", ["c3", "c2"])
def test_bench_line_single_gpu(dev, workload):
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--workload", workload, "--steps", "5", "--warmup", "2", "--no-cpu-baseline"]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=root)
    assert res.returncode == 0, res.stderr[-3000:]
    line = json.loads(res.stdout.strip().splitlines()[-1])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
                "config", "roofline", "cpu_baseline", "gemm_precision", "value_fp32_exact", "full_step"):
        assert key in line, key
    assert line["n_gpus"] == 1 and line["steps"] == 5 and line["value"] > 0 and line["vs_baseline"] is None and line["cpu_baseline"] is None
    roof = line["roofline"]
    assert roof["bound"] == "hbm" and 0 < roof["frac"] <= 1.2 and roof["timing"] == "in_step" and roof["backward"]["us_all_launches"] > 0
    assert line["full_step"]["ms_per_step"] > line["ms_per_step"]
    if workload == "c3":
        assert "one-launch" in line["gemm_precision"] and "x6" in line["gemm_precision"]          # the forward the step really took
    else:
        assert "staged" in line["gemm_precision"]
