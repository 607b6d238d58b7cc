#!/bin/bash
# Runs the bench on every single-GPU workload, eager and captured (hipGraph), and prints one condensed line per run.
cd "$(dirname "$0")/.."
for wl in c3 c1 c2 c4; do
  for mode in "--eager" "--graph"; do
    python bench.py --workload $wl $mode --no-cpu-baseline 2>gpurun_out/matrix_$wl$mode.err | tail -1 | python -c "
import sys, json
r = json.loads(sys.stdin.read())
f = r.get('full_step') or {}
print('$wl', '${mode}', 'scopeA_ms', r['ms_per_step'], 'Medges/s', r['value'], 'full_ms', f.get('ms_per_step'), 'roof', r['roofline']['kernel'], r['roofline']['us_per_launch'], r['roofline']['frac'], 'bwd', r['roofline']['backward']['us_all_launches'], r['roofline']['backward']['frac'], 'isolated', r['roofline'].get('us_isolated'), r['roofline']['backward'].get('us_isolated'), 'exact_fp32_ms', r.get('ms_per_step_fp32_exact'))
"
  done
done
