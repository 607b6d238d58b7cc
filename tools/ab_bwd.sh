#!/bin/bash
# A/B of the fused extractor backward (scope A): tools/ab_bwd.sh [workloads...]
mkdir -p gpurun_out
for wl in "${@:-c3 c1 c2 c4}"; do
  for f in 1 0; do
    GSAT_ATTN_BWD_FUSED=$f python bench.py --workload $wl --no-cpu-baseline --no-full-step --no-roofline --no-exact-rerun 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$wl bwd_fused=$f ms/step', d['ms_per_step'], 'Medges/s', d['value'])"
  done
done
