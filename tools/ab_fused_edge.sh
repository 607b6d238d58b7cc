#!/bin/bash
# edge-mode fused forward: P | Q through global memory (1) vs through the LDS image (0) vs staged: tools/ab_fused_edge.sh [workloads...]
for wl in "${@:-c4 c2}"; do
  for v in "1 1" "1 0" "0 1"; do
    set -- $v
    GSAT_ATTN_FUSED=$1 GSAT_ATTN_FUSED_PQG=$2 python bench.py --workload $wl --no-cpu-baseline --no-full-step --no-roofline --no-exact-rerun 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$wl fused=$1 pqg=$2 ms/step', d['ms_per_step'])"
  done
done
