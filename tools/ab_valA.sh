#!/bin/bash
# scope-A A/B over the VALUES of one environment variable: tools/ab_valA.sh VAR "v1 v2 ..." workloads...
var=$1; vals=$2; shift 2
for wl in "$@"; do
  for v in $vals; do
    env $var=$v python bench.py --workload $wl --no-cpu-baseline --no-full-step --no-roofline --no-exact-rerun 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$wl $var=$v scope A ms', d['ms_per_step'])"
  done
done
