#!/bin/bash
# A/B of one environment switch on the scope-A bench: tools/ab_env.sh VAR workloads...
var=$1; shift
mkdir -p gpurun_out
for wl in "$@"; do
  for f in 1 0; do
    env $var=$f python bench.py --workload $wl --no-cpu-baseline --no-full-step --no-roofline --no-exact-rerun 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$wl $var=$f ms/step', d['ms_per_step'], 'Medges/s', d['value'])"
  done
done
