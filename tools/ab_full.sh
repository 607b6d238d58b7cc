#!/bin/bash
# A/B of one environment switch on the WHOLE training step (full_step of the bench line): tools/ab_full.sh VAR workloads...
var=$1; shift
mkdir -p gpurun_out
for wl in "$@"; do
  for f in 1 0; do
    env $var=$f python bench.py --workload $wl --no-cpu-baseline --no-roofline --no-exact-rerun 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$wl $var=$f scope A ms', d['ms_per_step'], ' full step ms', d['full_step']['ms_per_step'])"
  done
done
