#!/bin/bash
# HBM traffic of the aggregation roofline leg from two separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE):
#   tools/pmc_pass.sh <workload> <kernel substring> [more substrings]    -> gpurun_out/pmc_<workload>.txt
wl=$1; shift
root=$(pwd)
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
: > $root/gpurun_out/pmc_$wl.txt
for ctr in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_${wl}_$ctr
  rocprofv3 --pmc $ctr -d /tmp/pmc_${wl}_$ctr -o r --output-format csv -- python3 $root/bench.py --workload $wl --roofline-only > $root/gpurun_out/pmc_${wl}_$ctr.log 2>&1
  csv=$(find /tmp/pmc_${wl}_$ctr -name "*counter_collection.csv" | head -1)
  python3 $root/tools/pmc_traffic.py "$csv" $ctr "$@" >> $root/gpurun_out/pmc_$wl.txt
  echo "pass $ctr done" 
done
cat $root/gpurun_out/pmc_$wl.txt
