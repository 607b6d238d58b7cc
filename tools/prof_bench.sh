#!/bin/bash
# rocprofv3 kernel-trace summary of a bench run: tools/prof_bench.sh <tag> <bench args...>   -> gpurun_out/prof_<tag>.txt
tag=$1; shift
root=$(pwd)
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_$tag
rocprofv3 --kernel-trace --stats -d /tmp/prof_$tag -o r -- python3 $root/bench.py "$@" > $root/gpurun_out/prof_$tag.log 2>&1
db=$(find /tmp/prof_$tag -name "*.db" | head -1)
cd $root
python3 tools/rocpd_summary.py "$db" gpurun_out/prof_$tag.txt "bench.py $*" 60
