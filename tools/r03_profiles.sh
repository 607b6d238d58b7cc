#!/bin/bash
# the rocprofv3 summaries committed under profiles/ for round 3 (run on the GPU box; copies land in gpurun_out/)
set -e
S="--steps 20 --warmup 5 --no-cpu-baseline --no-full-step --no-roofline --no-exact-rerun"
tools/prof_bench.sh r03_c3_scopeA --workload c3 $S
GSAT_ATTN_FUSED=0 tools/prof_bench.sh r03_c3_scopeA_staged_fwd --workload c3 $S
GSAT_NODE_ATT_LIFT=1 tools/prof_bench.sh r03_c3_scopeA_lifted_att --workload c3 $S
GSAT_ATTN_BWD_FUSED=1 tools/prof_bench.sh r03_c3_scopeA_fused_bwd --workload c3 $S
GSAT_DUAL_GEMM=1 tools/prof_bench.sh r03_c3_scopeA_dual_gemm --workload c3 $S
tools/prof_bench.sh r03_c3_fullstep --workload c3 --steps 6 --warmup 3 --no-cpu-baseline --no-roofline --no-exact-rerun
GSAT_PNA_COMPACT=1 tools/prof_bench.sh r03_c3_fullstep_compact --workload c3 --steps 6 --warmup 3 --no-cpu-baseline --no-roofline --no-exact-rerun
tools/prof_bench.sh r03_c2_fullstep --workload c2 --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-exact-rerun
tools/prof_bench.sh r03_c4_fullstep --workload c4 --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-exact-rerun
tools/bench_matrix.sh > gpurun_out/r03_bench_matrix.txt 2>gpurun_out/r03_bench_matrix.err
cat gpurun_out/r03_bench_matrix.txt
