#!/bin/bash
# round-end evidence in one gpurun call: kernel traces (+ PMC passes for the headline and NMFD) of the bench legs, the
# un-profiled default bench line.  Usage: bash tools/gpu_evidence.sh <tag>; copy what is judged into profiles/.
TAG=${1:-ev}; OUT=gpurun_out/$TAG; mkdir -p $OUT
bash tools/gpu_prof.sh ${TAG}/cfg1_f16 pmc > $OUT/cfg1_f16_prof.log 2>&1
BENCH_ARGS="--precision bf16" bash tools/gpu_prof.sh ${TAG}/cfg1_bf16 > $OUT/cfg1_bf16_prof.log 2>&1
for b in 2 0.5 0; do BENCH_ARGS="--beta $b" bash tools/gpu_prof.sh ${TAG}/beta$b > $OUT/beta${b}_prof.log 2>&1; done
BENCH_ARGS="--workload nmfd" bash tools/gpu_prof.sh ${TAG}/nmfd pmc > $OUT/nmfd_prof.log 2>&1
timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
python tools/bench_brief.py $OUT/bench.json
for d in cfg1_f16 cfg1_bf16 beta2 beta0.5 beta0 nmfd; do echo "== $d"; head -6 $OUT/$d/kernel_stats.csv 2>/dev/null | cut -c1-200; done
