#!/bin/bash
# last call of a round: full GPU suite, smoke, default bench, NMFD bench + kernel trace + PMC
TAG=${1:-final}; OUT=gpurun_out/$TAG; mkdir -p $OUT
bash tools/gpu_round.sh $TAG
timeout 300 python bench.py --workload nmfd --cpu-iters 1 > $OUT/bench_nmfd.json 2>> $OUT/bench.err; echo "nmfd rc=$?"
python -c "
import json; d=json.load(open('$OUT/bench_nmfd.json')); print('nmfd', d['iters_per_s'], d['ms_per_step'], d['roofline']['frac'], d['parity']['modes'], d['parity_mode']['iters_per_s'])"
BENCH_ARGS="--workload nmfd" bash tools/gpu_prof.sh $TAG/nmfd pmc > $OUT/nmfd_prof.log 2>&1
head -12 $OUT/nmfd/kernel_stats.csv | cut -c1-60,150-230
