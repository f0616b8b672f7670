#!/bin/bash
# NMFD: how much does the ragged 1025th channel cost?  bench at 1025 and 1024 rows, kernel traces of both
TAG=${1:-nmfdrows}; OUT=gpurun_out/$TAG; mkdir -p $OUT
for rows in 1025 1024; do
  timeout 300 python bench.py --workload nmfd --rows $rows --steps 40 --warmup 10 --cpu-iters 0 > $OUT/bench_$rows.json 2>> $OUT/bench.err
  python -c "
import json; d=json.load(open('$OUT/bench_$rows.json')); print('nmfd rows=$rows it/s=%.1f ms/step=%.4f' % (d['iters_per_s'], d['ms_per_step']))"
  BENCH_ARGS="--workload nmfd --rows $rows" bash tools/gpu_prof.sh ${TAG}_prof$rows 2>&1 | grep -E "nmfmu" | cut -c1-170 | head -10
done
