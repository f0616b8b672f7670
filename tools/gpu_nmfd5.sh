#!/bin/bash
# experiment: channels 1025 vs 1024 (tile quantisation), 256x128 / 128x256 tiles for the reconstruction GEMMs at 1024
TAG=${1:-nmfd8}; OUT=gpurun_out/$TAG; mkdir -p $OUT
for cfg in "1025 none" "1024 none" "1024 tw"; do
  set -- $cfg
  TORCHNMF_AMD_NMFD_SHAPE=$2 timeout 300 python bench.py --workload nmfd --rows $1 --steps 40 --warmup 10 --cpu-iters 0 > $OUT/bench_$1_$2.json 2>> $OUT/bench.err
  python -c "
import json; d=json.load(open('$OUT/bench_$1_$2.json')); print('nmfd rows=$1 shape=$2 it/s=%.1f ms/step=%.4f' % (d['iters_per_s'], d['ms_per_step']))"
  TORCHNMF_AMD_NMFD_SHAPE=$2 BENCH_ARGS="--workload nmfd --rows $1" bash tools/gpu_prof.sh ${TAG}_prof_$1_$2 2>&1 | grep -E "nt_gemm" | cut -c1-160 | head -6
done
