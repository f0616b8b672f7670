#!/bin/bash
# NMFD (configs[3]): parity tests of the NMFD / SIPLCA / PLCA paths, two bench lines, kernel trace.  Usage: bash tools/gpu_nmfd.sh <tag>
TAG=${1:-nmfd}; OUT=gpurun_out/$TAG; mkdir -p $OUT
timeout 900 python -m pytest tests -q -m gpu --tb=short -p no:cacheprovider -k "nmfd or siplca or rank_above_256 or plca" -x > $OUT/pytest_nmfd.log 2>&1; echo "pytest rc=$?"; tail -15 $OUT/pytest_nmfd.log
for mode in 128 128; do
  TORCHNMF_AMD_NMFD_TILE=$mode timeout 300 python bench.py --workload nmfd --steps 40 --warmup 10 --cpu-iters 0 > $OUT/bench_$mode.json 2>> $OUT/bench.err
  python -c "
import json; d=json.load(open('$OUT/bench_$mode.json')); print('nmfd tile=$mode it/s=%.1f ms/step=%.4f' % (d['iters_per_s'], d['ms_per_step']))"
done
BENCH_ARGS="--workload nmfd" bash tools/gpu_prof.sh ${TAG}_prof 2>&1 | grep -E "nmfmu" | cut -c1-200 | head -10
