#!/bin/bash
# One GPU-box session: smoke, GPU parity tests, a short bench, and a rocprofv3 kernel trace.
# Usage (from the repo root on the GPU box):  bash tools/gpu_round.sh [tag]
TAG=${1:-run}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
echo "== smoke" | tee $OUT/summary.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $OUT/summary.txt
tail -3 $OUT/smoke.log | tee -a $OUT/summary.txt
echo "== pytest -m gpu" | tee -a $OUT/summary.txt
timeout 1500 python -m pytest tests -q -m gpu --tb=short -p no:cacheprovider ${PYTEST_ARGS} > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/summary.txt
tail -40 $OUT/pytest.log | tee -a $OUT/summary.txt
echo "== bench" | tee -a $OUT/summary.txt
timeout 600 python bench.py --steps 30 --warmup 5 ${BENCH_ARGS} > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?" | tee -a $OUT/summary.txt
cat $OUT/bench.json | tee -a $OUT/summary.txt; tail -5 $OUT/bench.err | tee -a $OUT/summary.txt
