#!/bin/bash
# one gpurun call: the whole GPU test-suite, smoke, the default bench line.  Usage: bash tools/gpu_round.sh <tag>
TAG=${1:-round}; OUT=gpurun_out/$TAG; mkdir -p $OUT
timeout 2400 python -m pytest tests -q -m gpu -x > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -5 $OUT/pytest_gpu.log
grep -E "relW=|^cfg1|^f16 " $OUT/pytest_gpu.log | head -30
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; grep -v amdgpu.ids $OUT/smoke.log | tail -5
timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; cat $OUT/bench.json
