#!/bin/bash
# one gpurun call of round 3: GPU test-suite (all failures, not -x), smoke, default bench, then an interleaved A/B of
# library variants built with `make VARIANT=_x EXTRA=...`.  Usage: bash tools/gpu_r3.sh <tag> [variant suffixes ...]
TAG=${1:-r3}; shift
OUT=gpurun_out/$TAG; mkdir -p $OUT
rm -f gpurun_out/parity_measured.jsonl
if [ -z "$SKIP_TESTS" ]; then
  timeout ${TEST_TIMEOUT:-1500} python -m pytest tests -q -m gpu --timeout 900 ${PYTEST_ARGS} > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"
  tail -4 $OUT/pytest_gpu.log; grep -E "^(FAILED|ERROR)" $OUT/pytest_gpu.log | head -40
  cp gpurun_out/parity_measured.jsonl $OUT/ 2>/dev/null
  timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; grep -v amdgpu.ids $OUT/smoke.log | tail -8
fi
if [ -z "$SKIP_BENCH" ]; then
  timeout 900 python bench.py ${BENCH_ARGS} > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; tail -3 $OUT/bench.err
  python tools/bench_brief.py $OUT/bench.json
fi
for i in 1 2; do
  for v in "$@"; do
    [ "$v" = "base" ] && v=""
    lib=$PWD/pytorch-nmf_amd/torchnmf_amd/libnmfmu$v.so
    NMFMU_LIB=$lib timeout 300 python bench.py --steps 50 --warmup 10 --cpu-iters 0 --no-sweep ${AB_ARGS} > $OUT/v_${v}_$i.json 2>> $OUT/v.err
    echo -n "[${v:-base} #$i] "; python tools/bench_brief.py $OUT/v_${v}_$i.json short
  done
done
