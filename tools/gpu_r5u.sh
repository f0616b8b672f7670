#!/bin/bash
# round 5: apply kernel with one update item per thread (512-thread blocks): full GPU suite on the new build, the default
# bench line, then old / new interleaved on the headline, the beta = 2 path and BetaMu (libnmfmu_oldapply.so = the build before).
TAG=${1:-r5u}; OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
LIBD=$PWD/pytorch-nmf_amd/torchnmf_amd
timeout 900 python -m pytest tests -q -m gpu > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; grep -h "passed\|failed" $OUT/pytest_gpu.log | tail -2 | cut -c1-250
timeout 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; python tools/bench_brief.py $OUT/bench.json 2>&1 | head -3
for i in 1 2; do for v in _oldapply ""; do
  [ -f $LIBD/libnmfmu$v.so ] || continue
  NMFMU_LIB=$LIBD/libnmfmu$v.so timeout 200 python bench.py --cpu-iters 0 --no-sweep --no-parity-mode --repeats 3 > $OUT/b${v}_$i.json 2>> $OUT/err.log
  echo -n "[lib$v] "; python tools/bench_brief.py $OUT/b${v}_$i.json 1
done; done
for v in _oldapply ""; do
  [ -f $LIBD/libnmfmu$v.so ] || continue
  NMFMU_LIB=$LIBD/libnmfmu$v.so timeout 200 python bench.py --beta 2 --gram --cpu-iters 0 --no-sweep --no-parity-mode --repeats 3 > $OUT/g${v}.json 2>> $OUT/err.log
  echo -n "[lib$v beta2 gram] "; python tools/bench_brief.py $OUT/g${v}.json 1
  NMFMU_LIB=$LIBD/libnmfmu$v.so timeout 200 python bench.py --workload betamu --cpu-iters 0 --no-sweep --repeats 3 > $OUT/m${v}.json 2>> $OUT/err.log
  echo -n "[lib$v betamu] "; python tools/bench_brief.py $OUT/m${v}.json 1
done
