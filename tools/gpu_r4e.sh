#!/bin/bash
# round 4, call 5: NMFD tall tiles (256 x 128, four waves of 128 x 64) A/B + tests; Gram path after ring depth 4 / no colsum
TAG=${1:-r4e}; OUT=gpurun_out/$TAG; mkdir -p $OUT
KSEL="nmfd or beta2_without or gram_panel or cfg2_full_size_beta2"
timeout 1500 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "$KSEL" > $OUT/pytest.log 2>&1
echo "pytest rc=$?"; tail -8 $OUT/pytest.log
for i in 1 2; do
  for tall in 1 0; do
    TORCHNMF_AMD_NMFD_TALL=$tall timeout 300 python bench.py --workload nmfd --cpu-iters 0 --repeats 3 > $OUT/nmfd_tall${tall}_$i.json 2>> $OUT/bench.err
    python - <<PY
import json
d=json.load(open("$OUT/nmfd_tall${tall}_$i.json")); r=d['roofline']
print("tall=$tall it/s %.1f ms %.4f clock %s power %s" % (d['iters_per_s'], d['ms_per_step'], r.get('clock_mhz'), r.get('power_w')), {k:v['avg_launch_ms'] for k,v in r['per_gemm'].items()})
PY
  done
  timeout 300 python bench.py --beta 2 --gram --no-sweep --no-parity-mode --cpu-iters 0 --repeats 3 > $OUT/bench_gram_$i.json 2>> $OUT/bench.err
  python tools/bench_brief.py $OUT/bench_gram_$i.json short
done
timeout 300 python bench.py --beta 2 --gram --precision f16x --no-sweep --no-parity-mode --cpu-iters 0 --repeats 3 > $OUT/bench_gram_f16x.json 2>> $OUT/bench.err
python tools/bench_brief.py $OUT/bench_gram_f16x.json short
tail -3 $OUT/bench.err
