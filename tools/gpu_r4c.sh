#!/bin/bash
# round 4, call 3: beta == 2 without reconstruction (Gram path): tests + bench; async fit loop; f16x tests again
TAG=${1:-r4c}; OUT=gpurun_out/$TAG; mkdir -p $OUT
KSEL="beta2_without or gram_panel or f16x or fit_g1 or fit_g2 or g3 or g4 or fit_smoke or fit_f16 or auto_ or nmfd_fit_g5 or sparse_fit_g9 or plca_fit_g10 or nmf2d or cfg2"
timeout 1500 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "$KSEL" > $OUT/pytest.log 2>&1
echo "pytest rc=$?"; tail -12 $OUT/pytest.log
for i in 1 2; do
  timeout 300 python bench.py --beta 2 --gram --no-sweep --no-parity-mode --cpu-iters 0 --repeats 3 > $OUT/bench_gram_$i.json 2>> $OUT/bench.err
  python tools/bench_brief.py $OUT/bench_gram_$i.json short
  timeout 300 python bench.py --beta 2 --no-sweep --no-parity-mode --cpu-iters 0 --repeats 3 > $OUT/bench_b2_$i.json 2>> $OUT/bench.err
  python tools/bench_brief.py $OUT/bench_b2_$i.json short
done
timeout 300 python bench.py --beta 2 --gram --precision f16x --no-sweep --no-parity-mode --cpu-iters 0 --repeats 3 > $OUT/bench_gram_f16x.json 2>> $OUT/bench.err
python tools/bench_brief.py $OUT/bench_gram_f16x.json short
cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$OUT/trace_gram -- python $GRAFT_REPO_ROOT/bench.py --beta 2 --gram --no-sweep --no-parity-mode --cpu-iters 0 --repeats 3 --telemetry-s 0 > $GRAFT_REPO_ROOT/$OUT/bench_gram_traced.json 2>> $GRAFT_REPO_ROOT/$OUT/bench.err
cd $GRAFT_REPO_ROOT
f=$(find $OUT/trace_gram -name '*kernel_trace.csv' | head -1); python tools/trace_summary.py $f 2>/dev/null | head -14; cp $(find $OUT/trace_gram -name '*kernel_stats.csv' | head -1) $OUT/gram_kernel_stats.csv 2>/dev/null
timeout 900 python bench.py > $OUT/bench_default.json 2>> $OUT/bench.err
python - <<PY
import json
d=json.load(open("$OUT/bench_default.json"))
r=d['roofline']
print('default: it/s', d['iters_per_s'], 'frac', r['frac'], 'w/h', r['avg_launch_ms_w_step'], r['avg_launch_ms_h_step'], 'clock', r.get('clock_mhz'), 'power', r.get('power_w'))
print('fit', {k:v for k,v in d['fit'].items() if k!='note'})
print('real fit', {k:v for k,v in d['real_data_mode']['fit'].items() if k!='note'})
print('sweep2', d['beta_sweep']['betas']['2'])
print('nmfd', d['nmfd']['iters_per_s'])
PY
tail -5 $OUT/bench.err
