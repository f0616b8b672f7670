#!/bin/bash
# round 4, call 2: swapped G2 operand roles in the ping-pong kernel (parity + timeline + interleaved bench vs the un-swapped
# build), the f16x mode (tests + bench), fit / real-data legs of the default bench
TAG=${1:-r4b}; OUT=gpurun_out/$TAG; mkdir -p $OUT
T=pytorch-nmf_amd/torchnmf_amd
KSEL="half_steps or rank128 or shapes_and_ksplit or cfg1 or cfg2 or f16 or large_slice or fit_g1 or fit_g2 or g3 or g4 or auto_ or betamu_default or pack_x or sharded or rank_above_128"
timeout 1500 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "$KSEL" > $OUT/pytest.log 2>&1
echo "pytest rc=$?"; tail -15 $OUT/pytest.log
for i in 1 2; do
  for v in _dbg _nsdbg; do
    NMFMU_LIB=$PWD/$T/libnmfmu$v.so timeout 300 python tools/pp_timeline.py f16 > $OUT/timeline${v}_$i.txt 2>&1
    grep "cycles/tile" $OUT/timeline${v}_$i.txt
  done
done
NMFMU_LIB=$PWD/$T/libnmfmu_dbg.so timeout 300 python tools/pp_timeline.py bf16 > $OUT/timeline_dbg_bf16.txt 2>&1; grep "cycles/tile" $OUT/timeline_dbg_bf16.txt
BENCH_ARGS="--no-sweep --no-parity-mode --repeats 3" bash tools/gpu_variants.sh $TAG "" _ns
timeout 600 python bench.py --precision f16x --no-sweep --no-parity-mode --cpu-iters 0 > $OUT/bench_f16x.json 2>> $OUT/bench.err
python tools/bench_brief.py $OUT/bench_f16x.json 2>/dev/null | head -5
timeout 900 python bench.py > $OUT/bench_default.json 2>> $OUT/bench.err
python - <<PY
import json
d=json.load(open("$OUT/bench_default.json"))
r=d['roofline']
print('default: it/s', d['iters_per_s'], 'frac', r['frac'], 'w/h', r['avg_launch_ms_w_step'], r['avg_launch_ms_h_step'], 'clock', r.get('clock_mhz'), 'power', r.get('power_w'))
print('parity', d['parity'])
print('fit', d['fit'])
print('real', d['real_data_mode'])
print('sweep', {k:(v['iters_per_s'], v['kernel_frac'], v.get('parity')) for k,v in d['beta_sweep']['betas'].items()})
print('nmfd', d['nmfd']['iters_per_s'])
PY
tail -5 $OUT/bench.err
