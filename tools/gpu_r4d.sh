#!/bin/bash
# round 4, call 4: Gram path after the fixes (LDS size, deep streams, den slab, faster Gram kernels), f16x deep loop A/B,
# async fit, direct RCCL path at world 1
TAG=${1:-r4d}; OUT=gpurun_out/$TAG; mkdir -p $OUT
T=pytorch-nmf_amd/torchnmf_amd
KSEL="beta2_without or gram_panel or f16x or fit_g1 or g3 or g4 or fit_smoke or auto_ or sharded or comm_entries or cfg2 or cfg1_full_size_20 or rank_above_128"
timeout 1500 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "$KSEL" > $OUT/pytest.log 2>&1
echo "pytest rc=$?"; tail -8 $OUT/pytest.log
for i in 1 2; do
  timeout 300 python bench.py --beta 2 --gram --no-sweep --no-parity-mode --cpu-iters 0 --repeats 3 > $OUT/bench_gram_$i.json 2>> $OUT/bench.err
  python tools/bench_brief.py $OUT/bench_gram_$i.json short
  for v in "" _nd; do
    NMFMU_LIB=$PWD/$T/libnmfmu$v.so timeout 300 python bench.py --precision f16x --no-sweep --no-parity-mode --cpu-iters 0 --repeats 3 > $OUT/bench_f16x${v}_$i.json 2>> $OUT/bench.err
    echo "f16x lib=$v"; python tools/bench_brief.py $OUT/bench_f16x${v}_$i.json short
    NMFMU_LIB=$PWD/$T/libnmfmu$v.so timeout 300 python bench.py --precision f16x --beta 0.5 --no-sweep --no-parity-mode --cpu-iters 0 --repeats 3 > $OUT/bench_f16x_b05${v}_$i.json 2>> $OUT/bench.err
    echo "f16x beta=0.5 lib=$v"; python tools/bench_brief.py $OUT/bench_f16x_b05${v}_$i.json short
  done
done
timeout 300 python bench.py --beta 2 --gram --precision f16x --no-sweep --no-parity-mode --cpu-iters 0 --repeats 3 > $OUT/bench_gram_f16x.json 2>> $OUT/bench.err
python tools/bench_brief.py $OUT/bench_gram_f16x.json short
cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/trace_gram -o trace -- python $GRAFT_REPO_ROOT/bench.py --beta 2 --gram --no-sweep --no-parity-mode --cpu-iters 0 --repeats 3 --telemetry-s 0 > $GRAFT_REPO_ROOT/$OUT/bench_gram_traced.json 2>> $GRAFT_REPO_ROOT/$OUT/bench.err
cd $GRAFT_REPO_ROOT
f=$(find $OUT/trace_gram -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $OUT/gram_kernel_stats.csv && head -8 $OUT/gram_kernel_stats.csv | cut -c1-200
find $OUT -name "*.db" -delete; find $OUT -name "*kernel_trace.csv" -delete
timeout 900 python bench.py > $OUT/bench_default.json 2>> $OUT/bench.err
python - <<PY
import json
d=json.load(open("$OUT/bench_default.json"))
r=d['roofline']
print('default: it/s', d['iters_per_s'], 'frac', r['frac'], 'w/h', r['avg_launch_ms_w_step'], r['avg_launch_ms_h_step'], 'clock', r.get('clock_mhz'), 'power', r.get('power_w'))
print('fit', {k:v for k,v in d['fit'].items() if k!='note'})
print('real', {k:v for k,v in d['real_data_mode'].items() if k not in ('fit','hbm')})
print('real fit', {k:v for k,v in d['real_data_mode']['fit'].items() if k!='note'})
for b,e in d['beta_sweep']['betas'].items(): print('sweep', b, {k:v for k,v in e.items() if k not in ('gram_path','blocks_ms_per_step')})
g=d['beta_sweep']['betas']['2']['gram_path']; print('gram', {k:v for k,v in g.items() if k not in ('what',)})
n=d['nmfd']; print('nmfd', n['iters_per_s'], n['roofline'].get('clock_mhz'), n['roofline'].get('power_w'), n['roofline']['per_gemm'])
PY
tail -3 $OUT/bench.err
