#!/bin/bash
# round 6: the default bench line on the final build (host-side changes after the evidence call), with its wall time
OUT=gpurun_out/r6s; mkdir -p $OUT
T0=$(date +%s); timeout 1200 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$? wall $(( $(date +%s) - T0 )) s"; python tools/bench_brief.py $OUT/bench.json 2>&1 | head -8
python - <<'PY'
import json
d=json.load(open('gpurun_out/r6s/bench.json'))
print('fit', {k:d['fit'][k] for k in ('wall_ms','setup_ms','iters_per_s_whole_call','iters_per_s_loop','loop_over_engine_step')})
r=d['real_data_mode']; print('real', r['auto_picks'], r['iters_per_s'], r['parity']['rel_W'], {k:r['fit'][k] for k in ('wall_ms','setup_ms','iters_per_s_whole_call','loop_over_engine_step')})
print('traffic', d['roofline'].get('traffic'), d['roofline'].get('traffic_source','')[:80])
PY
