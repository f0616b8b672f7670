#!/bin/bash
# round 4: single panel image + transposing reads (FusedCfg::TR) in every single-plane four-wave MU instance: full GPU suite
# with the shipped library, then A/B against the two-image build (make VARIANT=_notr EXTRA=-DNMFMU_FUSED_TR_MINR=999)
TAG=${1:-r4q}; OUT=gpurun_out/$TAG; mkdir -p $OUT
T=$PWD/pytorch-nmf_amd/torchnmf_amd
timeout 1500 python -m pytest tests -q -m gpu > $OUT/pytest.log 2>&1
echo "pytest rc=$?"; tail -4 $OUT/pytest.log
run() { # label, lib suffix, args...
  local lab=$1 v=$2; shift 2
  NMFMU_LIB=$T/libnmfmu$v.so timeout 300 python bench.py --cpu-iters 0 --no-parity-mode --no-sweep --repeats 3 "$@" > $OUT/${lab}${v}.json 2>> $OUT/bench.err
  python - <<PY
import json
d=json.load(open("$OUT/${lab}${v}.json")); r=d['roofline']
print("%-10s lib=%-6s %7.1f it/s  kernel %.4f ms (w %.4f h %.4f) frac %.4f clock %s power %s" % ("$lab", "$v", d['iters_per_s'], r['avg_launch_ms'], r['avg_launch_ms_w_step'], r['avg_launch_ms_h_step'], r['frac'], r.get('clock_mhz'), r.get('power_w')))
PY
}
for v in "" _notr; do
  run b0.5 "$v" --beta 0.5
  run b2 "$v" --beta 2
  run f16x "$v" --precision f16x
  run cfg5 "$v" --config cfg5
  run r32b2 "$v" --beta 2 --rank 32
done
tail -3 $OUT/bench.err
