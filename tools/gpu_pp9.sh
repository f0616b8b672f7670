#!/bin/bash
TAG=${1:-pp9}; OUT=gpurun_out/$TAG; mkdir -p $OUT
KSEL="pack_x or pack_factor or half_steps or rank128 or shapes_and_ksplit or cfg1 or f16 or large_slice or sharded or betamu_g7 or plca_medium"
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "$KSEL" > $OUT/pytest.log 2>&1
echo "pytest rc=$?"; tail -4 $OUT/pytest.log; grep "^f16 \|^cfg1" $OUT/pytest.log | head -12
for pv in "bf16 128" "bf16 8320" "f16 128" "f16 8320"; do
  set -- $pv
  NMFMU_PP_VAR=$2 timeout 300 python tools/pp_timeline.py $1 2>&1 | grep -v amdgpu.ids | tee -a $OUT/timeline.txt
done
run() {
  name=$1; shift
  envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout 300 python bench.py --steps 40 --warmup 10 --cpu-iters 0 "$@" > $OUT/${name}_$i.json 2>> $OUT/bench.err
  python - <<PY
import json
try:
    d=json.load(open("$OUT/${name}_$i.json")); r=d["roofline"]; c=d["config"]
    print("[%-14s] it/s=%.1f ms/step=%.4f (w %.4f h %.4f) TF=%.0f" % ("$name", d["iters_per_s"], d["ms_per_step"], r["avg_launch_ms_w_step"], r["avg_launch_ms_h_step"], r["achieved"]))
except Exception as e: print("[$name] FAILED", e)
PY
}
for i in 1 2; do
  run old NMFMU_PP=0 -- --precision bf16
  run pp_bf16 NMFMU_PP_VAR=0 -- --precision bf16
  run pp_bf16_agpr NMFMU_PP_VAR=8192 -- --precision bf16
  run pp_f16 NMFMU_PP_VAR=0 -- --precision f16
  run pp_f16_agpr NMFMU_PP_VAR=8192 -- --precision f16
done
