#!/bin/bash
TAG=${1:-pp4}; OUT=gpurun_out/$TAG; mkdir -p $OUT
KSEL="half_steps_f16 or f16_range or fit_f16 or rank128"
for v in 0 256; do
  NMFMU_PP_VAR=$v timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "$KSEL" > $OUT/pytest_v$v.log 2>&1
  echo "pytest VAR=$v rc=$?"; tail -4 $OUT/pytest_v$v.log; grep "^f16 " $OUT/pytest_v$v.log | head -12
done
NMFMU_PP_VAR=384 timeout 300 python tools/pp_timeline.py bf16 > $OUT/timeline_xreg.txt 2>&1; grep -v amdgpu.ids $OUT/timeline_xreg.txt | head -30
run() {
  name=$1; shift
  envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout 300 python bench.py --steps 40 --warmup 10 --cpu-iters 0 "$@" > $OUT/${name}_$i.json 2>> $OUT/bench.err
  python - <<PY
import json
try:
    d=json.load(open("$OUT/${name}_$i.json")); r=d["roofline"]
    print("[%-14s] it/s=%.1f ms/step=%.4f fused_ms=%.4f (w %.4f h %.4f) TF=%.0f" % ("$name", d["iters_per_s"], d["ms_per_step"], r["avg_launch_ms"], r["avg_launch_ms_w_step"], r["avg_launch_ms_h_step"], r["achieved"]))
except Exception as e: print("[$name] FAILED", e)
PY
}
for i in 1 2; do
  run xreg NMFMU_PP_VAR=256 -- --precision bf16
  run x_no_x NMFMU_PP_VAR=264 -- --precision bf16
  run x_no_panel NMFMU_PP_VAR=272 -- --precision bf16
  run x_no_dma NMFMU_PP_VAR=280 -- --precision bf16
  run x_no_ew NMFMU_PP_VAR=288 -- --precision bf16
  run x_no_mfma NMFMU_PP_VAR=320 -- --precision bf16
  run x_2panel NMFMU_PP_VAR=1280 -- --precision bf16
  run x_2x NMFMU_PP_VAR=2304 -- --precision bf16
  run f16_xreg NMFMU_PP_VAR=256 -- --precision f16
done
