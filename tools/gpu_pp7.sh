#!/bin/bash
TAG=${1:-pp7}; OUT=gpurun_out/$TAG; mkdir -p $OUT
for v in 385; do
  PP_STEPS=w NMFMU_PP_VAR=$v timeout 300 python tools/pp_timeline.py bf16 2>&1 | grep -v amdgpu.ids | tee -a $OUT/timeline.txt
done
run() {
  name=$1; shift
  envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout 300 python bench.py --steps 40 --warmup 10 --cpu-iters 0 "$@" > $OUT/${name}_$i.json 2>> $OUT/bench.err
  python - <<PY
import json
try:
    d=json.load(open("$OUT/${name}_$i.json")); r=d["roofline"]; c=d["config"]
    print("[%-14s] it/s=%.1f ms/step=%.4f (w %.4f h %.4f)" % ("$name", d["iters_per_s"], d["ms_per_step"], r["avg_launch_ms_w_step"], r["avg_launch_ms_h_step"]))
except Exception as e: print("[$name] FAILED", e)
PY
}
for i in 1 2; do
  run xreg NMFMU_PP_VAR=256 -- --precision bf16
  run xreg_prio3 NMFMU_PP_VAR=257 -- --precision bf16
  run xreg_young NMFMU_PP_VAR=258 -- --precision bf16
done
