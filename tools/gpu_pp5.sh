#!/bin/bash
TAG=${1:-pp5}; OUT=gpurun_out/$TAG; mkdir -p $OUT
run() {
  name=$1; shift
  envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout 300 python bench.py --steps 40 --warmup 10 --cpu-iters 0 "$@" > $OUT/${name}_$i.json 2>> $OUT/bench.err
  python - <<PY
import json
try:
    d=json.load(open("$OUT/${name}_$i.json")); r=d["roofline"]; c=d["config"]
    print("[%-14s] it/s=%.1f ms/step=%.4f (w %.4f h %.4f) nsplit w/h %d/%d" % ("$name", d["iters_per_s"], d["ms_per_step"], r["avg_launch_ms_w_step"], r["avg_launch_ms_h_step"], c["nsplit_w"], c["nsplit_h"]))
except Exception as e: print("[$name] FAILED", e)
PY
}
for i in 1 2; do
  for cols in 65536 32768 16384 8192 2048; do
    run xreg_c$cols NMFMU_PP_VAR=256 NMFMU_FORCE_NSPLIT=1 -- --precision bf16 --cols $cols
    run nodma_c$cols NMFMU_PP_VAR=280 NMFMU_FORCE_NSPLIT=1 -- --precision bf16 --cols $cols
  done
done
