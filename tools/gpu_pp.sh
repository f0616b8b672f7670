#!/bin/bash
# One gpurun call for the ping-pong kernel: targeted parity tests, then an interleaved A/B of kernel variants.
# Usage: bash tools/gpu_pp.sh <tag> [rounds]
TAG=${1:-pp}; ROUNDS=${2:-2}
OUT=gpurun_out/$TAG; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu \
  -k "pack_x or pack_factor or half_steps or rank128 or shapes_and_ksplit or cfg1 or f16 or large_slice" > $OUT/pytest.log 2>&1
echo "pytest rc=$?" ; tail -5 $OUT/pytest.log
run() {  # name, env..., -- bench args
  name=$1; shift
  envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout 300 python bench.py --steps 40 --warmup 10 --cpu-iters 0 "$@" > $OUT/${name}_$i.json 2>> $OUT/bench.err
  python - <<PY
import json
try:
    d=json.load(open("$OUT/${name}_$i.json")); r=d["roofline"]
    print("[%-14s] it/s=%.1f ms/step=%.4f fused_ms=%.4f (w %.4f h %.4f) TF=%.0f frac=%.3f" % ("$name", d["iters_per_s"], d["ms_per_step"], r["avg_launch_ms"], r["avg_launch_ms_w_step"], r["avg_launch_ms_h_step"], r["achieved"], r["frac"]))
except Exception as e: print("[$name] FAILED", e)
PY
}
for i in $(seq 1 $ROUNDS); do
  run old_bf16 NMFMU_PP=0 -- --precision bf16
  run pp_bf16_v0 NMFMU_PP_VAR=0 -- --precision bf16
  run pp_bf16_v1 NMFMU_PP_VAR=1 -- --precision bf16
  run pp_bf16_v2 NMFMU_PP_VAR=2 -- --precision bf16
  run pp_bf16_v4 NMFMU_PP_VAR=4 -- --precision bf16
  run pp_f16_v0 NMFMU_PP_VAR=0 -- --precision f16
  run pp_f16_v1 NMFMU_PP_VAR=1 -- --precision f16
  run pp_f16_v4 NMFMU_PP_VAR=4 -- --precision f16
done
tail -5 $OUT/bench.err
