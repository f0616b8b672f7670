#!/bin/bash
# quick A/B of bench variants on one box.  Usage: bash tools/gpu_ab.sh tag "args1" "args2" ...
TAG=$1; shift
OUT=gpurun_out/$TAG; mkdir -p $OUT
for i in 1 2; do
  n=0
  for args in "$@"; do
    n=$((n+1))
    timeout 300 python bench.py --steps 40 --warmup 10 --cpu-iters 0 $args > $OUT/ab_${n}_$i.json 2>> $OUT/ab.err
    python - <<PY
import json
try:
    d=json.load(open("$OUT/ab_${n}_$i.json")); r=d["roofline"]
    print("[$args] it/s=%.1f ms/step=%.4f fused_ms=%.4f (w %.4f h %.4f) TF=%.0f" % (d["iters_per_s"], d["ms_per_step"], r["avg_launch_ms"], r["avg_launch_ms_w_step"], r["avg_launch_ms_h_step"], r["achieved"]))
except Exception as e: print("[$args] FAILED", e)
PY
  done
done
tail -3 $OUT/ab.err
