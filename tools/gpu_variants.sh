#!/bin/bash
# time kernel-variant builds (libnmfmu_<v>.so) against the default library on one box
TAG=$1; shift
OUT=gpurun_out/$TAG; mkdir -p $OUT
for i in 1 2; do
  for v in "$@"; do
    lib=$PWD/pytorch-nmf_amd/torchnmf_amd/libnmfmu$v.so
    NMFMU_LIB=$lib timeout 300 python bench.py --steps 40 --warmup 10 --cpu-iters 0 ${BENCH_ARGS} > $OUT/v_${v}_$i.json 2>> $OUT/v.err
    python - <<PY
import json
try:
    d=json.load(open("$OUT/v_${v}_$i.json")); r=d["roofline"]
    print("[%-4s] it/s=%.1f ms/step=%.4f fused_ms=%.4f (w %.4f h %.4f) TF=%.0f clock=%s MHz power=%s W" % ("$v" or "base", d["iters_per_s"], d["ms_per_step"], r["avg_launch_ms"], r["avg_launch_ms_w_step"], r["avg_launch_ms_h_step"], r["achieved"], r.get("clock_mhz"), r.get("power_w")))
except Exception as e: print("[$v] FAILED", e)
PY
  done
done
