#!/bin/bash
# round 5: what bounds the beta = 2 stream kernel (kModeXB, 0.60 of 8 TB/s)?  Shipped build against
# timing-only ablations (no MFMA / LDS reads; no X stream) and two knobs (five ring stages; default cache policy on X).
TAG=${1:-r5q}; OUT=gpurun_out/$TAG; mkdir -p $OUT
LIBD=$PWD/pytorch-nmf_amd/torchnmf_amd
VARS=${VARS:-"- _xbnogemm _xbnox _xbns5 _xbpol"}
for v in $VARS; do
  [ "$v" = "-" ] && v=""
  [ -f $LIBD/libnmfmu$v.so ] || { echo "no lib $v"; continue; }
  NMFMU_LIB=$LIBD/libnmfmu$v.so timeout 300 python bench.py $ARGS --beta 2 --gram --no-sweep --cpu-iters 0 --no-parity-mode --repeats 3 > $OUT/b${v}.json 2>> $OUT/err.log
  python - <<PY
import json
try:
    d=json.load(open("$OUT/b${v}.json")); r=d["roofline"]
    print("[lib%-10s $ARGS] it/s=%7.1f kernel_ms=%.4f (w %.4f h %.4f) frac=%.4f clock=%s power=%s" % ("$v", d["iters_per_s"], r["avg_launch_ms"], r["avg_launch_ms_w_step"], r["avg_launch_ms_h_step"], r["frac"], r.get("clock_mhz"), r.get("power_w")))
except Exception as e: print("[$v] FAILED", e)
PY
done
tail -2 $OUT/err.log
