#!/bin/bash
# round 5: is beta < 1 bound by its serial elementwise stage or by the socket's power limit?  The shipped
# kernel against a timing-only build without the transcendental / multiply chain (-DNMFMU_FUSED_ABL_NOELEM): clock, power, launch time.
TAG=${1:-r5g}; OUT=gpurun_out/$TAG; mkdir -p $OUT
LIBD=$PWD/pytorch-nmf_amd/torchnmf_amd
for v in "" _abl_noelem; do
  for args in "--beta 0.5" "--beta 0" "--beta 2" "--config cfg5 --steps 10"; do
    tag=$(echo $args | tr -d ' -' | tr '.' 'p')
    NMFMU_LIB=$LIBD/libnmfmu$v.so timeout 300 python bench.py $args --no-sweep --cpu-iters 0 --no-parity-mode --repeats 3 > $OUT/b${v}_$tag.json 2>> $OUT/err.log
    python - <<PY
import json
try:
    d=json.load(open("$OUT/b${v}_$tag.json")); r=d["roofline"]
    print("[lib%-12s %-22s] it/s=%7.1f kernel_ms=%.4f (w %.4f h %.4f) frac=%.4f clock=%s power=%s" % ("$v", "$args", d["iters_per_s"], r["avg_launch_ms"], r["avg_launch_ms_w_step"], r["avg_launch_ms_h_step"], r["frac"], r.get("clock_mhz"), r.get("power_w")))
except Exception as e: print("[$v $args] FAILED", e)
PY
  done
done
tail -2 $OUT/err.log
