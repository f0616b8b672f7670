#!/bin/bash
# round 5, third call: what bounds the NMFD GEMM's k loop -- timing-only ablations (no MFMA / no LDS-DMA / no fragment reads /
# no barrier in the loop; wrong results by construction) of the shipped kernel, chunk-major and window-staged
TAG=${1:-r5c}; OUT=gpurun_out/$TAG; mkdir -p $OUT
LIBD=$PWD/pytorch-nmf_amd/torchnmf_amd
for v in "" _abl_NOMFMA _abl_NODMA _abl_NOFRAG _abl_NOBAR; do
  for m in 1 0; do
    NMFMU_LIB=$LIBD/libnmfmu$v.so TORCHNMF_AMD_NMFD_WINSTAGE=$m timeout 200 python bench.py --workload nmfd --cpu-iters 0 --steps 30 --repeats 3 --telemetry-s 0.3 --no-parity-mode > $OUT/nmfd${v}_ws$m.json 2>> $OUT/err.log
    python - <<PY
import json
try:
    d=json.load(open("$OUT/nmfd${v}_ws$m.json")); r=d["roofline"]
    print("[lib%-12s winstage=$m] it/s=%7.1f gemms=%s clock=%s power=%s" % ("$v", d["iters_per_s"], {k:round(x["avg_launch_ms"]*1e3,1) for k,x in r["per_gemm"].items()}, r.get("clock_mhz"), r.get("power_w")))
except Exception as e: print("[$v $m] FAILED", e)
PY
  done
done
TORCHNMF_AMD_NMFD_WINSTAGE=0 timeout 300 python tools/nmfd_gemm_klen.py f16 > $OUT/klen_ws0.json 2> $OUT/klen_ws0.err; grep -v amdgpu $OUT/klen_ws0.err | tail -6
tail -3 $OUT/err.log
