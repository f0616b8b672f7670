#!/bin/bash
# round 5: launch diet of the window-operand path (NMF2D / NMF3D / short NMFD): tests + A/B
TAG=${1:-r5f}; OUT=gpurun_out/$TAG; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "launch_diet or nmf2d or nmf3d or shifted or several_shift or window_operand or convnd or siplca" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -8 $OUT/pytest.log | cut -c1-250
for i in 1 2; do
  for m in 1 0; do
    TORCHNMF_AMD_NMFD_ROWS_FUSED=$m timeout 200 python bench.py --workload nmf2d --precision auto --cpu-iters 0 --steps 50 --repeats 3 --telemetry-s 0 > $OUT/nmf2d_fused${m}_$i.json 2>> $OUT/err.log
    python - <<PY
import json
try:
    d=json.load(open("$OUT/nmf2d_fused${m}_$i.json")); r=d["roofline"]
    print("[nmf2d rows_fused=$m] it/s=%7.1f ms=%.4f gemms=%s fit=%s" % (d["iters_per_s"], d["ms_per_step"], {k:round(x["avg_launch_ms"]*1e3,1) for k,x in r["per_gemm"].items()}, (d.get("fit") or {}).get("iters_per_s_loop")))
except Exception as e: print("[$m] FAILED", e)
PY
  done
done
for m in 1 0; do TORCHNMF_AMD_NMFD_ROWS_FUSED=$m timeout 200 python tools/nmf3d_time.py > $OUT/nmf3d_fused$m.json 2>> $OUT/err.log; tail -c 400 $OUT/nmf3d_fused$m.json; echo; done
for m in 1 0; do TORCHNMF_AMD_NMFD_ROWS_FUSED=$m timeout 200 python tools/nmfd_short_time.py > $OUT/nmfd_short_fused$m.json 2>> $OUT/err.log; tail -c 400 $OUT/nmfd_short_fused$m.json; echo; done
tail -3 $OUT/err.log
