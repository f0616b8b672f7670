#!/bin/bash
# round 5, fifth call: three staging buffers (LDS-DMA latency) x four / eight waves x window / chunk-major: tests + A/B on configs[3]
TAG=${1:-r5e}; OUT=gpurun_out/$TAG; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "staging or nmfd_fit or nmfd_cfg4 or ragged" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -6 $OUT/pytest.log | cut -c1-250
for i in 1 2; do
  for m in 1 2 5 4 6 7; do
    TORCHNMF_AMD_NMFD_STAGE=$m timeout 200 python bench.py --workload nmfd --cpu-iters 0 --steps 50 --repeats 3 --telemetry-s 0.3 --no-parity-mode > $OUT/nmfd_stage${m}_$i.json 2>> $OUT/err.log
    python - <<PY
import json
try:
    d=json.load(open("$OUT/nmfd_stage${m}_$i.json")); r=d["roofline"]
    print("[stage_mode=$m] it/s=%7.1f gemms=%s clock=%s power=%s fit=%s" % (d["iters_per_s"], {k:round(x["avg_launch_ms"]*1e3,1) for k,x in r["per_gemm"].items()}, r.get("clock_mhz"), r.get("power_w"), (d.get("fit") or {}).get("iters_per_s_loop")))
except Exception as e: print("[$m] FAILED", e)
PY
  done
done
tail -3 $OUT/err.log
