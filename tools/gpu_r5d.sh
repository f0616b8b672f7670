#!/bin/bash
# round 5, fourth call: the eight-wave GEMM tile (two wave groups split every k-tile's contraction) -- staging-mode tests, NMFD
# tests against the oracle / goldens, A/B of nmfmu_gemm_desc.stage_mode 0..3 on configs[3]
TAG=${1:-r5d}; OUT=gpurun_out/$TAG; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "staging or nmfd or siplca or rank_above_256 or nmf2d" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -8 $OUT/pytest.log | cut -c1-250
for i in 1 2; do
  for m in 0 1 2 3; do
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
TORCHNMF_AMD_NMFD_STAGE=0 timeout 200 python bench.py --workload nmfd --precision bf16 --cpu-iters 3 --steps 50 --repeats 3 --telemetry-s 0 > $OUT/nmfd_bf16_stage0.json 2>> $OUT/err.log; python tools/bench_brief.py $OUT/nmfd_bf16_stage0.json 1
python - <<PY
import json
d=json.load(open("$OUT/nmfd_bf16_stage0.json")); print("bf16 stage0 it/s", d["iters_per_s"], "parity", d.get("parity"))
PY
tail -3 $OUT/err.log
