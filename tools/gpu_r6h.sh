#!/bin/bash
TAG=${1:-r6h}; OUT=gpurun_out/$TAG; mkdir -p $OUT
timeout 1200 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "betamu" 2>&1 | tail -15 | tee $OUT/pytest_betamu.txt
timeout 300 python bench.py --steps 20 --cpu-iters 0 --no-parity-mode --no-sweep > $OUT/bench_cfg1.json 2>> $OUT/err.log
python - <<PY
import json
d=json.load(open("$OUT/bench_cfg1.json")); r=d["roofline"]
print("cfg1 it/s", d["iters_per_s"], "frac", r["frac"])
print("in_kernel:", r.get("in_kernel"), r.get("in_kernel_error"))
print("ceiling:", {k:v for k,v in (r.get("ceiling") or {}).items() if k in ("with_stream","mfma_only")}, r.get("ceiling_error"))
PY
grep -v amdgpu.ids $OUT/err.log | tail -5
