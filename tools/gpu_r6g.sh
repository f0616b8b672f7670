#!/bin/bash
TAG=${1:-r6g}; OUT=gpurun_out/$TAG; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "cfg5_full_shard or abi or static" 2>&1 | tail -5 | tee $OUT/pytest_fullshard.txt
timeout 300 python bench.py --config cfg5 --steps 10 --cpu-iters 0 --no-parity-mode > $OUT/bench_cfg5.json 2>> $OUT/err.log
python tools/bench_brief.py $OUT/bench_cfg5.json 2>/dev/null | head -12
python - <<PY
import json
d=json.load(open("$OUT/bench_cfg5.json")); r=d["roofline"]
print("in_kernel:", r.get("in_kernel"), r.get("in_kernel_error"))
print("ceiling:", {k:v for k,v in (r.get("ceiling") or {}).items() if k in ("with_stream","mfma_only")}, r.get("ceiling_error"))
PY
timeout 900 python bench.py > $OUT/bench_default.json 2>> $OUT/err.log
python tools/bench_brief.py $OUT/bench_default.json 2>/dev/null | head -40
python - <<PY
import json
d=json.load(open("$OUT/bench_default.json")); r=d["roofline"]
print("in_kernel:", r.get("in_kernel"), r.get("in_kernel_error"))
print("ceiling:", {k:v for k,v in (r.get("ceiling") or {}).items() if k in ("with_stream","mfma_only")}, r.get("ceiling_error"))
PY
grep -v amdgpu.ids $OUT/err.log | tail -5
