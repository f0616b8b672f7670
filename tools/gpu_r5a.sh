#!/bin/bash
# round 5, first GPU call: full GPU suite (ABI 8, window staging, BetaMu / PLCA changes), NMFD A/B of the window staging,
# the Infinity-Cache go / no-go of configs[1] (tools/mall_probe.py, stamps + nt / default X loads), a short default bench.
TAG=${1:-r5a}; OUT=gpurun_out/$TAG; mkdir -p $OUT
timeout 1500 python -m pytest tests -q -m gpu > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -15 $OUT/pytest_gpu.log | cut -c1-300
for i in 1 2; do
  for m in 1 0; do
    TORCHNMF_AMD_NMFD_WINSTAGE=$m timeout 300 python bench.py --workload nmfd --cpu-iters 0 --steps 50 --telemetry-s 0.5 > $OUT/nmfd_ws${m}_$i.json 2>> $OUT/nmfd.err
    python - <<PY
import json
try:
    d=json.load(open("$OUT/nmfd_ws${m}_$i.json")); r=d["roofline"]
    print("[nmfd winstage=$m] it/s=%.1f ms=%.4f gemms=%s clock=%s power=%s fit=%s" % (d["iters_per_s"], d["ms_per_step"], {k:round(v["avg_launch_ms"]*1e3,1) for k,v in r["per_gemm"].items()}, r.get("clock_mhz"), r.get("power_w"), (d.get("fit") or {}).get("iters_per_s_loop")))
except Exception as e: print("[nmfd $m] FAILED", e)
PY
  done
done
tail -3 $OUT/nmfd.err
NMFMU_LIB=$PWD/pytorch-nmf_amd/torchnmf_amd/libnmfmu_dbg.so timeout 300 python tools/mall_probe.py f16 > $OUT/mall_nt.json 2> $OUT/mall_nt.err; echo "mall nt rc=$?"; cat $OUT/mall_nt.err | grep -v amdgpu.ids | tail -6
NMFMU_LIB=$PWD/pytorch-nmf_amd/torchnmf_amd/libnmfmu_dbgx.so timeout 300 python tools/mall_probe.py f16 > $OUT/mall_default.json 2> $OUT/mall_default.err; echo "mall default rc=$?"; cat $OUT/mall_default.err | grep -v amdgpu.ids | tail -6
timeout 900 python bench.py --cpu-iters 3 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; tail -3 $OUT/bench.err; python tools/bench_brief.py $OUT/bench.json 2>&1 | tail -40
