#!/bin/bash
# round 6: the software-pipelined rank-256 kernel -- its tests, bit-identity against the four-wave kernel, configs[4]'s shard A/B
TAG=${1:-r6c}; OUT=gpurun_out/$TAG; mkdir -p $OUT
LIBD=$PWD/pytorch-nmf_amd/torchnmf_amd
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "rank256 or rank_above_128 or cfg5_shard or sharded_path_world1" 2>&1 | tail -25 | tee $OUT/pytest.txt
for ns in 4 8; do
  TORCHNMF_AMD_NSPLIT=$ns NMFMU_LIB=$LIBD/libnmfmu_nosp.so timeout 300 python tools/sp_bitcompare.py --save /tmp/ref_$ns.pt 2>&1 | tail -1
  TORCHNMF_AMD_NSPLIT=$ns timeout 300 python tools/sp_bitcompare.py --compare /tmp/ref_$ns.pt 2>&1 | tail -3 | tee -a $OUT/bitcompare.txt
done
for i in 1 2; do
  for cfg in "_nosp:" ":" ":8" ${EXTRA_CFGS}; do
    v=${cfg%%:*}; ns=${cfg##*:}
    TORCHNMF_AMD_NSPLIT=$ns NMFMU_LIB=$LIBD/libnmfmu$v.so timeout 300 python bench.py --config cfg5 --steps 10 --no-sweep --cpu-iters 0 --no-parity-mode --repeats 3 > $OUT/cfg5${v}_ns${ns}_$i.json 2>> $OUT/err.log
    python - <<PY
import json
try:
    d=json.load(open("$OUT/cfg5${v}_ns${ns}_$i.json")); r=d["roofline"]
    print("[cfg5 lib%-6s nsplit=%-3s #$i] it/s=%7.1f kernel_ms=%.4f (w %.4f h %.4f) frac=%.4f clock=%s power=%s ceil=%s ns_h=%s" % ("$v", "$ns", d["iters_per_s"], r["avg_launch_ms"], r["avg_launch_ms_w_step"], r["avg_launch_ms_h_step"], r["frac"], r.get("clock_mhz"), r.get("power_w"), r.get("ceiling_tflops"), d["config"]["nsplit_h"]))
except Exception as e: print("[$cfg] FAILED", e)
PY
  done
done
grep -v amdgpu.ids $OUT/err.log | tail -5
