#!/bin/bash
# round 6: the two-accumulator software-pipelined kernel (nmfmu_sp2.h) -- tests, then configs[2]'s beta legs old / new
TAG=${1:-r6i}; OUT=gpurun_out/$TAG; mkdir -p $OUT
LIBD=$PWD/pytorch-nmf_amd/torchnmf_amd
timeout 1500 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "rank128_two_accumulator or half_steps_f16_every_beta or rank128_single_plane_beta_sweep or f16_scaled or cfg2_full_size" 2>&1 | tail -25 | tee $OUT/pytest.txt
for i in 1 2; do
  for v in _nosp2 ""; do
    for b in 0.5 0 1.5 0.3; do
      NMFMU_LIB=$LIBD/libnmfmu$v.so timeout 300 python bench.py --beta $b --steps 20 --no-sweep --cpu-iters 0 --no-parity-mode --repeats 3 > $OUT/b${v}_$b_$i.json 2>> $OUT/err.log
      python - <<PY
import json
try:
    d=json.load(open("$OUT/b${v}_$b_$i.json")); r=d["roofline"]
    print("[beta %-4s lib%-7s #$i] it/s=%7.1f kernel_ms=%.4f (w %.4f h %.4f) frac=%.4f clock=%s power=%s ns=(%s,%s)" % ("$b", "$v", d["iters_per_s"], r["avg_launch_ms"], r["avg_launch_ms_w_step"], r["avg_launch_ms_h_step"], r["frac"], r.get("clock_mhz"), r.get("power_w"), d["config"]["nsplit_w"], d["config"]["nsplit_h"]))
except Exception as e: print("[$b $v] FAILED", e)
PY
    done
  done
done
grep -v amdgpu.ids $OUT/err.log | tail -5
