#!/bin/bash
# fused-apply epilogue rewrite: parity subset, stamps (prologue / loop / epilogue), bench
TAG=${1:-pp}; OUT=gpurun_out/$TAG; mkdir -p $OUT
KSEL="half_steps or rank128 or shapes_and_ksplit or cfg1 or f16 or large_slice or sharded or g1_golden or g3 or g4 or graph_replay or fit"
timeout 1200 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "$KSEL" -x > $OUT/pytest.log 2>&1
echo "pytest rc=$?"; tail -3 $OUT/pytest.log
for p in bf16 f16; do
  NMFMU_PP_VAR=128 timeout 300 python tools/pp_timeline.py $p > $OUT/timeline_$p.txt 2>&1; tail -12 $OUT/timeline_$p.txt
done
for i in 1 2; do
  for p in bf16 f16; do
    timeout 300 python bench.py --steps 50 --warmup 20 --cpu-iters 0 --repeats 3 --no-parity-mode --precision $p > $OUT/b_${p}_$i.json 2>> $OUT/bench.err
    python - <<PY
import json
try:
    d=json.load(open("$OUT/b_${p}_$i.json")); r=d["roofline"]
    print("[%-5s] it/s=%.1f ms/step=%.4f (w %.4f h %.4f) outside %.4f TF=%.0f" % ("$p", d["iters_per_s"], d["ms_per_step"], r["avg_launch_ms_w_step"], r["avg_launch_ms_h_step"], r["outside_fused_kernels_ms"], r["achieved"]))
except Exception as e: print("[$p] FAILED", e)
PY
  done
done
