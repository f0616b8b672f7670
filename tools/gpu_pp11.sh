#!/bin/bash
TAG=${1:-pp11}; OUT=gpurun_out/$TAG; mkdir -p $OUT
KSEL="half_steps or rank128 or shapes_and_ksplit or cfg1 or f16 or large_slice or sharded or g1_golden or g3 or g4 or betamu_g7 or plca or sparse_fit_g9 or graph_replay"
timeout 1200 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "$KSEL" -x > $OUT/pytest.log 2>&1
echo "pytest rc=$?"; tail -3 $OUT/pytest.log
run() {
  name=$1; shift
  envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout 300 python bench.py --steps 50 --warmup 20 --cpu-iters 0 --repeats 3 --no-parity-mode "$@" > $OUT/${name}_$i.json 2>> $OUT/bench.err
  python - <<PY
import json
try:
    d=json.load(open("$OUT/${name}_$i.json")); r=d["roofline"]; c=d["config"]
    print("[%-14s] it/s=%.1f ms/step=%.4f (w %.4f h %.4f) outside %.4f TF=%.0f" % ("$name", d["iters_per_s"], d["ms_per_step"], r["avg_launch_ms_w_step"], r["avg_launch_ms_h_step"], r["outside_fused_kernels_ms"], r["achieved"]))
except Exception as e: print("[$name] FAILED", e)
PY
}
for i in 1 2; do
  run finalize NMFMU_PARTS=0 -- --precision bf16
  run parts NMFMU_PARTS=1 -- --precision bf16
  run parts_f16 NMFMU_PARTS=1 -- --precision f16
done
