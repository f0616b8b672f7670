#!/bin/bash
# diagnostics of the ping-pong kernel: fp16 hardware probe, segment timeline, ablation timings, PMC counters
TAG=${1:-pp2}; OUT=gpurun_out/$TAG; mkdir -p $OUT
./tools/ubench/f16_probe > $OUT/f16_probe.txt 2>&1; cat $OUT/f16_probe.txt
NMFMU_PP_VAR=128 timeout 300 python tools/pp_timeline.py bf16 > $OUT/timeline.txt 2>&1; cat $OUT/timeline.txt
run() {
  name=$1; shift
  envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout 300 python bench.py --steps 40 --warmup 10 --cpu-iters 0 "$@" > $OUT/${name}_$i.json 2>> $OUT/bench.err
  python - <<PY
import json
try:
    d=json.load(open("$OUT/${name}_$i.json")); r=d["roofline"]
    print("[%-14s] it/s=%.1f ms/step=%.4f fused_ms=%.4f (w %.4f h %.4f) TF=%.0f" % ("$name", d["iters_per_s"], d["ms_per_step"], r["avg_launch_ms"], r["avg_launch_ms_w_step"], r["avg_launch_ms_h_step"], r["achieved"]))
except Exception as e: print("[$name] FAILED", e)
PY
}
for i in 1 2; do
  run base NMFMU_PP_VAR=0 -- --precision bf16
  run no_x NMFMU_PP_VAR=8 -- --precision bf16
  run no_panel NMFMU_PP_VAR=16 -- --precision bf16
  run no_dma NMFMU_PP_VAR=24 -- --precision bf16
  run no_ew NMFMU_PP_VAR=32 -- --precision bf16
  run no_mfma NMFMU_PP_VAR=64 -- --precision bf16
  run mfma_only NMFMU_PP_VAR=56 -- --precision bf16
  run ew_only NMFMU_PP_VAR=88 -- --precision bf16
done
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM"; do
  n=$(echo $grp | cut -d' ' -f1)
  timeout 300 rocprofv3 --pmc $grp -d $R/$OUT/pmc_$n -o pmc -- python $R/bench.py --steps 5 --warmup 3 --cpu-iters 0 --no-roofline > $R/$OUT/pmc_$n.log 2>&1
done
cd $R; python tools/pmc_summary.py $OUT 2>&1 | tail -40
