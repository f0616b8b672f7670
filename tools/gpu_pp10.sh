#!/bin/bash
TAG=${1:-pp10}; OUT=gpurun_out/$TAG; mkdir -p $OUT
KSEL="pack_x or pack_factor or half_steps or rank128 or shapes_and_ksplit or cfg1 or f16 or large_slice or sharded or g1_golden or rank_above"
for v in 16384 0; do
  NMFMU_PP_VAR=$v timeout 1200 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "$KSEL" -x > $OUT/pytest_v$v.log 2>&1
  echo "pytest VAR=$v rc=$?"; tail -3 $OUT/pytest_v$v.log
done
for pv in "bf16 128" "bf16 16512"; do
  set -- $pv
  NMFMU_PP_VAR=$2 timeout 300 python tools/pp_timeline.py $1 2>&1 | grep -v amdgpu.ids | tee -a $OUT/timeline.txt
done
run() {
  name=$1; shift
  envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout 300 python bench.py --steps 40 --warmup 10 --cpu-iters 0 --repeats 3 --no-parity-mode "$@" > $OUT/${name}_$i.json 2>> $OUT/bench.err
  python - <<PY
import json
try:
    d=json.load(open("$OUT/${name}_$i.json")); r=d["roofline"]; c=d["config"]
    print("[%-14s] it/s=%.1f ms/step=%.4f (w %.4f h %.4f) TF=%.0f" % ("$name", d["iters_per_s"], d["ms_per_step"], r["avg_launch_ms_w_step"], r["avg_launch_ms_h_step"], r["achieved"]))
except Exception as e: print("[$name] FAILED", e)
PY
}
for i in 1 2; do
  run pp_bf16 NMFMU_PP_VAR=0 -- --precision bf16
  run pp_bf16_tr NMFMU_PP_VAR=16384 -- --precision bf16
  run pp_f16 NMFMU_PP_VAR=0 -- --precision f16
  run pp_f16_tr NMFMU_PP_VAR=16384 -- --precision f16
done
cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
NMFMU_PP_VAR=16384 timeout 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $R/$OUT/pmc_lds -o pmc -- python $R/bench.py --steps 5 --warmup 3 --cpu-iters 0 --repeats 1 --no-parity-mode --no-roofline > /dev/null 2> $R/$OUT/pmc.err
cd $R; python tools/pmc_summary.py $OUT 2>&1 | grep -A6 "pp_kernel"
