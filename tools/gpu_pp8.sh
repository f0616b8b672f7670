#!/bin/bash
TAG=${1:-pp8}; OUT=gpurun_out/$TAG; mkdir -p $OUT
for pv in "bf16 128" "f16 128"; do
  set -- $pv
  NMFMU_PP_VAR=$2 timeout 300 python tools/pp_timeline.py $1 2>&1 | grep -v amdgpu.ids | tee -a $OUT/timeline.txt
done
