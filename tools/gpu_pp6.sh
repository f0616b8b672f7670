#!/bin/bash
TAG=${1:-pp6}; OUT=gpurun_out/$TAG; mkdir -p $OUT
for cols in 65536 2048; do
for v in 384 408 416 440; do
  PP_STEPS=w PP_COLS=$cols NMFMU_FORCE_NSPLIT=1 NMFMU_PP_VAR=$v timeout 300 python tools/pp_timeline.py bf16 2>&1 | grep -v amdgpu.ids | tee -a $OUT/timeline.txt
done
done
