#!/bin/bash
# round 6: fused loss checkpoint (nmfmu_loss_checkpoint) -- fit()'s whole-call time with and without it
OUT=gpurun_out/r6o; mkdir -p $OUT
timeout 600 python tools/fit_ratio.py 2>&1 | grep -v amdgpu.ids | tee $OUT/fit.txt
