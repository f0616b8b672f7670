#!/bin/bash
# round 6: per-workgroup timeline of the two-accumulator pipelined kernel at configs[2] (beta = 0.5)
OUT=gpurun_out/r6w; mkdir -p $OUT
timeout 600 python tools/sp_timeline.py --rows 4096 --cols 65536 --rank 128 --beta 0.5 --iters 20 2>&1 | grep -v amdgpu.ids | tee $OUT/sp2_timeline_cfg2.txt
