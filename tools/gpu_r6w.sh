#!/bin/bash
# round 6: per-workgroup timeline of the ping-pong kernel at configs[1] (same stamps, same script as the rank-256 kernel)
OUT=gpurun_out/r6w; mkdir -p $OUT
timeout 600 python tools/sp_timeline.py --rows 4096 --cols 65536 --rank 128 --iters 30 2>&1 | grep -v amdgpu.ids | tee $OUT/pp_timeline_cfg1.txt
