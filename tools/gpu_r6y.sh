#!/bin/bash
# round 6: per-workgroup timeline of the rank-256 software-pipelined kernel on configs[4]'s shard (W against H half-step)
OUT=gpurun_out/r6y; mkdir -p $OUT
timeout 900 python tools/sp_timeline.py 2>&1 | grep -v amdgpu.ids | tee $OUT/sp_timeline.txt
