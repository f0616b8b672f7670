#!/bin/bash
# round 4: NMF3D engine-level timing, round-4 paths vs explicit operands + store-then-fold
mkdir -p gpurun_out/r4n
timeout 200 python tools/nmf3d_time.py > gpurun_out/r4n/nmf3d_new.json 2> gpurun_out/r4n/err.txt; cat gpurun_out/r4n/nmf3d_new.json
TORCHNMF_AMD_NMFD_EXPLICIT=1 TORCHNMF_AMD_NMFD_H_ROWS=0 TORCHNMF_AMD_NMFD_KSPLIT=0 PRECISIONS=bf16x3 STEPS=8 timeout 200 python tools/nmf3d_time.py > gpurun_out/r4n/nmf3d_old.json 2>> gpurun_out/r4n/err.txt; cat gpurun_out/r4n/nmf3d_old.json
tail -2 gpurun_out/r4n/err.txt
