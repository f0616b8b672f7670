#!/bin/bash
# round 6 (diagnostic, timing only): which part of the ping-pong kernel's fused-apply epilogue takes its 18.7 us at configs[1] --
# builds without the accumulator staging (1), the update pass (2), the transposed image (4), the master loads (8), the master stores
# (16), the row-major image stores (32), combinations; per-workgroup timeline of each
OUT=gpurun_out/r6z; mkdir -p $OUT
LIBD=$PWD/pytorch-nmf_amd/torchnmf_amd
for v in "" _pa1 _pa2 _pa4 _pa8 _pa16 _pa32 _pa56 _pa63; do
echo "--- lib$v" | tee -a $OUT/pp_epi_abl.txt
NMFMU_LIB=$LIBD/libnmfmu$v.so timeout 300 python tools/sp_timeline.py --rows 4096 --cols 65536 --rank 128 --iters 6 2>&1 | grep -v amdgpu.ids | grep "W half-step\|epilogue" | head -2 | tee -a $OUT/pp_epi_abl.txt
done
