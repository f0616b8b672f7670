#!/bin/bash
# round 6 (diagnostic, timing only): which part of the rank-256 kernel's fused-apply epilogue takes its 20 us -- builds without the
# accumulator staging (1), the update pass (2), the transposed image (4), all three (7); per-workgroup timeline of each
OUT=gpurun_out/r6z; mkdir -p $OUT
LIBD=$PWD/pytorch-nmf_amd/torchnmf_amd
for v in "" _ea8 _ea16 _ea32 _ea48 _ea56; do
echo "--- lib$v" | tee -a $OUT/epi_abl.txt
NMFMU_LIB=$LIBD/libnmfmu$v.so timeout 300 python tools/sp_timeline.py --iters 3 2>&1 | grep -v amdgpu.ids | grep "W half-step\|epilogue\|prologue\|a CU" | head -4 | tee -a $OUT/epi_abl.txt
done
