#!/bin/bash
OUT=gpurun_out/r6d; mkdir -p $OUT
for c in 256 512 1024; do SP_COLS=$c TORCHNMF_AMD_NSPLIT=1 timeout 300 python tools/sp_debug.py 2>&1 | grep -v amdgpu.ids | tee -a $OUT/sp_debug.txt; done
SP_COLS=512 TORCHNMF_AMD_NSPLIT=1 NMFMU_LIB=$PWD/pytorch-nmf_amd/torchnmf_amd/libnmfmu_nosp.so timeout 300 python tools/sp_debug.py 2>&1 | grep -v amdgpu.ids | tee -a $OUT/sp_debug.txt
