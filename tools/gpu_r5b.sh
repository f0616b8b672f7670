#!/bin/bash
# round 5, second call: the window-staging / BetaMu tests again (test fixes, tight k_len), NMFD suite subset, and where an NMFD
# GEMM launch spends its time (tools/nmfd_gemm_klen.py: intercept = epilogue, slope = k loop)
TAG=${1:-r5b}; OUT=gpurun_out/$TAG; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "window_staging or betamu_auto or nmfd or siplca or rank_above_256 or g13" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -8 $OUT/pytest.log | cut -c1-250
TORCHNMF_AMD_NMFD_WINSTAGE=0 timeout 300 python tools/nmfd_gemm_klen.py f16 > $OUT/klen_ws0.json 2> $OUT/klen_ws0.err; grep -v amdgpu $OUT/klen_ws0.err | tail -6
TORCHNMF_AMD_NMFD_WINSTAGE=1 timeout 300 python tools/nmfd_gemm_klen.py f16 > $OUT/klen_ws1.json 2> $OUT/klen_ws1.err; grep -v amdgpu $OUT/klen_ws1.err | tail -6
