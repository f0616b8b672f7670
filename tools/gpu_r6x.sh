#!/bin/bash
# round 6: beta == 1 denominators from the panel's partial column sums inside the consuming kernels (no colsum_finalize launches)
# against the finalized vectors (TORCHNMF_AMD_KL_PARTS=0): the whole GPU suite, then A/B of the headline and the 3-byte target
OUT=gpurun_out/r6x; mkdir -p $OUT
timeout 2400 python -m pytest tests -q -m gpu -x > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; grep -h "passed\|failed\|FAILED" $OUT/pytest_gpu.log | tail -5 | cut -c1-250
for rep in 1 2 3; do
for v in 0 1; do
for p in f16 f16r; do
TORCHNMF_AMD_KL_PARTS=$v timeout 300 python bench.py --precision $p --steps 30 --no-sweep --cpu-iters 0 --no-parity-mode --repeats 3 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); r=d['roofline']
print('kl_parts=$v $p: it/s', d['iters_per_s'], 'ms', d['ms_per_step'], 'kernel_ms', r['avg_launch_ms'], 'w/h', r.get('avg_launch_ms_w_step'), r.get('avg_launch_ms_h_step'), 'outside', r.get('outside_fused_kernels_ms'))" | tee -a $OUT/ab.txt
done; done; done
