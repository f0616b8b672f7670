#!/bin/bash
# round 6: the column-sum finalize inside the producing kernel (last workgroup) against the separate launches
# (-DNMFMU_INKERNEL_COLSUM=0): the whole GPU suite, then A/B of the headline and the 3-byte target, interleaved three times
OUT=gpurun_out/r6v; mkdir -p $OUT
LIBD=$PWD/pytorch-nmf_amd/torchnmf_amd
timeout 2400 python -m pytest tests -q -m gpu -x > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; grep -h "passed\|failed\|FAILED" $OUT/pytest_gpu.log | tail -5 | cut -c1-250
for rep in 1 2 3; do
for v in _nocs ""; do
for p in f16 f16r; do
NMFMU_LIB=$LIBD/libnmfmu$v.so timeout 300 python bench.py --precision $p --steps 30 --no-sweep --cpu-iters 0 --no-parity-mode --repeats 3 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); r=d['roofline']
print('lib$v $p: it/s', d['iters_per_s'], 'ms', d['ms_per_step'], 'kernel_ms', r['avg_launch_ms'], 'w/h', r.get('avg_launch_ms_w_step'), r.get('avg_launch_ms_h_step'), 'outside', r.get('outside_fused_kernels_ms'))" | tee -a $OUT/ab.txt
done; done; done
