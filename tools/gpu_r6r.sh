#!/bin/bash
# round 6: ratio stage of the ping-pong kernel's fp16 instances with v_fma_mixlo_f16 / v_fma_mixhi_f16 (8 VALU per four elements)
# against the v_fma_mix_f32 + v_cvt_pk_f16_f32 form (10): parity tests, then A/B of the headline and the 3-byte target
OUT=gpurun_out/r6r; mkdir -p $OUT
LIBD=$PWD/pytorch-nmf_amd/torchnmf_amd

for rep in 1 2; do
for v in _mix0 ""; do
for p in f16 f16r; do
NMFMU_LIB=$LIBD/libnmfmu$v.so timeout 300 python bench.py --precision $p --steps 30 --no-sweep --cpu-iters 0 --no-parity-mode --repeats 3 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); r=d['roofline']; k=r.get('in_kernel') or {}
print('lib$v $p: it/s', d['iters_per_s'], 'kernel_ms', r['avg_launch_ms'], 'w/h', r.get('avg_launch_ms_w_step'), r.get('avg_launch_ms_h_step'), 'frac', r['frac'], 'cyc/tile', k.get('cycles_per_tile'), 'MHz', k.get('clock_mhz_in_kernel'))" | tee -a $OUT/ab.txt
done; done; done
