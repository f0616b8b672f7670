#!/bin/bash
# round 6: the 3-byte target's ratio stage with two v_pk_mul_f32 per four elements (12 VALU) against four v_mul_f32 (14): f16r tests,
# then A/B on the f16r headline, interleaved three times
OUT=gpurun_out/r6u; mkdir -p $OUT
LIBD=$PWD/pytorch-nmf_amd/torchnmf_amd
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "f16r or riding or unrounded" 2>&1 | tail -3 | tee $OUT/tests.txt
for rep in 1 2 3; do
for v in _pk0 ""; do
NMFMU_LIB=$LIBD/libnmfmu$v.so timeout 300 python bench.py --precision f16r --steps 30 --no-sweep --cpu-iters 0 --no-parity-mode --repeats 3 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); r=d['roofline']; k=r.get('in_kernel') or {}
print('lib$v f16r: it/s', d['iters_per_s'], 'kernel_ms', r['avg_launch_ms'], 'w/h', r.get('avg_launch_ms_w_step'), r.get('avg_launch_ms_h_step'), 'frac', r['frac'], 'cyc/tile', k.get('cycles_per_tile'), 'MHz', k.get('clock_mhz_in_kernel'))" | tee -a $OUT/ab.txt
done; done
