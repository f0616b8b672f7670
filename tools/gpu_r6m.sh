#!/bin/bash
# round 6: where the ping-pong kernel issues its X loads -- end of the E segment (xm0) vs. the gaps of the next M segment (xm1),
# for the fp16 target (headline) and the 3-byte target; interleaved twice
OUT=gpurun_out/r6m; mkdir -p $OUT
LIBD=$PWD/pytorch-nmf_amd/torchnmf_amd
timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "f16r or cfg1_full or unrounded_target" 2>&1 | tail -3 | tee $OUT/tests.txt
for rep in 1 2; do
for v in xm0 xm1; do
for p in f16 f16r; do
NMFMU_LIB=$LIBD/libnmfmu_$v.so timeout 300 python bench.py --precision $p --steps 20 --no-sweep --cpu-iters 0 --no-parity-mode --repeats 3 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); r=d['roofline']; k=r['ceiling']['shipped_kernel'] if 'ceiling' in r and r['ceiling'] else {}
print('$v $p: it/s', d['iters_per_s'], 'kernel_ms', r['avg_launch_ms'], 'w/h', r.get('avg_launch_ms_w_step'), r.get('avg_launch_ms_h_step'), 'frac', r['frac'], 'cyc/tile', k.get('cycles_per_tile'), 'MHz', k.get('clock_mhz_in_kernel'))" | tee -a $OUT/ab.txt
done; done; done
