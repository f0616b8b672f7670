#!/bin/bash
# round 6: two-accumulator pipelined kernel, fused-apply epilogue with the master values of all four rank tiles fetched ahead of the
# first store (new) against one rank tile at a time (libnmfmu_old.so, same sources otherwise): tests, then beta = 0.5 / 0 legs
OUT=gpurun_out/r6z; mkdir -p $OUT
LIBD=$PWD/pytorch-nmf_amd/torchnmf_amd
timeout 1500 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "every_beta or cfg2 or two_accumulator or transposed_images or deterministic or golden or fuzz" 2>&1 | grep -E "passed|failed|^E " | tail -4 | tee $OUT/tests_sp2epi.txt
for rep in 1 2 3; do
for v in _old ""; do
for b in 0.5 0; do
TORCHNMF_AMD_NO_P2=0 NMFMU_LIB=$LIBD/libnmfmu$v.so timeout 300 python bench.py --beta $b --steps 20 --no-sweep --cpu-iters 0 --no-parity-mode --repeats 3 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); r=d['roofline']
print('lib$v beta=$b: it/s', d['iters_per_s'], 'kernel_ms', r['avg_launch_ms'], 'w/h', r.get('avg_launch_ms_w_step'), r.get('avg_launch_ms_h_step'), 'frac', r['frac'])" | tee -a $OUT/sp2_epi.txt
done; done; done
