#!/bin/bash
# round 6: NMFMU_STAGE_DMA_NOP2 (the software-pipelined kernels' factors: the transposed images are not kept) -- tests, then per
# configuration the bench with and without (TORCHNMF_AMD_NO_P2), interleaved twice
OUT=gpurun_out/r6z; mkdir -p $OUT
timeout 1500 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "rank256 or cfg5 or r256 or deterministic or transposed_images or sharded_path or every_beta or cfg2 or two_accumulator" 2>&1 | grep -E "passed|failed|^E " | tail -4 | tee $OUT/tests_nop2.txt
for rep in 1 2; do
for v in 0 1; do
for b in 0.5 0; do
TORCHNMF_AMD_NO_P2=$v timeout 300 python bench.py --beta $b --steps 20 --no-sweep --cpu-iters 0 --no-parity-mode --repeats 3 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); r=d['roofline']
print('no_p2=$v beta=$b: it/s', d['iters_per_s'], 'kernel_ms', r['avg_launch_ms'], 'w/h', r.get('avg_launch_ms_w_step'), r.get('avg_launch_ms_h_step'), 'frac', r['frac'])" | tee -a $OUT/nop2_sp2.txt
done; done; done
