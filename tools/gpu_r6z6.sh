#!/bin/bash
# round 6: two-accumulator pipelined kernel, staged fused-apply epilogue (new: both accumulator sets through LDS tiles, 16-byte
# master rows, row-major image from registers) against the per-element form (libnmfmu_epi0.so): tests, then beta = 0.5 / 0 / 1.5 legs
OUT=gpurun_out/r6z; mkdir -p $OUT
LIBD=$PWD/pytorch-nmf_amd/torchnmf_amd
timeout 1800 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -q -m gpu -x -k "every_beta or cfg2 or two_accumulator or deterministic or golden or fuzz or g1 or g2 or g3 or g4 or smoke" 2>&1 | grep -E "passed|failed|^E " | tail -4 | tee $OUT/tests_sp2epi.txt
for rep in 1 2 3; do
for v in _epi0 ""; do
for b in 0.5 0 1.5; do
NMFMU_LIB=$LIBD/libnmfmu$v.so timeout 300 python bench.py --beta $b --steps 20 --no-sweep --cpu-iters 0 --no-parity-mode --repeats 3 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); r=d['roofline']
print('lib$v beta=$b: it/s', d['iters_per_s'], 'kernel_ms', r['avg_launch_ms'], 'w/h', r.get('avg_launch_ms_w_step'), r.get('avg_launch_ms_h_step'), 'frac', r['frac'])" | tee -a $OUT/sp2_staged.txt
done; done; done
