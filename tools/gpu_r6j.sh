#!/bin/bash
# round 6: determinism of the two-accumulator software-pipelined kernel after the barrier / drain fix -- six runs per beta, each
# compared bit for bit with the four-wave kernel at the same contraction split; then its bench legs and tests
OUT=gpurun_out/r6j; mkdir -p $OUT
LIBD=$PWD/pytorch-nmf_amd/torchnmf_amd
for b in 0 0.5 0.3; do
  TORCHNMF_AMD_NSPLIT=8 NMFMU_LIB=$LIBD/libnmfmu_nosp2.so timeout 300 python tools/sp_bitcompare.py --beta $b --iters 2 --shapes 4096x65536x128 --save /tmp/ref.pt 2>&1 | tail -1
  for r in 1 2 3 4 5 6; do
    TORCHNMF_AMD_NSPLIT=8 timeout 300 python tools/sp_bitcompare.py --beta $b --iters 2 --shapes 4096x65536x128 --compare /tmp/ref.pt 2>&1 | tail -1 | cut -c1-150 | tee -a $OUT/bitcompare_fixed.txt
  done
done
for b in 0 0.5; do
timeout 300 python bench.py --beta $b --steps 20 --no-sweep --cpu-iters 0 --no-parity-mode --repeats 3 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); r=d['roofline']; print('beta=$b: it/s', d['iters_per_s'], 'kernel_ms', r['avg_launch_ms'], 'frac', r['frac'])"
done
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "rank128_two_accumulator or cfg2_full_size" 2>&1 | tail -3
