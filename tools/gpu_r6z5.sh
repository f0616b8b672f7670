#!/bin/bash
# round 6: NMFMU_STAGE_DMA_NOP2 (beta == 1, padded rank 256: the transposed images are not kept) -- rank-256 tests, then the shard
# bench and the W half-step timeline with and without (TORCHNMF_AMD_NO_P2), interleaved twice
OUT=gpurun_out/r6z; mkdir -p $OUT
timeout 1200 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "rank256 or cfg5 or r256 or deterministic or transposed_images or sharded_path" 2>&1 | tail -4 | tee $OUT/tests_nop2.txt
for rep in 1 2; do
for v in 0 1; do
echo "--- TORCHNMF_AMD_NO_P2=$v" | tee -a $OUT/nop2.txt
TORCHNMF_AMD_NO_P2=$v timeout 300 python tools/sp_timeline.py --iters 4 2>&1 | grep -v amdgpu.ids | grep "W half-step\|epilogue" | head -2 | tee -a $OUT/nop2.txt
TORCHNMF_AMD_NO_P2=$v timeout 300 python bench.py --config cfg5 --steps 10 --cpu-iters 0 --no-sweep --no-parity-mode --repeats 3 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); r=d['roofline']
print('no_p2=$v cfg5: it/s', d['iters_per_s'], 'kernel_ms', r['avg_launch_ms'], 'w/h', r.get('avg_launch_ms_w_step'), r.get('avg_launch_ms_h_step'), 'frac', r['frac'])" | tee -a $OUT/nop2.txt
done; done
