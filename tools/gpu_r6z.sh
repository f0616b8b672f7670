#!/bin/bash
# round 6: fused-apply epilogue of the rank-256 pipelined kernel with the master rows of 8 / 16 chunks in flight together
# (NMFMU_SP_EPI_BATCH) against one chunk at a time (=1, rounds 4-5): rank-256 tests, timeline and shard bench per variant
OUT=gpurun_out/r6z; mkdir -p $OUT
LIBD=$PWD/pytorch-nmf_amd/torchnmf_amd
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "rank256 or cfg5 or r256 or deterministic" 2>&1 | tail -3 | tee $OUT/tests.txt
for rep in 1 2; do
for v in _eb1 "" _eb16; do
echo "--- lib$v" | tee -a $OUT/ab2.txt
NMFMU_LIB=$LIBD/libnmfmu$v.so timeout 300 python tools/sp_timeline.py 2>&1 | grep -v amdgpu.ids | grep "W half-step\|epilogue\|a CU" | head -3 | tee -a $OUT/ab2.txt
NMFMU_LIB=$LIBD/libnmfmu$v.so timeout 300 python bench.py --config cfg5 --steps 10 --cpu-iters 0 --no-sweep --no-parity-mode --repeats 3 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); r=d['roofline']
print('lib$v cfg5: it/s', d['iters_per_s'], 'kernel_ms', r['avg_launch_ms'], 'w/h', r.get('avg_launch_ms_w_step'), r.get('avg_launch_ms_h_step'), 'frac', r['frac'])" | tee -a $OUT/ab2.txt
done; done
