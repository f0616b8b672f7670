#!/bin/bash
# round 6: non-temporal stores in the ping-pong kernel's fused-apply epilogue (1: master rows, 3: + row-major image): timeline
# of the W half-step and the headline bench per variant, interleaved twice
OUT=gpurun_out/r6z; mkdir -p $OUT
LIBD=$PWD/pytorch-nmf_amd/torchnmf_amd
for rep in 1 2; do
for v in "" _nt1 _nt3; do
echo "--- lib$v" | tee -a $OUT/pp_nt.txt
NMFMU_LIB=$LIBD/libnmfmu$v.so timeout 300 python tools/sp_timeline.py --rows 4096 --cols 65536 --rank 128 --iters 6 2>&1 | grep -v amdgpu.ids | grep "epilogue" | head -1 | tee -a $OUT/pp_nt.txt
NMFMU_LIB=$LIBD/libnmfmu$v.so timeout 300 python bench.py --steps 30 --no-sweep --cpu-iters 0 --no-parity-mode --repeats 3 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); r=d['roofline']
print('lib$v f16: it/s', d['iters_per_s'], 'kernel_ms', r['avg_launch_ms'], 'w/h', r.get('avg_launch_ms_w_step'), r.get('avg_launch_ms_h_step'))" | tee -a $OUT/pp_nt.txt
done; done
