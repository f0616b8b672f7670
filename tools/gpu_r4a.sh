#!/bin/bash
# round 4, call 1: amdsmi field probe, the Gray-code operand-port ablation of the ping-pong loop (VERDICT r3 item 1a:
# cycles per tile + core clock from the kernel's own stamps, bench it/s), interleaved with the shipped loop
TAG=${1:-r4a}; OUT=gpurun_out/$TAG; mkdir -p $OUT
T=pytorch-nmf_amd/torchnmf_amd
timeout 300 python tools/smi_probe.py > $OUT/smi_probe.txt 2>&1; tail -5 $OUT/smi_probe.txt | cut -c1-1500
for i in 1 2; do
  for v in _dbg _g1 _g2; do
    NMFMU_LIB=$PWD/$T/libnmfmu$v.so timeout 300 python tools/pp_timeline.py f16 > $OUT/timeline${v}_$i.txt 2>&1
    grep "cycles/tile" $OUT/timeline${v}_$i.txt
  done
done
BENCH_ARGS="--no-sweep --no-parity-mode --repeats 3" bash tools/gpu_variants.sh $TAG "" _g1 _g2
