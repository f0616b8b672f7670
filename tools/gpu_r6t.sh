#!/bin/bash
# round 6: smoke() with the f16r / riding-loss case, and the f16r half-step tests with the padded-rank-64 shape
OUT=gpurun_out/r6t; mkdir -p $OUT
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; grep -v amdgpu.ids $OUT/smoke.log | tail -14
timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "f16r" 2>&1 | tail -3
