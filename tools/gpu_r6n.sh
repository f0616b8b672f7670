#!/bin/bash
# round 6: the whole GPU suite + smoke (regression check between feature commits)
OUT=gpurun_out/r6n; mkdir -p $OUT
timeout 2400 python -m pytest tests -q -m gpu > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; grep -h "passed\|failed\|FAILED" $OUT/pytest_gpu.log | tail -8 | cut -c1-250
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; grep -v amdgpu.ids $OUT/smoke.log | tail -3
