#!/bin/bash
# round 6: the full configs[4] shard test, then the whole GPU suite, then the default bench line and the cfg5 line
TAG=${1:-r6f}; OUT=gpurun_out/$TAG; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "cfg5_full_shard" 2>&1 | tail -8 | tee $OUT/pytest_fullshard.txt
timeout 2400 python -m pytest tests -q -m gpu -x 2>&1 | tail -8 | tee $OUT/pytest_all.txt
timeout 300 python bench.py --config cfg5 --steps 10 --cpu-iters 0 --no-parity-mode > $OUT/bench_cfg5.json 2>> $OUT/err.log
python tools/bench_brief.py $OUT/bench_cfg5.json 2>/dev/null | head -20
grep -v amdgpu.ids $OUT/err.log | tail -5
