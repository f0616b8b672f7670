#!/bin/bash
# round 6: the 3-byte target on the ping-pong kernel -- packing test, then the bench with 'f16r' as the headline precision so that the
# in-kernel legs (clock stamps: cycles per tile, core clock) describe ITS launches
OUT=gpurun_out/r6l; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "f16r" 2>&1 | tail -3 | tee $OUT/tests.txt
timeout 900 python bench.py --precision f16r --no-sweep --cpu-iters 0 2>$OUT/bench.err | tee $OUT/bench_f16r.json | python -c "
import json,sys; d=json.load(sys.stdin)
print('it/s', d['iters_per_s'], 'roofline', json.dumps(d['roofline'])[:2500])"
tail -3 $OUT/bench.err
