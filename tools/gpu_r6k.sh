#!/bin/bash
# round 6: first GPU run of the 3-byte target 'f16r' -- its own tests, the auto-policy tests whose expectation moved to it, the
# determinism test of the pipelined kernels, then the default bench line (real_data_mode leg: auto must pick 'f16r')
OUT=gpurun_out/r6k; mkdir -p $OUT
timeout 1500 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "f16r or auto_precision or auto_f16x or betamu_auto or unrounded_target or betamu_conv or cfg1_full" 2>&1 | tail -15 | tee $OUT/tests.txt
timeout 900 python bench.py 2>$OUT/bench.err | tee $OUT/bench.json | python -c "
import json,sys; d=json.load(sys.stdin)
print('value', d['value'], 'frac', d['roofline']['frac'])
print('real_data_mode', json.dumps(d.get('real_data_mode'))[:1500])"
tail -5 $OUT/bench.err
