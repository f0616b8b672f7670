#!/bin/bash
# round 6: one-pass target statistics (admission test of 'auto' + the riding loss's sums) -- tests, then fit()'s whole call
OUT=gpurun_out/r6q; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "target_stats or riding or auto_precision or auto_f16x or betamu_auto" 2>&1 | tail -30 | tee $OUT/tests.txt

