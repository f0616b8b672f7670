#!/bin/bash
# round 6: the riding loss (nmfmu_mu_step_with_loss) -- tests, then fit()'s whole-call time with and without it
OUT=gpurun_out/r6p; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "riding or fused_loss_checkpoint or early_stop" 2>&1 | tail -12 | tee $OUT/tests.txt

