#!/bin/bash
# round 6: quick regression of the loss-checkpoint paths
timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "riding or fused_loss or early_stop or another_dtype or other_floating" 2>&1 | tail -6
