#!/bin/bash
# round 6: quick regression of selected tests
timeout 1500 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "rank256 or cfg5 or r256 or deterministic or transposed_images or sharded_path or riding or betamu_auto" 2>&1 | grep -E "passed|failed|^E |Error" | tail -8
