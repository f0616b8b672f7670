#!/bin/bash
# round 6: modules cast to another dtype (m.double(), m.bfloat16())
timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "another_dtype or g3_early or g4_frozen" 2>&1 | tail -25
