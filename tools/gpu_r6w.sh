#!/bin/bash
# round 6: quick regression of one or two tests
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "sharded_path_world1_rccl" 2>&1 | tail -12
