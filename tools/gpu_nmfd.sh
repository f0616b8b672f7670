#!/bin/bash
OUT=gpurun_out/${1:-nmfd}; mkdir -p $OUT
timeout 900 python -m pytest tests -q -m gpu --tb=short -p no:cacheprovider -k "nmfd" > $OUT/pytest_nmfd.log 2>&1; echo "pytest rc=$?"
tail -30 $OUT/pytest_nmfd.log
