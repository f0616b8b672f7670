#!/bin/bash
# round 5: padded rank 256 after the P1 swizzle change (nmfmu_layout.h: p1_swz at 32 slots per row) -- the tests that run the
# rank-256 kernels, the configs[4] shard bench, and the SQ LDS counters of its kernel.   bash tools/gpu_r5o.sh <tag>
TAG=${1:-r5o}; OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
ROOT=$PWD
timeout 900 python -m pytest tests -q -m gpu -x -k "rank256 or cfg5 or sharded or half_step or fuzz or rank_sweep or r256 or 256" > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -4 $OUT/pytest_gpu.log | cut -c1-250
for i in 1 2; do
timeout 300 python bench.py --config cfg5 --steps 10 --cpu-iters 0 --no-sweep --no-parity-mode > $OUT/bench_cfg5_shard_$i.json 2>> $OUT/bench.err; python tools/bench_brief.py $OUT/bench_cfg5_shard_$i.json 1
done
timeout 300 python bench.py --cpu-iters 0 --no-sweep --no-parity-mode > $OUT/bench_cfg1.json 2>> $OUT/bench.err; python tools/bench_brief.py $OUT/bench_cfg1.json 1
export TMPDIR=/tmp
cd /tmp
for grp in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  name=$(echo $grp | cut -d' ' -f1)
  timeout 400 rocprofv3 --pmc $grp --output-format csv -d $OUT/pmc_cfg5/pmc_$name -o pmc -- python $ROOT/bench.py --config cfg5 --steps 10 --warmup 3 --cpu-iters 0 --repeats 1 --max-repeats 1 --preroll-s 0.1 --no-parity-mode --no-sweep --telemetry-s 0 > /dev/null 2> $OUT/pmc_cfg5_$name.err
  echo "pmc cfg5 $name rc=$?"
done
python $ROOT/tools/pmc_summary.py $OUT/pmc_cfg5 > $OUT/cfg5_sq_pmc_summary.txt 2>&1; grep -A12 "fused_kernel" $OUT/cfg5_sq_pmc_summary.txt | head -30
find $OUT -name "*.db" -delete; find $OUT -size +8M -delete; find $OUT -type d -name "pmc_*" -exec rm -rf {} + 2>/dev/null
echo finished
