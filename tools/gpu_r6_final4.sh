#!/bin/bash
# round 6, after NMFMU_STAGE_DMA_NOP2: full GPU suite, smoke, and the configs[4] shard evidence again (bench lines, rocprofv3 kernel
# stats, FETCH_SIZE / WRITE_SIZE passes of the rank-256 kernel); the configs[1] evidence of gpu_r6_final.sh stays valid (same sources)
TAG=${1:-r6final4}; OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
ROOT=$PWD
timeout 2400 python -m pytest tests -q -m gpu > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; grep -h "passed\|failed" $OUT/pytest_gpu.log | tail -2 | cut -c1-250
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; grep -v amdgpu.ids $OUT/smoke.log | tail -2
timeout 300 python bench.py --config cfg5 --steps 10 --cpu-iters 0 --no-sweep --no-parity-mode > $OUT/bench_cfg5_shard.json 2>> $OUT/bench.err; python tools/bench_brief.py $OUT/bench_cfg5_shard.json 1
timeout 300 python bench.py --force-dist --config cfg5 --steps 10 --cpu-iters 0 --no-sweep --no-parity-mode > $OUT/bench_cfg5_world1_rccl.json 2>> $OUT/bench.err; python tools/bench_brief.py $OUT/bench_cfg5_world1_rccl.json 1
export TMPDIR=/tmp
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_cfg5 -o trace -- python $ROOT/bench.py --config cfg5 --steps 6 --warmup 2 --cpu-iters 0 --repeats 1 --max-repeats 1 --preroll-s 0.1 --no-parity-mode --no-sweep --telemetry-s 0 > $OUT/cfg5_shard_trace_bench.json 2> $OUT/trace_cfg5.err
f=$(find $OUT/trace_cfg5 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $OUT/cfg5_shard_kernel_stats.csv && head -3 $OUT/cfg5_shard_kernel_stats.csv | cut -c1-160
for grp in FETCH_SIZE WRITE_SIZE; do
  timeout 400 rocprofv3 --pmc $grp --output-format csv -d $OUT/pmc_cfg5/pmc_$grp -o pmc -- python $ROOT/bench.py --config cfg5 --steps 6 --warmup 2 --cpu-iters 0 --repeats 1 --max-repeats 1 --preroll-s 0.1 --no-parity-mode --no-sweep --telemetry-s 0 > /dev/null 2> $OUT/pmc_cfg5_$grp.err
  echo "pmc cfg5 $grp rc=$?"
done
python $ROOT/tools/pmc_summary.py $OUT/pmc_cfg5 > $OUT/cfg5_shard_pmc_summary.txt 2>&1; grep -A2 "sp_kernel" $OUT/cfg5_shard_pmc_summary.txt | head -4
find $OUT -name "*.db" -delete; find $OUT -size +8M -delete; find $OUT -type d -name "trace_*" -exec rm -rf {} + 2>/dev/null; find $OUT -type d -name "pmc_*" -exec rm -rf {} + 2>/dev/null
echo finished
