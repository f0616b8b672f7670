#!/bin/bash
# round 6 evidence call: full GPU suite, smoke, the default bench (what the driver runs), configs[4]'s shard, rocprofv3 kernel
# stats per leg, FETCH_SIZE / WRITE_SIZE passes of the headline kernel and of the software-pipelined rank-256 kernel.
#   bash tools/gpu_r6_final.sh <tag>
TAG=${1:-r6final}; OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
ROOT=$PWD
timeout 2400 python -m pytest tests -q -m gpu > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; grep -h "passed\|failed" $OUT/pytest_gpu.log | tail -2 | cut -c1-250
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; grep -v amdgpu.ids $OUT/smoke.log | tail -3
timeout 1200 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; python tools/bench_brief.py $OUT/bench.json 2>&1 | tail -30
timeout 300 python bench.py --config cfg5 --steps 10 --cpu-iters 0 --no-sweep --no-parity-mode > $OUT/bench_cfg5_shard.json 2>> $OUT/bench.err; python tools/bench_brief.py $OUT/bench_cfg5_shard.json 1
timeout 300 python bench.py --force-dist --config cfg5 --steps 10 --cpu-iters 0 --no-sweep --no-parity-mode > $OUT/bench_cfg5_world1_rccl.json 2>> $OUT/bench.err; python tools/bench_brief.py $OUT/bench_cfg5_world1_rccl.json 1
timeout 300 python bench.py --workload betamu --cpu-iters 0 --no-sweep > $OUT/bench_betamu.json 2>> $OUT/bench.err; python tools/bench_brief.py $OUT/bench_betamu.json 1
export TMPDIR=/tmp
cd /tmp
prof() {  # name, bench args
  local name=$1; shift
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$name -o trace -- python $ROOT/bench.py --steps 20 --warmup 5 --cpu-iters 0 --repeats 1 --max-repeats 1 --preroll-s 0.1 --no-parity-mode --no-sweep --telemetry-s 0 "$@" > $OUT/${name}_trace_bench.json 2> $OUT/trace_$name.err
  f=$(find $OUT/trace_$name -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $OUT/${name}_kernel_stats.csv && head -4 $OUT/${name}_kernel_stats.csv | cut -c1-160
}
prof cfg1_f16
prof cfg1_f16r --precision f16r
prof cfg5_shard --config cfg5 --steps 6 --warmup 2
prof beta05 --beta 0.5
prof beta0 --beta 0
pmc() {  # name, bench args
  local name=$1; shift
  for grp in FETCH_SIZE WRITE_SIZE; do
    timeout 400 rocprofv3 --pmc $grp --output-format csv -d $OUT/pmc_$name/pmc_$grp -o pmc -- python $ROOT/bench.py --steps 20 --warmup 5 --cpu-iters 0 --repeats 1 --max-repeats 1 --preroll-s 0.1 --no-parity-mode --no-sweep --telemetry-s 0 "$@" > /dev/null 2> $OUT/pmc_${name}_$grp.err
    echo "pmc $name $grp rc=$?"
  done
  python $ROOT/tools/pmc_summary.py $OUT/pmc_$name > $OUT/${name}_pmc_summary.txt 2>&1; grep -A2 "pp_kernel\|sp_kernel\|sp2_kernel" $OUT/${name}_pmc_summary.txt | head -12
}
pmc cfg1_f16
pmc cfg1_f16r --precision f16r
pmc cfg5_shard --config cfg5 --steps 6 --warmup 2
find $OUT -name "*.db" -delete; find $OUT -size +8M -delete; find $OUT -type d -name "trace_*" -exec rm -rf {} + 2>/dev/null; find $OUT -type d -name "pmc_*" -exec rm -rf {} + 2>/dev/null
echo finished
