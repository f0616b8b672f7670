#!/bin/bash
# round 4, evidence call: full GPU suite, smoke, the default bench line, rocprofv3 kernel stats of its legs (copy what
# should be judged from gpurun_out/<tag>/ into profiles/)
TAG=${1:-r4z}; OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
timeout 1500 python -m pytest tests -q -m gpu --durations=12 > $OUT/pytest.log 2>&1
echo "pytest rc=$?"; tail -20 $OUT/pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -12 $OUT/smoke.log
timeout 900 python bench.py > $OUT/bench_default.json 2>> $OUT/bench.err; echo "bench rc=$?"
python tools/bench_brief.py $OUT/bench_default.json | cut -c1-400
export TMPDIR=/tmp
prof() {  # tag, bench args
  local t=$1; shift
  ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$t -o trace -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --cpu-iters 0 --repeats 1 --max-repeats 1 --preroll-s 0.1 --no-parity-mode --no-sweep --telemetry-s 0 "$@" > $OUT/${t}_trace_bench.json 2> $OUT/trace_$t.err )
  f=$(find $OUT/trace_$t -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $OUT/${t}_kernel_stats.csv && head -5 $OUT/${t}_kernel_stats.csv | cut -c1-160
  rm -rf $OUT/trace_$t
}
prof cfg1_f16
prof cfg1_f16x --precision f16x
prof beta2_gram --beta 2 --gram
prof beta2 --beta 2
prof beta0.5 --beta 0.5
prof nmfd --workload nmfd
prof nmf2d --workload nmf2d --precision auto
# HBM traffic of the streaming kernels (FETCH_SIZE x 2 gfx950 correction + WRITE_SIZE), separate --pmc passes
for grp in FETCH_SIZE WRITE_SIZE; do
  ( cd /tmp && timeout 600 rocprofv3 --pmc $grp --output-format csv -d $OUT/pmc_gram_$grp -o pmc -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --cpu-iters 0 --repeats 1 --max-repeats 1 --preroll-s 0.05 --no-parity-mode --no-sweep --telemetry-s 0 --no-roofline --beta 2 --gram > /dev/null 2> $OUT/pmc_gram_$grp.err )
done
python tools/pmc_summary.py $OUT > $OUT/pmc_summary.txt 2>&1; grep -A3 "fused_kernel<128, 1, 2, 3>" $OUT/pmc_summary.txt | head -12
find $OUT -name "*.db" -delete; find $OUT -size +8M -delete
