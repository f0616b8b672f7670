#!/bin/bash
# round 5, final build: rocprofv3 kernel stats of the headline and of the beta = 2 path (the apply kernel changed after the second evidence call)
TAG=${1:-r5v}; OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT; ROOT=$PWD
export TMPDIR=/tmp; cd /tmp
prof() {
  local name=$1; shift
  timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$name -o trace -- python $ROOT/bench.py --steps 20 --warmup 5 --cpu-iters 0 --repeats 1 --max-repeats 1 --preroll-s 0.1 --no-parity-mode --no-sweep --telemetry-s 0 "$@" > $OUT/${name}_trace_bench.json 2> $OUT/trace_$name.err
  f=$(find $OUT/trace_$name -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $OUT/${name}_kernel_stats.csv && head -7 $OUT/${name}_kernel_stats.csv | cut -c1-150
  rm -rf $OUT/trace_$name
}
prof cfg1_f16
prof beta2_gram --beta 2 --gram
