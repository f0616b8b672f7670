#!/bin/bash
# rocprofv3 kernel trace (+ optional PMC passes) of the bench command.  Usage: bash tools/gpu_prof.sh <tag> [pmc]
TAG=${1:-prof}
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
BENCH="python $PWD/bench.py --steps 20 --warmup 5 --cpu-iters 0 --repeats 1 --max-repeats 1 --preroll-s 0.1 --no-parity-mode --no-sweep ${BENCH_ARGS}"
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- $BENCH > $OUT/trace_bench.json 2> $OUT/trace.err
echo "trace rc=$?"
find $OUT/trace -name "*kernel_stats*" | head -3
f=$(find $OUT/trace -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp "$f" $OUT/kernel_stats.csv && head -20 $OUT/kernel_stats.csv
if [ "$2" = "pmc" ]; then
  for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum"; do
    name=$(echo $grp | cut -d' ' -f1)
    timeout 600 rocprofv3 --pmc $grp --output-format csv -d $OUT/pmc_$name -o pmc -- $BENCH > /dev/null 2> $OUT/pmc_$name.err
    echo "pmc $name rc=$?"
  done
  python $OLDPWD/tools/pmc_summary.py $OUT > $OUT/pmc_summary.txt 2>&1; cat $OUT/pmc_summary.txt
fi
# keep the merged-back payload small
find $OUT -name "*.db" -delete; find $OUT -size +8M -delete
