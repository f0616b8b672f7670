#!/bin/bash
# round-end evidence in one gpurun call: kernel traces + PMC passes of the headline and of NMFD, un-profiled bench lines,
# the ingest microbenchmark.  Usage: bash tools/gpu_evidence.sh <tag>; copy what is judged into profiles/.
TAG=${1:-ev}; OUT=gpurun_out/$TAG; mkdir -p $OUT
bash tools/gpu_prof.sh ${TAG}/cfg1 pmc > $OUT/cfg1_prof.log 2>&1
BENCH_ARGS="--workload nmfd" bash tools/gpu_prof.sh ${TAG}/nmfd pmc > $OUT/nmfd_prof.log 2>&1
BENCH_ARGS="--precision f16" bash tools/gpu_prof.sh ${TAG}/cfg1_f16 > $OUT/cfg1_f16_prof.log 2>&1
timeout 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
timeout 300 python bench.py --workload nmfd --cpu-iters 1 > $OUT/bench_nmfd.json 2>> $OUT/bench.err; echo "nmfd rc=$?"
timeout 100 tools/ubench/dma_bw > $OUT/dma_bw.txt 2>&1
python - <<PY
import json
for f in ("bench", "bench_nmfd"):
    d = json.load(open("$OUT/%s.json" % f)); print(f, d.get("iters_per_s"), d.get("ms_per_step"), (d.get("roofline") or {}).get("frac"))
PY
