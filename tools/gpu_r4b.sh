#!/bin/bash
# round 4: the default bench line once more (driver's command), brief + a copy for profiles/
OUT=gpurun_out/r4b; mkdir -p $OUT
timeout 900 python bench.py > $OUT/bench_default.json 2> $OUT/bench.err; echo "bench rc=$?"
python tools/bench_brief.py $OUT/bench_default.json | cut -c1-420
tail -3 $OUT/bench.err
