#!/bin/bash
OUT=gpurun_out/r3i; mkdir -p $OUT
rm -f gpurun_out/parity_measured.jsonl
timeout 1200 python -m pytest tests -q -m gpu --timeout 900 > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"
tail -3 $OUT/pytest_gpu.log; grep -E "^(FAILED|ERROR)" $OUT/pytest_gpu.log | head -20
cp gpurun_out/parity_measured.jsonl $OUT/ 2>/dev/null
for i in 1 2 3; do for m in 1 0; do TORCHNMF_AMD_NMFD_TAIL_SPLIT=$m timeout 200 python bench.py --workload nmfd --steps 50 --warmup 10 --cpu-iters 0 > $OUT/nmfd_${m}_$i.json 2>/dev/null; echo -n "[nmfd tail=$m #$i] "; python -c "
import json,sys; d=json.load(open('$OUT/nmfd_${m}_$i.json')); print(d['iters_per_s'], d['ms_per_step'], {k:v['avg_launch_ms'] for k,v in d['roofline']['per_gemm'].items()})"; done; done
for i in 1 2; do for cfg in "--beta 2" "--beta 0.5" "--beta 0" "--beta 3"; do f=$OUT/b_$(echo $cfg | tr -d ' -')_$i.json; timeout 300 python bench.py --steps 50 --warmup 10 --cpu-iters 0 --no-sweep --no-parity-mode $cfg > $f 2>> $OUT/v.err; echo -n "[$cfg #$i] "; python tools/bench_brief.py $f short; done; done
