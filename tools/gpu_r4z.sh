#!/bin/bash
# round 4, NMF2D: window-operand H numerator, N-D implicit operands, split-K W numerator: focused tests + benches + trace
OUT=gpurun_out/r4z; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "h_numerator or several_shift or nmf2d or siplca or SIPLCA or fold_from_tile or fp16_operands" > $OUT/pytest.log 2>&1; echo "pytest rc=$?" ; tail -4 $OUT/pytest.log
show() { python - <<PY
import json
d=json.load(open("$1")); print("$2", d["config"]["precision"], "it/s", d["iters_per_s"], {k: v["avg_launch_ms"] for k, v in d["roofline"].get("per_gemm", {}).items()}, (d.get("parity") or {}).get("modes"))
PY
}
for prec in bf16 bf16x3 auto; do
  timeout 300 python bench.py --workload nmf2d --precision $prec --steps 30 --warmup 5 --cpu-iters $([ $prec = auto ] && echo 3 || echo 0) --telemetry-s 0 > $OUT/bench_nmf2d_$prec.json 2>$OUT/bench_$prec.err; show $OUT/bench_nmf2d_$prec.json "nmf2d $prec"
done
TORCHNMF_AMD_NMFD_EXPLICIT=1 timeout 300 python bench.py --workload nmf2d --precision bf16 --steps 30 --warmup 5 --cpu-iters 0 --telemetry-s 0 > $OUT/bench_nmf2d_explicit.json 2>>$OUT/bench_bf16.err; show $OUT/bench_nmf2d_explicit.json "nmf2d bf16 explicit"
timeout 300 python bench.py --workload nmfd --steps 30 --warmup 5 --cpu-iters 0 --no-parity-mode --telemetry-s 0 > $OUT/bench_nmfd.json 2>$OUT/bench_nmfd.err; show $OUT/bench_nmfd.json nmfd
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof -- python $GRAFT_REPO_ROOT/bench.py --workload nmf2d --precision bf16 --steps 20 --warmup 5 --cpu-iters 0 --telemetry-s 0 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
f=$(ls -t $(find $OUT/prof -name "*kernel_stats.csv") | head -1); cp $f $OUT/nmf2d_kernel_stats.csv
python - <<PY
import csv
for r in list(csv.DictReader(open("$OUT/nmf2d_kernel_stats.csv")))[:12]:
    print(r['Name'][:75], r['Calls'], round(float(r['AverageNs'])/1e3,1), r['Percentage'])
PY
