#!/bin/bash
# NMFD tests + an interleaved A/B of one environment switch of the NMFD engine on the bench's nmfd workload.
# Usage: bash tools/gpu_nmfd_ab.sh <tag> <ENV_NAME> [pytest -k expression]
TAG=${1:-nmfd}; VAR=${2:-TORCHNMF_AMD_NMFD_RAGGED_IN_GRID}; KEXPR=${3:-nmfd}
OUT=gpurun_out/$TAG; mkdir -p $OUT
timeout 1200 python -m pytest tests -q -m gpu --timeout 600 -k "$KEXPR" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"
tail -3 $OUT/pytest.log; grep -E "^(FAILED|ERROR)" $OUT/pytest.log | head -20
for i in 1 2; do for v in 0 1; do
  env $VAR=$v timeout 300 python bench.py --workload nmfd --steps 50 --warmup 10 --cpu-iters 0 > $OUT/ab_${v}_$i.json 2>> $OUT/ab.err
  echo -n "[$VAR=$v #$i] "; python - $OUT/ab_${v}_$i.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
g = d.get('roofline', {}).get('per_gemm') or d.get('per_gemm') or {}
print(round(d.get('iters_per_s', 0) or 1000.0 / d['ms_per_step']), 'it/s', d['ms_per_step'], 'ms', {k: v.get('avg_launch_ms') for k, v in g.items()} if isinstance(g, dict) else '')
PY
done; done
