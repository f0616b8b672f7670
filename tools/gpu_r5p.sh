#!/bin/bash
# round 5: LDS bank-conflict scan (SQ_LDS_BANK_CONFLICT against SQ_LDS_IDX_ACTIVE) over every kernel the bench legs launch --
# the check that found the rank-256 swizzle.   bash tools/gpu_r5p.sh <tag>
TAG=${1:-r5p}; OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
ROOT=$PWD
export TMPDIR=/tmp
cd /tmp
COMMON="--steps 8 --warmup 3 --cpu-iters 0 --repeats 1 --max-repeats 1 --preroll-s 0.1 --telemetry-s 0"
scan() {  # name, bench args
  local name=$1; shift
  timeout 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES --output-format csv -d $OUT/pmc_$name/pmc_lds -o pmc -- python $ROOT/bench.py $COMMON "$@" > /dev/null 2> $OUT/pmc_$name.err
  echo "scan $name rc=$?"
  python $ROOT/tools/pmc_summary.py $OUT/pmc_$name > $OUT/${name}_lds_pmc_summary.txt 2>&1
}
scan default
scan beta2_gram --beta 2 --gram --no-sweep --no-parity-mode
scan bf16x3 --precision bf16x3 --no-sweep --no-parity-mode
scan r32 --rank 32 --cols 16384 --no-sweep --no-parity-mode
scan r64 --rank 64 --cols 16384 --no-sweep --no-parity-mode
scan r64_kl --rank 64 --cols 16384 --beta 0.5 --no-sweep --no-parity-mode
scan plca --workload plca --precision bf16x3
scan sparse --workload sparse
scan betamu --workload betamu --no-sweep
python - <<PY
import glob, re
for f in sorted(glob.glob("$OUT/*_lds_pmc_summary.txt")):
    name = None; vals = {}
    def flush():
        if name and vals.get('SQ_LDS_IDX_ACTIVE', 0) > 0:
            c, a = vals.get('SQ_LDS_BANK_CONFLICT', 0), vals['SQ_LDS_IDX_ACTIVE']
            if c / a > 0.05: print("%-14s %5.1f%%  conflict=%.3g active=%.3g  %s" % (f.split('/')[-1][:14], 100 * c / a, c, a, name[:110]))
    for ln in open(f):
        m = re.match(r"\s+(SQ_\w+)\s+n=\s*\d+\s+mean=(\S+)", ln)
        if m: vals[m.group(1)] = float(m.group(2))
        elif ln.strip() and not ln.startswith(' '):
            flush(); name = ln.strip(); vals = {}
    flush()
PY
find $OUT -name "*.db" -delete; find $OUT -size +8M -delete; find $OUT -type d -name "pmc_*" -exec rm -rf {} + 2>/dev/null
echo finished
