#!/bin/bash
# round 5, evidence call: full GPU suite, smoke, the default bench (what the driver runs), rocprofv3 kernel stats of its legs,
# HBM-traffic PMC passes of the two streaming kernels, the secondary workloads.   bash tools/gpu_r5_final.sh <tag>
TAG=${1:-r5final}; OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
ROOT=$PWD
timeout 1500 python -m pytest tests -q -m gpu > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -4 $OUT/pytest_gpu.log | cut -c1-250
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; grep -v amdgpu.ids $OUT/smoke.log | tail -3
timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; python tools/bench_brief.py $OUT/bench.json 2>&1 | tail -30
# secondary workloads (bench lines only)
timeout 300 python bench.py --workload betamu --cpu-iters 0 --no-sweep > $OUT/bench_betamu.json 2>> $OUT/bench.err; python tools/bench_brief.py $OUT/bench_betamu.json 1
timeout 300 python bench.py --workload plca --precision bf16x3 --cpu-iters 0 > $OUT/bench_plca_bf16x3.json 2>> $OUT/bench.err; python -c "import json;d=json.load(open('$OUT/bench_plca_bf16x3.json'));print('plca bf16x3', d['iters_per_s'], 'EM it/s')"
timeout 300 python bench.py --config cfg5 --steps 10 --cpu-iters 0 --no-sweep --no-parity-mode > $OUT/bench_cfg5_shard.json 2>> $OUT/bench.err; python tools/bench_brief.py $OUT/bench_cfg5_shard.json 1
timeout 300 python bench.py --force-dist --config cfg5 --steps 10 --cpu-iters 0 --no-sweep --no-parity-mode > $OUT/bench_cfg5_world1_rccl.json 2>> $OUT/bench.err; python tools/bench_brief.py $OUT/bench_cfg5_world1_rccl.json 1
# rocprofv3: kernel stats per leg, then the PMC traffic passes (separate runs, MI355X_MICROARCH.md)
export TMPDIR=/tmp
cd /tmp
prof() {  # name, bench args
  local name=$1; shift
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$name -o trace -- python $ROOT/bench.py --steps 20 --warmup 5 --cpu-iters 0 --repeats 1 --max-repeats 1 --preroll-s 0.1 --no-parity-mode --no-sweep --telemetry-s 0 "$@" > $OUT/${name}_trace_bench.json 2> $OUT/trace_$name.err
  f=$(find $OUT/trace_$name -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $OUT/${name}_kernel_stats.csv && head -4 $OUT/${name}_kernel_stats.csv | cut -c1-160
}
prof cfg1_f16
prof nmfd --workload nmfd
prof nmf2d --workload nmf2d --precision auto
prof beta2_gram --beta 2 --gram
pmc() {  # name, bench args
  local name=$1; shift
  for grp in FETCH_SIZE WRITE_SIZE; do
    timeout 400 rocprofv3 --pmc $grp --output-format csv -d $OUT/pmc_$name/pmc_$grp -o pmc -- python $ROOT/bench.py --steps 20 --warmup 5 --cpu-iters 0 --repeats 1 --max-repeats 1 --preroll-s 0.1 --no-parity-mode --no-sweep --telemetry-s 0 "$@" > /dev/null 2> $OUT/pmc_${name}_$grp.err
    echo "pmc $name $grp rc=$?"
  done
  python $ROOT/tools/pmc_summary.py $OUT/pmc_$name > $OUT/${name}_pmc_summary.txt 2>&1; grep -A2 "pp_kernel\|fused_kernel" $OUT/${name}_pmc_summary.txt | head -12
}
pmc cfg1_f16
pmc beta2_gram --beta 2 --gram
# NMFD GEMMs: where the waves' cycles go (the k loop as a chain of latencies)
for grp in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum FETCH_SIZE"; do
  name=$(echo $grp | cut -d' ' -f1)
  timeout 400 rocprofv3 --pmc $grp --output-format csv -d $OUT/pmc_nmfd/pmc_$name -o pmc -- python $ROOT/bench.py --workload nmfd --steps 20 --warmup 5 --cpu-iters 0 --repeats 1 --max-repeats 1 --preroll-s 0.1 --no-parity-mode --telemetry-s 0 > /dev/null 2> $OUT/pmc_nmfd_$name.err
  echo "pmc nmfd $name rc=$?"
done
python $ROOT/tools/pmc_summary.py $OUT/pmc_nmfd > $OUT/nmfd_pmc_summary.txt 2>&1; grep -A12 "nt_gemm_kernel" $OUT/nmfd_pmc_summary.txt | head -30
find $OUT -name "*.db" -delete; find $OUT -size +8M -delete; find $OUT -type d -name "trace_*" -exec rm -rf {} + 2>/dev/null; find $OUT -type d -name "pmc_*" -exec rm -rf {} + 2>/dev/null
echo finished
