#!/bin/bash
# round 5, beta = 2 path: its tests, the stream kernel's per-workgroup timeline (diagnostic build, if present),
# the bench line and the per-kernel times of one iteration.   bash tools/gpu_xbtest.sh
OUT=$PWD/gpurun_out/xbtest; mkdir -p $OUT; ROOT=$PWD
timeout 600 python -m pytest tests -q -m gpu -x -k "beta2 or gram or xb or apply" 2>&1 | tail -3 | cut -c1-200
[ -f pytorch-nmf_amd/torchnmf_amd/libnmfmu_dbg.so ] && NMFMU_LIB=$PWD/pytorch-nmf_amd/torchnmf_amd/libnmfmu_dbg.so timeout 100 python tools/xb_timeline.py f16 2>&1 | grep -v amdgpu.ids | grep "step rows\|prologue\|us (min"
for i in 1 2; do timeout 200 python bench.py --beta 2 --gram --no-sweep --cpu-iters 0 --no-parity-mode --repeats 3 2>/dev/null > $OUT/b$i.json; python tools/bench_brief.py $OUT/b$i.json 1; done
export TMPDIR=/tmp; cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python $ROOT/bench.py --beta 2 --gram --steps 20 --warmup 5 --cpu-iters 0 --repeats 1 --max-repeats 1 --preroll-s 0.1 --no-parity-mode --no-sweep --telemetry-s 0 > /dev/null 2> $OUT/trace.err
f=$(find $OUT/trace -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $OUT/kernel_stats.csv && head -8 $OUT/kernel_stats.csv | cut -c1-150
rm -rf $OUT/trace
