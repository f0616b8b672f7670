#!/bin/bash
# round 6, first call: the joule budget of the ping-pong kernel by result-preserving DUPLICATION.
# Builds (all with the clock stamps): make VARIANT=_dbg EXTRA=-DNMFMU_DEBUG_HOOKS VUNITS="nmfmu_capi nmfmu_inst_pp", and
# _dup{1,2,4,8,16,32} with -DNMFMU_PP_DUP=<bit> on top.  Per variant, interleaved twice: the headline bench leg (launch ms,
# amdsmi clock and power) and the in-kernel cycles per tile / core clock (tools/pp_timeline.py).
TAG=${1:-r6a}; OUT=gpurun_out/$TAG; mkdir -p $OUT
LIBD=$PWD/pytorch-nmf_amd/torchnmf_amd
tools/ubench/dma_off_probe > $OUT/dma_off_probe.txt 2>&1; cat $OUT/dma_off_probe.txt
for i in 1 2; do
  for v in "" _dbg _dup1 _dup2 _dup4 _dup8 _dup16 _dup32; do
    [ -f $LIBD/libnmfmu$v.so ] || { echo "missing $v"; continue; }
    NMFMU_LIB=$LIBD/libnmfmu$v.so timeout 300 python bench.py --steps 40 --warmup 10 --cpu-iters 0 --no-sweep --no-parity-mode --repeats 3 > $OUT/b${v}_$i.json 2>> $OUT/err.log
    python - <<PY
import json
try:
    d=json.load(open("$OUT/b${v}_$i.json")); r=d["roofline"]
    print("[lib%-7s #$i] it/s=%7.1f kernel_ms=%.4f (w %.4f h %.4f) frac=%.4f clock=%s power=%s" % ("$v", d["iters_per_s"], r["avg_launch_ms"], r["avg_launch_ms_w_step"], r["avg_launch_ms_h_step"], r["frac"], r.get("clock_mhz"), r.get("power_w")))
except Exception as e: print("[$v] FAILED", e)
PY
    if [ -n "$v" ]; then
      NMFMU_LIB=$LIBD/libnmfmu$v.so PP_STEPS=h timeout 200 python tools/pp_timeline.py f16 2>> $OUT/err.log | tail -1 | tee -a $OUT/timeline${v}.txt
    fi
  done
done
# baselines of the kernels this round works on (shipped library): configs[4]'s shard, beta sweep legs
for args in "--config cfg5 --steps 10" "--beta 0.5" "--beta 0"; do
  tag=$(echo $args | tr -d ' -' | tr '.' 'p')
  timeout 300 python bench.py $args --no-sweep --cpu-iters 0 --no-parity-mode --repeats 3 > $OUT/base_$tag.json 2>> $OUT/err.log
  python - <<PY
import json
try:
    d=json.load(open("$OUT/base_$tag.json")); r=d["roofline"]
    print("[base %-24s] it/s=%7.1f kernel_ms=%.4f (w %.4f h %.4f) frac=%.4f clock=%s power=%s ceil=%s" % ("$args", d["iters_per_s"], r["avg_launch_ms"], r["avg_launch_ms_w_step"], r["avg_launch_ms_h_step"], r["frac"], r.get("clock_mhz"), r.get("power_w"), r.get("ceiling_tflops")))
except Exception as e: print("[$args] FAILED", e)
PY
done
tail -3 $OUT/err.log
