#!/bin/bash
OUT=gpurun_out/r3g; mkdir -p $OUT
rm -f gpurun_out/parity_measured.jsonl
timeout 1200 python -m pytest tests -q -m gpu --timeout 900 > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"
tail -3 $OUT/pytest_gpu.log; grep -E "^(FAILED|ERROR)" $OUT/pytest_gpu.log | head -20
cp gpurun_out/parity_measured.jsonl $OUT/ 2>/dev/null
for i in 1 2; do
  for cfg in "--beta 2" "--beta 0.5" "--beta 0" "--beta 0.5 --rank 64" "--config cfg5 --steps 10 --warmup 3" "--beta 2 --precision bf16"; do
    for v in "" _noil; do
      lib=$PWD/pytorch-nmf_amd/torchnmf_amd/libnmfmu$v.so
      f=$OUT/v_${v}_$(echo $cfg | tr -d ' -')_$i.json
      NMFMU_LIB=$lib timeout 300 python bench.py --steps 50 --warmup 10 --cpu-iters 0 --no-sweep --no-parity-mode $cfg > $f 2>> $OUT/v.err
      echo -n "[${v:-base} $cfg #$i] "; python tools/bench_brief.py $f short
    done
  done
done
