#!/bin/bash
# round 3, call d: parity subset for the pipelined four-wave loop, A/B SP2 on/off over the beta sweep, ablation control
OUT=gpurun_out/r3d; mkdir -p $OUT
rm -f gpurun_out/parity_measured.jsonl
timeout 900 python -m pytest tests -q -m gpu --timeout 600 -k "half_steps or every_beta or scaled_terms or rank128 or rank_above or fuzz_dense or random_dense or g1_golden or g3 or g4 or shapes_and_ksplit or cfg2 or betamu_g7 or beta_trainer or sharded or sparse_equals_dense_generic" > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"
tail -3 $OUT/pytest_gpu.log; grep -E "^(FAILED|ERROR)" $OUT/pytest_gpu.log | head -20
for i in 1 2; do
  for cfg in "--beta 2" "--beta 0.5" "--beta 0" "--beta 2 --precision bf16" "--beta 2 --rank 64"; do
    for v in "" _nosp; do
      lib=$PWD/pytorch-nmf_amd/torchnmf_amd/libnmfmu$v.so
      f=$OUT/v_${v}_$(echo $cfg | tr -d ' -')_$i.json
      NMFMU_LIB=$lib timeout 300 python bench.py --steps 50 --warmup 10 --cpu-iters 0 --no-sweep --no-parity-mode $cfg > $f 2>> $OUT/v.err
      echo -n "[${v:-base} $cfg #$i] "; python tools/bench_brief.py $f short
    done
  done
done
for v in abl2 abl; do NMFMU_LIB=$PWD/pytorch-nmf_amd/torchnmf_amd/libnmfmu_$v.so timeout 200 python tools/pp_timeline.py f16 > $OUT/tl_$v.txt 2>&1; grep "cycles/tile" $OUT/tl_$v.txt; done
for v in abl2 abl dbg; do NMFMU_LIB=$PWD/pytorch-nmf_amd/torchnmf_amd/libnmfmu_$v.so python bench.py --cpu-iters 0 --no-sweep --steps 50 > $OUT/bench_$v.json 2>/dev/null; echo -n "[$v] "; python tools/bench_brief.py $OUT/bench_$v.json short; done
