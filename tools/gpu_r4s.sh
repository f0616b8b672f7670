mkdir -p gpurun_out/r4s
python tools/siplca2_time.py > gpurun_out/r4s/siplca2_new.json 2> gpurun_out/r4s/err_new.txt; cat gpurun_out/r4s/siplca2_new.json
PRECISION=bf16 python tools/siplca2_time.py > gpurun_out/r4s/siplca2_new_bf16.json 2>> gpurun_out/r4s/err_new.txt; cat gpurun_out/r4s/siplca2_new_bf16.json
TORCHNMF_AMD_NMFD_EXPLICIT=1 TORCHNMF_AMD_NMFD_H_ROWS=0 TORCHNMF_AMD_NMFD_KSPLIT=0 STEPS=10 python tools/siplca2_time.py > gpurun_out/r4s/siplca2_old.json 2> gpurun_out/r4s/err_old.txt; cat gpurun_out/r4s/siplca2_old.json
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
tail -3 gpurun_out/r4s/err_new.txt
