#!/bin/bash
# round 4: engine-level timings of the convolutive models outside the bench shapes (NMF3D; NMFD with a short kernel),
# round-4 paths vs the paths of rounds 1-3 (env switches)
mkdir -p gpurun_out/r4n
for t in ${TOOLS:-nmf3d_time nmfd_short_time}; do
  timeout 200 python tools/$t.py > gpurun_out/r4n/${t}_new.json 2> gpurun_out/r4n/err.txt; cat gpurun_out/r4n/${t}_new.json
  TORCHNMF_AMD_NMFD_EXPLICIT=$([ $t = nmf3d_time ] && echo 1 || echo 0) TORCHNMF_AMD_NMFD_H_ROWS=0 TORCHNMF_AMD_NMFD_KSPLIT=0 PRECISIONS=bf16x3 STEPS=8 timeout 200 python tools/$t.py > gpurun_out/r4n/${t}_old.json 2>> gpurun_out/r4n/err.txt; cat gpurun_out/r4n/${t}_old.json
done
