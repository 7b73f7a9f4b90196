#!/bin/bash
# row padding of K1s' bf16 operands (elements): L2 channel spread of the 16 rows x 64 B one LDS-DMA instruction fetches
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
for pad in 0 32 64 96 128 192 8; do
  for ab in 0 4; do
    MCD_GEMM_EXP_KPAD=$pad MCD_GEMM_EXP_ABLATE=$ab rocprofv3 --kernel-trace --stats -d gpurun_out/prof_kpad_${pad}_$ab -o x --output-format csv -- python3 scripts/prof_gemm_exp.py 25000 10000 20 > /dev/null 2>&1
    echo "KPAD=$pad ablate=$ab: $(grep gemm_nt_bf16_exp gpurun_out/prof_kpad_${pad}_$ab/x_kernel_stats.csv | sed 's/.*)",//' | cut -d, -f1-3)"
  done
done
