#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
for cfg in "256 2" "256 1" "192 2"; do
  set -- $cfg
  for ab in 0 1 4; do
    MCD_GEMM_EXP_TM=$1 MCD_GEMM_EXP_SPB=$2 MCD_GEMM_EXP_ABLATE=$ab rocprofv3 --kernel-trace --stats -d gpurun_out/prof_spb_$1_$2_$ab -o x --output-format csv -- python3 scripts/prof_gemm_exp.py 25000 10000 20 > /dev/null 2>&1
    echo "TM=$1 SPB=$2 ablate=$ab: $(grep gemm_nt_bf16_exp gpurun_out/prof_spb_$1_$2_$ab/x_kernel_stats.csv | sed 's/.*)",//' | cut -d, -f1-3)"
  done
done
