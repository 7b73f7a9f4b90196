#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
for cfg in "192 1" "192 0" "256 0"; do
  set -- $cfg
  for ab in 0 4; do
    MCD_GEMM_EXP_TM=$1 MCD_GEMM_EXP_PIPE=$2 MCD_GEMM_EXP_ABLATE=$ab rocprofv3 --kernel-trace --stats -d gpurun_out/prof_pipe_$1_$2_$ab -o x --output-format csv -- python3 scripts/prof_gemm_exp.py 25000 10000 20 > /dev/null 2>&1
    echo "TM=$1 PIPE=$2 ablate=$ab: $(grep gemm_nt_bf16_exp gpurun_out/prof_pipe_$1_$2_$ab/x_kernel_stats.csv | sed 's/.*)",//' | cut -d, -f1-3)"
  done
done
