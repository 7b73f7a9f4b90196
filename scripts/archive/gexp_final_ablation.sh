#!/bin/bash
# final ablation matrix of K1s for profiles/r02_gemm_exp_ablation.txt (run on the GPU box from the repo root)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/r02_gemm_exp_ablation2.txt
echo "gemm_nt_bf16_exp_kernel at 25000 x 10000 x 512, buffer-load LDS-DMA loaders: rocprofv3 --kernel-trace --stats, 22 launches each: calls,total_ns,avg_ns" > $OUT
echo "MCD_GEMM_EXP_ABLATE: 0 product; 1 no output stores; 2 no exp; 4 K loop only; 20 K loop only, every workgroup stages tile (0,0); 32 sc1 (write-through) stores" >> $OUT
for cfg in "256 5 0" "256 4 0" "192 5 1" "192 5 0"; do
  set -- $cfg
  for ab in 0 1 2 4 20 32; do
    MCD_GEMM_EXP_TM=$1 MCD_GEMM_EXP_STAGES=$2 MCD_GEMM_EXP_PIPE=$3 MCD_GEMM_EXP_ABLATE=$ab rocprofv3 --kernel-trace --stats -d gpurun_out/prof_fin_$1_$2_$3_$ab -o x --output-format csv -- python3 scripts/prof_gemm_exp.py 25000 10000 20 > /dev/null 2>&1
    echo "TM=$1 stages=$2 PIPE=$3 ablate=$ab: $(grep gemm_nt_bf16_exp gpurun_out/prof_fin_$1_$2_$3_$ab/x_kernel_stats.csv | sed 's/.*)",//' | cut -d, -f1-3)" >> $OUT
  done
done
echo "other kernels of the call (TM=256 stages=5 ablate=0):" >> $OUT
grep -v gemm_nt_bf16_exp gpurun_out/prof_fin_256_5_0_0/x_kernel_stats.csv | grep "anonymous" | sed 's/(anonymous namespace):://; s/(.*)"/"/' | cut -d, -f1-4 >> $OUT
cat $OUT
