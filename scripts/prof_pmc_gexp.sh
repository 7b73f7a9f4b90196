#!/bin/bash
# PMC passes over K1s (gemm_nt_bf16_exp_kernel) at one rank's share of configs[4]; run on the GPU box from the repo root:
#   bash scripts/prof_pmc_gexp.sh            -> gpurun_out/r02_gemm_exp_pmc.txt + profiles-ready JSON
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
D=gpurun_out/pmc_gexp_final
rm -rf $D
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_LDS -d $D/sq -- python3 scripts/prof_gemm_exp.py 25000 10000 6 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE -d $D/tcc -- python3 scripts/prof_gemm_exp.py 25000 10000 6 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $D/fetch -- python3 scripts/prof_gemm_exp.py 25000 10000 6 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $D/write -- python3 scripts/prof_gemm_exp.py 25000 10000 6 > /dev/null 2>&1
python3 scripts/pmc_db.py $D gemm_nt_bf16_exp > gpurun_out/r02_gemm_exp_pmc.txt 2>&1
cat gpurun_out/r02_gemm_exp_pmc.txt
