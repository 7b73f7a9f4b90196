#!/bin/bash
# LDS counters of K1s (bank conflicts, LDS-array cycles) at one rank's share of configs[4]; whole kernel and K loop only
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
for ab in 0 4; do
  D=gpurun_out/pmc_gexp_lds$ab
  rm -rf $D
  MCD_GEMM_EXP_ABLATE=$ab rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL GRBM_GUI_ACTIVE -d $D -- python3 scripts/prof_gemm_exp.py 25000 10000 6 > /dev/null 2>&1
  echo "ablate=$ab"
  python3 scripts/pmc_db.py $D gemm_nt_bf16_exp
  rm -rf $D
done
