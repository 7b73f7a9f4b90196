#!/bin/bash
# What do K1s' stores cost the memory path?  PMC of the default (w4, folded) kernel: product against no-stores (ABLATE 1).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03_gexp_stores_pmc.txt
: > $O
for ab in 0 1; do
  echo "== ablate $ab" >> $O
  for grp in "TA_BUSY_avr TA_TA_BUSY_sum" "TCP_TCC_WRITE_REQ_sum TCP_TCC_READ_REQ_sum" "TCP_PENDING_STALL_CYCLES_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_STALL_sum" "TCC_EA0_RDREQ_sum" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_WRITE_sum" "FETCH_SIZE" "WRITE_SIZE"; do
    D=gpurun_out/pmc_st; rm -rf $D
    MCD_GEMM_EXP_ABLATE=$ab rocprofv3 --kernel-trace --pmc $grp -d $D -- python3 scripts/prof_gemm_exp.py 25000 10000 6 > /dev/null 2>&1 || echo "   ($grp: not collected)" >> $O
    python3 scripts/pmc_db.py $D gemm_nt_bf16_exp 2>/dev/null | grep -v "^gpurun_out" >> $O
    rm -rf $D
  done
done
cat $O
