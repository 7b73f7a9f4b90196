#!/bin/bash
# K1s (v6): what a store costs the texture-address unit under different conditions (TA busy cycles per E store, kernel time under the
# counters): stagger off / on, cache-policy bits of the E stores, all stores into an L2-resident window, no stores.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
export MCD_LIB_PATH=$PWD/mammo-clip-dissect_amd/csrc/libmcd_hip_dev.so MCD_PROF_LIBRARY=0
O=gpurun_out/r05_store_exp.txt
: > $O
run() {  # label, env...
  D=gpurun_out/pmc_st; rm -rf $D
  env "${@:2}" rocprofv3 --kernel-trace --pmc TA_TA_BUSY_sum SQ_VMEM_TA_ADDR_FIFO_FULL GRBM_GUI_ACTIVE -d $D -- python3 scripts/prof_gemm_exp.py 25000 10000 6 > /dev/null 2>&1
  echo "== $1" >> $O
  python3 scripts/pmc_db.py $D gemm_nt_bf16_exp >> $O 2>&1
  rm -rf $D
}
run "product" X=1
run "stagger 0" MCD_GEMM_EXP_STAGGER=0
run "stagger 52000" MCD_GEMM_EXP_STAGGER=52000
run "nt stores" MCD_GEMM_EXP_STAUX=2
run "sc1 stores" MCD_GEMM_EXP_STAUX=16
run "sc0 stores" MCD_GEMM_EXP_STAUX=1
run "L2 window" MCD_GEMM_EXP_ABLATE=2
run "no stores" MCD_GEMM_EXP_ABLATE=1
cat $O
