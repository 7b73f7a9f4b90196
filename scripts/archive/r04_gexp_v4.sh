#!/bin/bash
# Round 4, K1s v4 (k_gexp_v4.inc): correctness on edge shapes, then kernel times (rocprofv3 --kernel-trace --stats) of the product
# and its ablations (MCD_GEMM_EXP_ABLATE: 1 no stores, 4 K loop only; MCD_GEMM_EXP_STAGGER=0: no late starts), round 3's w4
# beside them, then PMC passes of the product and of its K loop.   bash scripts/r04_gexp_v4.sh [nopmc]
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
export MCD_LIB_PATH=$PWD/mammo-clip-dissect_amd/csrc/libmcd_hip_dev.so   # the ablation kernels live in the dev build (make dev)
O=gpurun_out/r04_gexp_v4.txt
: > $O
timeout -k 10 300 python3 scripts/gexp_check.py v4 >> $O 2>&1 || { echo "check v4 FAILED" >> $O; tail -30 $O; exit 1; }
run() {   # layout ablate stagger-env
  D=gpurun_out/gexp_$1_$2_$3; rm -rf $D
  if [ "$3" = "-" ]; then unset MCD_GEMM_EXP_STAGGER; else export MCD_GEMM_EXP_STAGGER=$3; fi
  MCD_PROF_LIBRARY=0 MCD_GEMM_EXP_LAYOUT=$1 MCD_GEMM_EXP_ABLATE=$2 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $D -- python3 scripts/prof_gemm_exp.py 25000 10000 20 > $D.log 2>&1
  python3 - $D $1 $2 $3 >> $O <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "gemm_nt_bf16_exp" in r["Name"]:
            us = float(r["AverageNs"]) / 1e3
            print("layout %-3s ablate %-2s stagger %-6s %-42s calls %3s  avg %7.1f us  %6.0f TFLOP/s  %.3f of 2.5 PF" % (
                sys.argv[2], sys.argv[3], sys.argv[4], r["Name"].split("(anonymous namespace)::")[-1][:42], r["Calls"], us, 2 * 25000 * 10000 * 512 / us / 1e6, 2 * 25000 * 10000 * 512 / us / 1e6 / 2500))
PY
  grep "embed_gemm_exp" $D.log >> $O
  rm -rf $D
  unset MCD_GEMM_EXP_STAGGER
}
for ab in 0 1 4; do run v4 $ab -; done
for ab in 0 1; do run v4 $ab 0; done
run v4 0 13000; run v4 0 40000
run w4 0 -
cat $O
[ "$1" = nopmc ] && exit 0
for ab in 0 4; do
  D=gpurun_out/pmc_gexp_v4_$ab; rm -rf $D
  export MCD_PROF_LIBRARY=0 MCD_GEMM_EXP_LAYOUT=v4 MCD_GEMM_EXP_ABLATE=$ab
  rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_LDS -d $D/sq -- python3 scripts/prof_gemm_exp.py 25000 10000 6 > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE -d $D/tcc -- python3 scripts/prof_gemm_exp.py 25000 10000 6 > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc TA_TA_BUSY_sum SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_LEVEL_VMEM -d $D/ta -- python3 scripts/prof_gemm_exp.py 25000 10000 6 > /dev/null 2>&1
  if [ $ab = 0 ]; then
    rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $D/fetch -- python3 scripts/prof_gemm_exp.py 25000 10000 6 > /dev/null 2>&1
    rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $D/write -- python3 scripts/prof_gemm_exp.py 25000 10000 6 > /dev/null 2>&1
  fi
  echo "== PMC, v4, ablate $ab (mean per dispatch)" >> $O
  python3 scripts/pmc_db.py $D gemm_nt_bf16_exp >> $O 2>&1
  rm -rf $D
done
tail -60 $O
