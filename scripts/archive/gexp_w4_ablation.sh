#!/bin/bash
# K1s at 25 000 x 10 000 x 512: round 3's one-wave-per-SIMD kernel (w4) against round 2's (w12), product and ablations
# (MCD_GEMM_EXP_ABLATE: 1 no stores, 2 no exp, 4 K loop only), kernel times from rocprofv3 --kernel-trace --stats.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03_gexp_w4_ablation.txt
: > $O
for layout in w4 w12; do
  for ab in 0 1 2 4; do
    D=gpurun_out/gexp_${layout}_$ab; rm -rf $D
    MCD_GEMM_EXP_LAYOUT=$layout MCD_GEMM_EXP_ABLATE=$ab rocprofv3 --kernel-trace --stats --output-format csv -d $D -- python3 scripts/prof_gemm_exp.py 25000 10000 20 > $D.log 2>&1
    python3 - $D $layout $ab >> $O <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "gemm_nt_bf16_exp" in r["Name"]:
            us = float(r["AverageNs"]) / 1e3
            print("layout %-3s ablate %-3s  %-60s calls %3s  avg %7.1f us  %6.0f TFLOP/s  %.3f of 2.5 PF" % (
                sys.argv[2], sys.argv[3], r["Name"].split("(anonymous namespace)::")[-1][:60], r["Calls"], us, 2 * 25000 * 10000 * 512 / us / 1e6, 2 * 25000 * 10000 * 512 / us / 1e6 / 2500))
PY
    grep "embed_gemm_exp" $D.log >> $O
    rm -rf $D
  done
done
cat $O
# PMC of the w4 kernel: K loop only and product
for ab in 4 0; do
  D=gpurun_out/pmc_gexp_w4_$ab; rm -rf $D
  MCD_GEMM_EXP_LAYOUT=w4 MCD_GEMM_EXP_ABLATE=$ab rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_LDS -d $D/sq -- python3 scripts/prof_gemm_exp.py 25000 10000 6 > /dev/null 2>&1
  MCD_GEMM_EXP_LAYOUT=w4 MCD_GEMM_EXP_ABLATE=$ab rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INST_LEVEL_VMEM SQ_WAIT_INST_VMEM GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD -d $D/sq2 -- python3 scripts/prof_gemm_exp.py 25000 10000 6 > /dev/null 2>&1
  MCD_GEMM_EXP_LAYOUT=w4 MCD_GEMM_EXP_ABLATE=$ab rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum -d $D/tcc -- python3 scripts/prof_gemm_exp.py 25000 10000 6 > /dev/null 2>&1
  echo "== PMC, w4 layout, ablate $ab" >> $O
  python3 scripts/pmc_db.py $D gemm_nt_bf16_exp >> $O 2>&1
  rm -rf $D
done
tail -40 $O
