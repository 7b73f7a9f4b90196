#!/bin/bash
# PMC passes over K1s (the default kernel: gemm_nt_bf16_exp_w4_kernel<4,4,4,0,fold,LT>) at one rank's share of configs[4]; run on
# the GPU box from the repo root:   bash scripts/prof_pmc_gexp.sh   -> gpurun_out/r03_gemm_exp_pmc.txt
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
D=gpurun_out/pmc_gexp_final
rm -rf $D
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_LDS -d $D/sq -- python3 scripts/prof_gemm_exp.py 25000 10000 6 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE -d $D/tcc -- python3 scripts/prof_gemm_exp.py 25000 10000 6 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc TA_TA_BUSY_sum SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $D/ta -- python3 scripts/prof_gemm_exp.py 25000 10000 6 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $D/fetch -- python3 scripts/prof_gemm_exp.py 25000 10000 6 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $D/write -- python3 scripts/prof_gemm_exp.py 25000 10000 6 > /dev/null 2>&1
python3 scripts/pmc_db.py $D gemm_nt_bf16_exp > gpurun_out/r03_gemm_exp_pmc.txt 2>&1
rm -rf $D
rocprofv3 --kernel-trace --stats --output-format csv -d $D -- python3 scripts/prof_gemm_exp.py 25000 10000 20 > /dev/null 2>&1
python3 - $D >> gpurun_out/r03_gemm_exp_pmc.txt <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "gemm_nt_bf16_exp" in r["Name"]:
            us = float(r["AverageNs"]) / 1e3
            print("un-profiled kernel trace: %s: %s calls, average %.1f us = %.0f TFLOP/s" % (r["Name"].split("::")[-1][:70], r["Calls"], us, 2 * 25000 * 10000 * 512 / us / 1e6))
PY
rm -rf $D
cat gpurun_out/r03_gemm_exp_pmc.txt
