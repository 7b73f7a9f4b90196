#!/bin/bash
# quick A/B of K1s layouts: correctness of the default, then kernel times.   bash scripts/r04_gexp_quick.sh "v4 v5" "0 1 4"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
export MCD_LIB_PATH=$PWD/mammo-clip-dissect_amd/csrc/libmcd_hip_dev.so   # the ablation kernels live in the dev build (make dev)
O=gpurun_out/r04_gexp_quick.txt
: > $O
for lay in $1; do
  timeout -k 10 300 python3 scripts/gexp_check.py $lay > gpurun_out/quick_check_$lay.log 2>&1 || { echo "check $lay FAILED" >> $O; tail -20 gpurun_out/quick_check_$lay.log >> $O; }
  tail -1 gpurun_out/quick_check_$lay.log >> $O
done
for rep in 1 2; do for lay in $1; do for ab in $2; do
  D=gpurun_out/gexp_q; rm -rf $D
  MCD_PROF_LIBRARY=0 MCD_GEMM_EXP_LAYOUT=$lay MCD_GEMM_EXP_ABLATE=$ab timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $D -- python3 scripts/prof_gemm_exp.py 25000 10000 20 > $D.log 2>&1
  python3 - $D $lay $ab >> $O <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "gemm_nt_bf16_exp" in r["Name"]:
            us = float(r["AverageNs"]) / 1e3
            print("layout %-3s ablate %-2s %-42s calls %3s  avg %7.1f us  %6.0f TFLOP/s  %.3f of 2.5 PF" % (
                sys.argv[2], sys.argv[3], r["Name"].split("(anonymous namespace)::")[-1][:42], r["Calls"], us, 2 * 25000 * 10000 * 512 / us / 1e6, 2 * 25000 * 10000 * 512 / us / 1e6 / 2500))
PY
  rm -rf $D
done; done; done
cat $O
