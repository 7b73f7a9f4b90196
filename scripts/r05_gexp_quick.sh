#!/bin/bash
# K1s (round 5, k_gexp_v6.inc): correctness on the edge shapes, then kernel times from rocprofv3 traces of the dev build.
#   bash scripts/r05_gexp_quick.sh "<ablate values>" [reps]      variants through MCD_GEMM_EXP_OVERLAP / MCD_GEMM_EXP_STAUX in the environment
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
export MCD_LIB_PATH=$PWD/mammo-clip-dissect_amd/csrc/libmcd_hip_dev.so   # the ablation kernels live in the dev build (make dev)
O=gpurun_out/r05_gexp_quick.txt
REPS=${2:-20}
: > $O
timeout -k 10 300 python3 scripts/gexp_check.py > gpurun_out/quick_check.log 2>&1 || { echo "check FAILED" >> $O; tail -20 gpurun_out/quick_check.log >> $O; }
tail -1 gpurun_out/quick_check.log >> $O
for rep in 1 2; do for ab in $1; do
  D=gpurun_out/gexp_q; rm -rf $D
  MCD_PROF_LIBRARY=0 MCD_GEMM_EXP_ABLATE=$ab timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $D -- python3 scripts/prof_gemm_exp.py 25000 10000 $REPS > $D.log 2>&1
  python3 - $D $ab >> $O <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "gemm_nt_bf16_exp" in r["Name"]:
            us = float(r["AverageNs"]) / 1e3
            print("ablate %-2s %-46s calls %3s  avg %7.1f us  %6.0f TFLOP/s  %.3f of 2.5 PF" % (
                sys.argv[2], r["Name"].split("(anonymous namespace)::")[-1][:46], r["Calls"], us, 2 * 25000 * 10000 * 512 / us / 1e6, 2 * 25000 * 10000 * 512 / us / 1e6 / 2500))
PY
  rm -rf $D
done; done
cat $O
