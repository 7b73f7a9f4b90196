#!/bin/bash
# Round 4, K1s v4 (k_gexp_v4.inc): correctness on edge shapes, then kernel times (rocprofv3 --kernel-trace --stats) of the product
# and its ablations (MCD_GEMM_EXP_ABLATE: 1 no stores, 4 K loop only) for both sync forms, round 3's w4 beside them.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04_gexp_v4.txt
: > $O
timeout -k 10 300 python3 scripts/gexp_check.py v4 0 >> $O 2>&1 || { echo "check v4 sync0 FAILED" >> $O; tail -30 $O; exit 1; }
timeout -k 10 300 python3 scripts/gexp_check.py v4 1 >> $O 2>&1 || { echo "check v4 sync1 FAILED" >> $O; }
run() {   # layout sync ablate
  D=gpurun_out/gexp_$1_$2_$3; rm -rf $D
  MCD_PROF_LIBRARY=0 MCD_GEMM_EXP_LAYOUT=$1 MCD_GEMM_EXP_SYNC=$2 MCD_GEMM_EXP_ABLATE=$3 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $D -- python3 scripts/prof_gemm_exp.py 25000 10000 20 > $D.log 2>&1
  python3 - $D $1 $2 $3 >> $O <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "gemm_nt_bf16_exp" in r["Name"]:
            us = float(r["AverageNs"]) / 1e3
            print("layout %-3s sync %s ablate %-3s  %-50s calls %3s  avg %7.1f us  %6.0f TFLOP/s  %.3f of 2.5 PF" % (
                sys.argv[2], sys.argv[3], sys.argv[4], r["Name"].split("(anonymous namespace)::")[-1][:50], r["Calls"], us, 2 * 25000 * 10000 * 512 / us / 1e6, 2 * 25000 * 10000 * 512 / us / 1e6 / 2500))
PY
  grep "embed_gemm_exp" $D.log >> $O
  rm -rf $D
}
for ab in 0 1 4; do run v4 0 $ab; done
for ab in 0 1 4; do run v4 1 $ab; done
for ab in 0 4; do run w4 0 $ab; done
cat $O
