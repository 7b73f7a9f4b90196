#!/bin/bash
# K1s v4: placement of the k-step's memory instructions (MCD_GEMM_EXP_PLACE, g4_op_after): product and K loop only.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
export MCD_LIB_PATH=$PWD/mammo-clip-dissect_amd/csrc/libmcd_hip_dev.so   # the ablation kernels live in the dev build (make dev)
O=gpurun_out/r04_gexp_place.txt
: > $O
for pl in 0 1 2; do
  MCD_GEMM_EXP_PLACE=$pl timeout -k 10 300 python3 scripts/gexp_check.py v4 > gpurun_out/place_check_$pl.log 2>&1 || { echo "check place $pl FAILED" >> $O; tail -20 gpurun_out/place_check_$pl.log >> $O; }
done
for rep in 1 2; do
for pl in 0 1 2; do for ab in 0 4; do
  D=gpurun_out/gexp_pl; rm -rf $D
  MCD_PROF_LIBRARY=0 MCD_GEMM_EXP_LAYOUT=v4 MCD_GEMM_EXP_PLACE=$pl MCD_GEMM_EXP_ABLATE=$ab timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $D -- python3 scripts/prof_gemm_exp.py 25000 10000 20 > $D.log 2>&1
  python3 - $D $pl $ab >> $O <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "gemm_nt_bf16_exp" in r["Name"]:
            us = float(r["AverageNs"]) / 1e3
            print("place %s ablate %-2s %-42s calls %3s  avg %7.1f us  %6.0f TFLOP/s  %.3f of 2.5 PF" % (
                sys.argv[2], sys.argv[3], r["Name"].split("(anonymous namespace)::")[-1][:42], r["Calls"], us, 2 * 25000 * 10000 * 512 / us / 1e6, 2 * 25000 * 10000 * 512 / us / 1e6 / 2500))
PY
  rm -rf $D
done; done; done
cat $O
