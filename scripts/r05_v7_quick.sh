#!/bin/bash
# K1s, the two-accumulator-set kernel (k_gexp_v7.inc, dev build, MCD_GEMM_EXP_V7=1): correctness on the edge shapes, then kernel
# times of v6 and v7 from rocprofv3 traces in ONE call.   bash scripts/r05_v7_quick.sh "<v7 ablate values>" [reps]
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
export MCD_LIB_PATH=$PWD/mammo-clip-dissect_amd/csrc/libmcd_hip_dev.so
O=gpurun_out/r05_v7_quick.txt
REPS=${2:-20}
: > $O
MCD_GEMM_EXP_V7=1 timeout -k 10 300 python3 scripts/gexp_check.py > gpurun_out/v7_check.log 2>&1 || { echo "v7 check FAILED" >> $O; grep -c FAIL gpurun_out/v7_check.log >> $O; }
tail -1 gpurun_out/v7_check.log >> $O
for rep in 1 2; do for cfg in "0 0" $(for a in $1; do echo "1 $a"; done | tr '\n' ';' | sed 's/;$//' | tr ';' '\n' | sed 's/ /_/'); do
  v7=${cfg%%[_ ]*}; ab=${cfg##*[_ ]}
  D=gpurun_out/gexp_q7; rm -rf $D
  MCD_PROF_LIBRARY=0 MCD_GEMM_EXP_V7=$v7 MCD_GEMM_EXP_ABLATE=$ab timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $D -- python3 scripts/prof_gemm_exp.py 25000 10000 $REPS > $D.log 2>&1
  python3 - $D $v7 $ab >> $O <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "gemm_nt_bf16_exp" in r["Name"]:
            us = float(r["AverageNs"]) / 1e3
            print("v7=%s ablate %-2s %-40s calls %3s  avg %7.1f us  %6.0f TFLOP/s  %.3f of 2.5 PF" % (
                sys.argv[2], sys.argv[3], r["Name"].split("(anonymous namespace)::")[-1][:40], r["Calls"], us, 2 * 25000 * 10000 * 512 / us / 1e6, 2 * 25000 * 10000 * 512 / us / 1e6 / 2500))
PY
  rm -rf $D
done; done
cat $O; grep -E "FAIL|Error|error" gpurun_out/v7_check.log | head -20
