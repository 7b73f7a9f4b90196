#!/bin/bash
# K1s: kernel time by the WALK of a plain k-step's 64 MFMAs over the 8 x 8 block grid (dev knob MCD_GEMM_EXP_PLACE = placement + 16 x walk:
# 1 row-major, 17 serpentine, 33 2 x 2 quads, 49 column-major, 65 column serpentine), correctness first, then interleaved rounds of the
# 22-launch trace (the sustained regime).   bash scripts/r05_walk.sh ["1 17 33 49 65"] [rounds]
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
export MCD_LIB_PATH=$PWD/mammo-clip-dissect_amd/csrc/libmcd_hip_dev.so
LIST=${1:-"1 17 33 49 65"}; ROUNDS=${2:-3}
O=gpurun_out/r05_walk.txt; : > $O
for pl in $LIST; do MCD_GEMM_EXP_PLACE=$pl timeout -k 10 300 python3 scripts/gexp_check.py > gpurun_out/walk_check_$pl.log 2>&1; echo "place $pl check: $(tail -1 gpurun_out/walk_check_$pl.log)" >> $O; done
for rep in $(seq $ROUNDS); do for pl in $LIST; do
  D=gpurun_out/gexp_wk; rm -rf $D
  MCD_GEMM_EXP_PLACE=$pl MCD_PROF_LIBRARY=0 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $D -- python3 scripts/prof_gemm_exp.py 25000 10000 20 > $D.log 2>&1
  python3 - $D $pl >> $O <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "gemm_nt_bf16_exp" in r["Name"]:
            us = float(r["AverageNs"]) / 1e3
            print("place %2s  calls %3s  avg %7.1f us  %.3f of 2.5 PF" % (sys.argv[2], r["Calls"], us, 2 * 25000 * 10000 * 512 / us / 1e6 / 2500))
PY
  rm -rf $D
done; done
cat $O
