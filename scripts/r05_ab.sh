#!/bin/bash
# K1s A/B in ONE call: kernel times (rocprofv3 kernel trace, 22 back-to-back launches) of several dev libraries, interleaved, three rounds.
#   bash scripts/r05_ab.sh "<label=path.so> ..."   (paths relative to the repository root)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05_ab.txt; : > $O
for rep in 1 2 3; do for item in $1; do
  lab=${item%%=*}; lib=${item#*=}
  D=gpurun_out/gexp_ab; rm -rf $D
  MCD_LIB_PATH=$PWD/$lib MCD_PROF_LIBRARY=0 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $D -- python3 scripts/prof_gemm_exp.py 25000 10000 20 > $D.log 2>&1
  python3 - $D $lab >> $O <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "gemm_nt_bf16_exp" in r["Name"]:
            us = float(r["AverageNs"]) / 1e3
            print("%-10s calls %3s  avg %7.1f us  %.3f of 2.5 PF" % (sys.argv[2], r["Calls"], us, 2 * 25000 * 10000 * 512 / us / 1e6 / 2500))
PY
  rm -rf $D
done; done
cat $O
