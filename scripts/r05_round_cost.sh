#!/bin/bash
# K1s: what the workgroups' LAST, partly filled round of tiles costs: kernel time at 25 000 images (15.3 tiles per workgroup: 16 rounds)
# against 24 576 images (15 rounds exactly) and 24 832 (15.16: the last round holds half as many tiles), same 10 000 concepts.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05_round_cost.txt; : > $O
for rep in 1 2; do for n in 25000 24576 24832 22016; do
  D=gpurun_out/gexp_rc; rm -rf $D
  MCD_PROF_LIBRARY=0 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $D -- python3 scripts/prof_gemm_exp.py $n 10000 20 > $D.log 2>&1
  python3 - $D $n >> $O <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "gemm_nt_bf16_exp" in r["Name"]:
            us = float(r["AverageNs"]) / 1e3; n = int(sys.argv[2]); t = ((n + 255) // 256) * 40 / 256.0
            print("images %6d  tiles per workgroup %.2f  avg %7.1f us  per round of the longest walk %.2f us" % (n, t, us, us / -(-t // 1)))
PY
  rm -rf $D
done; done
cat $O
