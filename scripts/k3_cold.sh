#!/bin/bash
# K3 kernel time by rocprofv3: dense / pitched rows, warm / cold reads (scripts/k3_cold.py)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04_k3_cold.txt
: > $O
for mode in dense_warm pitch_warm dense_cold pitch_cold; do
  D=gpurun_out/k3c; rm -rf $D
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $D -- python3 scripts/k3_cold.py $mode > $D.log 2>&1
  python3 - $D $mode >> $O <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "neuron_topk_fast" in r["Name"]:
            print("%-12s %-40s calls %3s avg %7.1f min %7.1f max %7.1f us" % (sys.argv[2], r["Name"].split("::")[-1][:40], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
PY
  rm -rf $D
done
cat $O
