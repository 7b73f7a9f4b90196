#!/bin/bash
# K3 (neuron_topk_fast_kernel): staggered first generation of workgroups (MCD_TOPK_STAGGER = quarter of a workgroup's life in
# cycles, MCD_TOPK_STAGGER_SHIFT = which bits of the workgroup id pick the class), kernel times by rocprofv3.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04_k3_stagger.txt
: > $O
cat > /tmp/k3_run.py <<'PY'
import os, sys
sys.path.insert(0, os.getcwd())
import torch
import mammo_clip_dissect_amd
from mammo_clip_dissect_amd import core
dev = torch.device("cuda:0")
N = int(sys.argv[1]); U = 9216
g = torch.Generator(device=dev).manual_seed(0)
At = torch.randn(U, N, device=dev, generator=g)
for _ in range(12):
    vals, idx = core.col_topk(At, 100, neuron_major=True)
torch.cuda.synchronize()
PY
for N in 10000 25000; do
for cfg in "0 3" "5000 3" "8000 3" "12000 3" "8000 8" "8000 0" "16000 3" "24000 3"; do
  set -- $cfg
  D=gpurun_out/k3s; rm -rf $D
  MCD_TOPK_STAGGER=$1 MCD_TOPK_STAGGER_SHIFT=$2 timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $D -- python3 /tmp/k3_run.py $N > $D.log 2>&1
  python3 - $D $N $1 $2 >> $O <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "neuron_topk_fast" in r["Name"]:
            us = float(r["AverageNs"]) / 1e3
            N = int(sys.argv[2])
            print("N %6d stagger %6s shift %s  %-40s calls %3s avg %7.1f us  %.2f TB/s" % (N, sys.argv[3], sys.argv[4], r["Name"].split("::")[-1][:40], r["Calls"], us, (4.0 * N * 9216 + 8 * 100 * 9216) / us / 1e6))
PY
  rm -rf $D
done; done
cat $O
