#!/bin/bash
# K3 round 4: the ring kernel (next row by LDS-DMA during the selection; MCD_TOPK_RING=1 default, 2 = larger buffer at 25 000 images)
# against the workgroup-per-row kernel (=0): the top-K tests, then kernel times by rocprofv3 at 10 000 and 25 000 images.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04_k3_ring.txt
: > $O
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "${K3_TESTS:-topk or fuzz or multirank_hip or dissector}" > gpurun_out/k3_tests.log 2>&1; echo "tests rc=$? $(tail -1 gpurun_out/k3_tests.log)" >> $O
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
for rep in 1 2; do for N in 10000 25000; do for sel in ${RINGS:-0 1}; do
  D=gpurun_out/k3s; rm -rf $D
  MCD_TOPK_RING=$sel timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $D -- python3 /tmp/k3_run.py $N > $D.log 2>&1
  python3 - $D $N $sel >> $O <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "neuron_topk" in r["Name"]:
            us = float(r["AverageNs"]) / 1e3
            N = int(sys.argv[2])
            print("N %6d ring %s  %-46s calls %3s avg %7.1f us  %.2f TB/s" % (N, sys.argv[3], r["Name"].split("::")[-1][:46], r["Calls"], us, (4.0 * N * 9216 + 8 * 100 * 9216) / us / 1e6))
PY
  rm -rf $D
done; done; done
cat $O
