#!/bin/bash
# K3 A/B by rocprofv3 kernel times: VARIANTS="label=ENV1=v,ENV2=v ..." (default: the product library alone), N in SIZES.
# e.g. VARIANTS="buf= glob=MCD_LIB_PATH=$PWD/mammo-clip-dissect_amd/csrc/libmcd_hip_g.so" bash scripts/k3_ab.sh
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/${OUT:-r04_k3_ab.txt}
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
for rep in 1 2; do for N in ${SIZES:-10000 25000}; do for var in ${VARIANTS:-product=}; do
  label=${var%%=*}; envs=${var#*=}
  D=gpurun_out/k3s; rm -rf $D
  ( for kv in ${envs//,/ }; do export "$kv"; done
    timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $D -- python3 /tmp/k3_run.py $N > $D.log 2>&1 )
  python3 - $D $N $label >> $O <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "neuron_topk_fast" in r["Name"]:
            us = float(r["AverageNs"]) / 1e3
            N = int(sys.argv[2])
            print("N %6d %-10s %-46s calls %3s avg %7.1f us  %.2f TB/s" % (N, sys.argv[3], r["Name"].split("::")[-1][:46], r["Calls"], us, (4.0 * N * 9216 + 8 * 100 * 9216) / us / 1e6))
PY
  rm -rf $D
done; done; done
cat $O
