#!/bin/bash
# K1s v5 (k_gexp_v5.inc: stores deferred into the next tile's K loop) against v4: correctness, bit-identity, kernel times.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
export MCD_LIB_PATH=$PWD/mammo-clip-dissect_amd/csrc/libmcd_hip_dev.so   # the ablation kernels live in the dev build (make dev)
O=gpurun_out/r04_gexp_v5.txt
: > $O
timeout -k 10 300 python3 scripts/gexp_check.py v5 >> $O 2>&1 || { echo "check v5 FAILED" >> $O; cat $O; exit 1; }
timeout -k 10 300 python3 - >> $O 2>&1 <<'PY'
import os, sys, torch
sys.path.insert(0, os.getcwd())
import mammo_clip_dissect_amd
from mammo_clip_dissect_amd import core
dev = torch.device("cuda:0")
bad = 0
for (N, C) in [(3000, 2000), (25000, 10000), (513, 511), (17, 33), (5000, 763)]:
    g = torch.Generator().manual_seed(N + C)
    I = torch.randn(N, 512, generator=g).to(dev); T = torch.randn(C, 512, generator=g).to(dev)
    os.environ["MCD_GEMM_EXP_LAYOUT"] = "v4"; E4, r4 = core.embed_gemm_exp(I, T, 10.0, normalize=True); E4 = E4.clone(); r4 = r4.clone()
    os.environ["MCD_GEMM_EXP_LAYOUT"] = "v5"; E5, r5 = core.embed_gemm_exp(I, T, 10.0, normalize=True)
    torch.cuda.synchronize()
    same = torch.equal(E4.view(torch.int16), E5.view(torch.int16)) and torch.equal(r4, r5)
    print("v5 == v4 bit for bit at N=%d C=%d: %s" % (N, C, same)); bad += 0 if same else 1
sys.exit(1 if bad else 0)
PY
run() {   # layout ablate
  D=gpurun_out/gexp_$1_$2; rm -rf $D
  MCD_PROF_LIBRARY=0 MCD_GEMM_EXP_LAYOUT=$1 MCD_GEMM_EXP_ABLATE=$2 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $D -- python3 scripts/prof_gemm_exp.py 25000 10000 20 > $D.log 2>&1
  python3 - $D $1 $2 >> $O <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "gemm_nt_bf16_exp" in r["Name"]:
            us = float(r["AverageNs"]) / 1e3
            print("layout %-3s ablate %-2s %-42s calls %3s  avg %7.1f us  %6.0f TFLOP/s  %.3f of 2.5 PF" % (
                sys.argv[2], sys.argv[3], r["Name"].split("(anonymous namespace)::")[-1][:42], r["Calls"], us, 2 * 25000 * 10000 * 512 / us / 1e6, 2 * 25000 * 10000 * 512 / us / 1e6 / 2500))
PY
  grep "embed_gemm_exp" $D.log >> $O
  rm -rf $D
}
for rep in 1 2; do run v5 0; run v4 0; done
run v5 1; run v5 4; run v4 4
cat $O
