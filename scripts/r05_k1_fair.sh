#!/bin/bash
# K1 (fp32 MFMA GEMM, parity mode) at 10 000 x 763 x 512: kernel time with the two workgroups of a CU at equal priority (MCD_GEMM_K1_FAIR=0)
# against alternating priority by K-tile and hardware wave slot (1, the product) and a four-level ladder by progress (2); dev library,
# rocprofv3 kernel trace, interleaved, three rounds; then the per-workgroup stamps of modes 0 and 1.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
export MCD_LIB_PATH=$PWD/mammo-clip-dissect_amd/csrc/libmcd_hip_dev.so
O=gpurun_out/r05_k1_fair.txt; : > $O
for rep in 1 2 3; do for f in 0 1 2; do
  D=gpurun_out/k1f; rm -rf $D
  MCD_GEMM_K1_FAIR=$f timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $D -- python3 scripts/k1_ksweep.py 512 > $D.log 2>&1
  python3 - $D $f >> $O <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "gemm_nt_f32_dma" in r["Name"]:
            us = float(r["AverageNs"]) / 1e3
            print("fair %s  calls %3s  avg %6.2f us  %.3f of 157.3 TF" % (sys.argv[2], r["Calls"], us, 2 * 10000 * 763 * 512 / us / 1e6 / 157.3))
PY
  rm -rf $D
done; done
for f in 0 1; do echo "--- stamps, MCD_GEMM_K1_FAIR=$f" >> $O; MCD_GEMM_K1_FAIR=$f timeout -k 10 300 python3 scripts/k1_stamps.py 512 2>&1 | grep -v amdgpu.ids >> $O; done
cat $O
