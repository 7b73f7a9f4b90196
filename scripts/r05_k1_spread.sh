#!/bin/bash
# K1 (fp32 MFMA GEMM, parity mode) at 10 000 x 763 x 512: the next K-tile's eight DMA instructions all behind the barrier (MCD_GEMM_K1_SPREAD=0,
# rounds 4-5) against spread between the tile's MFMAs (1, the product): bit-exactness tests first, then kernel time from rocprofv3 kernel
# traces of the dev library, interleaved, then the per-workgroup stamps of both.   bash scripts/r05_k1_spread.sh [rounds]
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05_k1_spread.txt; : > $O
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py -q -x -k "gemm" > gpurun_out/k1_spread_tests.log 2>&1; echo "gemm tests (product library): rc=$? $(tail -1 gpurun_out/k1_spread_tests.log)" >> $O
timeout -k 10 600 python3 scripts/fuzz_front.py 150 5 > gpurun_out/k1_spread_fuzz.log 2>&1; echo "fuzz_front 150 5: rc=$? $(tail -1 gpurun_out/k1_spread_fuzz.log)" >> $O
export MCD_LIB_PATH=$PWD/mammo-clip-dissect_amd/csrc/libmcd_hip_dev.so
for rep in $(seq ${1:-4}); do for f in 0 1 2; do
  D=gpurun_out/k1s; rm -rf $D
  MCD_GEMM_K1_SPREAD=$f timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $D -- python3 scripts/k1_ksweep.py 512 > $D.log 2>&1
  python3 - $D $f >> $O <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "gemm_nt_f32_dma" in r["Name"]:
            us = float(r["AverageNs"]) / 1e3
            print("spread %s  calls %3s  avg %6.2f us  %.3f of 157.3 TF" % (sys.argv[2], r["Calls"], us, 2 * 10000 * 763 * 512 / us / 1e6 / 157.3))
PY
  rm -rf $D
done; done
for f in 0 2; do echo "--- stamps, MCD_GEMM_K1_SPREAD=$f" >> $O; MCD_GEMM_K1_SPREAD=$f MCD_GEMM_K1_STAMPS=1 timeout -k 10 300 python3 scripts/k1_stamps.py 512 2>&1 | grep -v amdgpu.ids >> $O; done
for f in 0 2; do echo "--- depth sweep, MCD_GEMM_K1_SPREAD=$f" >> $O; MCD_GEMM_K1_SPREAD=$f timeout -k 10 300 python3 scripts/k1_ksweep.py 2>&1 | grep -v amdgpu.ids >> $O; done
cat $O
