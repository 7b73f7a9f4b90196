#!/bin/bash
# K1 (fp32 MFMA GEMM): the output through a buffer descriptor (MCD_GEMM_K1_BST=1, the product: one v_add + one store per value) against plain
# stores with 64-bit address arithmetic, a compare and a branch per value (0): bit-exactness first (product library), then interleaved traces.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05_k1_bst.txt; : > $O
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_configs.py tests/test_gpu_e2e.py -q -x -k "gemm or golden" > gpurun_out/k1_bst_tests.log 2>&1; echo "gemm / golden tests (product library): rc=$? $(tail -1 gpurun_out/k1_bst_tests.log)" >> $O
timeout -k 10 600 python3 scripts/fuzz_front.py 400 21 > gpurun_out/k1_bst_fuzz.log 2>&1; echo "fuzz_front 400 21: rc=$? $(tail -1 gpurun_out/k1_bst_fuzz.log)" >> $O
export MCD_LIB_PATH=$PWD/mammo-clip-dissect_amd/csrc/libmcd_hip_dev.so
for rep in $(seq ${1:-4}); do for f in 0 1; do
  D=gpurun_out/k1b; rm -rf $D
  MCD_GEMM_K1_BST=$f timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $D -- python3 scripts/k1_ksweep.py 512 > $D.log 2>&1
  python3 - $D $f >> $O <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "gemm_nt_f32_dma" in r["Name"]:
            us = float(r["AverageNs"]) / 1e3
            print("bst %s  calls %3s  avg %6.2f us  %.3f of 157.3 TF" % (sys.argv[2], r["Calls"], us, 2 * 10000 * 763 * 512 / us / 1e6 / 157.3))
PY
  rm -rf $D
done; done
for f in 0 1; do echo "--- K1_REPS=2000, MCD_GEMM_K1_BST=$f: $(K1_REPS=2000 MCD_GEMM_K1_BST=$f timeout -k 10 300 python3 scripts/k1_ksweep.py 512 2>&1 | tail -1)" >> $O; done
cat $O
