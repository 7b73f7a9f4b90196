#!/bin/bash
# K1 (fp32 MFMA GEMM, parity mode): the WIDE form (256 x 128 tiles, one 512-thread workgroup per CU, three-stage ring: two K-tiles in flight)
# against the 128 x 128 form (two workgroups per CU, one K-tile in flight).  Bit-exactness first -- the product library's tests, then the
# randomised front test with the wide form FORCED on every shape that meets the DMA preconditions (dev knob MCD_GEMM_K1_WIDE=1) --, then
# kernel times from rocprofv3 traces, interleaved, at the configs[1] shape and over the reduction depth.   bash scripts/r05_k1_wide.sh [rounds]
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05_k1_wide.txt; : > $O
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_configs.py -q -x -k "gemm or golden" > gpurun_out/k1_wide_tests.log 2>&1; echo "gemm / golden tests (product library): rc=$? $(tail -1 gpurun_out/k1_wide_tests.log)" >> $O
export MCD_LIB_PATH=$PWD/mammo-clip-dissect_amd/csrc/libmcd_hip_dev.so
MCD_GEMM_K1_WIDE=1 timeout -k 10 600 python3 scripts/fuzz_front.py 300 11 > gpurun_out/k1_wide_fuzz.log 2>&1; echo "fuzz_front 300 11, wide forced: rc=$? $(tail -1 gpurun_out/k1_wide_fuzz.log)" >> $O
for rep in $(seq ${1:-4}); do for f in 0 1; do
  D=gpurun_out/k1w; rm -rf $D
  MCD_GEMM_K1_WIDE=$f timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $D -- python3 scripts/k1_ksweep.py 512 > $D.log 2>&1
  python3 - $D $f >> $O <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "gemm_nt_f32_" in r["Name"]:
            us = float(r["AverageNs"]) / 1e3
            print("wide %s  %-28s calls %3s  avg %6.2f us  %.3f of 157.3 TF" % (sys.argv[2], r["Name"].split("::")[-1][:28], r["Calls"], us, 2 * 10000 * 763 * 512 / us / 1e6 / 157.3))
PY
  rm -rf $D
done; done
for f in 0 1; do echo "--- depth sweep, MCD_GEMM_K1_WIDE=$f" >> $O; MCD_GEMM_K1_WIDE=$f timeout -k 10 300 python3 scripts/k1_ksweep.py 2>&1 | grep -v amdgpu.ids >> $O; done
cat $O
