#!/bin/bash
# dev-build compile of k_gemm.hip with its assembly kept (/tmp/isa), for the static audits of the K1s kernels
mkdir -p /tmp/isa && cd /root/repo/mammo-clip-dissect_amd/csrc && /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -DMCD_DEV_KNOBS -c k_gemm.hip -o /tmp/isa/k_gemm_dev.o -save-temps=obj 2>&1 | grep -E "error|warning" -A5 | head -30
