#!/bin/bash
# Build check of the generated overlapped K1s kernel: its accumulators are physical registers named in inline asm, which is only
# safe while the compiler itself never touches an AGPR and never spills (a spill would go to an AGPR or to scratch).
cd "$(dirname "$0")/../../mammo-clip-dissect_amd/csrc" || exit 1
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -Wno-inline-asm -mllvm -amdgpu-spill-vgpr-to-agpr=0 -S --cuda-device-only -o /tmp/k_gemm.s k_gemm.hip 2>&1 | grep -E "error" -A3 | head -12
rc=0
# HISTORICAL (round 3): the w4o kernels are not in the build; without them there is nothing to check and the script says so
if ! grep -q "w4o_[0-9]x[0-9]_kernel" /tmp/k_gemm.s; then echo "no w4o kernel in this build (scripts/archive/gen_gexp_w4o.py generates them): nothing checked"; exit 2; fi
for K in $(grep -o "^_ZN[A-Za-z0-9_]*w4o_[0-9]x[0-9]_kernel[A-Za-z0-9_]*:" /tmp/k_gemm.s | tr -d ':'); do
  S=$(grep -n "^$K:" /tmp/k_gemm.s | head -1 | cut -d: -f1); E=$(awk -v s=$S 'NR>s && /s_endpgm/ {print NR; exit}' /tmp/k_gemm.s)
  sed -n "${S},${E}p" /tmp/k_gemm.s > /tmp/w4o_one.s
  sc=$(grep -c scratch_ /tmp/w4o_one.s); aw=$(grep -c v_accvgpr_write /tmp/w4o_one.s); hv=$(grep -v "^\s*v_mfma\|^\s*v_mov_b32_e32 v[0-9]*, v[12][0-9][0-9]$" /tmp/w4o_one.s | grep -c "\bv\(12[89]\|1[3-9][0-9]\|2[0-5][0-9]\)\b")
  echo "$K: lines $(wc -l < /tmp/w4o_one.s) mfma $(grep -c v_mfma /tmp/w4o_one.s) scratch $sc accvgpr_write $aw compiler uses of v128+ $hv  $(awk -v s=$E 'NR>s && /codeLenInByte/ {print; exit}' /tmp/k_gemm.s) $(awk -v s=$E 'NR>s && /; NumVgprs/ {print; exit}' /tmp/k_gemm.s)"
  # the compiler may park values in AGPRs only in the un-overlapped flush after the K loops (behind the last s_barrier), where it
  # reuses registers whose accumulators have been consumed already (ascending order; tests/test_gpu_kernels.py checks the bits)
  lb=$(grep -n "s_barrier" /tmp/w4o_one.s | tail -1 | cut -d: -f1); awl=$(head -n ${lb:-0} /tmp/w4o_one.s | grep -c v_accvgpr_write)
  echo "   v_accvgpr_write inside the K loops: $awl"
  if [ "$sc" != 0 ] || [ "$awl" != 0 ]; then rc=1; fi
done
exit $rc
