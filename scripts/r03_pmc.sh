#!/bin/bash
# Round-3 PMC evidence on the CURRENT tree: HBM traffic of every core kernel at configs[1]'s shape and at the
# stress shape (FETCH_SIZE / WRITE_SIZE in separate passes), K4's SQ / LDS counters, K4s' gather counters.
# Run on the GPU box from the repo root:  bash scripts/r03_pmc.sh [core|stress|k4|all]
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
what=${1:-all}
O=gpurun_out
if [ $what = core ] || [ $what = all ]; then
  D=$O/pmc_core; rm -rf $D
  rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $D/fetch -- python3 bench.py --config core --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1 &&
  rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $D/write -- python3 bench.py --config core --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1 &&
  python3 scripts/pmc_traffic_json.py $D core $O/r03_pmc_traffic.json > /dev/null || exit 1
  rm -rf $D
fi
if [ $what = stress ] || [ $what = all ]; then
  D=$O/pmc_stress; rm -rf $D
  rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $D/fetch -- python3 bench.py --config stress --steps 3 --warmup 1 > /dev/null 2>&1 &&
  rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $D/write -- python3 bench.py --config stress --steps 3 --warmup 1 > /dev/null 2>&1 &&
  python3 scripts/pmc_traffic_json.py $D stress $O/r03_stress_pmc_traffic.json > /dev/null || exit 1
  rm -rf $D
fi
if [ $what = k4 ] || [ $what = all ]; then
  D=$O/pmc_k4; rm -rf $D
  rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS -d $D/sq -- python3 scripts/prof_k4.py trusted > /dev/null 2>&1 &&
  rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVES GRBM_GUI_ACTIVE -d $D/sq2 -- python3 scripts/prof_k4.py trusted > /dev/null 2>&1 &&
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_RD SQ_WAIT_INST_VMEM SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_ACTIVE_INST_SCA -d $D/sq3 -- python3 scripts/prof_k4.py trusted > /dev/null 2>&1
  python3 scripts/pmc_db.py $D wpmi_slice > $O/r03_k4_pmc_sq.txt 2>&1
  rm -rf $D
  cat $O/r03_k4_pmc_sq.txt
fi
