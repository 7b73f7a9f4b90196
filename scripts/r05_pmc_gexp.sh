#!/bin/bash
# PMC passes over K1s (round 5: gemm_nt_bf16_exp_v6_kernel) at one rank's share of configs[4]: where the wave's cycles go.
#   bash scripts/r05_pmc_gexp.sh [ablate] [overlap] -> gpurun_out/r05_gexp_pmc_a<ablate>_o<overlap>.txt     (dev build)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
export MCD_LIB_PATH=$PWD/mammo-clip-dissect_amd/csrc/libmcd_hip_dev.so
AB=${1:-0}; OV=${2:-2}
export MCD_GEMM_EXP_ABLATE=$AB MCD_GEMM_EXP_OVERLAP=$OV MCD_PROF_LIBRARY=0
D=gpurun_out/pmc_gexp_r05
O=gpurun_out/r05_gexp_pmc_a${AB}_o${OV}.txt
rm -rf $D
P="python3 scripts/prof_gemm_exp.py 25000 10000 6"
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_LDS -d $D/sq1 -- $P > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_VALU_MFMA_COEXEC_CYCLES -d $D/sq2 -- $P > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU -d $D/sq3 -- $P > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc TA_TA_BUSY_sum SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE -d $D/ta -- $P > /dev/null 2>&1
echo "ablate $AB overlap $OV" > $O
python3 scripts/pmc_db.py $D gemm_nt_bf16_exp >> $O 2>&1
rm -rf $D
cat $O
