#!/bin/bash
# PMC passes over K3 (neuron_topk_fast_kernel) at the config-2 shape; run on the GPU box from the repo root
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
D=gpurun_out/pmc_k3
rm -rf $D
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU -d $D/sq -- python3 scripts/prof_k34.py > /dev/null 2>&1 &&
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU SQ_WAVES GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD -d $D/sq2 -- python3 scripts/prof_k34.py > /dev/null 2>&1 &&
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ATOMIC_RETURN SQ_IFETCH SQ_WAIT_INST_VMEM SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM SQ_LEVEL_WAVES -d $D/sq3 -- python3 scripts/prof_k34.py > /dev/null 2>&1
python3 scripts/pmc_db.py $D neuron_topk_fast > gpurun_out/r02_k3_pmc.txt 2>&1
rm -rf $D
cat gpurun_out/r02_k3_pmc.txt
