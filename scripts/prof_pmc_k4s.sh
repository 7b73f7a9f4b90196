#!/bin/bash
# PMC passes over K4s (wpmi_bf16_kernel) at one rank's share of configs[4]; run on the GPU box from the repo root
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
D=gpurun_out/pmc_k4s
rm -rf $D
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD -d $D/sq -- python3 scripts/prof_k4s.py > /dev/null 2>&1 &&
rocprofv3 --kernel-trace --pmc SQ_INST_CYCLES_VMEM_RD SQ_WAIT_INST_VMEM SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_WAVES GRBM_GUI_ACTIVE -d $D/sq2 -- python3 scripts/prof_k4s.py > /dev/null 2>&1 &&
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum -d $D/tcc -- python3 scripts/prof_k4s.py > /dev/null 2>&1 &&
rocprofv3 --kernel-trace --pmc TA_BUSY_avr TA_TA_BUSY_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum -d $D/ta -- python3 scripts/prof_k4s.py > /dev/null 2>&1 &&
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $D/fetch -- python3 scripts/prof_k4s.py > /dev/null 2>&1
python3 scripts/pmc_db.py $D wpmi_bf16 > gpurun_out/r03_k4s_pmc.txt 2>&1
cat gpurun_out/r03_k4s_pmc.txt
rm -rf $D    # the counter databases are tens of MB: gpurun copies at most 64 MiB back
