#!/bin/bash
# Round-5 PMC evidence on the CURRENT tree: HBM traffic of every core kernel at configs[1]'s shape and at the stress shape
# (FETCH_SIZE / WRITE_SIZE in separate passes), K4's SQ counters, K1s' counters -> the JSON files bench.py reads (profiles/r05_*).
# Run on the GPU box from the repo root:  bash scripts/r05_pmc.sh [core|stress|k4|gexp|all]
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
what=${1:-all}
O=gpurun_out
if [ $what = core ] || [ $what = all ]; then
  D=$O/pmc_core; rm -rf $D
  rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $D/fetch -- python3 bench.py --config core --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1 &&
  rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $D/write -- python3 bench.py --config core --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1 &&
  python3 scripts/pmc_traffic_json.py $D core $O/r05_pmc_traffic.json > /dev/null || exit 1
  rm -rf $D
fi
if [ $what = stress ] || [ $what = all ]; then
  D=$O/pmc_stress; rm -rf $D
  rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $D/fetch -- python3 bench.py --config stress --steps 3 --warmup 1 > /dev/null 2>&1 &&
  rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $D/write -- python3 bench.py --config stress --steps 3 --warmup 1 > /dev/null 2>&1 &&
  python3 scripts/pmc_traffic_json.py $D stress $O/r05_stress_pmc_traffic.json > /dev/null || exit 1
  rm -rf $D
fi
if [ $what = k4 ] || [ $what = all ]; then
  D=$O/pmc_k4; rm -rf $D
  rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS -d $D/sq -- python3 scripts/prof_k4.py trusted > /dev/null 2>&1 &&
  rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVES GRBM_GUI_ACTIVE -d $D/sq2 -- python3 scripts/prof_k4.py trusted > /dev/null 2>&1
  # the shape of scripts/prof_k4.py: 12 x 768 neurons, K = 100, C = 763 -> U K C logs
  python3 scripts/pmc_to_json.py $D wpmi_slice $O/r05_k4_pmc.json logs=703180800 shape=10000x763x9216xK100 > /dev/null
  rm -rf $D
fi
if [ $what = gexp ] || [ $what = all ]; then
  D=$O/pmc_gexp; rm -rf $D
  export MCD_PROF_LIBRARY=0
  rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_LDS -d $D/sq -- python3 scripts/prof_gemm_exp.py 25000 10000 6 > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE -d $D/tcc -- python3 scripts/prof_gemm_exp.py 25000 10000 6 > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc TA_TA_BUSY_sum SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_VMEM_TA_ADDR_FIFO_FULL SQ_INSTS_SALU -d $D/ta -- python3 scripts/prof_gemm_exp.py 25000 10000 6 > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $D/fetch -- python3 scripts/prof_gemm_exp.py 25000 10000 6 > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $D/write -- python3 scripts/prof_gemm_exp.py 25000 10000 6 > /dev/null 2>&1
  python3 scripts/pmc_to_json.py $D gemm_nt_bf16_exp $O/r05_gemm_stress_pmc_raw.json shape=25000x10000x512 > /dev/null
  python3 - $O/r05_gemm_stress_pmc_raw.json $O/r05_gemm_stress_pmc.json <<'PY'
import json, sys
r = json.load(open(sys.argv[1]))
us = r["mean_dur_us_under_counters"]
clk = r["GRBM_GUI_ACTIVE"] / 8 / us / 1e3            # GHz: the counter sums the 8 XCDs
out = {"pmc_source": "profiles/r05_gemm_stress_pmc.json: rocprofv3 --pmc passes over scripts/prof_gemm_exp.py 25000 10000 (scripts/r05_pmc.sh gexp)",
       "pmc_kernel": r["kernel"], "pmc_kernel_ms": round(us / 1e3, 4),
       "pmc_kernel_tflops": round(2 * 25000 * 10000 * 512 / us / 1e6, 1),
       "sustained_clock_ghz": round(clk, 2),
       "mfma_busy_frac": round(r["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024 / (r["GRBM_GUI_ACTIVE"] / 8), 3),
       "l2_hit_rate": round(r["TCC_HIT_sum"] / (r["TCC_HIT_sum"] + r["TCC_MISS_sum"]), 3),
       "hbm_write_bytes": r["WRITE_SIZE"] * 1024, "fetch_size_kb_raw": r["FETCH_SIZE"],
       "ta_busy_cycles_per_cu": round(r["TA_TA_BUSY_sum"] / 256, 0), "lds_bank_conflict_cycles": r["SQ_LDS_BANK_CONFLICT"],
       "valu_insts_per_wave": round(r["SQ_INSTS_VALU"] / 1024, 0)}
json.dump(out, open(sys.argv[2], "w"), indent=1)
print(json.dumps(out, indent=1))
PY
  rm -rf $D
fi
ls -la $O/r05_*pmc*.json
