#!/bin/bash
# Round-5 evidence on the final tree (run on the GPU box from the repo root): the MFMA ceiling micro, PMC passes -> the JSON files
# bench.py reads, GPU tests with parity statistics, headline bench + kernel stats, stress bench + kernel stats.
#   bash scripts/r05_final.sh [quick]      quick: no PMC passes (they are copied from profiles/ as they are)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out
hipcc --offload-arch=gfx950 -O3 -w -o /tmp/mfma_fill scripts/micro/mfma_fill.hip && for i in 1 2 3; do /tmp/mfma_fill D; done > $O/r05_mfma_ceiling.txt 2>&1
python3 - $O/r05_mfma_ceiling.txt $O/r05_mfma_ceiling.json <<'PY'
import json, re, sys
t = open(sys.argv[1]).read()
v16 = [float(x) for x in re.findall(r"16x16x32 random: [\d.]+ ms = (\d+) TFLOP/s", t)]
v32 = [float(x) for x in re.findall(r"32x32x16 random: [\d.]+ ms = (\d+) TFLOP/s", t)]
json.dump({"_how": "scripts/micro/mfma_fill.hip D, three runs (scripts/r05_final.sh): bare bf16 MFMA loops on random operands, operands in registers, one wave per SIMD, 256 CUs",
           "tflops_16x16x32_random": v16, "tflops_32x32x16_random": v32}, open(sys.argv[2], "w"), indent=1)
print(open(sys.argv[2]).read())
PY
cp $O/r05_mfma_ceiling.json profiles/r05_mfma_ceiling.json
if [ "$1" != quick ]; then
  bash scripts/r05_pmc.sh all > $O/r05_pmc.log 2>&1; echo "pmc rc=$?"
  for f in r05_pmc_traffic.json r05_stress_pmc_traffic.json r05_k4_pmc.json r05_gemm_stress_pmc.json; do [ -s $O/$f ] && cp $O/$f profiles/$f; done
fi
rm -f $O/r05_parity_stats.txt
MCD_STATS_FILE=$PWD/$O/r05_parity_stats.txt timeout -k 10 1100 python -m pytest tests -m gpu -x -q > $O/r05_gputests.log 2>&1; echo "tests rc=$?"; tail -2 $O/r05_gputests.log
python bench.py > $O/r05_bench.json 2> $O/r05_bench.err; echo "bench rc=$?"
python bench.py --config stress --steps 5 > $O/r05_stress_bench.json 2> $O/r05_stress.err; echo "stress rc=$?"
python bench.py --config cfg2 --no-cpu-baseline > $O/r05_bench_cfg2.json 2> $O/r05_cfg2.err; echo "cfg2 rc=$?"
rm -rf $O/kt_head $O/kt_stress
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_head -- python3 bench.py --no-cpu-baseline > /dev/null 2>&1
cp $(ls $O/kt_head/*/*kernel_stats.csv | head -1) $O/r05_bench_kernel_stats.csv
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_stress -- python3 bench.py --config stress --steps 5 > /dev/null 2>&1
cp $(ls $O/kt_stress/*/*kernel_stats.csv | head -1) $O/r05_stress_kernel_stats.csv
rm -rf $O/kt_head $O/kt_stress
ls -la $O/r05_*
