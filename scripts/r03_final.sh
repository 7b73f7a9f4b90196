#!/bin/bash
# Round-3 evidence on the final tree (run on the GPU box from the repo root): parity stats, headline bench + kernel stats,
# stress bench + kernel stats, PMC traffic of both shapes, K4 SQ counters.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out
rm -f $O/r03_parity_stats.txt
MCD_STATS_FILE=$PWD/$O/r03_parity_stats.txt timeout -k 10 600 python -m pytest tests -m gpu -x -q > $O/r03_gputests.log 2>&1; echo "tests rc=$?"; tail -2 $O/r03_gputests.log
bash scripts/r03_pmc.sh all > $O/r03_pmc.log 2>&1; echo "pmc rc=$?"
python bench.py > $O/r03_bench.json 2> $O/r03_bench.err; echo "bench rc=$?"
python bench.py --config stress --steps 5 > $O/r03_stress_bench.json 2> $O/r03_stress.err; echo "stress rc=$?"
rm -rf $O/kt_head $O/kt_stress
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_head -- python3 bench.py --no-cpu-baseline > /dev/null 2>&1
cp $(ls $O/kt_head/*/*kernel_stats.csv | head -1) $O/r03_bench_kernel_stats.csv
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_stress -- python3 bench.py --config stress --steps 5 > /dev/null 2>&1
cp $(ls $O/kt_stress/*/*kernel_stats.csv | head -1) $O/r03_stress_kernel_stats.csv
rm -rf $O/kt_head $O/kt_stress
ls -la $O/r03_*
