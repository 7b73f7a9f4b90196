#!/bin/bash
# K1s: cache-policy bits on the E stores (built as libmcd_hip variants with -DMCD_GEXP_STORE_AUX=n under scripts/micro/_build),
# the GEMM call alone and the whole stress pass (K4s gathers E out of L2 / the Infinity Cache afterwards).
set -e
out=gpurun_out/r03_gexp_store_policy.txt
: > $out
for a in 0 1 2 16 17; do
  if [ $a = 0 ]; then unset MCD_LIB_PATH; else export MCD_LIB_PATH=$PWD/scripts/micro/_build/libmcd_hip_aux$a.so; fi
  echo "== store aux $a" >> $out
  MCD_PROF_LIBRARY=0 timeout -k 10 120 python3 scripts/prof_gemm_exp.py 25000 10000 20 2>/dev/null | grep embed_gemm_exp >> $out
  timeout -k 10 200 python3 bench.py --config stress --steps 5 --warmup 2 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('   stress ms', d['ms_per_step'], d['stage_ms'])" >> $out
done
cat $out
