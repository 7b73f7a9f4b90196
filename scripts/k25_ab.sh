#!/bin/bash
# K2 / K5 timing (scripts/core_timing.py lines) per library variant: VARIANTS="label=path ..." (default: the product library)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04_k25_ab.txt
: > $O
for var in ${VARIANTS:-product=}; do
  label=${var%%=*}; lib=${var#*=}
  for rep in 1 2; do
    ( [ -n "$lib" ] && export MCD_LIB_PATH=$PWD/mammo-clip-dissect_amd/csrc/$lib
      timeout -k 10 200 python scripts/core_timing.py 2>/dev/null | grep "K2 row_softmax\|K5 logsumexp\|K6 row_topk k=10\|core, eager" | sed "s/^/$label  /" >> $O )
  done
done
cat $O
