#!/bin/bash
# does K4 (parity mode) run on its LDS bank conflicts or on VALU issue?  PMC pair: the product build against a
# TIMING-ONLY build whose table lookups cannot conflict (-DMCD_K4_ABL_NOCONFLICT: every lane reads its own lane's entry
# through the same instruction; results are wrong, instruction counts identical).  Run on the GPU box from the repo root.
# Build of the ablation library (in the build container):
#   cd mammo-clip-dissect_amd/csrc && hipcc $(HIPFLAGS) -DMCD_K4_ABL_NOCONFLICT -c k_wpmi.hip -o /tmp/k_wpmi_noconf.o &&
#   hipcc --offload-arch=gfx950 -shared -fPIC -o ../../scripts/micro/libmcd_k4_noconf.so <the other objects> /tmp/k_wpmi_noconf.o
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03_k4_lds_ablation.txt
: > $O
for which in product noconflict; do
  if [ $which = noconflict ]; then export MCD_LIB_PATH=$PWD/scripts/micro/libmcd_k4_noconf.so; else unset MCD_LIB_PATH; fi
  D=gpurun_out/pmc_k4_$which; rm -rf $D
  rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE -d $D/sq -- python3 scripts/prof_k4.py trusted > /dev/null 2>&1 &&
  rocprofv3 --kernel-trace --stats --output-format csv -d $D/trace -- python3 scripts/prof_k4.py trusted > /dev/null 2>&1
  echo "== $which build" >> $O
  python3 scripts/pmc_db.py $D/sq wpmi_slice >> $O 2>&1
  python3 - $D/trace >> $O <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "wpmi_slice" in r["Name"]:
            print("   un-profiled kernel trace: %s calls, average %.1f us" % (r["Calls"], float(r["AverageNs"]) / 1e3))
PY
  rm -rf $D
done
cat $O
