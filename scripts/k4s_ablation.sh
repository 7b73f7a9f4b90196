#!/bin/bash
# K4s ablations (dev builds under scripts/micro/, see MCD_K4S_ABL in k_wpmi.hip): 1 = no rinv gather, 2 = no product/log
python scripts/prof_k4s.py 10 || exit 1
for a in 1 2 3; do
  MCD_LIB_PATH=$PWD/scripts/micro/libmcd_k4s_abl$a.so python scripts/prof_k4s.py 10 || exit 1
done
