#!/bin/bash
# K4s: one v_log_f32 per row (MCD_WPMI_BF16_GROUP=1) against one per product of four rows' arguments (default)
export MCD_LIB_PATH=$PWD/mammo-clip-dissect_amd/csrc/libmcd_hip_dev.so   # the knob lives in the dev build (make dev)
for g in 1 4; do
  MCD_WPMI_BF16_GROUP=$g python bench.py --config stress --steps 5 2>/dev/null > /tmp/s_g.json || exit 1
  python - <<PY
import json
j = json.load(open("/tmp/s_g.json"))
print("group $g: wpmi stage %.4f  K4s launch %.4f  core %.4f" % (j["stage_ms"]["wpmi"], j["roofline"]["avg_launch_ms"], j["core_ms"]))
PY
done
