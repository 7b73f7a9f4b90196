#!/bin/bash
# K4s slice width experiment: MCD_WPMI_BF16_LPN = 16 (128-concept slices) / 8 (64-concept slices)
for lpn in 16 8; do
  MCD_WPMI_BF16_LPN=$lpn python bench.py --config stress --steps 5 2>/dev/null > /tmp/s_$lpn.json
  python - <<PY
import json
j = json.load(open("/tmp/s_$lpn.json"))
print("lpn $lpn: wpmi stage %.4f  K4s launch %.4f  core %.4f" % (j["stage_ms"]["wpmi"], j["roofline"]["avg_launch_ms"], j["core_ms"]))
PY
done
