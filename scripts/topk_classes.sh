#!/bin/bash
# K3 workgroup classes (threads, 16-byte quads per thread) at N = 10 000 and 20 000 / 25 000 (MCD_TOPK_CLASS dev knob)
export MCD_LIB_PATH=$PWD/mammo-clip-dissect_amd/csrc/libmcd_hip_dev.so   # the knob lives in the dev build (make dev)
cat > /tmp/k3c.py <<'PY'
import sys, os
sys.path.insert(0, os.getcwd())
import torch
import mammo_clip_dissect_amd
from mammo_clip_dissect_amd import core
N = int(sys.argv[1]); U = 9216
At = torch.randn(U, N, device="cuda:0")
for _ in range(3): core.col_topk(At, 100, neuron_major=True)
torch.cuda.synchronize()
s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(10): core.col_topk(At, 100, neuron_major=True)
e.record(); torch.cuda.synchronize()
ms = s.elapsed_time(e) / 10
print("N=%d class %s: %.4f ms %.2f TB/s" % (N, os.environ.get("MCD_TOPK_CLASS", "default"), ms, 4.0 * N * U / ms / 1e9))
PY
for c in default 256,10 512,5 512,6; do
  if [ $c = default ]; then python /tmp/k3c.py 10000; else MCD_TOPK_CLASS=$c python /tmp/k3c.py 10000; fi
done
for c in default 512,10 768,7 1024,5; do
  if [ $c = default ]; then python /tmp/k3c.py 20000; else MCD_TOPK_CLASS=$c python /tmp/k3c.py 20000; fi
done
for c in default 512,13 768,9 1024,7; do
  if [ $c = default ]; then python /tmp/k3c.py 25000; else MCD_TOPK_CLASS=$c python /tmp/k3c.py 25000; fi
done
