"""K3 (neuron_topk_fast_kernel) phase anatomy from in-kernel s_memtime stamps.  Needs the stamped dev build:
hipcc ... -DMCD_K3_STAMPS -c k_topk.hip, linked to scripts/micro/libmcd_k3_stamps.so; run with MCD_LIB_PATH pointing to it.
argv: [N] [U]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import mammo_clip_dissect_amd  # noqa: F401
from mammo_clip_dissect_amd import core

N = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
U = int(sys.argv[2]) if len(sys.argv) > 2 else 9216
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
At = torch.randn(U, N, device=dev, generator=g)
for _ in range(3):
    vals, idx = core.col_topk(At, 100, neuron_major=True)
torch.cuda.synchronize()
st = vals.cpu().numpy().view(np.int64)[:, :6].astype(np.float64)      # 6 stamps per workgroup (s_memtime ticks)
t0 = st[:, 0].min()
names = ["load+keys+max", "barrier 1", "search (1 wave) + barrier 2", "compaction + barrier 3", "rank + store"]
d = np.diff(st, axis=1)
print("N=%d U=%d: workgroup lifetime median %.0f ticks, p10 %.0f, p90 %.0f" % ((N, U) + tuple(np.percentile(st[:, 5] - st[:, 0], [50, 10, 90]))))
for i, n in enumerate(names):
    print("  %-32s median %7.0f  p10 %7.0f  p90 %7.0f" % ((n,) + tuple(np.percentile(d[:, i], [50, 10, 90]))))
span = st[:, 5].max() - t0
print("kernel span %.0f ticks; sum of lifetimes / span = %.1f workgroups resident on average" % (span, (st[:, 5] - st[:, 0]).sum() / span))
