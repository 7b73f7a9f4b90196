#!/bin/bash
# K4s: gathers through a per-wave LDS ring (MCD_WPMI_BF16_RING = 2 / 4 quads) against the register path (0): bit-equality of
# the stress chain's result and the stage times.  HISTORICAL: the ring variant (and its environment switch) only exists on the
# commit "K4s: per-(neuron, rank) meta array ..."; it was removed after profiles/r03_k4s_ring.txt / r03_gather_path.txt.
# scripts/k4s_hash.py checks the current kernel against the same output hashes.
set -e
out=gpurun_out/r03_k4s_ring.txt
: > $out
for r in 0 2 4; do
  echo "== MCD_WPMI_BF16_RING=$r" >> $out
  MCD_WPMI_BF16_RING=$r timeout -k 10 200 python3 bench.py --config stress --steps 5 --warmup 2 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('   stress ms', d['ms_per_step'], d['stage_ms'])" >> $out
  MCD_WPMI_BF16_RING=$r timeout -k 10 200 python3 - >> $out <<'PY'
import torch, hashlib, sys
sys.path.insert(0, '.')
import mammo_clip_dissect_amd
from mammo_clip_dissect_amd import core
dev = torch.device('cuda:0')
g = torch.Generator(device=dev).manual_seed(5)
for (N, C, U, K) in [(3000, 1000, 300, 100), (777, 333, 50, 37), (5000, 2560, 129, 16), (64, 128, 7, 3)]:
    I = core.normalize_rows(torch.randn(N, 512, device=dev, generator=g)); T = core.normalize_rows(torch.randn(C, 512, device=dev, generator=g))
    E, rinv = core.embed_gemm_exp(I, T, 10.0)
    idx = torch.stack([torch.randperm(N, device=dev, generator=g)[:K] for _ in range(U)]).to(torch.int32)
    p = torch.linspace(0.998, 0.97, K, device=dev)
    outs = core.wpmi_score_bf16(E, rinv, idx, p, 1e-9, soft=True)
    outh = core.wpmi_score_bf16(E, rinv, idx, None, 1e-9, soft=False)
    print('   ', (N, C, U, K), hashlib.sha256(outs.cpu().numpy().tobytes()).hexdigest()[:16], hashlib.sha256(outh.cpu().numpy().tobytes()).hexdigest()[:16])
PY
done
cat $out
