"""Randomised differential test of the five similarity functions of the drop-in module (concept_vit/similarity.py mirror:
soft_wpmi, wpmi, cos_similarity, cos_similarity_cubed, rank_reorder under one seed) against the CPU oracle on random
(N, C, U) -- tolerances as in tests/test_gpu_e2e.py.  argv: [cases] [seed]"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import mammo_clip_dissect_amd  # noqa: F401
from mammo_clip_dissect_amd.concept_vit import similarity as sim
import oracle as O
import util

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
rng = np.random.default_rng(seed)
dev = "cuda:0"
bad = 0
for c in range(cases):
    N = int(rng.integers(100, 3000))
    C = int(rng.choice([5, 31, 32, 33, 64, 100, 255, 763, 1000, 1030]))
    U = int(rng.integers(1, 60))
    g = torch.Generator().manual_seed(seed * 7919 + c)
    tag = "N=%d C=%d U=%d" % (N, C, U)
    try:
        P = torch.randn(N, C, generator=g) * 0.05
        A = torch.randn(N, U, generator=g)
        K = int(rng.choice([28, 100]))
        util.assert_sim_boundary(sim.soft_wpmi(P, A, top_k=K, device=dev).cpu().numpy(), O.soft_wpmi(P.numpy(), A.numpy(), top_k=K), "soft " + tag)
        util.assert_sim_boundary(sim.wpmi(P, A, top_k=28, device=dev).cpu().numpy(), O.wpmi(P.numpy(), A.numpy(), top_k=28), "wpmi " + tag)
        d = np.abs(sim.cos_similarity(P, A, device=dev).cpu().numpy() - O.cos_similarity(P.numpy(), A.numpy())).max()
        d3 = np.abs(sim.cos_similarity_cubed(P, A, device=dev).cpu().numpy() - O.cos_similarity_cubed(P.numpy(), A.numpy())).max()
        assert d <= 5e-7 and d3 <= 5e-7, ("cos", d, d3)
        Ps = torch.softmax(4 * torch.randn(N, C, generator=g), dim=1)
        torch.manual_seed(c)
        ref = O.rank_reorder(Ps.numpy(), A.numpy())
        torch.manual_seed(c)
        got = sim.rank_reorder(Ps, A, device=dev).cpu().numpy()
        assert np.array_equal(np.isnan(got), np.isnan(ref)), "rank_reorder NaN pattern"
        m = ~np.isnan(ref)
        err = np.abs(got[m] - ref[m]) - 5e-6 * np.abs(ref[m])
        assert err.max() <= 1e-9, ("rank_reorder", float(err.max()))
    except AssertionError as e:
        bad += 1
        print("MISMATCH %s: %s" % (tag, str(e)[:200]), flush=True)
    if (c + 1) % 20 == 0:
        print("%d cases, %d mismatches" % (c + 1, bad), flush=True)
print("done: %d cases, %d mismatches" % (cases, bad))
sys.exit(1 if bad else 0)
