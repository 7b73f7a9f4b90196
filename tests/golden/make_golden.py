#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ from the REFERENCE itself.

Run in the build container only (needs /root/reference; the GPU box never has it):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

What is imported from the reference: concept_vit/similarity.py (the five similarity functions;
it needs only torch/scipy/tqdm, all present).  The orchestration files (utils.py, describe_*.py)
cannot be imported here (torchvision / timm / wandb / albumentations are absent), so the few torch
calls they make around the similarity function are issued directly below, each next to the
reference line it stands for.  Outputs are data only: inputs, expected outputs, CSV bytes.

The reference ships no golden vectors of its own, so these files are the pin
for oracle/ and for the HIP path.  All inputs are asserted tie-free (torch.topk's tie order is
unspecified; the build defines lowest-index-first and does not claim parity on ties).
"""
import io
import json
import os
import sys

sys.dont_write_bytecode = True
REF = os.environ.get("MCD_REFERENCE", "/root/reference")
sys.path.insert(0, os.path.join(REF, "concept_vit"))

import numpy as np
import pandas as pd
import torch

import similarity as ref_sim  # noqa: E402  (the reference module)

HERE = os.path.dirname(os.path.abspath(__file__))
torch.set_num_threads(8)


def unit_embeddings(n, d, seed):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(n, d, generator=g)


def make_inputs(N, C, U, D, seed, act="gauss"):
    """E_img [N,D], E_txt [C,D], A [N,U]; P as utils.py:577-594 computes it."""
    E_img = unit_embeddings(N, D, seed)
    E_txt = unit_embeddings(C, D, seed + 1)
    g = torch.Generator().manual_seed(seed + 2)
    A = torch.randn(N, U, generator=g)
    if act == "relu_like":  # positive, skewed activations like pooled CNN channels; still tie-free
        A = torch.nn.functional.softplus(A * 2.0) + 1e-3 * torch.rand(N, U, generator=g)
    image_features = E_img.clone().float()
    text_features = E_txt.clone().float()
    with torch.no_grad():
        image_features /= image_features.norm(dim=-1, keepdim=True)  # utils.py:577
        text_features /= text_features.norm(dim=-1, keepdim=True)    # utils.py:578
        P = image_features @ text_features.T                         # utils.py:594
    return E_img, E_txt, A, P


def assert_tie_free(A, sim=None, k=10):
    for u in range(A.shape[1]):
        assert len(torch.unique(A[:, u])) == A.shape[0], "tie in activation column %d" % u
    if sim is not None:
        v, _ = torch.topk(sim, k=min(k + 1, sim.shape[1]), dim=1)
        # similarity rows are NOT forced tie-free: gaps of one fp32 ulp (3e-5) occur naturally and the
        # U == 1 case is identically zero.  The smallest gap is recorded so the tests can state on which
        # rows an exact integer match is well defined.
        return (v[:, :-1] - v[:, 1:]).min().item()
    return None


def run_quiet(fn, *a, **kw):
    out, err = sys.stdout, sys.stderr
    sys.stdout = io.StringIO()
    sys.stderr = io.StringIO()
    try:
        return fn(*a, **kw)
    finally:
        sys.stdout, sys.stderr = out, err


def post_og(sim, A):
    # describe_og_neurons.py:99-100 / describe_broad_neurons.py:101-102
    vals, ids = torch.topk(sim, k=min(10, sim.shape[1]), dim=1)
    _, top_ids = torch.topk(A, k=5, dim=0)
    return vals, ids, top_ids


def post_clip(sim, A):
    # describe_clip_neurons.py:64-66
    vals, ids = torch.max(sim, dim=1)
    _, top_ids = torch.topk(A, k=5, dim=0)
    return vals, ids, top_ids


def csv_og(layers, words):
    """describe_broad_neurons.py:79-122 (rows appended per layer, then DataFrame.to_csv)."""
    outputs = {"layer": [], "unit": [], "description": [], "similarity": [], "images": []}
    for name, sim, A in layers:
        vals, ids, top_ids = post_og(sim, A)
        descriptions = []
        for id in ids:
            descriptions.append([words[int(idx)] for idx in id])
        outputs["unit"].extend([i for i in range(len(vals))])
        outputs["layer"].extend([name] * len(vals))
        outputs["description"].extend(descriptions)
        outputs["similarity"].extend(vals.cpu().numpy())
        outputs["images"].extend(top_ids.T.cpu().numpy())
    buf = io.StringIO()
    pd.DataFrame(outputs).to_csv(buf, index=False)
    return buf.getvalue()


def csv_clip(layers, words):
    """describe_clip_neurons.py:49-91."""
    outputs = {"layer": [], "unit": [], "description": [], "similarity": [], "images": []}
    for name, sim, A in layers:
        vals, ids, top_ids = post_clip(sim, A)
        descriptions = [words[int(idx)] for idx in ids]
        outputs["unit"].extend([i for i in range(len(vals))])
        outputs["layer"].extend([name] * len(vals))
        outputs["description"].extend(descriptions)
        outputs["similarity"].extend(vals.cpu().numpy())
        outputs["images"].extend(top_ids.T.cpu().numpy())
    buf = io.StringIO()
    pd.DataFrame(outputs).to_csv(buf, index=False)
    return buf.getvalue()


def main():
    meta = {"torch": torch.__version__, "cpu_capability": torch.backends.cpu.get_cpu_capability(),
            "threads": torch.get_num_threads(), "cases": {}}
    with open(os.path.join(REF, "Concepts", "Specific_concepts_sorted.txt")) as f:
        words = f.read().split("\n")  # describe_clip_neurons.py:50-51
    assert len(words) == 763

    # ---- p_in_examples (similarity.py:58) -------------------------------------------------------
    top_k, p_start, p_end = 100, 0.998, 0.97
    p100 = (p_start - (torch.arange(start=0, end=top_k) / top_k * (p_start - p_end)).unsqueeze(1)).numpy()
    np.savez(os.path.join(HERE, "p_in_examples.npz"), p100=p100.astype(np.float32))

    # ---- similarity-function cases -------------------------------------------------------------
    cases = [
        # name         N     C    U   D    K    seed  act
        ("tiny",       128,  5,   7,  16,  5,   11,   "gauss"),
        ("main",       256,  763, 64, 512, 100, 21,   "gauss"),
        ("relu",       300,  763, 33, 512, 100, 31,   "relu_like"),
        ("kfull",      100,  40,  9,  64,  100, 41,   "gauss"),   # K == N: every image selected
        ("one_neuron", 160,  763, 1,  512, 100, 51,   "gauss"),   # U == 1: logsumexp over one row
        ("n1000",      1000, 763, 48, 512, 100, 61,   "gauss"),
    ]
    for name, N, C, U, D, K, seed, act in cases:
        E_img, E_txt, A, P = make_inputs(N, C, U, D, seed, act)
        out = run_quiet(ref_sim.soft_wpmi, P, A, top_k=K, device="cpu")
        pdge = run_quiet(ref_sim.soft_wpmi, P, A, top_k=K, lam=0, device="cpu")  # lam=0 -> prob_d_given_e
        S = torch.nn.functional.softmax(10 * P, dim=1)           # similarity.py:54
        inds = torch.topk(A, dim=0, k=K)[1]                      # similarity.py:55
        gap = assert_tie_free(A, out)
        vals10, ids10, top5 = post_og(out, A)
        vmax, imax, _ = post_clip(out, A)
        d = dict(E_img=E_img.numpy(), E_txt=E_txt.numpy(), A=A.numpy(), P=P.numpy(), S=S.numpy(),
                 inds=inds.numpy(), pdge=pdge.numpy(), soft_wpmi=out.numpy(), vals10=vals10.numpy(),
                 ids10=ids10.numpy(), top5=top5.numpy(), vmax=vmax.numpy(), imax=imax.numpy(),
                 top_k=np.int64(K))
        info = {"N": N, "C": C, "U": U, "D": D, "K": K, "seed": seed, "act": act, "min_top10_gap": gap}
        if name in ("main", "tiny", "relu"):
            Kw = min(28, N)
            w = run_quiet(ref_sim.wpmi, P, A, top_k=Kw, device="cpu")
            d["wpmi"] = w.numpy()
            d["wpmi_top_k"] = np.int64(Kw)
            d["cos_similarity"] = run_quiet(ref_sim.cos_similarity, P, A, device="cpu").numpy()
            d["cos_similarity_cubed"] = run_quiet(ref_sim.cos_similarity_cubed, P, A, device="cpu").numpy()
        if name == "main":
            torch.manual_seed(1234)  # rank_reorder uses torch.randperm (similarity.py:119)
            d["rank_reorder_seed1234"] = run_quiet(ref_sim.rank_reorder, P, A, device="cpu").numpy()
        if name == "n1000":
            # large inputs are regenerated from the seed by the tests; keep outputs + the 512-d embeddings out
            for k_ in ("E_img", "E_txt", "P", "S"):
                d.pop(k_)
            d["P_checksum"] = np.float64(P.double().sum().item())
            d["A_checksum"] = np.float64(A.double().sum().item())
            d.pop("A")
        np.savez_compressed(os.path.join(HERE, "sim_%s.npz" % name), **d)
        meta["cases"][name] = info
        print(name, info)

    # ---- CSV contract (both driver variants), two "layers" from the main + relu cases ------------
    layers = []
    for lname, cname in (("layer_a", "main"), ("layer_b", "relu")):
        z = np.load(os.path.join(HERE, "sim_%s.npz" % cname))
        layers.append((lname, torch.from_numpy(z["soft_wpmi"]), torch.from_numpy(z["A"])))
    with open(os.path.join(HERE, "descriptions_og.csv"), "w", newline="") as f:
        f.write(csv_og(layers, words))
    with open(os.path.join(HERE, "descriptions_clip.csv"), "w", newline="") as f:
        f.write(csv_clip(layers, words))

    with open(os.path.join(HERE, "golden_meta.json"), "w") as f:
        json.dump(meta, f, indent=2)
    main_round2(words)


def _sha(t):
    import hashlib
    return hashlib.sha256(np.ascontiguousarray(t.numpy() if hasattr(t, "numpy") else t).tobytes()).hexdigest()


def main_round2(words=None):
    """Round-2 additions (`python make_golden.py --round2` writes only these; the round-1 files stay untouched):

    shared2  the DRIVER's shape -- one probe set, one P, two target layers (describe_broad_neurons.py:83-116 loops
             the layers over the same clip/text files): probe set and layer 0 are the `main` case, layer 1 is a second
             activation matrix over the same 256 images.  CSV bytes of both driver variants for the two layers.
    n10k     one layer at a REAL size (configs[1]: 10 000 images x 768 neurons x 763 concepts, top_k 100): outputs
             only; the tests regenerate the inputs from the seed and P by the oracle's restatement of this host's
             normalise + matmul order, pinned here by sha256 of P's bytes.
    """
    if words is None:
        with open(os.path.join(REF, "Concepts", "Specific_concepts_sorted.txt")) as f:
            words = f.read().split("\n")
    meta_path = os.path.join(HERE, "golden_meta.json")
    meta = json.load(open(meta_path))

    # ---- shared2 -----------------------------------------------------------------------------------
    N, C, U0, D, K, seed = 256, 763, 64, 512, 100, 21               # == the `main` case
    E_img, E_txt, A0, P = make_inputs(N, C, U0, D, seed, "gauss")
    zmain = np.load(os.path.join(HERE, "sim_main.npz"))
    assert np.array_equal(zmain["P"], P.numpy()) and np.array_equal(zmain["A"], A0.numpy())
    g = torch.Generator().manual_seed(seed + 7)
    A1 = torch.nn.functional.softplus(torch.randn(N, 24, generator=g) * 2.0) + 1e-3 * torch.rand(N, 24, generator=g)
    sim0 = torch.from_numpy(zmain["soft_wpmi"])
    sim1 = run_quiet(ref_sim.soft_wpmi, P, A1, top_k=K, device="cpu")
    gap = assert_tie_free(A1, sim1)
    v10, i10, t5 = post_og(sim1, A1)
    np.savez_compressed(os.path.join(HERE, "sim_shared2.npz"), A1=A1.numpy(), soft_wpmi1=sim1.numpy(), vals10=v10.numpy(),
                        ids10=i10.numpy(), top5=t5.numpy(), top_k=np.int64(K))
    layers = [("layer_a", sim0, A0), ("layer_b", sim1, A1)]
    with open(os.path.join(HERE, "descriptions_shared2_og.csv"), "w", newline="") as f:
        f.write(csv_og(layers, words))
    with open(os.path.join(HERE, "descriptions_shared2_clip.csv"), "w", newline="") as f:
        f.write(csv_clip(layers, words))
    meta["cases"]["shared2"] = {"N": N, "C": C, "U": [U0, 24], "D": D, "K": K, "seed": seed,
                                "probe": "the main case's E_img/E_txt/P; layer_a = main's A", "min_top10_gap": gap}
    print("shared2", meta["cases"]["shared2"])

    # ---- n10k --------------------------------------------------------------------------------------
    # 10 000 gaussian fp32 values per column collide somewhere with high probability; what torch.topk's (unspecified)
    # tie order can touch is the top K+1 of a column only, so that is what must be tie-free: take the first seed
    # from 71 on for which it is.
    N, C, U, D, K = 10000, 763, 768, 512, 100
    for seed in range(71, 91):
        E_img, E_txt, A, P = make_inputs(N, C, U, D, seed, "gauss")
        top = torch.topk(A, K + 1, dim=0).values
        if bool((top[:-1] > top[1:]).all()):
            break
    else:
        raise SystemExit("no tie-free seed")
    out = run_quiet(ref_sim.soft_wpmi, P, A, top_k=K, device="cpu")
    v_, _ = torch.topk(out, k=11, dim=1)
    gap = (v_[:, :-1] - v_[:, 1:]).min().item()
    v10, i10, t5 = post_og(out, A)
    vmax, imax, _ = post_clip(out, A)
    np.savez_compressed(os.path.join(HERE, "sim_n10k.npz"), soft_wpmi=out.numpy(), vals10=v10.numpy(), ids10=i10.numpy(),
                        top5=t5.numpy(), vmax=vmax.numpy(), imax=imax.numpy(), top_k=np.int64(K),
                        P_sha256=np.array(_sha(P)), A_sha256=np.array(_sha(A)), E_img_sha256=np.array(_sha(E_img)),
                        E_txt_sha256=np.array(_sha(E_txt)))
    meta["cases"]["n10k"] = {"N": N, "C": C, "U": U, "D": D, "K": K, "seed": seed, "act": "gauss", "min_top10_gap": gap}
    print("n10k", meta["cases"]["n10k"])
    with open(meta_path, "w") as f:
        json.dump(meta, f, indent=2)


def main_fuzz(cases_from_campaign=42, unit_cases=10):
    """Round-5 addition (`python make_golden.py --fuzz`): the reference's OWN outputs on off-golden shapes, so that
    the HIP path is pinned to the reference directly there and not through the oracle (whose exp / log differ from MKL's in the
    last bit about once per 2e5 sums).

    The shapes and inputs are those of scripts/fuzz_sim.py's campaign `500 77` (same numpy stream for (N, C, U, K), same torch
    generator seed 77 * 7919 + c for P = 0.05 randn, A = randn): every 12th case plus the three the round-4 campaign flagged
    (N, C, U) = (1651, 763, 45), (1107, 763, 38), (410, 255, 48); and `unit_cases` more with P from unit-norm 512-d embeddings
    (make_inputs, i.e. what the drivers feed).  Stored per case: the generator recipe, sha256 of the regenerated P and A, and the
    reference's soft_wpmi (top_k K) and wpmi (top_k 28) outputs with torch.max / torch.topk(10) of each (data only)."""
    rng = np.random.default_rng(77)
    flagged = {(1651, 763, 45), (1107, 763, 38), (410, 255, 48)}
    picked, recipes = 0, []
    for c in range(500):
        N = int(rng.integers(100, 3000))
        C = int(rng.choice([5, 31, 32, 33, 64, 100, 255, 763, 1000, 1030]))
        U = int(rng.integers(1, 60))
        K = int(rng.choice([28, 100]))
        if (N, C, U) in flagged or (c % 12 == 0 and picked < cases_from_campaign):
            recipes.append(dict(kind="scaled", c=c, N=N, C=C, U=U, K=K, gen_seed=77 * 7919 + c))
            picked += (N, C, U) not in flagged
    rng2 = np.random.default_rng(7705)
    for j in range(unit_cases):
        N = int(rng2.integers(150, 2500))
        C = int(rng2.choice([255, 763, 1000, 1030]))
        U = int(rng2.integers(2, 48))
        K = int(rng2.choice([28, 100]))
        recipes.append(dict(kind="unit", c=j, N=N, C=C, U=U, K=K, gen_seed=9100 + 3 * j))
    out, meta = {}, []
    for i, r in enumerate(recipes):
        if r["kind"] == "scaled":
            g = torch.Generator().manual_seed(r["gen_seed"])
            P = torch.randn(r["N"], r["C"], generator=g) * 0.05
            A = torch.randn(r["N"], r["U"], generator=g)
        else:
            _, _, A, P = make_inputs(r["N"], r["C"], r["U"], 512, r["gen_seed"], "gauss")
        top = torch.topk(A, r["K"] + 1, dim=0).values if r["N"] > r["K"] else None
        r["topk_tie_free"] = bool((top[:-1] > top[1:]).all()) if top is not None else True
        assert r["topk_tie_free"], r
        soft = run_quiet(ref_sim.soft_wpmi, P, A, top_k=r["K"], device="cpu")
        hard = run_quiet(ref_sim.wpmi, P, A, top_k=28, device="cpu")
        kk = min(10, r["C"])
        sv, si = torch.topk(soft, k=kk, dim=1)
        hv, hi = torch.topk(hard, k=kk, dim=1)
        r["P_sha256"], r["A_sha256"] = _sha(P), _sha(A)
        out["soft_%d" % i], out["wpmi_%d" % i] = soft.numpy(), hard.numpy()
        out["soft_ids10_%d" % i], out["wpmi_ids10_%d" % i] = si.numpy().astype(np.int32), hi.numpy().astype(np.int32)
        out["soft_vals10_%d" % i], out["wpmi_vals10_%d" % i] = sv.numpy(), hv.numpy()
        meta.append(r)
        print(i, r["kind"], r["N"], r["C"], r["U"], r["K"], flush=True)
    np.savez_compressed(os.path.join(HERE, "sim_fuzz.npz"), **out)
    with open(os.path.join(HERE, "sim_fuzz_meta.json"), "w") as f:
        json.dump({"torch": torch.__version__, "cpu_capability": torch.backends.cpu.get_cpu_capability(), "cases": meta}, f, indent=1)
    print("fuzz cases:", len(meta))


if __name__ == "__main__":
    if "--fuzz" in sys.argv:
        main_fuzz()
    elif "--round2" in sys.argv:
        main_round2()
    else:
        main()
