"""Shared helpers for the parity tests: golden loading and the stated tolerances."""
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CASES = ["tiny", "main", "relu", "kfull", "one_neuron", "n1000"]

# ---- tolerances (DESIGN.md section 6 explains each) ------------------------------------------------
# north_star: "similarity scores within 1e-4 fp32".  The scores are differences of two fp32 numbers of
# magnitude 256..1024 (prob_d_given_e and prob_d), whose ulp is 3.05e-5 .. 6.1e-5, so 1e-4 is 1.6 ulp of
# the intermediates: an implementation whose exp/log differ from SLEEF's in the last bit lands exactly on
# the oracle for >99.9% of the entries and 1 or 2 intermediate ulps away on the rest.
SIM_BOUNDARY_ATOL = 6.2e-5   # soft_wpmi(P, A) at the drop-in boundary (same P as the reference): ONE ulp of the
                             # intermediates; nothing may exceed it, and SIM_BOUNDARY_EXACT of the entries are bit-identical
SIM_BOUNDARY_EXACT = 0.9995   # measured: 99.998 % (main: 1 entry of 48 832), 100 % on every other case
SIM_ATOL = 1e-4          # the stated tolerance; must hold for all but SIM_OUTLIER_FRAC of the entries
SIM_OUTLIER_FRAC = 1e-3  # entries allowed between SIM_ATOL and SIM_HARD_ATOL
SIM_HARD_ATOL = 2.5e-4   # 4 ulp of an intermediate in [512, 1024): nothing may exceed this
SIM_MEAN_ATOL = 2e-5
ARGMAX_GAP = 2.5e-4      # integer matches are asserted on rows whose reference gap exceeds this
P_ATOL = 5e-7            # P = I_hat @ T_hat^T entries are in [-1, 1]; fp32 dot of 512 terms
S_RTOL = 2e-6            # softmax entries, relative
PDGE_ATOL = 1.3e-4       # 2 ulp at [512,1024) for the un-normalised sums given identical inputs


def golden(name):
    return np.load(os.path.join(GOLDEN, "sim_%s.npz" % name))


def regen_n1000():
    """Inputs of the n1000 case are regenerated from the seed (make_golden.py: make_inputs).  P is recomputed with
    the oracle's restatement of the fixture host's normalise + matmul order (bit-identical to every stored P): torch's
    own matmul on the host running the tests may cut K differently (MKL picks per CPU model)."""
    import sys
    import torch
    sys.path.insert(0, os.path.join(os.path.dirname(GOLDEN), os.pardir, "oracle"))
    import oracle as O
    N, C, U, D, seed = 1000, 763, 48, 512, 61
    g = torch.Generator().manual_seed(seed)
    E_img = torch.randn(N, D, generator=g)
    g = torch.Generator().manual_seed(seed + 1)
    E_txt = torch.randn(C, D, generator=g)
    g = torch.Generator().manual_seed(seed + 2)
    A = torch.randn(N, U, generator=g)
    P = O.embed_gemm(E_img.numpy(), E_txt.numpy(), blas=False)
    return E_img.numpy(), E_txt.numpy(), A.numpy(), P


def regen_inputs(N, C, U, D, seed):
    """make_golden.py: make_inputs(..., act="gauss") regenerated from the seed; P by the oracle's restatement of the
    fixture host's normalise + matmul order."""
    import sys
    import torch
    sys.path.insert(0, os.path.join(os.path.dirname(GOLDEN), os.pardir, "oracle"))
    import oracle as O
    E_img = torch.randn(N, D, generator=torch.Generator().manual_seed(seed)).numpy()
    E_txt = torch.randn(C, D, generator=torch.Generator().manual_seed(seed + 1)).numpy()
    A = torch.randn(N, U, generator=torch.Generator().manual_seed(seed + 2)).numpy()
    return E_img, E_txt, A, O.embed_gemm(E_img, E_txt, blas=False)


def sha256(a):
    import hashlib
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


_N10K = {}


def n10k_inputs():
    """The real-size golden (configs[1]'s layer shape: 10 000 images x 768 neurons x 763 concepts): inputs from the
    seed, every one of them checked against the sha256 the generator stored (P: the bits torch computed there)."""
    if not _N10K:
        z = golden("n10k")
        meta = __import__("json").load(open(os.path.join(GOLDEN, "golden_meta.json")))["cases"]["n10k"]
        E_img, E_txt, A, P = regen_inputs(meta["N"], meta["C"], meta["U"], meta["D"], meta["seed"])
        for name, arr in (("E_img", E_img), ("E_txt", E_txt), ("A", A), ("P", P)):
            assert sha256(arr) == str(z[name + "_sha256"]), name
        _N10K.update(z=z, E_img=E_img, E_txt=E_txt, A=A, P=P, K=int(z["top_k"]))
    return _N10K


def assert_top10_decided(got_ids, got_vals, ref_ids, ref_vals, what=""):
    """Top-k lists given only the reference's top-k VALUES (no full similarity matrix): rank j is decided when the
    reference's gaps to rank j-1 and rank j+1 both exceed ARGMAX_GAP (the last rank needs only the gap above: what
    lies below it is unknown, so it is compared as a set member instead)."""
    ref_vals = np.asarray(ref_vals)
    gaps = ref_vals[:, :-1] - ref_vals[:, 1:]
    clear = gaps > ARGMAX_GAP                                    # [U, k-1]
    above = np.concatenate([np.ones((clear.shape[0], 1), bool), clear], axis=1)[:, :-1]
    decided = above & clear                                      # ranks 0..k-2
    same = np.asarray(got_ids)[:, :decided.shape[1]] == np.asarray(ref_ids)[:, :decided.shape[1]]
    bad = decided & ~same
    assert not bad.any(), "%s: %d decided ranks differ" % (what, int(bad.sum()))
    assert np.abs(np.asarray(got_vals) - ref_vals).max() <= SIM_ATOL, what
    return float(decided.mean())


SCORE_ULP = 2.0 ** -15   # one ulp of soft_wpmi's fp32 intermediates (prob_d_given_e and prob_d are ~ -420): the
                         # reference's scores are multiples of it, and so are the gaps between them


def assert_topk_against_full_sim(got_ids, ref_sim, k, what=""):
    """Top-k lists against the reference's FULL similarity matrix.  Rank j of a row is decided when the reference's gaps to
    rank j-1 and to rank j+1 both exceed ONE ulp of the scores (SCORE_ULP; a 1-ulp gap is what a single last-bit difference
    in one log can close -- the reference's own `torch.log` is not correctly rounded in 0.02 % of its results);
    every decided rank must carry the reference's concept, and rank 0 must do so on every row that has a decided rank 0.
    Returns (decided ranks, undecided ranks)."""
    ref = np.asarray(ref_sim, np.float32)
    got = np.asarray(got_ids)
    order = np.argsort(-ref, axis=1, kind="stable")[:, :k + 1]
    v = np.take_along_axis(ref, order, axis=1).astype(np.float64)
    gaps = v[:, :-1] - v[:, 1:]                     # [U, k]: gap between rank j and rank j+1
    below = gaps > SCORE_ULP * 1.001
    above = np.concatenate([np.ones((ref.shape[0], 1), bool), below[:, :-1]], axis=1)
    decided = above & below                         # [U, k]
    bad = decided & (got[:, :k] != order[:, :k])
    n_dec, n_und = int(decided.sum()), int((~decided).sum())
    msg = "%s: top-%d ranks decided %d, undecided %d (gap <= 1 ulp = %.3g), decided-but-different %d" % (
        what, k, n_dec, n_und, SCORE_ULP, int(bad.sum()))
    if os.environ.get("MCD_STATS_FILE"):
        with open(os.environ["MCD_STATS_FILE"], "a") as f:
            f.write("ranks " + msg + "\n")
    assert not bad.any(), msg
    return n_dec, n_und


def case_inputs(name):
    z = golden(name)
    if name == "n1000":
        E_img, E_txt, A, P = regen_n1000()
        assert abs(float(P.astype(np.float64).sum()) - float(z["P_checksum"])) < 1e-3
        assert abs(float(A.astype(np.float64).sum()) - float(z["A_checksum"])) < 1e-6
        return z, E_img, E_txt, A, P
    return z, z["E_img"], z["E_txt"], z["A"], z["P"]


def assert_sim_boundary(got, ref, what=""):
    """Same inputs as the reference (its own P / S): bit-exact softmax and top-K, near correctly rounded log."""
    got = np.asarray(got, np.float32)
    ref = np.asarray(ref, np.float32)
    d = np.abs(got.astype(np.float64) - ref.astype(np.float64))
    exact = float((d == 0).mean())
    msg = "%s max=%.3e mean=%.3e exact=%.4f" % (what, d.max(), d.mean(), exact)
    if os.environ.get("MCD_STATS_FILE"):
        with open(os.environ["MCD_STATS_FILE"], "a") as f:
            f.write("boundary " + msg + "\n")
    assert d.max() <= SIM_BOUNDARY_ATOL, msg
    assert exact >= SIM_BOUNDARY_EXACT or d.size < 2000, msg
    return d.max(), exact


def assert_sim_close(got, ref, what=""):
    got = np.asarray(got, np.float32)
    ref = np.asarray(ref, np.float32)
    assert got.shape == ref.shape, (got.shape, ref.shape)
    d = np.abs(got.astype(np.float64) - ref.astype(np.float64))
    frac = float((d > SIM_ATOL).mean())
    msg = "%s max=%.3e mean=%.3e frac>1e-4=%.2e exact=%.4f" % (what, d.max(), d.mean(), frac, float((d == 0).mean()))
    if os.environ.get("MCD_STATS_FILE"):
        with open(os.environ["MCD_STATS_FILE"], "a") as f:
            f.write(msg + "\n")
    assert d.max() <= SIM_HARD_ATOL, msg
    assert (d > SIM_ATOL).sum() <= max(2, SIM_OUTLIER_FRAC * d.size), msg   # small cases: 2 entries at most
    assert d.mean() <= SIM_MEAN_ATOL, msg
    return d.max(), d.mean(), frac


def assert_topk_ids(got_ids, got_sim, ref_ids, ref_sim, k, what=""):
    """Integer match of the top-k concept ids per neuron wherever the reference ranking is separated by
    more than ARGMAX_GAP; elsewhere the two orders may swap neighbours that are closer than the gap."""
    ref_sim = np.asarray(ref_sim)
    order = np.sort(ref_sim, axis=1)[:, ::-1][:, :k + 1]
    gaps = order[:, :-1] - order[:, 1:]            # [U,k] gap below each of the k ranks
    if gaps.shape[1] < k:                          # k == C: nothing ranks below the last entry
        gaps = np.concatenate([gaps, np.full((gaps.shape[0], k - gaps.shape[1]), np.inf, gaps.dtype)], axis=1)
    clear = gaps > ARGMAX_GAP
    same = np.asarray(got_ids)[:, :k] == np.asarray(ref_ids)[:, :k]
    # a rank is "decided" when the gaps above and below it are clear
    above = np.concatenate([np.ones((gaps.shape[0], 1), bool), clear[:, :-1]], axis=1)
    decided = clear & above
    bad = decided & ~same
    assert not bad.any(), "%s: %d decided ranks differ" % (what, int(bad.sum()))
    return float(decided.mean())


def host_staged_gather(group=None):
    """A `gather` for pipeline.Dissector that rehearses several ranks on ONE GPU: RCCL cannot put two ranks on one
    device, so the ranks rendezvous over gloo and the payload is staged through the host.  Test/bench rehearsal only --
    the package's own transport is rccl_all_gather_rows (device memory, backend "nccl")."""
    import torch
    import torch.distributed as dist

    def gather(t):
        world = dist.get_world_size(group)
        host = torch.empty((world * t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype)
        dist.all_gather_into_tensor(host, t.contiguous().cpu(), group=group)
        return host.to(t.device)
    return gather


# ---- the reference's own outputs on off-golden shapes (make_golden.py --fuzz; round 5) -------------------------
FUZZ_ATOL = 1e-4          # north_star's tolerance, asserted against the REFERENCE's output directly (not through the oracle)
FUZZ_TOP1_GAP = 1.3e-4    # a top-1 / top-10 rank is decided when the reference's own gaps exceed two ulps of its sums (6.1e-5 each)
_FUZZ = {}


def fuzz_cases():
    import json
    return json.load(open(os.path.join(GOLDEN, "sim_fuzz_meta.json")))["cases"]


def fuzz_case(i):
    """Case i of tests/golden/sim_fuzz.npz: inputs regenerated from the stored recipe (torch's CPU generator; every byte checked
    against the sha256 the generator script stored), the reference's soft_wpmi / wpmi outputs and top-10 lists."""
    import torch
    if "z" not in _FUZZ:
        _FUZZ["z"] = np.load(os.path.join(GOLDEN, "sim_fuzz.npz"))
        _FUZZ["meta"] = fuzz_cases()
    z, r = _FUZZ["z"], _FUZZ["meta"][i]
    if r["kind"] == "scaled":
        g = torch.Generator().manual_seed(r["gen_seed"])
        P = (torch.randn(r["N"], r["C"], generator=g) * 0.05).numpy()
        A = torch.randn(r["N"], r["U"], generator=g).numpy()
    else:
        _, _, A, P = regen_inputs(r["N"], r["C"], r["U"], 512, r["gen_seed"])
    assert sha256(A) == r["A_sha256"], "A of fuzz case %d" % i
    assert sha256(P) == r["P_sha256"], "P of fuzz case %d (the oracle's restatement of the fixture host's normalise + matmul)" % i
    ref = {k: z["%s_%d" % (k, i)] for k in ("soft", "wpmi", "soft_ids10", "wpmi_ids10", "soft_vals10", "wpmi_vals10")}
    return r, P, A, ref


def fuzz_compare(got, ref_full, what, stats):
    """got against the reference's output: everything within FUZZ_ATOL; the top concept of every neuron whose reference gap to
    the runner-up exceeds FUZZ_TOP1_GAP; the top-10 ranks decided by that gap.  Appends (entries, bit-identical, max |diff|) to stats."""
    got = np.asarray(got, np.float32)
    ref = np.asarray(ref_full, np.float32)
    assert got.shape == ref.shape, what
    d = np.abs(got.astype(np.float64) - ref.astype(np.float64))
    assert np.isfinite(got).all() and float(d.max()) <= FUZZ_ATOL, "%s: max |diff| %.3g" % (what, float(d.max()))
    k = min(10, ref.shape[1])
    order = np.argsort(-ref, axis=1, kind="stable")[:, :min(k + 1, ref.shape[1])]
    v = np.take_along_axis(ref, order, axis=1).astype(np.float64)
    gorder = np.argsort(-got, axis=1, kind="stable")[:, :k]
    n_dec = n_bad = 0
    if ref.shape[1] > 1:
        gaps = v[:, :-1] - v[:, 1:]
        below = gaps > FUZZ_TOP1_GAP
        if below.shape[1] < k:
            below = np.concatenate([below, np.ones((ref.shape[0], k - below.shape[1]), bool)], axis=1)
        above = np.concatenate([np.ones((ref.shape[0], 1), bool), below[:, :-1]], axis=1)
        decided = (above & below)[:, :k]
        bad = decided & (gorder != order[:, :k])
        n_dec, n_bad = int(decided.sum()), int(bad.sum())
        assert not bad[:, 0].any(), "%s: the top concept differs on %d neurons whose reference gap exceeds %.2g" % (what, int(bad[:, 0].sum()), FUZZ_TOP1_GAP)
        assert n_bad == 0, "%s: %d decided top-10 ranks differ" % (what, n_bad)
    stats.append((what, int(d.size), int((got == ref).sum()), float(d.max()), n_dec))
