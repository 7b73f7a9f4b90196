"""CPU: the oracle (oracle/) against every golden vector generated from the reference
(tests/golden/make_golden.py imports /root/reference/concept_vit/similarity.py).  This is the pin."""
import numpy as np
import pytest

import util
from util import CASES


def test_p_in_examples(oracle):
    p = np.load(util.GOLDEN + "/p_in_examples.npz")["p100"][:, 0]
    mine = oracle.p_in_examples(100)
    assert mine.dtype == np.float32 and np.array_equal(mine, p)
    # first 0x1.fef9dcp-1, last 0x1.f0c88cp-1
    assert float(p[0]).hex() == "0x1.fef9dc0000000p-1" and float(p[-1]).hex() == "0x1.f0c88c0000000p-1"


def test_sum_split_rule(oracle):
    assert [oracle.sum_split(c) for c in (1, 3, 5, 7, 8, 20, 33, 763, 10000)] == [0, 0, 4, 4, 0, 0, 32, 736, 9984]


@pytest.mark.parametrize("name", CASES)
def test_stages_against_golden(oracle, name):
    z, E_img, E_txt, A, P = util.case_inputs(name)
    K = int(z["top_k"])
    # utils.py:577-594
    # ATen's norm order + MKL's K-blocked fma chains restated: the reference's P to the bit
    assert np.array_equal(oracle.embed_gemm(E_img, E_txt, blas=False), P)   # (n1000: trivially, util.regen_n1000)
    assert np.abs(oracle.embed_gemm(E_img, E_txt, blas=True) - P).max() <= util.P_ATOL  # numpy's own BLAS order
    # similarity.py:54
    if "S" in z:
        S = oracle.row_softmax(P, 10.0)
        assert (np.abs(S - z["S"]) / z["S"]).max() <= util.S_RTOL
    # similarity.py:55 -- integer, exact (inputs are tie-free)
    _, idx = oracle.col_topk(A, K)
    assert np.array_equal(idx, z["inds"])
    # similarity.py:59-65 given the reference's own S and indices: ATen's summation order restated
    if "S" in z:
        pdge = oracle.wpmi_score(z["S"], z["inds"], oracle.p_in_examples(K), np.float32(1e-7), 1)
        d = np.abs(pdge - z["pdge"])
        assert d.max() <= util.PDGE_ATOL
        assert (pdge != z["pdge"]).mean() <= 2e-4  # correctly rounded log vs MKL vsLn: rare last-ulp differences
    # similarity.py:70-72 given the reference's own prob_d_given_e
    out = oracle.logsumexp_sub(z["pdge"], 1.0)
    assert np.abs(out - z["soft_wpmi"]).max() <= 6.2e-5
    # describe_broad_neurons.py:101-102 on the reference's own similarities
    if name != "one_neuron":  # U == 1 makes every similarity 0: all ties
        k = min(10, z["soft_wpmi"].shape[1])
        v10, i10 = oracle.row_topk(z["soft_wpmi"], k)
        assert np.array_equal(i10, z["ids10"]) and np.array_equal(v10, z["vals10"])
        v1, i1 = oracle.row_topk(z["soft_wpmi"], 1)
        assert np.array_equal(i1[:, 0], z["imax"]) and np.array_equal(v1[:, 0], z["vmax"])
    _, t5 = oracle.col_topk(A, 5)
    assert np.array_equal(t5, z["top5"])


@pytest.mark.parametrize("name", CASES)
def test_soft_wpmi_end_to_end(oracle, name):
    z, E_img, E_txt, A, P = util.case_inputs(name)
    K = int(z["top_k"])
    util.assert_sim_close(oracle.soft_wpmi(P, A, top_k=K), z["soft_wpmi"], "from P")
    got = oracle.soft_wpmi(oracle.embed_gemm(E_img, E_txt), A, top_k=K)
    util.assert_sim_close(got, z["soft_wpmi"], "from embeddings")
    if name != "one_neuron":
        k = min(10, got.shape[1])
        _, ids = oracle.row_topk(got, k)
        util.assert_topk_ids(ids, got, z["ids10"], z["soft_wpmi"], k)


def test_real_layer_size_golden(oracle):
    """One reference-made case at configs[1]'s layer size (N = 10 000, U = 768, C = 763, top_k = 100; make_golden.py
    main_round2): the oracle from the embeddings and from P against the reference's outputs, at the north_star
    tolerance (1e-4) -- the score ulp is 3e-5, so the reference's own top-10 lists hold exact ties
    (golden_meta.json: min_top10_gap 0.0) and ranks are compared where the reference decides them."""
    g = util.n10k_inputs()
    z = g["z"]
    got = oracle.soft_wpmi(g["P"], g["A"], top_k=g["K"])
    d = np.abs(got.astype(np.float64) - z["soft_wpmi"])
    assert d.max() <= util.SIM_ATOL and (d == 0).mean() >= 0.999, (d.max(), (d == 0).mean())
    _, t5 = oracle.col_topk(g["A"], 5)
    assert np.array_equal(t5, z["top5"])
    v10, i10 = oracle.row_topk(got, 10)
    assert util.assert_top10_decided(i10, v10, z["ids10"], z["vals10"], "n10k") > 0.5
    sep = z["vals10"][:, 0] - z["vals10"][:, 1] > util.ARGMAX_GAP
    assert np.array_equal(i10[sep, 0], z["imax"][sep])           # top concept: exact wherever the reference decides it


def test_shared_probe_two_layer_golden(oracle):
    """The driver's shape (one probe set, one P, two layers; make_golden.py: shared2): layer 1 of the fixture."""
    zm, z2 = util.golden("main"), util.golden("shared2")
    got = oracle.soft_wpmi(zm["P"], z2["A1"], top_k=int(z2["top_k"]))
    util.assert_sim_close(got, z2["soft_wpmi1"], "shared2 layer_b")
    _, t5 = oracle.col_topk(z2["A1"], 5)
    assert np.array_equal(t5, z2["top5"])
    v10, i10 = oracle.row_topk(z2["soft_wpmi1"], 10)
    assert np.array_equal(i10, z2["ids10"]) and np.array_equal(v10, z2["vals10"])


@pytest.mark.parametrize("name", ["tiny", "main", "relu"])
def test_other_similarity_fns(oracle, name):
    z = util.golden(name)
    w = oracle.wpmi(z["P"], z["A"], top_k=int(z["wpmi_top_k"]))
    util.assert_sim_close(w, z["wpmi"], "wpmi")
    assert np.abs(oracle.cos_similarity(z["P"], z["A"]) - z["cos_similarity"]).max() <= 5e-7
    assert np.abs(oracle.cos_similarity_cubed(z["P"], z["A"]) - z["cos_similarity_cubed"]).max() <= 5e-7


def test_rank_reorder_seeded(oracle):
    """similarity.py:99-132 under torch.manual_seed(1234): the oracle draws the reference's torch.randperm stream."""
    import torch
    z = util.golden("main")
    torch.manual_seed(1234)
    r = oracle.rank_reorder(z["P"], z["A"])
    g = z["rank_reorder_seed1234"]
    assert np.array_equal(np.isnan(r), np.isnan(g))       # mean of the gathered similarities < 0 -> sqrt -> NaN
    m = ~np.isnan(g)
    rel = np.abs(r[m] - g[m]) / np.abs(g[m])
    assert rel.max() <= 2e-6 and np.median(rel) <= 1e-7   # column means in ATen's order: no cancellation left over


def test_topk_k_out_of_range(oracle):
    A = np.random.default_rng(0).standard_normal((10, 3)).astype(np.float32)
    with pytest.raises(RuntimeError, match="selected index k out of range"):
        oracle.col_topk(A, 11)


def test_ties_lowest_index_first(oracle):
    A = np.zeros((9, 2), np.float32)
    A[[1, 2, 4], 0] = 1.0
    _, idx = oracle.col_topk(A, 5)
    assert idx[:, 0].tolist() == [1, 2, 4, 0, 3] and idx[:, 1].tolist() == [0, 1, 2, 3, 4]
