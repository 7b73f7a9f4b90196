"""GPU: the drop-in similarity functions (mirror of the reference's concept_vit/similarity.py) against
the golden vectors produced by the reference itself, end to end through the C ABI."""
import numpy as np
import pytest
import torch

import util
from util import CASES

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def sim(mcd):
    from mammo_clip_dissect_amd.concept_vit import similarity
    return similarity


@pytest.mark.parametrize("name", CASES)
def test_soft_wpmi_matches_reference(sim, dev, name):
    z, E_img, E_txt, A, P = util.case_inputs(name)
    K = int(z["top_k"])
    Pt, At = torch.from_numpy(P), torch.from_numpy(A)     # CPU tensors in, like the reference's callers
    out = sim.soft_wpmi(Pt, At, top_k=K, device=str(dev))
    assert out.shape == z["soft_wpmi"].shape and out.dtype == torch.float32 and out.is_cuda
    util.assert_sim_close(out.cpu().numpy(), z["soft_wpmi"], name)
    # bit-identical input: softmax and top-K are bit-exact, the log near correctly rounded
    util.assert_sim_boundary(out.cpu().numpy(), z["soft_wpmi"], name)
    # inputs are not modified (reference contract)
    assert torch.equal(Pt, torch.from_numpy(P)) and torch.equal(At, torch.from_numpy(A))
    # lam = 0 returns prob_d_given_e itself
    pd = sim.soft_wpmi(Pt.to(dev), At.to(dev), top_k=K, lam=0, device=str(dev))
    util.assert_sim_close(pd.cpu().numpy(), z["pdge"], name + " lam=0")
    if name != "one_neuron":
        from mammo_clip_dissect_amd import core
        k = min(10, out.shape[1])
        v, ids = core.row_topk(out, k)
        util.assert_topk_ids(ids.cpu().numpy(), out.cpu().numpy(), z["ids10"], z["soft_wpmi"], k, name)


@pytest.mark.parametrize("name", ["tiny", "main", "relu"])
def test_wpmi_matches_reference(sim, dev, name):
    z = util.golden(name)
    out = sim.wpmi(torch.from_numpy(z["P"]), torch.from_numpy(z["A"]), top_k=int(z["wpmi_top_k"]), device=str(dev))
    util.assert_sim_close(out.cpu().numpy(), z["wpmi"], name)
    util.assert_sim_boundary(out.cpu().numpy(), z["wpmi"], "wpmi " + name)


def test_topk_error_matches_torch(sim, dev):
    # N < top_k: torch.topk raises inside the reference
    with pytest.raises(RuntimeError, match="selected index k out of range"):
        sim.soft_wpmi(torch.randn(50, 7), torch.randn(50, 3), device=str(dev))


def test_full_size_properties(sim, dev):
    """BASELINE config 2 shape (N=10000, C=763, one ViT-B layer of 768 neurons): size-independent
    properties -- permuting the images leaves the scores unchanged up to the relabelling of indices,
    every similarity column has logsumexp == log U (the normalisation of similarity.py:70-72), and the
    result is deterministic run to run."""
    g = torch.Generator().manual_seed(7)
    N, C, U, D = 10000, 763, 768, 512
    I = torch.nn.functional.normalize(torch.randn(N, D, generator=g), dim=1)
    Tt = torch.nn.functional.normalize(torch.randn(C, D, generator=g), dim=1)
    P = (I @ Tt.T).to(dev)
    # tie-free activations by construction (a permuted strictly increasing grid per neuron): with ties the
    # lowest-index rule would make the ranking depend on the image order
    base = torch.linspace(-3.0, 3.0, N)
    A = torch.stack([base[torch.randperm(N, generator=g)] * (1.0 + 0.001 * u) for u in range(U)], dim=1).to(dev)
    out = sim.soft_wpmi(P, A, device=str(dev))
    out2 = sim.soft_wpmi(P, A, device=str(dev))
    assert torch.equal(out, out2)
    perm = torch.randperm(N, generator=g).to(dev)
    outp = sim.soft_wpmi(P[perm], A[perm], device=str(dev))
    assert torch.equal(out, outp)   # same rows gathered in the same rank order: bit-identical
    lse = torch.logsumexp(out.double(), dim=0)
    assert float((lse - np.log(U)).abs().max()) < 2e-4


@pytest.mark.parametrize("name", ["tiny", "main", "relu"])
def test_cos_similarities_match_reference(sim, dev, name):
    """cos_similarity / cos_similarity_cubed (reference similarity.py:7-47): column normalisation + one MFMA GEMM.
    Entries are cosines in [-1, 1]; fp32 dot products over N images: 5e-7."""
    z = util.golden(name)
    P, A = torch.from_numpy(z["P"]), torch.from_numpy(z["A"])
    out = sim.cos_similarity(P, A, device=str(dev))
    assert out.shape == z["cos_similarity"].shape and out.is_cuda
    assert np.abs(out.cpu().numpy() - z["cos_similarity"]).max() <= 5e-7
    out = sim.cos_similarity_cubed(P, A, device=str(dev))
    assert np.abs(out.cpu().numpy() - z["cos_similarity_cubed"]).max() <= 5e-7
    assert torch.equal(P, torch.from_numpy(z["P"]))   # "Does not modify any tensors in place" (reference :8-11)


def _assert_rank_reorder_close(got, ref, what):
    """NaN pattern identical (sqrt of a negative mean, as in the reference); elsewhere 5e-6 relative.  The column
    means, which can cancel, follow ATen's summation order in K8; the remaining differences are the (positive,
    well-conditioned) error sums, reduced in another order than ATen's."""
    assert got.shape == ref.shape
    assert np.array_equal(np.isnan(got), np.isnan(ref)), what
    m = ~np.isnan(ref)
    rel = np.abs(got[m] - ref[m]) / np.abs(ref[m])
    assert rel.max() <= 5e-6, (what, rel.max())


def test_rank_reorder_matches_reference_under_the_same_seed(sim, dev):
    """similarity.py:99-132 with torch.manual_seed(1234): the mirror draws the reference's permutations."""
    z = util.golden("main")
    P, A = torch.from_numpy(z["P"]), torch.from_numpy(z["A"])
    torch.manual_seed(1234)
    out = sim.rank_reorder(P, A, device=str(dev))
    assert out.is_cuda and out.dtype == torch.float32
    _assert_rank_reorder_close(out.cpu().numpy(), z["rank_reorder_seed1234"], "golden main")
    torch.manual_seed(99)                      # another stream of permutations: another baseline
    other = sim.rank_reorder(P, A, device=str(dev)).cpu().numpy()
    m = ~np.isnan(other)
    assert (other[m] != out.cpu().numpy()[m]).mean() > 0.9


@pytest.mark.parametrize("shape", [(4000, 763, 24, 3, 0.5), (2400, 130, 7, 2, 0.5), (1000, 37, 5, 2.5, 0.3), (20000, 64, 3, 3, 0.5),
                                   (50000, 24, 2, 3, 0.5)])
def test_rank_reorder_against_oracle(sim, dev, oracle, shape):
    """Larger top_n (200 / 120 / 50 / 1000 / 2500 images), a padded and an unpadded concept count, general exponents.
    clip_feats are softmax rows here (positive means: no NaNs), as in the reference's CLIP-Dissect lineage."""
    N, C, U, p, sp = shape
    g = torch.Generator().manual_seed(N + C)
    P = torch.softmax(4 * torch.randn(N, C, generator=g), dim=1)
    A = torch.randn(N, U, generator=g)
    torch.manual_seed(7)
    ref = oracle.rank_reorder(P.numpy(), A.numpy(), p=p, scale_p=sp)
    torch.manual_seed(7)
    got = sim.rank_reorder(P, A, device=str(dev), p=p, scale_p=sp).cpu().numpy()
    _assert_rank_reorder_close(got, ref, str(shape))


def test_rank_reorder_errors(sim, dev):
    with pytest.raises(RuntimeError):
        sim.rank_reorder(torch.randn(10, 3), torch.randn(10, 2), device=str(dev))   # int(10 * 0.05) == 0 images
    with pytest.raises(RuntimeError, match="GPU only"):
        sim.rank_reorder(torch.randn(100, 3), torch.randn(100, 2), device="cpu")


@pytest.mark.parametrize("shape", [(12000, 1000, 40, 100), (3000, 96, 17, 28), (700, 1500, 9, 100), (20000, 763, 6, 100),
                                   (600, 10000, 5, 100), (500, 2000, 20, 100), (300, 100, 33, 20)])
def test_other_shapes_against_oracle(sim, dev, shape):
    """Shapes off the config-2 fast paths: N in the 1024-thread top-K class, C > 1024 (multi-pass softmax), K = 28;
    C = 10 000 (the stress configuration: ATen's row_sum columns start on a slice boundary, so the sliced kernel
    takes [0, 9984) and the tail kernel the last 16), C = 2000 (row_sum group in the middle of a slice: generic
    kernels), C = 100 with K = 20."""
    import os, sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
    import oracle as O
    N, C, U, K = shape
    g = torch.Generator().manual_seed(N + C)
    P = (torch.randn(N, C, generator=g) * 0.05)
    A = torch.randn(N, U, generator=g)
    out = sim.soft_wpmi(P, A, top_k=K, device=str(dev)).cpu().numpy()
    ref = O.soft_wpmi(P.numpy(), A.numpy(), top_k=K)
    util.assert_sim_boundary(out, ref, "shape %s" % (shape,))


def test_full_size_properties_config2(sim, dev):
    """Size-independent properties at BASELINE config-2 size (10 000 images x 763 concepts x 768 neurons, K = 100),
    where the oracle would take minutes:
      * relabelling the probe images (the same permutation of the rows of P and A) changes nothing, bit for bit --
        the sums run over the activation RANK of an image, never over its index;
      * scaling the activations by a positive power of two changes nothing, bit for bit (only their order matters);
      * every row of the result is finite, and the top-1 concept of a neuron whose top images all carry the same
        planted concept is that concept."""
    N, C, U, K = 10000, 763, 768, 100
    g = torch.Generator().manual_seed(2024)
    P = torch.randn(N, C, generator=g) * 0.0442
    # tie-free activations (10 000 gaussian floats per column do collide now and then, and a tie is broken by the
    # image index, which the relabelling changes): every column is a random arrangement of N distinct values
    grid = torch.linspace(-3.0, 3.0, N)
    A = torch.stack([grid[torch.randperm(N, generator=g)] for _ in range(U)], dim=1)
    # plant: neuron 5 fires on images 0..149, whose similarity to concept 321 is high
    A[:150, 5] += 10.0
    P[:150, 321] += 0.35
    base = sim.soft_wpmi(P, A, top_k=K, device=str(dev))
    assert torch.isfinite(base).all()
    assert int(base[5].argmax()) == 321
    perm = torch.randperm(N, generator=g)
    assert torch.equal(sim.soft_wpmi(P[perm], A[perm], top_k=K, device=str(dev)), base)
    assert torch.equal(sim.soft_wpmi(P, A * 4.0, top_k=K, device=str(dev)), base)
    # hard WPMI: same two invariances
    bw = sim.wpmi(P, A, device=str(dev))
    assert torch.equal(sim.wpmi(P[perm], A[perm], device=str(dev)), bw)
    # the cosine scores are invariant to the relabelling up to summation order (an MFMA dot product over the images)
    cs = sim.cos_similarity(P, A, device=str(dev))
    assert float((sim.cos_similarity(P[perm], A[perm], device=str(dev)) - cs).abs().max()) <= 2e-6


@pytest.mark.parametrize("shape", [(50000, 763, 512, 100), (25000, 10000, 768, 100)])
def test_full_size_properties_configs_3_and_4(sim, dev, shape):
    """The same size-independent properties at the other BASELINE sizes: configs[3]'s whole probe set (50 000 images,
    763 concepts, a 512-channel EfficientNet-B5 block: the two-pass top-K kernel) and one rank's share of configs[4]
    (25 000 images x 10 000 concepts, 768 neurons: the sliced + tail scoring kernels, 1 GB of similarities)."""
    N, C, U, K = shape
    g = torch.Generator().manual_seed(N + C)
    P = torch.randn(N, C, generator=g) * 0.0442
    grid = torch.linspace(-3.0, 3.0, N)
    A = torch.stack([grid[torch.randperm(N, generator=g)] for _ in range(U)], dim=1)
    A[:150, 5] += 10.0
    P[:150, C // 2] += 0.35
    Pd, Ad = P.to(dev), A.to(dev)
    del P, A
    base = sim.soft_wpmi(Pd, Ad, top_k=K, device=str(dev))
    assert torch.isfinite(base).all()
    assert int(base[5].argmax()) == C // 2
    assert torch.equal(sim.soft_wpmi(Pd, Ad, top_k=K, device=str(dev)), base)            # run to run
    assert torch.equal(sim.soft_wpmi(Pd, Ad * 0.5, top_k=K, device=str(dev)), base)      # only the order of A matters
    lse = torch.logsumexp(base.double(), dim=0)                                          # similarity.py:70-72
    assert float((lse - np.log(U)).abs().max()) < 2e-4
    perm = torch.randperm(N, generator=g).to(dev)
    Pp, Ap = Pd[perm], Ad[perm]
    del Pd, Ad
    assert torch.equal(sim.soft_wpmi(Pp, Ap, top_k=K, device=str(dev)), base)            # relabelled images
    assert torch.equal(sim.wpmi(Pp, Ap, device=str(dev)), sim.wpmi(Pp, Ap, device=str(dev)))


def test_edge_shapes_like_the_reference(sim, dev):
    """Degenerate inputs behave as in the reference: no neurons -> torch.cat([]) error; more top images than images ->
    torch.topk's error; one concept (softmax == 1: every term is log(1 + 1e-7), all neurons equal, similarity 0);
    K == N; and inputs are left untouched."""
    g = torch.Generator().manual_seed(1)
    P, A = torch.randn(200, 9, generator=g) * 0.1, torch.randn(200, 4, generator=g)
    with pytest.raises(RuntimeError, match="non-empty list"):
        sim.soft_wpmi(P, A[:, :0], device=str(dev))
    with pytest.raises(RuntimeError, match="selected index k out of range"):
        sim.soft_wpmi(P, A, top_k=201, device=str(dev))
    one = sim.soft_wpmi(P[:, :1].contiguous(), A, device=str(dev))
    assert one.shape == (4, 1) and float(one.abs().max()) <= 2e-7     # (log U + m) - log U: fp32 rounding of m ~ 1e-5
    full = sim.soft_wpmi(P, A, top_k=200, device=str(dev))      # K == N: every image, in activation order
    assert full.shape == (4, 9) and torch.isfinite(full).all()
    Pc, Ac = P.clone(), A.clone()
    sim.wpmi(P, A, device=str(dev)); sim.cos_similarity(P, A, device=str(dev)); sim.cos_similarity_cubed(P, A, device=str(dev))
    assert torch.equal(P, Pc) and torch.equal(A, Ac)


def test_nan_inputs_propagate_like_the_reference(sim, dev):
    """A NaN similarity row poisons exactly the neurons whose top images include that image (and, through the
    layer-wide logsumexp, the layer's normalisation): NaN out, never a finite made-up value."""
    g = torch.Generator().manual_seed(3)
    P, A = torch.randn(300, 20, generator=g) * 0.1, torch.randn(300, 6, generator=g)
    A[7, 2] = 50.0                      # image 7 is neuron 2's top image
    P[7, 5] = float("nan")
    out = sim.soft_wpmi(P, A, device=str(dev))
    assert torch.isnan(out[2]).all()    # softmax of a row with a NaN is all NaN


def test_center_cube_normalize_in_place_equals_out_of_place(sim, dev):
    """cos_similarity_cubed runs K7 in place (similarity.py mirror: out=t); the kernel must not assume its input and
    output are distinct buffers."""
    from mammo_clip_dissect_amd import core
    g = torch.Generator().manual_seed(3)
    x = torch.randn(37, 1000, generator=g).to(dev)
    want = core.center_cube_normalize_rows(x.clone())
    y = x.clone()
    got = core.center_cube_normalize_rows(y, out=y)
    assert got.data_ptr() == y.data_ptr() and torch.equal(got, want)
    bad = torch.empty(37, 2000, device=dev)[:, ::2]
    with pytest.raises(ValueError):
        core.center_cube_normalize_rows(x, out=bad)        # an output the ABI cannot address is an error, not a copy


def test_core_calls_follow_the_tensors_device(sim, dev):
    """core wrappers make the tensors' device current for the call (and refuse tensors on two devices); with one GPU
    this checks the guard is transparent and that mixed CPU/GPU arguments still raise."""
    from mammo_clip_dissect_amd import core
    x = torch.randn(8, 16, device=dev)
    y = core.normalize_rows(x)
    assert y.device == x.device and torch.allclose(y.norm(dim=1), torch.ones(8, device=dev), atol=1e-6)
    with pytest.raises(TypeError):
        core.normalize_rows(x.cpu())
