"""GPU: every HIP kernel, called through the C ABI (mammo_clip_dissect_amd.core -> libmcd_hip.so),
against the oracle and the golden vectors on the same inputs.  Integer results are exact; floating
point tolerances are the ones stated in tests/util.py."""
import numpy as np
import os

import pytest
import torch

import util
from util import CASES

pytestmark = pytest.mark.gpu


def T(x, dev):
    return torch.from_numpy(np.ascontiguousarray(x)).to(dev)


@pytest.fixture(scope="module")
def core(mcd):
    from mammo_clip_dissect_amd import core as c
    return c


@pytest.mark.parametrize("name", CASES)
def test_normalize_and_gemm(core, dev, oracle, name):
    z, E_img, E_txt, A, P = util.case_inputs(name)
    I = core.normalize_rows(T(E_img, dev))
    Tt = core.normalize_rows(T(E_txt, dev))
    assert np.array_equal(I.cpu().numpy(), oracle.normalize_rows(E_img))   # ATen's accumulation order: exact
    # in place, like the reference's `image_features /= ...`
    x = T(E_img, dev)
    core.normalize_rows(x, out=x)
    assert torch.equal(x, I)
    Pg = core.embed_gemm(I, Tt).cpu().numpy()
    assert Pg.shape == P.shape
    # the reference's own torch CPU matmul on the fixture host (MKL's K-blocked fma chains), to the bit (n1000: its
    # restatement, see util.regen_n1000)
    assert np.array_equal(Pg, P)


def test_normalize_rows_matches_aten_order(core, dev, oracle):
    """Every row length class of ATen's norm kernel: no full 8-vector, full vectors only, tails of 1..7 (the tail's
    first 4*(tail/4) elements unfused, the rest fused), padded leading dimension, a row count that is not a multiple
    of the 32 rows a workgroup takes."""
    rng = np.random.default_rng(21)
    for d in list(range(1, 34)) + [100, 512, 515, 768, 1027, 1029, 1031]:
        E = rng.standard_normal((77, d)).astype(np.float32)
        pad = T(np.concatenate([E, np.full((77, 3), 9.0, np.float32)], axis=1), dev)
        for x in (T(E, dev), pad[:, :d]):
            assert np.array_equal(core.normalize_rows(x).cpu().numpy(), oracle.normalize_rows(E)), d


def test_gemm_follows_mkl_k_blocks(core, dev, oracle):
    """D <= 384: one fma chain; 384 < D <= 768: two blocks, the first roundup4(ceil(D/2)) long; above: blocks of 384.
    The oracle's C restatement (checked against the golden P, D = 512, and -- in the build container -- against live
    torch for the other widths) and the MFMA kernel must agree to the bit, ragged tiles and unaligned rows included."""
    rng = np.random.default_rng(22)
    for (N, C, D) in [(130, 70, 384), (130, 70, 388), (257, 129, 512), (200, 763, 520), (64, 100, 600),
                      (64, 100, 700), (150, 130, 767), (140, 133, 768), (129, 64, 1024), (64, 64, 1000), (33, 5, 2049)]:
        a = oracle.normalize_rows(rng.standard_normal((N, D)).astype(np.float32))
        b = oracle.normalize_rows(rng.standard_normal((C, D)).astype(np.float32))
        ref = np.empty((N, C), np.float32)
        oracle.lib().mcd_o_gemm_nt(oracle._f(a), oracle._f(b), oracle._i64(N), oracle._i64(C), oracle._i64(D),
                                   oracle._f(ref))
        got = core.embed_gemm(T(a, dev), T(b, dev)).cpu().numpy()
        assert np.array_equal(got, ref), (N, C, D)
        if D % 4:   # unaligned leading dimension: the scalar-load variant of the kernel
            continue
        a1 = T(np.concatenate([a, np.ones((N, 1), np.float32)], axis=1), dev)[:, :D]
        assert np.array_equal(core.embed_gemm(a1, T(b, dev)).cpu().numpy(), ref), (N, C, D, "ld")


def test_gemm_output_paths_agree_across_the_31_bit_limit(core, dev):
    """K1 writes P through ONE buffer descriptor per tile while N * ldp * 4 < 2^31 and with plain 64-bit stores above it: both paths
    on the same rows must give the same bits -- a 704 000 x 763 output (2.15 GB, just past the limit) against the same rows
    computed in slices that take the descriptor path, partial tiles at both ends included."""
    import torch
    N, C, D = 704000, 763, 64
    assert N * C * 4 >= 2 ** 31
    g = torch.Generator(device=dev).manual_seed(3)
    a = core.normalize_rows(torch.randn(N, D, device=dev, generator=g))
    b = core.normalize_rows(torch.randn(C, D, device=dev, generator=g))
    big = core.embed_gemm(a, b)
    assert big.shape == (N, C)
    for r0, r1 in ((0, 1000), (351900, 353001), (N - 777, N)):
        part = core.embed_gemm(a[r0:r1].contiguous(), b)
        assert torch.equal(big[r0:r1], part), (r0, r1)
    # and a padded pitch on the descriptor path: the columns between C and ldp stay untouched
    out = torch.full((3000, 768), 7.0, device=dev)
    core.embed_gemm(a[:3000].contiguous(), b, out=out[:, :C])
    assert torch.equal(out[:, :C], big[:3000]) and bool((out[:, C:] == 7.0).all())


def test_gemm_is_an_exact_fp32_fma_chain(core, dev):
    """MCD_GEMM_F32 = v_mfma_f32_32x32x2_f32: bit-for-bit fma(a_k, b_k, acc) in k order.  Integer data
    makes every order exact, so this checks layout; an asymmetric B catches a transposed C write."""
    rng = np.random.default_rng(5)
    for (N, C, D) in [(1, 1, 1), (130, 70, 33), (257, 129, 512), (64, 763, 100)]:
        a = rng.integers(-3, 4, (N, D)).astype(np.float32)
        b = rng.integers(-3, 4, (C, D)).astype(np.float32) + np.arange(C, dtype=np.float32)[:, None] % 2
        got = core.embed_gemm(T(a, dev), T(b, dev)).cpu().numpy()
        assert np.array_equal(got, a @ b.T), (N, C, D)
    # and the fma chain itself on real data
    a = rng.standard_normal((40, 96)).astype(np.float32)
    b = rng.standard_normal((50, 96)).astype(np.float32)
    ref = np.zeros((40, 50), np.float32)
    for k in range(96):
        ref = (a[:, k:k + 1].astype(np.float64) * b[None, :, k].astype(np.float64) + ref.astype(np.float64)).astype(
            np.float32)  # fma: one rounding (double holds the exact product + sum of two floats to within fp32 rounding)
    got = core.embed_gemm(T(a, dev), T(b, dev)).cpu().numpy()
    assert (got != ref).mean() < 1e-3  # double rounding in the emulation can differ in rare last bits
    assert np.abs(got - ref).max() <= 2e-6


@pytest.mark.parametrize("name", ["main", "relu", "kfull"])
def test_gemm_bf16_modes(core, dev, name):
    """MCD_GEMM_BF16X3 (split bf16, three MFMAs per product, fp32 accumulate): fp32-like accuracy on the unit-norm
    embeddings; MCD_GEMM_BF16 (single pass): bf16 input rounding, ~4e-3, no parity claim (stress configuration)."""
    z = util.golden(name)
    I = core.normalize_rows(T(z["E_img"], dev))
    Tt = core.normalize_rows(T(z["E_txt"], dev))
    ref = core.embed_gemm(I, Tt, mode="f32").cpu().numpy()
    x3 = core.embed_gemm(I, Tt, mode="bf16x3").cpu().numpy()
    x1 = core.embed_gemm(I, Tt, mode="bf16").cpu().numpy()
    # dropped lo*lo term + rounding of lo: <= ~1.2e-5 * sum_k |a_k b_k| (<= 1 for unit vectors); observed 1e-6 at
    # D = 512 and 2.4e-6 at D = 64
    assert np.abs(x3 - ref).max() <= 4e-6, np.abs(x3 - ref).max()
    assert np.abs(x3 - z["P"]).max() <= 4e-6
    assert 1e-5 < np.abs(x1 - ref).max() <= 8e-3
    # exact on data that is exactly representable in bf16 (layout check with an asymmetric B)
    rng = np.random.default_rng(9)
    a = rng.integers(-8, 9, (200, 96)).astype(np.float32)
    b = rng.integers(-8, 9, (70, 96)).astype(np.float32) + (np.arange(70, dtype=np.float32)[:, None] % 3)
    for mode in ("bf16", "bf16x3"):
        assert np.array_equal(core.embed_gemm(T(a, dev), T(b, dev), mode=mode).cpu().numpy(), a @ b.T), mode


@pytest.mark.parametrize("shape", [(8192, 4096, 512), (8200, 4100, 500), (12000, 6000, 96), (4097, 8193, 64)])
def test_gemm_bf16_large_kernel(core, dev, shape):
    """The 256x256-tile DMA-staged kernel (taken when the tiles outnumber the CUs and a workspace is given) against
    the 128x128 kernel of the same mode (converts while staging): the products and the k-order inside an MFMA are
    the same, so the two agree to accumulation-order rounding; ragged edges, K padding (D not a multiple of 64);
    integer data (exact in bf16) must be exact, with an asymmetric B so that a transposed tile would show."""
    N, C, D = shape
    g = torch.Generator(device=dev).manual_seed(N + C)
    I = core.normalize_rows(torch.randn(N, D, device=dev, generator=g))
    Tt = core.normalize_rows(torch.randn(C, D, device=dev, generator=g))
    for mode, tol in (("bf16", 2e-6), ("bf16x3", 2e-6)):
        big = core.embed_gemm(I, Tt, mode=mode)
        small = core.embed_gemm(I, Tt, mode=mode, use_workspace=False)
        assert float((big - small).abs().max()) <= tol, mode
    ref = core.embed_gemm(I, Tt, mode="f32")
    assert float((core.embed_gemm(I, Tt, mode="bf16x3") - ref).abs().max()) <= 6e-6
    a = torch.randint(-8, 9, (N, D), device=dev, generator=g).float()
    b = torch.randint(-8, 9, (C, D), device=dev, generator=g).float() + (torch.arange(C, device=dev)[:, None] % 3).float()
    want = core.embed_gemm(a, b, mode="f32")          # exact on integers (test_gemm_is_an_exact_fp32_fma_chain)
    for mode in ("bf16", "bf16x3"):
        assert torch.equal(core.embed_gemm(a, b, mode=mode), want), mode


@pytest.mark.parametrize("name", CASES)
def test_row_softmax(core, dev, oracle, name):
    z, E_img, E_txt, A, P = util.case_inputs(name)
    S = core.row_softmax(T(P, dev), 10.0)
    assert S.shape == P.shape and S.stride(0) % 64 == 0
    ref = z["S"] if "S" in z else oracle.row_softmax(P, 10.0)
    got = S.cpu().numpy()
    assert (np.abs(got - ref) / ref).max() <= util.S_RTOL
    # K2 restates ATen's CPU kernel exactly (SLEEF expf_u10, 16-lane vector sum, reciprocal multiply)
    assert np.array_equal(got, ref), int((got != ref).sum())
    # padding columns are exactly zero
    full = torch.as_strided(S, (S.shape[0], S.stride(0)), (S.stride(0), 1))
    assert float(full[:, P.shape[1]:].abs().max()) == 0.0 if S.stride(0) > P.shape[1] else True
    # rows sum to one
    assert np.abs(got.sum(1, dtype=np.float64) - 1).max() < 1e-6


@pytest.mark.parametrize("name", CASES)
@pytest.mark.parametrize("neuron_major", [False, True])
def test_col_topk_matches_torch_topk(core, dev, name, neuron_major):
    z, E_img, E_txt, A, P = util.case_inputs(name)
    for K in (int(z["top_k"]), 5, 1):
        Ain = T(A.T.copy() if neuron_major else A, dev)
        vals, idx = core.col_topk(Ain, K, neuron_major=neuron_major)
        ref_v, ref_i = torch.topk(torch.from_numpy(A), k=K, dim=0)
        assert np.array_equal(idx.cpu().numpy().T, ref_i.numpy())       # integer: exact
        assert np.array_equal(vals.cpu().numpy().T, ref_v.numpy())       # values are copied, not computed
    assert np.array_equal(core.col_topk(T(A, dev), int(z["top_k"]))[1].cpu().numpy().T, z["inds"])
    assert np.array_equal(core.col_topk(T(A, dev), 5)[1].cpu().numpy().T, z["top5"])


def test_col_topk_edges(core, dev, oracle):
    rng = np.random.default_rng(3)
    # k out of range -> torch's error
    with pytest.raises(RuntimeError, match="selected index k out of range"):
        core.col_topk(T(rng.standard_normal((10, 3)).astype(np.float32), dev), 11)
    # ties: lowest image index first (the build's definition; torch leaves it unspecified)
    A = np.zeros((9, 2), np.float32)
    A[[1, 2, 4], 0] = 1.0
    _, idx = core.col_topk(T(A, dev), 5)
    assert idx.cpu().numpy()[0].tolist() == [1, 2, 4, 0, 3] and idx.cpu().numpy()[1].tolist() == [0, 1, 2, 3, 4]
    # dead channel (every image ties) and a half-dead ReLU channel, large N: the exact tie path
    N = 3000
    A = np.zeros((N, 3), np.float32)
    A[:, 1] = np.maximum(rng.standard_normal(N), 1.2).astype(np.float32)   # ~88% tie at 1.2
    A[:, 2] = rng.standard_normal(N).astype(np.float32)
    for K in (100, 7):
        vals, idx = core.col_topk(T(A, dev), K)
        rv, ri = oracle.col_topk(A, K)
        assert np.array_equal(idx.cpu().numpy().T, ri) and np.array_equal(vals.cpu().numpy().T, rv)
    # NaN ranks above +inf, -inf last (torch.topk's rule)
    A = rng.standard_normal((300, 2)).astype(np.float32)
    A[7, 0] = np.nan; A[9, 0] = np.inf; A[11, 0] = -np.inf
    vals, idx = core.col_topk(T(A, dev), 300)
    i0 = idx.cpu().numpy()[0]
    assert i0[0] == 7 and i0[1] == 9 and i0[-1] == 11
    tv, ti = torch.topk(torch.from_numpy(A), k=300, dim=0)
    assert np.array_equal(idx.cpu().numpy().T, ti.numpy())
    # the register-resident kernel compares floats: a row with NaNs (they rank first), -inf / +inf, and +0.0 / -0.0
    # beside positive values, at the sizes of several classes
    # (10 001 / 32 003: the quad across the end of a row whose length is not a multiple of 4; 32 003 / 50 000: the round-4 classes)
    for N in (1000, 10000, 10001, 20000, 25000, 32003, 50000):
        A = rng.standard_normal((N, 4)).astype(np.float32)
        A[5, 0] = np.nan; A[N - 1, 0] = np.nan; A[17, 0] = np.inf
        A[:, 1] = -np.abs(A[:, 1]); A[3, 1] = -np.inf; A[N // 2, 1] = np.inf
        A[::3, 2] = 0.0; A[1::3, 2] = -0.0; A[2::3, 2] = -np.abs(A[2::3, 2]) - 1.0; A[[2, 11], 2] = 3.0
        vals, idx = core.col_topk(T(A, dev), 100)
        rv, ri = oracle.col_topk(A, 100)
        iv, ii = vals.cpu().numpy().T, idx.cpu().numpy().T
        assert np.array_equal(ii[:, [0, 1, 3]], ri[:, [0, 1, 3]]), N
        assert np.array_equal(iv[:, [1, 3]], rv[:, [1, 3]]) and np.isnan(iv[:2, 0]).all() and np.array_equal(iv[2:, 0], rv[2:, 0])
        # zeros of either sign tie for torch.topk; this build puts +0.0 before -0.0 and orders each by image index
        assert ii[:2, 2].tolist() == [2, 11] and (A[ii[2:, 2], 2] == 0.0).all()
    # denormals keep their order (float comparisons in IEEE mode, like the integer keys): descending 3e-40 .. 1e-45, then negatives
    N = 10000
    A = -np.abs(rng.standard_normal((N, 2)).astype(np.float32)) - 1.0
    tiny = np.array([3e-40, 2e-40, 1e-40, 5e-41, 1e-45], np.float32)
    pos = [77, 9000, 4123, 5, 6001]
    A[pos, 0] = tiny
    vals, idx = core.col_topk(T(A, dev), 100)
    assert idx.cpu().numpy()[0, :5].tolist() == pos and np.array_equal(vals.cpu().numpy()[0, :5], tiny)
    tv, ti = torch.topk(torch.from_numpy(A), k=100, dim=0)
    assert np.array_equal(idx.cpu().numpy().T, ti.numpy())
    # every register-resident size class, ragged N, K = N
    for N in (1, 2, 63, 64, 65, 1000, 1025, 2049, 4097, 10000, 10241, 16385, 25000, 26625, 33000, 70000):
        A = rng.standard_normal((N, 2)).astype(np.float32)
        K = min(N, 100)
        vals, idx = core.col_topk(T(A, dev), K)
        tv, ti = torch.topk(torch.from_numpy(A), k=K, dim=0)
        assert np.array_equal(idx.cpu().numpy().T, ti.numpy()), N
    # K up to 1024 (rank_reorder's top_fraction*N)
    A = rng.standard_normal((10000, 3)).astype(np.float32)
    vals, idx = core.col_topk(T(A, dev), 500)
    tv, ti = torch.topk(torch.from_numpy(A), k=500, dim=0)
    assert np.array_equal(idx.cpu().numpy().T, ti.numpy())
    # 1024 < K <= 4096: streamed threshold + LDS bitonic sort (rank_reorder at 50 000 images: K = 2500); with a block
    # of exact ties straddling the K-th place (lower image index first) and K = N
    A = rng.standard_normal((50000, 3)).astype(np.float32)
    A[100:3000, 1] = 0.25
    for K in (1025, 2500, 4096):
        vals, idx = core.col_topk(T(A, dev), K)
        order = np.lexsort((np.arange(A.shape[0])[:, None].repeat(3, 1), -A), axis=0)[:K]   # value desc, index asc
        assert np.array_equal(idx.cpu().numpy().T, order), K
        assert np.array_equal(vals.cpu().numpy().T, np.take_along_axis(A, order, 0))
    A = rng.standard_normal((3000, 2)).astype(np.float32)
    vals, idx = core.col_topk(T(A, dev), 3000)
    assert np.array_equal(idx.cpu().numpy().T, np.argsort(-A, axis=0, kind="stable"))
    with pytest.raises(Exception, match="4096"):
        core.col_topk(T(rng.standard_normal((6000, 2)).astype(np.float32), dev), 5000)


@pytest.mark.parametrize("name", CASES)
def test_wpmi_score_given_reference_inputs(core, dev, oracle, name):
    """K4 on the reference's own S and indices: isolates the gather/log/cascade-sum kernel.
    The sums are 100 logs of magnitude ~5 (ulp 4.8e-7) accumulated to ~-420 (ulp 3.05e-5).  torch.log is
    MKL's vsLn (practically correctly rounded); the kernel's default log agrees with it on 99.9 % of the
    arguments, so nearly every sum lands on the reference's bits and the rest one ulp away.  The optional
    fast log (v_log_f32 based, <= ~1.5 ulp) is checked at the looser statistical tolerance."""
    z, E_img, E_txt, A, P = util.case_inputs(name)
    K = int(z["top_k"])
    S = z["S"] if "S" in z else oracle.row_softmax(P, 10.0)
    C = S.shape[1]
    Sp = np.zeros((S.shape[0], (C + 63) // 64 * 64), np.float32)
    Sp[:, :C] = S
    idx = T(z["inds"].T.astype(np.int32), dev)
    p = T(oracle.p_in_examples(K), dev)
    got = core.wpmi_score(T(Sp, dev)[:, :C], idx, p, 1e-7, soft=True).cpu().numpy()
    ref = z["pdge"] if "S" in z else oracle.wpmi_score(S, z["inds"], oracle.p_in_examples(K), np.float32(1e-7), 1)
    util.assert_sim_boundary(got, ref, "pdge " + name)
    assert (got == ref).mean() >= 0.999 or got.size < 5000
    fast = core.wpmi_score(T(Sp, dev)[:, :C], idx, p, 1e-7, soft=True, fast_log=True).cpu().numpy()
    util.assert_sim_close(fast, ref, "pdge fast-log " + name)
    assert np.abs(fast - ref).max() <= 5 * np.spacing(np.abs(ref).max())
    # unpadded S (odd leading dimension): the 1-concept-per-lane variant gives the same bits
    got1 = core.wpmi_score(T(S, dev), idx, p, 1e-7, soft=True).cpu().numpy()
    assert np.array_equal(got1, got)
    # hard WPMI terms, K = 28
    K2 = min(28, S.shape[0])
    _, i2 = oracle.col_topk(A, K2)
    got2 = core.wpmi_score(T(Sp, dev)[:, :C], T(i2.T.astype(np.int32), dev), None, 1e-7, soft=False).cpu().numpy()
    ref2 = oracle.wpmi_score(S, i2, None, np.float32(1e-7), 0)
    util.assert_sim_boundary(got2, ref2, "hard pdge " + name)


def test_accurate_log_matches_torch(core, dev):
    """The accurate log of K4 in isolation.  Hard WPMI with min_prob = 0 and K = 1 makes the output the log of the
    gathered entry (generic kernel); K = 4 with three gathers of an all-ones row (log 1 = 0, and x + 0 is exact)
    does the same through the XCD-sliced kernel.  Both must be within 1 ulp of the correctly rounded log and equal it
    on >= 99.99 % of the arguments.  torch.log itself depends on the host: in the build container (where the goldens
    were made) it equals the correctly rounded log on 99.993 % of these arguments, on the GPU box's CPU on 97.4 %
    (tests/test_log_table_cpu.py pins the former); here it is only required to stay within 1 ulp of ours."""
    rng = np.random.default_rng(9)
    N, C = 4096, 768
    x = np.concatenate([np.exp(rng.uniform(np.log(2.0 ** -25), np.log(1.99), (N // 2, C))),
                        rng.uniform(0.002, 0.06, (N // 2, C))]).astype(np.float32)
    x[N - 1, :] = 1.0
    x[0, :8] = [1.0, np.nextafter(np.float32(1), np.float32(0)), np.nextafter(np.float32(1), np.float32(2)), 0.5,
                2.0 ** -25, 1e-7, 1.5, 0.99999]
    cr = np.log(x.astype(np.float64)).astype(np.float32)
    t = torch.log(torch.from_numpy(x)).numpy()
    rows = np.arange(N, dtype=np.int32)
    idx1 = T(rows[:, None].copy(), dev)                                              # [U = N, K = 1]
    idx4 = T(np.stack([rows, np.full(N, N - 1), np.full(N, N - 1), np.full(N, N - 1)], 1).astype(np.int32), dev)
    S = T(x, dev)
    for idx in (idx1, idx4):
        got = core.wpmi_score(S, idx, None, 0.0, soft=False, split=C).cpu().numpy()
        ulp = np.abs(got.view(np.int32).astype(np.int64) - cr.view(np.int32).astype(np.int64))
        assert ulp.max() <= 1
        assert (got == cr).mean() >= 0.9999
        assert np.abs(got.view(np.int32).astype(np.int64) - t.view(np.int32).astype(np.int64)).max() <= 1 + (t != cr).any()
        assert got[0, 0] == 0.0
    # arguments outside the table (>= 2, zero, denormal, inf) take the libm path
    y = np.array([[2.0, 3.5, 1e6, 0.0, 1e-40, np.inf, 2.0 ** -31, 1.0]], np.float32).repeat(4, 0)
    y[3, :] = 1.0
    g = core.wpmi_score(T(y, dev), T(np.array([[0]], np.int32), dev), None, 0.0, soft=False, split=8).cpu().numpy()
    with np.errstate(divide="ignore"):
        ref = np.log(y[0].astype(np.float64)).astype(np.float32)
    assert np.array_equal(g[0, 3], ref[3]) and np.all(np.abs(g[0, [0, 1, 2, 4, 6]] - ref[[0, 1, 2, 4, 6]]) <= 2e-6 * np.abs(ref[[0, 1, 2, 4, 6]]))
    assert g[0, 5] == np.inf and g[0, 7] == 0.0


def test_wpmi_score_trusted_equals_checked(core, dev, oracle):
    """MCD_WPMI_S_IS_PROB (no range check in front of the log table) gives the same bits as the checked kernel."""
    z = util.golden("main")
    K = int(z["top_k"])
    S = z["S"]
    C = S.shape[1]
    Sp = np.zeros((S.shape[0], 768), np.float32)
    Sp[:, :C] = S
    idx = T(z["inds"].T.astype(np.int32), dev)
    p = T(oracle.p_in_examples(K), dev)
    a = core.wpmi_score(T(Sp, dev)[:, :C], idx, p, 1e-7, soft=True)
    b = core.wpmi_score(T(Sp, dev)[:, :C], idx, p, 1e-7, soft=True, s_is_prob=True)
    assert torch.equal(a, b)
    a = core.wpmi_score(T(Sp, dev)[:, :C], idx[:, :28].contiguous(), None, 1e-7, soft=False)
    b = core.wpmi_score(T(Sp, dev)[:, :C], idx[:, :28].contiguous(), None, 1e-7, soft=False, s_is_prob=True)
    assert torch.equal(a, b)
    util.assert_sim_boundary(b.cpu().numpy(), oracle.wpmi_score(S, z["inds"][:28], None, np.float32(1e-7), 0), "trusted hard")


@pytest.mark.parametrize("shape", [(400, 100, 5, 100), (300, 763, 3, 100), (200, 40, 4, 28), (600, 70, 2, 333),
                                   (64, 7, 3, 50), (64, 5, 2, 17)])
def test_wpmi_score_summation_order_is_atens(core, dev, oracle, shape):
    """Bit-exact check of the summation order.  Hard-WPMI terms log(g + min_prob) with min_prob = 2^-30 and
    g = 2^-k - 2^-30 make every log argument an exact power of two, whose log both sides round
    identically; what remains is the ORDER of the K additions, which must be ATen's: cascade below
    `split`, row_sum (4 interleaved partials) from `split` on."""
    N, C, U, K = shape
    rng = np.random.default_rng(11)
    k = rng.integers(6, 26, (N, C))   # log arguments stay inside the log table (>= 2^-25)
    S = (np.ldexp(1.0, -k) - 2.0 ** -30).astype(np.float32)
    mp = np.float32(2.0 ** -30)
    assert np.all(np.log2((S + mp).astype(np.float64)) == -k)
    A = rng.standard_normal((N, U)).astype(np.float32)
    _, idx = oracle.col_topk(A, K)
    ref = oracle.wpmi_score(S, idx, None, mp, 0)
    split = oracle.sum_split(C)
    all_cascade = oracle.wpmi_score(S, idx, None, mp, 0, split=C)
    # the two orders do differ on these data (a handful of sums is too few to be sure of it)
    assert C == split or ref[:, split:].size < 8 or (ref[:, split:] != all_cascade[:, split:]).any()
    d_idx = T(idx.T.astype(np.int32), dev)
    got = core.wpmi_score(T(S, dev), d_idx, None, float(mp), soft=False).cpu().numpy()
    assert np.array_equal(got, ref)
    got_c = core.wpmi_score(T(S, dev), d_idx, None, float(mp), soft=False, split=C).cpu().numpy()
    assert np.array_equal(got_c, all_cascade)
    Sp = np.zeros((N, (C + 63) // 64 * 64), np.float32)     # padded leading dimension: 2 concepts per lane
    Sp[:, :C] = S
    got_p = core.wpmi_score(T(Sp, dev)[:, :C], d_idx, None, float(mp), soft=False).cpu().numpy()
    assert np.array_equal(got_p, ref)


@pytest.mark.parametrize("name", CASES)
def test_logsumexp_sub_given_reference_inputs(core, dev, oracle, name):
    z = util.golden(name)
    got = core.logsumexp_sub(T(z["pdge"], dev), 1.0).cpu().numpy()
    assert np.abs(got - z["soft_wpmi"]).max() <= 6.2e-5     # one ulp of prob_d at [512,1024)
    # lam = 0.6 (wpmi), against the oracle
    lam = float(np.float32(0.6))
    got = core.logsumexp_sub(T(z["pdge"], dev), lam).cpu().numpy()
    assert np.abs(got - oracle.logsumexp_sub(z["pdge"], lam)).max() <= 6.2e-5


def test_logsumexp_sub_segments_and_ragged_rows(core, dev, oracle):
    """Several layers in one launch, row counts that exercise every remainder of ATen's order."""
    rng = np.random.default_rng(2)
    C = 100
    sizes = [1, 2, 3, 4, 5, 15, 16, 17, 63, 64, 65, 100, 255, 256, 257, 1100]
    x = (rng.standard_normal((sum(sizes), C)) * 20 - 450).astype(np.float32)
    offs = np.concatenate([[0], np.cumsum(sizes)]).tolist()
    got = core.logsumexp_sub(T(x, dev), 1.0, seg_offsets=offs).cpu().numpy()
    for s, (a, b) in enumerate(zip(offs[:-1], offs[1:])):
        ref = oracle.logsumexp_sub(x[a:b], 1.0)
        d = np.abs(got[a:b] - ref)
        assert d.max() <= 6.2e-5, (sizes[s], d.max())


@pytest.mark.parametrize("name", ["tiny", "main", "relu", "kfull", "n1000"])
def test_row_topk(core, dev, name):
    z = util.golden(name)
    sim = z["soft_wpmi"]
    k = min(10, sim.shape[1])
    v, i = core.row_topk(T(sim, dev), k)
    assert np.array_equal(i.cpu().numpy(), z["ids10"]) and np.array_equal(v.cpu().numpy(), z["vals10"])
    v, i = core.row_topk(T(sim, dev), 1)   # torch.max(sim, dim=1)
    assert np.array_equal(i.cpu().numpy()[:, 0], z["imax"]) and np.array_equal(v.cpu().numpy()[:, 0], z["vmax"])


def test_row_topk_edges(core, dev):
    sim = torch.zeros(3, 763, device=dev)          # U == 1 style all-ties rows: lowest concept index first
    v, i = core.row_topk(sim, 10)
    assert i.cpu().numpy().tolist() == [list(range(10))] * 3
    with pytest.raises(RuntimeError, match="selected index k out of range"):
        core.row_topk(torch.zeros(2, 4, device=dev), 5)
    rng = np.random.default_rng(0)
    for C in (1, 2, 63, 64, 65, 763, 10000):
        s = rng.standard_normal((5, C)).astype(np.float32)
        k = min(10, C)
        v, i = core.row_topk(T(s, dev), k)
        tv, ti = torch.topk(torch.from_numpy(s), k=k, dim=1)
        assert np.array_equal(i.cpu().numpy(), ti.numpy()) and np.array_equal(v.cpu().numpy(), tv.numpy())
    # long rows take 16-byte loads: a tail that is not a multiple of 4, an unaligned leading dimension (scalar loads),
    # ties across the vectorised part (lowest column first) and an all-ties row (insertion-list path)
    for C, ld in ((1027, 1028), (4099, 4101), (10000, 10112)):
        buf = torch.from_numpy(rng.standard_normal((6, ld)).astype(np.float32)).to(dev)
        s = buf[:, :C]
        s[1, 5] = s[1, 900] = s[1, C - 1] = 9.0        # three-way tie for rank 0..2
        s[2] = 0.25                                     # constant row
        v, i = core.row_topk(s, 10)
        sc = s.cpu()
        for r in range(6):
            order = sorted(range(C), key=lambda c: (-float(sc[r, c]), c))[:10]     # ties -> lower column
            assert i[r].cpu().tolist() == order, (C, ld, r)
            assert v[r].cpu().tolist() == [float(sc[r, c]) for c in order]
    # rows of 4 096 concepts or more run on K3's workgroup-per-row kernels (the "streaming kernel" mark lives in the row's
    # first output index): NaN rows, constant rows, k = 1, and a row length beyond the register-resident classes
    for C in (4096, 10000, 30000):
        s = torch.from_numpy(rng.standard_normal((7, C)).astype(np.float32)).to(dev)
        s[1, 17] = float("nan"); s[1, C - 2] = float("nan"); s[1, 3] = float("inf")
        s[2] = -1.5
        s[3, ::2] = 2.0                                  # half the row ties at the top
        for k in (1, 10):
            v, i = core.row_topk(s, k)
            sc = s.cpu()
            for r in range(7):
                key = lambda c: (0 if sc[r, c] != sc[r, c] else 1, -float(sc[r, c]) if sc[r, c] == sc[r, c] else 0.0, c)
                order = sorted(range(C), key=key)[:k]   # NaN first, then value descending, ties -> lower column
                assert i[r].cpu().tolist() == order, (C, k, r)
                got, want = v[r].cpu(), sc[r, order]
                assert torch.equal(torch.isnan(got), torch.isnan(want)) and torch.equal(got[~torch.isnan(got)], want[~torch.isnan(want)])


def _topk_order(row, k):
    """torch.topk's order as the kernels define it where torch leaves it open: NaN first, then value descending, ties to the
    lower column.  The kernels compare order-preserving integer keys (mcd_f2key), which refine the float order on signed zeros:
    +0 ranks above -0 (IEEE totalOrder).  torch's own order on that tie is arbitrary (topk([0, -0, 0, -0, 0], 4) -> [2, 4, 3, 0]
    on the fixture host), so no parity exists to keep there."""
    C = row.shape[0]
    key = lambda c: (0 if row[c] != row[c] else 1, -float(row[c]) if row[c] == row[c] else 0.0,
                     1 if (row[c] == 0 and np.signbit(row[c])) else 0, c)
    return sorted(range(C), key=key)[:k]


@pytest.mark.parametrize("C", [5, 65, 763, 1024])
@pytest.mark.parametrize("k", [1, 4, 10, 16])
def test_row_topk_short_rows_with_nan_inf_and_sliced_views(core, dev, C, k):
    """row_topk_short_kernel takes every C <= 1 024 -- the production 763-concept K6 path -- and had been tested with
    finite data and all-ties rows only.  NaN (ranks first), +-inf, -0 / +0 (+0 first: the key order), rows of one value, NaN-heavy rows that overflow
    the 64-slot list (the insertion fallback together with NaNs), k up to the KK = 16 instantiation; and the same on a sliced view
    (leading dimension C + 1, base offset of one element: rows that are not 16-byte aligned)."""
    if k > C:
        pytest.skip("k > C")
    rng = np.random.default_rng(C * 31 + k)
    R = 12
    s = rng.standard_normal((R, C)).astype(np.float32)
    s[1, 0] = np.nan
    s[1, C - 1] = np.inf
    s[2, C // 2] = -np.inf
    s[2, 1 % C] = np.nan
    s[3, :] = np.inf                                  # all +inf: ties to the lower column
    s[4, :] = -np.inf
    s[5, ::2] = 0.0
    s[5, 1::2] = -0.0                                 # +0 above -0 (the integer keys' order), columns ascending within each
    s[6, :] = np.nan                                  # all NaN
    s[7, : max(1, C // 2)] = np.nan                   # more NaNs than the compaction list holds (C = 763, 1024)
    s[8, -1] = np.nan                                 # the row's last element (the tail quad when C % 4 != 0)
    s[9, :] = 3.5
    s[9, C - 1] = np.nan
    s[10, : min(C, 70)] = np.inf                      # > 64 ties at the top
    for view in ("dense", "sliced"):
        if view == "dense":
            t = torch.from_numpy(s).to(dev)
        else:
            buf = torch.full((R * (C + 1) + 8,), 7e30, dtype=torch.float32, device=dev)      # poison around the rows
            t = torch.as_strided(buf, (R, C), (C + 1, 1), 1)
            t.copy_(torch.from_numpy(s))
        v, i = core.row_topk(t, k)
        v, i = v.cpu().numpy(), i.cpu().numpy()
        for r in range(R):
            order = _topk_order(s[r], k)
            assert i[r].tolist() == order, (view, C, k, r, i[r].tolist(), order)
            want = s[r, order]
            assert np.array_equal(np.isnan(v[r]), np.isnan(want)) and np.array_equal(v[r][~np.isnan(want)], want[~np.isnan(want)])
            assert np.array_equal(np.signbit(v[r][~np.isnan(want)]), np.signbit(want[~np.isnan(want)]))


@pytest.mark.parametrize("C", [257, 763, 1030])
def test_row_softmax_and_normalize_on_sliced_views(core, dev, oracle, C):
    """the buffer-load paths of row_softmax_lds_kernel<128> and normalize_to_bf16_kernel at a row tail with C % 4 != 0 on
    a SLICED view (leading dimension C + 1, base offset of one element): same bits as on the dense copy, nothing read past a row."""
    rng = np.random.default_rng(C)
    N = 37
    P = (rng.standard_normal((N, C)) * 0.3).astype(np.float32)
    buf = torch.full((N * (C + 1) + 8,), float("nan"), dtype=torch.float32, device=dev)      # NaN around the rows: a stray read shows
    Pv = torch.as_strided(buf, (N, C), (C + 1, 1), 1)
    Pv.copy_(torch.from_numpy(P))
    S_dense = core.row_softmax(torch.from_numpy(P).to(dev), 10.0)
    S_view = core.row_softmax(Pv, 10.0)
    assert torch.equal(S_dense, S_view) and bool(torch.isfinite(S_view).all())
    assert np.array_equal(S_view.cpu().numpy(), oracle.row_softmax(P, 10.0))
    # embed_gemm_exp(normalize=True): the images operand as a sliced view
    D = C if C <= 1030 else 512
    I = rng.standard_normal((N, D)).astype(np.float32)
    T = torch.from_numpy(rng.standard_normal((50, D)).astype(np.float32)).to(dev)
    bufI = torch.full((N * (D + 1) + 8,), float("nan"), dtype=torch.float32, device=dev)
    Iv = torch.as_strided(bufI, (N, D), (D + 1, 1), 1)
    Iv.copy_(torch.from_numpy(I))
    E0, r0 = core.embed_gemm_exp(torch.from_numpy(I).to(dev), T, 10.0, normalize=True)
    E1, r1 = core.embed_gemm_exp(Iv, T, 10.0, normalize=True)
    assert torch.equal(E0.view(torch.int16), E1.view(torch.int16)) and torch.equal(r0, r1) and bool(torch.isfinite(r1).all())


def test_hook_pool(core, dev, oracle):
    rng = np.random.default_rng(4)
    N, Utot = 37, 90
    for neuron_major in (True, False):
        dst = torch.full((Utot, N) if neuron_major else (N, Utot), -7.0, device=dev)
        ref = np.full((N, Utot), -7.0, np.float32)
        col = 0
        for shape, mode in [((5, 24, 7, 7), "avg"), ((5, 10, 12, 12), "avg"), ((5, 16, 3, 5), "max"),
                            ((5, 13, 8), "avg"), ((5, 11), "avg")]:
            x = rng.standard_normal(shape).astype(np.float32)
            n = core.hook_pool(T(x, dev), mode, dst, 20, col, neuron_major)
            ref[20:25, col:col + n] = oracle.hook_pool(x, mode)
            col += n
        got = dst.cpu().numpy().T if neuron_major else dst.cpu().numpy()
        assert np.abs(got - ref).max() <= 1e-6      # mean over H*W: fp32 sum order differs from the oracle's fp64
        assert np.array_equal(got[:20], ref[:20])   # untouched rows stay untouched
    with pytest.raises(IndexError):
        core.hook_pool(torch.zeros(5, 4, device=dev), "avg", torch.zeros(3, 3, device=dev), 0, 0, False)
    # NaN in a plane: output.mean / output.amax (reference utils.py:33-52) both propagate it
    for hw in ((4, 4), (3, 5)):                     # the float4 and the scalar load paths
        x = torch.randn(2, 3, *hw, device=dev)
        x[1, 2, 1, 1] = float("nan")
        for mode, f in (("avg", lambda t: t.mean(dim=[2, 3])), ("max", lambda t: t.amax(dim=[2, 3]))):
            dst = torch.zeros(2, 3, device=dev)
            core.hook_pool(x, mode, dst, 0, 0, False)
            want = f(x)
            assert torch.equal(torch.isnan(dst), torch.isnan(want)) and bool(torch.isnan(dst[1, 2]))
            assert torch.allclose(dst[~torch.isnan(dst)], want[~torch.isnan(want)], atol=1e-6)


def test_transpose(core, dev):
    rng = np.random.default_rng(6)
    for (N, U) in [(1, 1), (65, 3), (130, 257), (1000, 48)]:
        A = rng.standard_normal((N, U)).astype(np.float32)
        assert np.array_equal(core.transpose(T(A, dev)).cpu().numpy(), A.T)


def test_logsumexp_panel_kernel_equals_three_pass(core, dev, tmp_path):
    """K5's one-pass LDS-panel kernel performs the same operations in the same order as the three-launch path
    (forced in a child process on the dev library with MCD_LSE_NO_PANEL=1): identical bits, for cascade and row_sum columns, ragged
    segments, and the 32 / 16 / 8-column panel widths."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cases = [([0, 768, 1536, 1600, 1607], 763), ([0, 1500], 100), ([0, 70, 2700], 40), ([0, 5, 6], 7),
             ([0, 768, 1000, 1001], 64), ([0, 130, 898], 208)]   # pitches of whole 16-byte groups: the panel kernel's float4 path
    code = r"""
import sys, numpy as np, torch
sys.path.insert(0, %r)
import mammo_clip_dissect_amd
from mammo_clip_dissect_amd import core
cases = %r
dev = torch.device("cuda:0")
for k, (offs, C) in enumerate(cases):
    g = torch.Generator().manual_seed(100 + k)
    x = (torch.randn(offs[-1], C, generator=g) * 3 - 400).to(dev)
    out = core.logsumexp_sub(x, 0.6, seg_offsets=offs)
    np.save(sys.argv[1] + "/lse_%%d.npy" %% k, out.cpu().numpy())
""" % (root, cases)
    # the knob exists in the dev build of the library only (make dev); the child loads that one, this process the product
    dev_lib = os.path.join(root, "mammo-clip-dissect_amd", "csrc", "libmcd_hip_dev.so")
    assert os.path.exists(dev_lib), "libmcd_hip_dev.so not built (make -C mammo-clip-dissect_amd/csrc dev)"
    env = dict(os.environ, MCD_LSE_NO_PANEL="1", MCD_LIB_PATH=dev_lib)
    r = subprocess.run([sys.executable, "-c", code, str(tmp_path)], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    for k, (offs, C) in enumerate(cases):
        g = torch.Generator().manual_seed(100 + k)
        x = (torch.randn(offs[-1], C, generator=g) * 3 - 400).to(dev)
        got = core.logsumexp_sub(x, 0.6, seg_offsets=offs).cpu().numpy()
        assert np.array_equal(got, np.load(str(tmp_path) + "/lse_%d.npy" % k)), (offs, C)


@pytest.mark.parametrize("shape", [(300, 1500), (257, 4099), (130, 10000), (64, 16384), (40, 16400), (33, 1025)])
def test_row_softmax_long_rows_bit_exact(core, dev, oracle, shape):
    """Rows longer than 1024 concepts (one workgroup per row, the row in LDS; 16 400 > the LDS kernel's limit takes
    the generic kernel): the same bits as the oracle's restatement of ATen's softmax, aligned and unaligned pitches."""
    N, C = shape
    g = torch.Generator().manual_seed(C)
    P = torch.randn(N, C, generator=g) * 0.3
    ref = oracle.row_softmax(P.numpy(), 10.0)
    got = core.row_softmax(P.to(dev), 10.0)
    assert np.array_equal(got.cpu().numpy(), ref)
    assert got.stride(0) % 192 == 0 and float(torch.as_strided(got, (N, got.stride(0)), (got.stride(0), 1))[:, C:].abs().sum()) == 0.0
    Pp = torch.zeros(N, C + 3)           # a pitch that is not a multiple of 4 floats: the scalar path
    Pp[:, :C] = P
    got2 = core.row_softmax(Pp.to(dev)[:, :C], 10.0)
    assert np.array_equal(got2.cpu().numpy(), ref)


@pytest.mark.parametrize("shape", [(3, 197, 12), (2, 1, 2), (2, 5, 1), (1, 32, 3), (2, 33, 2), (1, 64, 1), (2, 100, 4),
                                   (1, 224, 2), (1, 256, 2), (5, 50, 12)])
def test_vit_attention_matches_sdpa(core, dev, shape):
    """K9 against PyTorch's fp32 attention (math definition, computed in float64): every tile class -- one partial
    tile, exact multiples of 32, a last tile of 1 / 5 keys, the 8-wave maximum -- and large-magnitude scores
    (the online softmax must rescale)."""
    B, T, H = shape
    g = torch.Generator(device=dev).manual_seed(B * 1000 + T)
    for scale in (1.0, 6.0):
        qkv = torch.randn(B, T, 3 * H * 64, device=dev, generator=g) * scale
        out = core.vit_attention(qkv, H)
        q, k, v = qkv.double().view(B, T, 3, H, 64).permute(2, 0, 3, 1, 4)
        ref = (torch.softmax(q @ k.transpose(-1, -2) / 8.0, dim=-1) @ v).transpose(1, 2).reshape(B, T, H * 64)
        err = (out.double() - ref).abs().max().item()
        # the same product chain in fp32 (bmm, softmax, bmm): the error fp32 attention has on these inputs -- it
        # grows with the score magnitude (|s| * 2^-24 relative per probability), so it is the yardstick
        q32, k32, v32 = qkv.view(B, T, 3, H, 64).permute(2, 0, 3, 1, 4)
        ref32 = (torch.softmax(q32 @ k32.transpose(-1, -2) / 8.0, dim=-1) @ v32).transpose(1, 2).reshape(B, T, H * 64)
        err32 = (ref32.double() - ref).abs().max().item()
        assert err <= 3e-6 * max(1.0, ref.abs().max().item()) + 3 * err32, (shape, scale, err, err32)
    with pytest.raises(Exception):
        core.vit_attention(torch.zeros(1, 257, 3 * 64, device=dev), 1)     # one wave per 32 queries, 8 waves
    with pytest.raises((TypeError, ValueError)):
        core.vit_attention(torch.zeros(1, 8, 3 * 64 + 4, device=dev), 1)


def test_vit_tower_uses_the_hip_attention(mcd, dev):
    """The image tower with K9 against the same tower on PyTorch's SDPA: same features to fp32 accuracy."""
    from mammo_clip_dissect_amd.concept_vit import data_utils
    torch.manual_seed(0)
    tower = data_utils.ViTTower(image_size=64, depth=2).to(dev).eval()
    for p in tower.parameters():
        torch.nn.init.normal_(p, std=0.05)
    x = torch.randn(3, 3, 64, 64, device=dev)
    with torch.no_grad():
        assert data_utils.HIP_ATTENTION
        a = tower(x)
        data_utils.HIP_ATTENTION = False
        try:
            b = tower(x)
        finally:
            data_utils.HIP_ATTENTION = True
    assert (a - b).abs().max().item() <= 1e-4 * max(1.0, b.abs().max().item())


@pytest.mark.parametrize("shape", [(197 * 5, 768, 768), (64, 128, 3072), (1000, 768, 3072), (33, 40, 100)])
def test_linear_residual_matches_torch(core, dev, shape):
    """libmcd_blaslt.so: res + h @ W^T + b in one hipBLASLt GEMM against PyTorch's linear + add (both fp32 MFMA, so
    they differ by accumulation order only); new output, in place, without residual, without bias."""
    if not core.linear_residual_available():
        pytest.skip("libmcd_blaslt.so not built")
    M, N, K = shape
    g = torch.Generator(device=dev).manual_seed(M + N)
    h = torch.randn(M, K, device=dev, generator=g)
    W = torch.randn(N, K, device=dev, generator=g) * 0.05
    b = torch.randn(N, device=dev, generator=g)
    res = torch.randn(M, N, device=dev, generator=g)
    ref = (res.double() + h.double() @ W.double().T + b.double())
    tol = 2e-5 * float(ref.abs().max())
    keep = res.clone()
    out = core.linear_residual(res, h, W, b)
    assert torch.equal(res, keep) and out.data_ptr() != res.data_ptr()        # the residual input is not touched
    assert float((out.double() - ref).abs().max()) <= tol
    r2 = res.clone()
    assert core.linear_residual(r2, h, W, b, out=r2) is r2                    # in place
    assert float((r2.double() - ref).abs().max()) <= tol
    nores = core.linear_residual(None, h, W, b)
    assert float((nores.double() - (ref - res.double())).abs().max()) <= tol
    nobias = core.linear_residual(res, h, W, None)
    assert float((nobias.double() - (ref - b.double())).abs().max()) <= tol
    h3 = h.view(1, M, K)                                                      # leading batch dimensions
    assert core.linear_residual(res.view(1, M, N), h3, W, b).shape == (1, M, N)
    with pytest.raises(ValueError):
        core.linear_residual(res, h, W[:, :K - 1].contiguous(), b)


def test_vit_tower_fused_residual(mcd, dev):
    """The tower with the fused residual GEMMs against the same tower on nn.Linear + add."""
    from mammo_clip_dissect_amd import core
    from mammo_clip_dissect_amd.concept_vit import data_utils
    if not core.linear_residual_available():
        pytest.skip("libmcd_blaslt.so not built")
    torch.manual_seed(1)
    tower = data_utils.ViTTower(image_size=64, depth=2).to(dev).eval()
    for p in tower.parameters():
        torch.nn.init.normal_(p, std=0.05)
    x = torch.randn(3, 3, 64, 64, device=dev)
    seen = []
    hk = tower.encoder.layer[0].register_forward_hook(lambda m, i, o: seen.append((i[0].clone(), i[0], o)))
    with torch.no_grad():
        assert data_utils.FUSED_RESIDUAL
        a = tower(x)
        data_utils.FUSED_RESIDUAL = False
        try:
            b = tower(x)
        finally:
            data_utils.FUSED_RESIDUAL = True
    hk.remove()
    assert (a - b).abs().max().item() <= 1e-4 * max(1.0, b.abs().max().item())
    before, inp, outp = seen[0]
    assert torch.equal(before, inp) and outp.data_ptr() != inp.data_ptr()     # a block never writes into its input


@pytest.mark.parametrize("shape", [(197 * 3, 768), (5, 4), (1000, 512), (77, 1024), (33, 2048), (64, 260), (9, 1540)])
def test_layer_norm_matches_torch(core, dev, shape):
    """K10 against torch.nn.functional.layer_norm computed in float64: every registers-per-lane class, rows that do
    not fill a workgroup, a large common offset (two-pass statistics do not cancel)."""
    R, D = shape
    g = torch.Generator(device=dev).manual_seed(R + D)
    for offset in (0.0, 50.0):
        x = torch.randn(R, D, device=dev, generator=g) * 2.0 + offset
        w = torch.randn(D, device=dev, generator=g)
        b = torch.randn(D, device=dev, generator=g)
        ref = torch.nn.functional.layer_norm(x.double(), (D,), w.double(), b.double(), 1e-12)
        got = core.layer_norm(x, w, b, 1e-12)
        assert float((got.double() - ref).abs().max()) <= 3e-6 * max(1.0, float(ref.abs().max())) * (1 + offset)
        x3 = x.view(1, R, D)
        assert torch.equal(core.layer_norm(x3, w, b, 1e-12).view(R, D), got)
    with pytest.raises(Exception):
        core.layer_norm(torch.zeros(4, 6, device=dev), torch.ones(6, device=dev), torch.zeros(6, device=dev), 1e-5)


@pytest.mark.parametrize("shape", [(3, 3, 224, 224, 16), (2, 3, 64, 64, 16), (1, 1, 32, 64, 8), (5, 3, 48, 48, 4)])
def test_patchify_is_the_conv_operand(core, dev, shape):
    """K11: a pure permutation (exact); times the conv weight viewed as [dim, Cin*P*P] it is the patch-embedding conv."""
    B, Cin, H, W, P = shape
    g = torch.Generator(device=dev).manual_seed(H + P)
    x = torch.randn(B, Cin, H, W, device=dev, generator=g)
    got = core.patchify(x, P)
    nH, nW = H // P, W // P
    ref = x.view(B, Cin, nH, P, nW, P).permute(0, 2, 4, 1, 3, 5).reshape(B, nH * nW, Cin * P * P)
    assert got.shape == (B, 1 + nH * nW, Cin * P * P)
    assert torch.equal(got[:, 1:], ref) and float(got[:, 0].abs().max()) == 0.0
    conv = torch.nn.Conv2d(Cin, 24, P, P).to(dev)
    with torch.no_grad():
        want = conv(x).flatten(2).transpose(1, 2)
        have = got[:, 1:] @ conv.weight.view(24, -1).T + conv.bias
    assert float((want - have).abs().max()) <= 1e-4


def test_vit_tower_embed_as_gemm(mcd, dev):
    """ViTTower.embed on K11 + the fused GEMM against Conv2d + cat + add."""
    from mammo_clip_dissect_amd import core
    from mammo_clip_dissect_amd.concept_vit import data_utils
    if not core.linear_residual_available():
        pytest.skip("libmcd_blaslt.so not built")
    torch.manual_seed(2)
    tower = data_utils.ViTTower(image_size=64, depth=1).to(dev).eval()
    for p in tower.parameters():
        torch.nn.init.normal_(p, std=0.05)
    x = torch.randn(4, 3, 64, 64, device=dev)
    with torch.no_grad():
        a = tower.embed(x)
        a2 = tower.embed(x[:3])                      # another batch size: the cached residual operand is rebuilt
        data_utils.FUSED_RESIDUAL = False
        try:
            b = tower.embed(x)
        finally:
            data_utils.FUSED_RESIDUAL = True
    assert a.shape == b.shape == (4, 17, 768)
    assert float((a - b).abs().max()) <= 2e-5 * max(1.0, float(b.abs().max()))
    assert float((a2 - b[:3]).abs().max()) <= 2e-5 * max(1.0, float(b.abs().max()))


# ---- the stress chain: K1s (GEMM + exp epilogue, bf16 out) and K4s (bf16 gather) ---------------------------------
@pytest.mark.parametrize("shape", [(1000, 763, 512, 10.0), (600, 10000, 512, 10.0), (257, 193, 70, 2.0), (5, 3, 8, 10.0),
                                   (3000, 1000, 512, 10.0)])
def test_embed_gemm_exp(core, dev, shape):
    """E = bf16(exp(a (P - 1))) and rinv = 1 / rowsum straight from the MFMA accumulators (no fp32 P, no K2).  bf16
    operands move P by up to ~4e-3 (=> a * 4e-3 relative in E) and the bf16 store rounds to 2^-8 relative; the padding
    columns of the row-padded buffer are exactly 0; the row sums cover exactly the C real concepts."""
    N, C, D, a = shape
    g = torch.Generator().manual_seed(N + C)
    I = core.normalize_rows(torch.randn(N, D, generator=g).to(dev))
    T = core.normalize_rows(torch.randn(C, D, generator=g).to(dev))
    E, rinv = core.embed_gemm_exp(I, T, a)
    assert E.dtype == torch.bfloat16 and tuple(E.shape) == (N, C) and E.stride(0) % 128 == 0
    P = (I.double() @ T.double().t())
    ref = torch.exp(a * (P - 1.0))
    rel = (E.double() / ref - 1.0).abs()
    assert float(rel.max()) <= a * 8e-3 + 2.0 ** -7, float(rel.max())
    full = torch.as_strided(E, (N, E.stride(0)), (E.stride(0), 1))
    assert float(full[:, C:].float().abs().max()) == 0.0 if E.stride(0) > C else True
    rs = ref.sum(dim=1)
    assert float((rinv.double() * rs - 1.0).abs().max()) <= a * 4e-3 + 1e-3
    S = E.float() * rinv[:, None]
    assert float((S.sum(dim=1) - 1.0).abs().max()) <= 1e-4     # round 5: the sums are taken over the stored bf16 values (fp32 product rounding only)
    # raw embeddings + normalize=True (K1a folded into the bf16 conversion) give the same thing up to bf16 rounding
    g2 = torch.Generator().manual_seed(N + C)
    Iraw, Traw = torch.randn(N, D, generator=g2).to(dev) * 3.0, torch.randn(C, D, generator=g2).to(dev) * 0.2
    E2, rinv2 = core.embed_gemm_exp(Iraw, Traw, a, normalize=True)
    In, Tn = core.normalize_rows(Iraw), core.normalize_rows(Traw)
    ref2 = torch.exp(a * (In.double() @ Tn.double().t() - 1.0))
    assert float((E2.double() / ref2 - 1.0).abs().max()) <= a * 8e-3 + 2.0 ** -7
    assert float((rinv2.double() * ref2.sum(dim=1) - 1.0).abs().max()) <= a * 4e-3 + 1e-3


@pytest.mark.parametrize("shape", [(16, 32, 128, 10.0), (17, 33, 256, 10.0), (255, 257, 384, 5.0), (1, 1, 128, 10.0), (513, 511, 512, 10.0),
                                   (4100, 300, 640, 3.0), (9000, 9000, 512, 10.0), (25000, 2048, 512, 10.0),
                                   (777, 1300, 200, 4.0), (260, 520, 64, 10.0), (5, 3, 8, 10.0), (20000, 3000, 416, 6.0),
                                   (700, 4000, 1024, 10.0), (257, 193, 70, 2.0), (300, 763, 2048, 10.0)])
def test_embed_gemm_exp_tile_edges(core, dev, shape):
    """The K1s kernel (k_gexp_v6.inc: v_mfma_f32_16x16x32_bf16 on accumulators and fragments named in asm statements, fragment-major
    operands with paired concept rows, the epilogue of a tile inside the next tile's first k-steps, concepts past the last one
    masked in the packed pieces): shapes that walk its 32-concept pair / 16-image block / 256 x 256 tile / row-pitch edges and
    reduction depths that are NOT multiples of 128 (the library pads the operand image: round 5 retired the fallback kernels that
    served them), against float64, three launches each (a DMA or sync race would show as a run-to-run difference)."""
    N, C, D, a = shape
    g = torch.Generator().manual_seed(N * 7 + C)
    I = torch.randn(N, D, generator=g).to(dev)
    T = torch.randn(C, D, generator=g).to(dev)
    In, Tn = core.normalize_rows(I), core.normalize_rows(T)
    ref = torch.exp(a * (In.double() @ Tn.double().t() - 1.0))
    E0 = r0 = None
    for rep in range(3):
        E, rinv = core.embed_gemm_exp(I, T, a, normalize=True)
        torch.cuda.synchronize()
        if rep == 0:
            E0, r0 = E.clone(), rinv.clone()
            assert float((E.double() / ref - 1.0).abs().max()) <= a * 8e-3 + 2.0 ** -7
            assert float((rinv.double() * ref.sum(dim=1) - 1.0).abs().max()) <= a * 4e-3 + 1e-3
            full = torch.as_strided(E, (N, E.stride(0)), (E.stride(0), 1))
            if E.stride(0) > C:
                assert float(full[:, C:].float().abs().max()) == 0.0          # the padding columns: masked pieces, exactly 0
            # the row sums are sums of the STORED bf16 values (row-sum MFMAs on the packed pieces): S = E * rinv sums to 1 up to
            # fp32 rounding, which the fp32-summed-then-rounded form of rounds 2-4 only did to 5e-3
            assert float(((E.double().sum(dim=1) * rinv.double()) - 1.0).abs().max()) <= 1e-5
        else:
            assert torch.equal(E.view(torch.int16), E0.view(torch.int16)) and torch.equal(rinv, r0)


def test_embed_gemm_exp_is_repeatable_at_the_stress_shape(core, dev):
    """The K1s kernel hand-counts its vector-memory and LDS waits across a 4-slot DMA ring, a sync point inside the boundary phase
    and fragment refills of registers that MFMAs of the previous k-step still read (k_gexp_v6.inc): a mis-count shows as a rare
    wrong tile that comes and goes with timing.  60 launches at one rank's share of configs[4] (25 000 x 10 000 x 512: 16 tiles per
    workgroup, every boundary-phase variant) must give the same bits every time, under different amounts of other work in flight."""
    N, C, D, a = 25000, 10000, 512, 10.0
    g = torch.Generator(device=dev).manual_seed(99)
    I = torch.randn(N, D, device=dev, generator=g)
    T = torch.randn(C, D, device=dev, generator=g)
    E0, r0 = core.embed_gemm_exp(I, T, a, normalize=True)
    h0 = (E0.view(torch.int16).to(torch.int64).sum().item(), float(r0.double().sum()))
    E0 = E0.clone()
    filler = torch.empty(64 << 20, dtype=torch.float32, device=dev)
    for rep in range(60):
        if rep % 3 == 1:
            filler.fill_(float(rep))              # a 256 MB write in front: cold caches, busy memory
        elif rep % 3 == 2:
            torch.cuda.synchronize()              # an idle chip in front
        E, r = core.embed_gemm_exp(I, T, a, normalize=True)
        assert (E.view(torch.int16).to(torch.int64).sum().item(), float(r.double().sum())) == h0, rep
        if rep % 20 == 19:
            assert torch.equal(E.view(torch.int16), E0.view(torch.int16))
    # and it is right: a sampled block against float64
    In, Tn = core.normalize_rows(I[:512]), core.normalize_rows(T[:777])
    ref = torch.exp(a * (In.double() @ Tn.double().t() - 1.0))
    assert float((E0[:512, :777].double() / ref - 1.0).abs().max()) <= a * 8e-3 + 2.0 ** -7


def test_embed_gemm_exp_two_set_experiment_gives_the_product_bits(core, dev, tmp_path):
    """Round 5's two-accumulator-set form of K1s (k_gexp_v7.inc: 256 x 128 tiles, the epilogue of tile i spread over the 16
    k-steps of tile i + 1; VERDICT r4 #1b) lost to the product kernel by 20 % and lives in the dev library only
    (MCD_GEMM_EXP_V7=1).  It is kept honest: same operand image, same k order, same pieces, same row-sum MFMAs -- so it must
    produce the product's E and rinv bit for bit, on whole, ragged and masked tiles and with reductions deeper than 512."""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    shapes = [(1000, 763, 512, 10.0), (600, 2000, 512, 10.0), (4100, 300, 640, 3.0), (700, 1300, 1024, 10.0), (129, 257, 512, 5.0),
              (9000, 2048, 512, 10.0)]
    code = r"""
import sys, numpy as np, torch
sys.path.insert(0, %r)
import mammo_clip_dissect_amd
from mammo_clip_dissect_amd import core
dev = torch.device("cuda:0")
for k, (N, C, D, a) in enumerate(%r):
    g = torch.Generator().manual_seed(N * 7 + C)
    I = torch.randn(N, D, generator=g).to(dev); T = torch.randn(C, D, generator=g).to(dev)
    E, rinv = core.embed_gemm_exp(I, T, a, normalize=True)
    np.save(sys.argv[1] + "/e_%%d.npy" %% k, E.view(torch.int16).cpu().numpy()); np.save(sys.argv[1] + "/r_%%d.npy" %% k, rinv.cpu().numpy())
""" % (root, shapes)
    dev_lib = os.path.join(root, "mammo-clip-dissect_amd", "csrc", "libmcd_hip_dev.so")
    assert os.path.exists(dev_lib), "libmcd_hip_dev.so not built (make -C mammo-clip-dissect_amd/csrc dev)"
    env = dict(os.environ, MCD_GEMM_EXP_V7="1", MCD_LIB_PATH=dev_lib)
    r = subprocess.run([sys.executable, "-c", code, str(tmp_path)], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    for k, (N, C, D, a) in enumerate(shapes):
        g = torch.Generator().manual_seed(N * 7 + C)
        I = torch.randn(N, D, generator=g).to(dev)
        T = torch.randn(C, D, generator=g).to(dev)
        E, rinv = core.embed_gemm_exp(I, T, a, normalize=True)
        assert np.array_equal(E.view(torch.int16).cpu().numpy(), np.load(str(tmp_path) + "/e_%d.npy" % k)), (N, C, D)
        assert np.array_equal(rinv.cpu().numpy(), np.load(str(tmp_path) + "/r_%d.npy" % k)), (N, C, D)


def test_embed_gemm_exp_rejects_an_odd_pitch(core, dev):
    """The C ABI refuses an E pitch that is not a multiple of 16 elements (the kernel stores 16-byte pieces at 16-concept steps;
    rounds 2-4 kept a second kernel for such callers) with MCD_E_UNSUPPORTED and a message; the binding's own pitch is a multiple
    of 128."""
    from mammo_clip_dissect_amd import _lib
    L = _lib.load()
    N, C, D = 64, 40, 128
    I = torch.randn(N, D).to(dev)
    T = torch.randn(C, D).to(dev)
    nws = L.mcd_embed_gemm_exp_workspace(N, C, D)
    ws = torch.empty(max(nws, 16), dtype=torch.uint8, device=dev)
    E = torch.empty((N, 56), dtype=torch.bfloat16, device=dev)
    rinv = torch.empty(N, device=dev)
    rc = L.mcd_embed_gemm_exp(I.data_ptr(), D, T.data_ptr(), D, N, C, D, 10.0, 1, E.data_ptr(), 56, rinv.data_ptr(), ws.data_ptr(), nws, None)
    assert rc != 0 and b"multiple of 16" in L.mcd_last_error()
    E = torch.empty((N, 48), dtype=torch.bfloat16, device=dev)
    core.check(L.mcd_embed_gemm_exp(I.data_ptr(), D, T.data_ptr(), D, N, C, D, 10.0, 1, E.data_ptr(), 48, rinv.data_ptr(), ws.data_ptr(), nws, None))
    torch.cuda.synchronize()
    ref = torch.exp(10.0 * (core.normalize_rows(I).double() @ core.normalize_rows(T).double().t() - 1.0))
    assert float((E[:, :C].double() / ref - 1.0).abs().max()) <= 10.0 * 8e-3 + 2.0 ** -7
    assert float(E[:, C:].float().abs().max()) == 0.0


def test_embed_gemm_exp_kernel_timing_hook(core, dev):
    """include/mcd_hip.h's measurement hook: off by default (< 0: nothing recorded on a fresh process would be -1; here: the value
    does not move while timing is off), on -> a positive kernel time smaller than the whole call's."""
    from mammo_clip_dissect_amd import _lib
    L = _lib.load()
    g = torch.Generator().manual_seed(5)
    I = torch.randn(4096, 512, generator=g).to(dev)
    T = torch.randn(2048, 512, generator=g).to(dev)
    core.embed_gemm_exp(I, T, 10.0, normalize=True)
    before = float(L.mcd_embed_gemm_exp_kernel_ms())
    core.embed_gemm_exp(I, T, 10.0, normalize=True)
    assert float(L.mcd_embed_gemm_exp_kernel_ms()) == before                   # timing off: no new event pair
    assert L.mcd_embed_gemm_exp_time_kernel(4) == 0        # 4 back-to-back launches of the kernel between the event pair
    try:
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        core.embed_gemm_exp(I, T, 10.0, normalize=True)
        e.record()
        torch.cuda.synchronize()
        k_ms = float(L.mcd_embed_gemm_exp_kernel_ms())
    finally:
        L.mcd_embed_gemm_exp_time_kernel(0)
    assert 0.0 < k_ms <= s.elapsed_time(e) / 4 * 1.05          # per launch
    # the quotient uses the launches ISSUED: a reduction depth that is not a multiple of 128 takes the same kernel now
    I2, T2 = torch.randn(4096, 416, generator=g).to(dev), torch.randn(2048, 416, generator=g).to(dev)
    assert L.mcd_embed_gemm_exp_time_kernel(4) == 0
    try:
        core.embed_gemm_exp(I2, T2, 10.0, normalize=True)
        torch.cuda.synchronize()
        s.record()
        core.embed_gemm_exp(I2, T2, 10.0, normalize=True)
        e.record()
        torch.cuda.synchronize()
        k2 = float(L.mcd_embed_gemm_exp_kernel_ms())
    finally:
        L.mcd_embed_gemm_exp_time_kernel(0)
    assert 0.25 * s.elapsed_time(e) / 4 <= k2 <= s.elapsed_time(e) / 4 * 1.05


def test_product_library_refuses_the_ablation_knob(core, dev):
    """MCD_GEMM_EXP_ABLATE selects timing kernels that return wrong results by design; they live in the dev build only
    (libmcd_hip_dev.so) and the product library says so instead of silently measuring something else."""
    from mammo_clip_dissect_amd import _lib
    if "dev" in os.path.basename(_lib.LIB_PATH):
        pytest.skip("running against the dev build")
    I = torch.randn(64, 128).to(dev)
    T = torch.randn(64, 128).to(dev)
    os.environ["MCD_GEMM_EXP_ABLATE"] = "4"
    try:
        with pytest.raises(Exception) as ei:
            core.embed_gemm_exp(I, T, 10.0, normalize=True)
        assert "product library" in str(ei.value)
    finally:
        os.environ.pop("MCD_GEMM_EXP_ABLATE", None)


@pytest.mark.parametrize("soft", [True, False])
def test_wpmi_score_bf16(core, dev, soft):
    """K4s against the same expression in float64 on the same bf16 E and rinv: log(1 + p (S - 1) + min_prob) summed over
    the K gathered rows (hard WPMI: log(S + min_prob)); v_log_f32 (1 ulp of log2) + fp32 sums: <= 2e-3 absolute on sums
    of magnitude ~500."""
    N, C, D, U, K = 2000, 1000, 512, 150, 100
    g = torch.Generator().manual_seed(77)
    I = core.normalize_rows(torch.randn(N, D, generator=g).to(dev))
    T = core.normalize_rows(torch.randn(C, D, generator=g).to(dev))
    E, rinv = core.embed_gemm_exp(I, T, 10.0)
    A = torch.randn(N, U, generator=g).to(dev)
    _, idx = core.col_topk(A, K)
    p = (0.998 - (torch.arange(0, K) / K * (0.998 - 0.97))).float().to(dev)
    out = core.wpmi_score_bf16(E, rinv, idx, p if soft else None, 1e-7, soft)
    S = E.double() * rinv.double()[:, None]
    gsel = S[idx.long()]                                        # [U, K, C]
    if soft:
        w = 1.0 + p.double()[None, :, None] * (gsel - 1.0) + 1e-7
    else:
        w = gsel + 1e-7
    ref = torch.log(w).sum(dim=1)
    assert float((out.double() - ref).abs().max()) <= 2e-3, float((out.double() - ref).abs().max())
    # a ragged neuron count and a K that is not a multiple of 8
    out2 = core.wpmi_score_bf16(E, rinv, idx[:37, :91].contiguous(), p[:91].contiguous() if soft else None, 1e-7, soft)
    if soft:
        w2 = 1.0 + p.double()[None, :91, None] * (gsel[:37, :91] - 1.0) + 1e-7
    else:
        w2 = gsel[:37, :91] + 1e-7
    assert float((out2.double() - torch.log(w2).sum(dim=1)).abs().max()) <= 2e-3


@pytest.mark.parametrize("K", [1, 3, 4, 5, 7, 8, 9, 12, 13, 16, 17, 20, 24, 25, 31, 48])
def test_wpmi_score_bf16_batch_edges(core, dev, K):
    """K4s walks the K rows in batches of 16 and takes one log per product of four arguments; the rows that pad the last
    group are neutral.  Every K around those boundaries, a neuron count that leaves lanes idle, C over several slices."""
    N, C, U = 300, 333, 21
    g = torch.Generator().manual_seed(K)
    E = (torch.rand(N, 384, generator=g) * 0.9 + 0.05).to(torch.bfloat16).to(dev)[:, :C]
    rinv = (torch.rand(N, generator=g) * 0.01 + 0.001).to(dev)
    idx = torch.stack([torch.randperm(N, generator=g)[:K] for _ in range(U)]).int().to(dev)
    p = torch.linspace(0.998, 0.97, K).float().to(dev)
    S = E.double() * rinv.double()[:, None]
    gsel = S[idx.long()]
    for soft in (True, False):
        out = core.wpmi_score_bf16(E, rinv, idx, p if soft else None, 1e-7, soft)
        w = 1.0 + p.double()[None, :, None] * (gsel - 1.0) + 1e-7 if soft else gsel + 1e-7
        ref = torch.log(w).sum(dim=1)
        assert float((out.double() - ref).abs().max()) <= 1e-5 * K * float(torch.log(w).abs().max()) + 1e-6


def test_wpmi_score_bf16_needs_its_workspace(core, dev):
    """The C entry point refuses a missing or short workspace (the per-(neuron, rank) {row, p rinv[row]} array) with the
    library's workspace status instead of writing past it; the size it asks for is 8 bytes per (neuron, rank) + flags."""
    from mammo_clip_dissect_amd import _lib
    L = _lib.load()
    N, C, U, K = 64, 128, 5, 12
    g = torch.Generator().manual_seed(3)
    E = (torch.rand(N, 128, generator=g) * 0.5 + 0.1).to(torch.bfloat16).to(dev)
    rinv = torch.full((N,), 1e-2, device=dev)
    idx = torch.stack([torch.randperm(N, generator=g)[:K] for _ in range(U)]).int().to(dev)
    out = torch.empty(U, C, device=dev)
    need = int(L.mcd_wpmi_score_bf16_workspace(U, K))
    assert need >= 8 * U * K
    ws = torch.empty(need // 8 + 1, dtype=torch.int64, device=dev)
    args = (E.data_ptr(), E.stride(0), N, C, rinv.data_ptr(), idx.data_ptr(), K, U, K, None, 1e-7, 0, out.data_ptr(), C)
    assert L.mcd_wpmi_score_bf16(*args, ws.data_ptr(), need - 8, None) != 0
    assert b"workspace" in L.mcd_last_error()
    assert L.mcd_wpmi_score_bf16(*args, None, need, None) != 0
    assert L.mcd_wpmi_score_bf16(*args, ws.data_ptr(), need, None) == 0
    torch.cuda.synchronize()
    ref = core.wpmi_score_bf16(E[:, :C], rinv, idx, None, 1e-7, False)
    assert torch.equal(out, ref)


def test_wpmi_score_bf16_tiny_min_prob_and_wide_pitch(core, dev):
    """min_prob below 2^-30: four arguments' product could leave the normal range, so the kernel takes one log per row;
    a row pitch of 2^24 bytes or more: 64-bit row offsets instead of the 24-bit multiply."""
    N, C, U, K = 64, 200, 9, 20
    g = torch.Generator().manual_seed(5)
    Eh = (torch.rand(N, C, generator=g) * 1e-3).to(torch.bfloat16)
    rinv = torch.full((N,), 1e-9)
    idx = torch.stack([torch.randperm(N, generator=g)[:K] for _ in range(U)]).int().to(dev)
    S = Eh.double() * rinv.double()[:, None]
    ref = torch.log(S[idx.cpu().long()] + 1e-12).sum(dim=1)
    Epad = torch.zeros(N, 256, dtype=torch.bfloat16)
    Epad[:, :C] = Eh
    out = core.wpmi_score_bf16(Epad.to(dev)[:, :C], rinv.to(dev), idx, None, 1e-12, False)
    assert float((out.double().cpu() - ref).abs().max()) <= 1e-4 * float(ref.abs().max())
    # wide pitch: 8 rows of 2^23 bf16 columns (16 MiB each); only the first C columns are scored
    wide = torch.zeros(8, 1 << 23, dtype=torch.bfloat16, device=dev)
    wide[:, :C] = Eh[:8].to(dev)
    idx8 = torch.stack([torch.randperm(8, generator=g)[:5] for _ in range(U)]).int().to(dev)
    out8 = core.wpmi_score_bf16(wide[:, :C], rinv[:8].to(dev) * 1e6, idx8, None, 1e-7, False)
    S8 = Eh[:8].double() * (rinv[:8].double() * 1e6)[:, None]
    ref8 = torch.log(S8[idx8.cpu().long()] + 1e-7).sum(dim=1)
    assert float((out8.double().cpu() - ref8).abs().max()) <= 1e-4 * float(ref8.abs().max())


def test_encoder_gemm_picks_can_be_read_and_forced(core, dev):
    """libmcd_blaslt.so keeps one hipBLASLt algorithm per GEMM shape, picked by timing; a multi-rank run forces rank 0's
    picks on the other ranks (pipeline.sync_encoder_gemm_picks).  Read the pick of a shape, force another index, and the
    same call must run on it (recorded pick changes, result still the GEMM), then force the first one back: same bits
    as the first run (same algorithm => same summation order)."""
    g = torch.Generator().manual_seed(3)
    M, N, K = 394, 96, 160
    h, W, b, res = (torch.randn(M, K, generator=g).to(dev), torch.randn(N, K, generator=g).to(dev),
                    torch.randn(N, generator=g).to(dev), torch.randn(M, N, generator=g).to(dev))
    ref = res.double() + h.double() @ W.double().t() + b.double()
    out0 = core.linear_residual(res, h, W, b).clone()
    picks = [p for p in core.encoder_gemm_picks() if p[:4] == (M, N, K, 1)]
    assert len(picks) == 1 and picks[0][4] >= 0
    first = picks[0][4]
    other = 0 if first != 0 else 1
    core.set_encoder_gemm_picks([(M, N, K, 1, other)])
    out1 = core.linear_residual(res, h, W, b).clone()
    now = [p for p in core.encoder_gemm_picks() if p[:4] == (M, N, K, 1)][0][4]
    assert now == other
    assert float((out1.double() - ref).abs().max()) <= 1e-4 * float(ref.abs().max())
    core.set_encoder_gemm_picks([(M, N, K, 1, first)])
    out2 = core.linear_residual(res, h, W, b)
    assert torch.equal(out2, out0)
    core.set_encoder_gemm_picks([(M, N, K, 1, -1)])          # un-force: back to the timed choice on the next plan
