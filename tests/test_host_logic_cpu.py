"""CPU: host-side logic that needs no GPU -- the reference's naming contracts, the CSV formatting path,
the hook bookkeeping, and the image-sharded multi-rank path under gloo (world_size 2) with the
oracle-backed test backend (tests/cpu_ops.py)."""
import io
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

import util

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_get_save_names_match_reference(mcd):
    """reference concept_vit/utils.py:54-62 (same in og_utils.py:58-66, CLIP_og_utils.py:38-46)."""
    from mammo_clip_dissect_amd.concept_vit import utils, og_utils, CLIP_og_utils
    for m in (utils, og_utils, CLIP_og_utils):
        t, c, x = m.get_save_names("ViT-B/16", "breastclip", "image_encoder._blocks[3]", "vindr",
                                   "/s/Concepts/Specific_concepts_sorted.txt", "avg", "saved_activations")
        assert t == "saved_activations/vindr_breastclip_image_encoder._blocks[3].pt"
        assert c == "saved_activations/vindr_ViT-B16.pt"
        assert x == "saved_activations/Specific_concepts_sorted_ViT-B16.pt"
        t, _, _ = m.get_save_names("RN50", "resnet50", "{}", "broden", "data/20k.txt", "max", "d")
        assert t == "d/broden_resnet50_{}_max.pt"
    # writer prefixes, reference utils.py:456-468 and og_utils.py:394-406
    assert utils.save_prefix("vindr", None, None) == "/Latest_vindr_not_mammo_pretrained_"
    assert utils.save_prefix("vindr", "a.tar", None) == "/latest_vindr_mammo_pretrained_"
    assert utils.save_prefix("vindr", "a.tar", "b.pth") == "/newest_vindr_cancer_finetuned_"
    assert og_utils.save_prefix("breastclip", "imagenet_subsets") == \
        "/clip_dissector_breastclip_target_imagenet_subsets_small_not_mammo_pretrained_"


def test_driver_flags_match_reference(mcd):
    """argparse flag names and defaults of the three drivers (reference describe_*_neurons.py)."""
    from mammo_clip_dissect_amd.concept_vit import describe_clip_neurons as c, describe_og_neurons as o, \
        describe_broad_neurons as b
    d = vars(c.parser.parse_args([]))
    assert d == {"clip_model": "ViT-B/16", "target_model": "resnet50",
                 "target_layers": "conv1,layer1,layer2,layer3,layer4", "d_probe": "broden",
                 "concept_set": "data/20k.txt", "batch_size": 200, "device": "cuda",
                 "activation_dir": "saved_activations", "result_dir": "", "pool_mode": "avg",
                 "similarity_fn": "soft_wpmi"}
    for m in (o, b):
        d = vars(m.parser.parse_args([]))
        for k in ("clip_model", "num_class", "target_model", "target_layers", "d_probe", "concept_set", "batch_size",
                  "device", "activation_dir", "result_dir", "pool_mode", "similarity_fn", "Breast_clip_chkpt",
                  "finetuned_img_classifier_chkpt", "arch"):
            assert k in d, k
    assert vars(b.parser.parse_args([]))["top_k"] == 100


def test_resolve_layer(mcd):
    from mammo_clip_dissect_amd.concept_vit import utils, data_utils
    m, _ = data_utils.get_target_model("resnet50", "cpu")
    assert utils.resolve_layer(m, "layer3") is m.layer3
    assert utils.resolve_layer(m, "layer3[2].conv1") is m.layer3[2].conv1


def test_model_factory_shapes(mcd):
    """Hook points and widths the reference's launch scripts name (run_clipdissect.sh, run_og_clip.sh)."""
    from mammo_clip_dissect_amd.concept_vit import data_utils
    eff, _ = data_utils.get_target_model("breastclip", "cpu")
    w = [b._project_conv.out_channels for b in eff.image_encoder._blocks]
    assert len(w) == 39 and sum(w) == 6992 and w[:3] == [24] * 3 and w[-3:] == [512] * 3
    vit, _ = data_utils.get_target_model("breastclip_vit", "cpu")
    assert len(vit.image_encoder.encoder.layer) == 12
    clip, _ = data_utils.get_target_model("clip", "cpu")
    assert len(clip.vision_model.encoder.layers) == 12
    with pytest.raises(ValueError):
        data_utils.get_target_model("breastclip_classifier", "cpu")          # needs n_class, like the reference
    tok = eff.tokenize(["mass", "BI-RADS density A"])
    assert set(tok) >= {"input_ids", "attention_mask"} and tok["input_ids"].shape[0] == 2


def _golden_result(name_a="main", name_b="relu"):
    """A DissectResult assembled from the reference's own similarities (two 'layers')."""
    from mammo_clip_dissect_amd.pipeline import DissectResult
    import cpu_ops
    sims, A = [], []
    for n in (name_a, name_b):
        z = util.golden(n)
        sims.append(torch.from_numpy(z["soft_wpmi"]))
        A.append(z["A"])
    vals, ids, tops = [], [], []
    for s, a in zip(sims, A):
        v, i = cpu_ops.row_topk(s, 10)
        _, t = cpu_ops.col_topk(torch.from_numpy(a), 5)
        vals.append(v); ids.append(i); tops.append(t)
    return DissectResult(["layer_a", "layer_b"], [s.shape[0] for s in sims], torch.cat(sims), torch.cat(vals),
                         torch.cat(ids), torch.cat(tops), None, 256)


def test_csv_bytes_match_reference_golden(mcd):
    """DataFrame -> CSV is byte-identical to the CSV the reference's driver code wrote for the same
    similarities (tests/golden/descriptions_{og,clip}.csv; make_golden.py: csv_og / csv_clip)."""
    from mammo_clip_dissect_amd.pipeline import results_to_dataframe
    with open(os.path.join(ROOT, "mammo-clip-dissect_amd", "Concepts", "Specific_concepts_sorted.txt")) as f:
        words = f.read().split("\n")
    assert len(words) == 763
    res = _golden_result()
    for variant in ("og", "clip"):
        want = open(os.path.join(util.GOLDEN, "descriptions_%s.csv" % variant), newline="").read()
        for fast in (True, False):
            buf = io.StringIO()
            results_to_dataframe(res, words, variant, fast_format=fast).to_csv(buf, index=False)
            assert buf.getvalue() == want, (variant, fast)
        from mammo_clip_dissect_amd.pipeline import write_descriptions_csv
        buf = io.StringIO()
        write_descriptions_csv(res, words, buf, variant)      # the direct writer: same bytes
        assert buf.getvalue() == want, variant
    # words that force quoting / escaping through both writers
    odd = list(words)
    odd[3], odd[7], odd[11] = 'a "quoted" word', "it's, with a comma", "line\nbreak"
    res.ids[:, 0] = torch.tensor([3, 7, 11] * (res.ids.shape[0] // 3 + 1))[:res.ids.shape[0]].to(res.ids.dtype)
    a, b = io.StringIO(), io.StringIO()
    results_to_dataframe(res, odd, "og").to_csv(a, index=False)
    write_descriptions_csv(res, odd, b, "og")
    assert a.getvalue() == b.getvalue()


def test_fast_cell_formatting_equals_numpy(mcd):
    """format_f32_rows / format_i64_rows must give str(ndarray) character for character: random rows over the
    regimes numpy treats differently (signs, magnitudes, exponent notation, wrapping, zeros, non-finite)."""
    from mammo_clip_dissect_amd.pipeline import format_f32_rows, format_i64_rows
    rng = np.random.default_rng(0)
    blocks = []
    for scale in (1.0, 3.0, 0.3, 0.05, 30.0, 1e3, 1e5, 1e-3, 1e7, 5e7, 2e8):
        blocks.append((rng.standard_normal((1500, 10)) * scale).astype(np.float32))
        blocks.append(np.sort(rng.standard_normal((500, 10)) * scale, axis=1)[:, ::-1].astype(np.float32))
    blocks.append(rng.uniform(0.5, 2.5, (3000, 10)).astype(np.float32))
    blocks.append((rng.uniform(0.5, 2.5, (500, 10)) * 10.0 ** rng.integers(-6, 9, (500, 1))).astype(np.float32))
    a = np.concatenate(blocks)
    a[5, 3] = 0.0; a[6, 0] = np.nan; a[7, 9] = np.inf; a[8, 2] = -0.0; a[9] = 1.5; a[10] = 2.0; a[11, :] = [1e-4] * 10
    want = [str(row) for row in a]
    for native in (True, False):            # csrc/libmcd_host.so (exact-double Dragon4 equivalent) and the Python path
        got = format_f32_rows(a, native=native)
        for g, w_ in zip(got, want):
            assert g == w_, (native, g, w_)
    from mammo_clip_dissect_amd import pipeline
    assert pipeline._load_host_lib(), "libmcd_host.so must be built (make -C mammo-clip-dissect_amd/csrc)"
    for k in (1, 2, 7, 13, 40, 64):
        b = (rng.standard_normal((800, k)) * 2).astype(np.float32)
        assert format_f32_rows(b) == [str(r) for r in b]
    # the cases Dragon4 decides on margins: powers of two (unequal margins), exact decimals, values whose 8-digit cut
    # rounds up with a carry, half-way 8th digits, the edges of the native regime
    special = np.array([[1.0, 2.0, 0.5, 0.25, 4.0, 8.0, 1024.0, 0.125, 0.0625, 16.0],
                        [0.1, 0.2, 0.3, 0.7, 1.1, 2.5, 3.75, 9.999999, 0.99999994, 1.0000001],
                        [1.9999999, 2.9999998, 0.49999997, 0.019999999, 99.99999, 0.0009765625, 0.001, 0.5000001, 7.0, 3.0],
                        [16777215.0, 16777214.0, 8388608.5, 123456.79, 65536.0, 32768.5, 4194304.5, 2097152.2, 1e6, 1e7],
                        [1.001e-4, 1.5e-4, 0.00012207031, 0.0999, 0.00999, 0.01, 0.0625, 0.000244140625, 3e-4, 5e-4]], np.float32)
    special = np.concatenate([special, -special, special[:, ::-1] * np.float32(-1.0)])
    assert format_f32_rows(special) == [str(r) for r in special]
    # any mantissa, every binade of the native regime: per row a base exponent, elements within 2^8 of it
    e_row = rng.integers(127 - 13, 127 + 15, (40000, 1))
    bits = (((e_row + rng.integers(0, 9, (40000, 10))) << 23) | rng.integers(0, 1 << 23, (40000, 10))).astype(np.uint32)
    vals = bits.view(np.float32) * np.where(rng.random((40000, 10)) < 0.5, -1, 1).astype(np.float32)
    assert format_f32_rows(vals) == [str(r) for r in vals]
    import ctypes
    L = pipeline._load_host_lib()
    lens = np.empty((40000,), np.int32)
    buf = np.empty((40000, 400), np.uint8)
    vc = np.ascontiguousarray(vals)
    L.mcd_fmt_f32_rows(vc.ctypes.data, 40000, 10, buf.ctypes.data, 400, lens.ctypes.data)
    assert (lens >= 0).mean() > 0.9          # the native path, not the fallback, produced these
    i = rng.integers(0, 10 ** rng.integers(1, 9, (4000, 1)), (4000, 5)).astype(np.int64)
    i[3] = [0, 0, 0, 0, 0]; i[4, 1] = -17
    for native in (True, False):
        assert format_i64_rows(i, native=native) == [str(r) for r in i]
    w = rng.integers(0, 10 ** 12, (50, 9)).astype(np.int64)   # wide rows: handed to numpy
    assert format_i64_rows(w) == [str(r) for r in w]
    neg = rng.integers(-10 ** 6, 10 ** 6, (2000, 5)).astype(np.int64)
    assert format_i64_rows(neg) == [str(r) for r in neg]


def test_hooks_fill_the_activation_matrix(mcd):
    """get_activation semantics (reference utils.py:27-52) through the Dissector hooks, CPU test backend."""
    import cpu_ops
    from mammo_clip_dissect_amd.pipeline import Dissector
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Conv2d(3, 6, 3), torch.nn.ReLU(), torch.nn.Conv2d(6, 4, 3))
    N, B = 10, 4
    dis = Dissector(N, ["0", "2"], [6, 4], 5, 8, "cpu", top_k=3, ops=cpu_ops)
    hs = [net[0].register_forward_hook(dis.hook(0)), net[2].register_forward_hook(dis.hook(1))]
    x = torch.randn(N, 3, 9, 9)
    ref0, ref1 = [], []
    for i in range(0, N, B):
        xb = x[i:i + B]
        with torch.no_grad():
            y0 = net[0](xb); y1 = net(xb)
        ref0.append(y0.mean(dim=[2, 3])); ref1.append(y1.mean(dim=[2, 3]))
        dis.advance(xb.shape[0])
    for h in hs:
        h.remove()
    At = dis.At[:, :N]
    assert torch.allclose(At[:6].t(), torch.cat(ref0), atol=1e-6) and torch.allclose(At[6:].t(), torch.cat(ref1), atol=1e-6)
    with pytest.raises(RuntimeError):
        dis.advance(1)


# ---- world_size 2 under gloo ------------------------------------------------------------------------
def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _make_problem(N, widths, C, D, seed):
    g = torch.Generator().manual_seed(seed)
    U = sum(widths)
    At = torch.randn(U, N, generator=g)
    E_img = torch.randn(N, D, generator=g)
    E_txt = torch.randn(C, D, generator=g)
    return At, E_img, E_txt


def _run_dissect(world, rank, N, widths, C, D, K, seed, group=None):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import mammo_clip_dissect_amd  # noqa
    import cpu_ops
    from mammo_clip_dissect_amd.pipeline import Dissector
    from mammo_clip_dissect_amd.pipeline import shard_bounds
    At, E_img, E_txt = _make_problem(N, widths, C, D, seed)
    lo, hi = shard_bounds(N, world, rank)        # uneven when world does not divide N
    n_l = hi - lo
    dis = Dissector(n_l, ["l%d" % i for i in range(len(widths))], widths, C, D, "cpu", top_k=K, ops=cpu_ops,
                    group=group)
    assert dis.n_total == N and dis.row0 == lo
    dis.At[:, :n_l] = At[:, lo:hi]
    dis.E_img[:] = E_img[lo:hi]
    dis.cursor = n_l
    r = dis.finish(E_txt, k_desc=10, k_img=min(5, N))
    return r.sim, r.vals, r.ids, r.top_ids, r.top_vals


def _worker(rank, world, port, args, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    out = _run_dissect(world, rank, *args)
    q.put((rank, [o.numpy() for o in out]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,case", [(2, (240, [7, 12], 37, 16, 50, 5)), (2, (200, [5], 763, 32, 100, 9)),
                                        (2, (241, [7, 12], 37, 16, 50, 6)),      # 121 + 120 images
                                        (3, (130, [9, 4], 37, 16, 50, 7)),       # 44 + 43 + 43: every shard < top_k
                                        (3, (2, [3], 11, 8, 2, 8)),              # N < ranks: rank 2 holds no image
                                        (8, (810, [13, 8], 37, 16, 100, 11))])   # the scaling run's 8 ranks: 21 neurons over 8 ranks (rank 7: none)
def test_ranks_bit_identical_to_one(mcd, world, case):
    """S shards all-gathered, local top-K merged to the global top-K (ties -> lower global index),
    neurons split for scoring, prob_d_given_e all-gathered.  No float is reduced across ranks, so the G-rank
    result must equal the 1-rank result bit for bit -- for any N, divisible by G or not (the reference walks any
    N, utils.py:174-181)."""
    single = [o.numpy() for o in _run_dissect(1, 0, *case)]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, case, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=300) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for r in range(world):
        for a, b in zip(single, got[r]):
            assert np.array_equal(a, b)


def test_shard_bounds_cover_every_image_once(mcd):
    from mammo_clip_dissect_amd.pipeline import shard_bounds
    for n, g in ((10000, 8), (10001, 8), (7, 8), (0, 3), (256, 1), (50000, 6)):
        b = [shard_bounds(n, g, r) for r in range(g)]
        assert b[0][0] == 0 and b[-1][1] == n and all(b[i][1] == b[i + 1][0] for i in range(g - 1))
        sizes = [hi - lo for lo, hi in b]
        assert max(sizes) - min(sizes) <= 1 and sizes == sorted(sizes, reverse=True)


def test_shard_bounds_aligned_to_the_encoder_batch(mcd, monkeypatch):
    """MCD_SHARD_ALIGN = the encoder batch size: every boundary is a batch multiple (each batch of the global image order is
    encoded whole by one rank: what encoder-inclusive bit-identity across rank counts needs, pipeline.shard_align), the shards
    still cover every image once, in order, and differ by at most one batch."""
    from mammo_clip_dissect_amd.pipeline import shard_bounds
    for n, g, a in ((10000, 8, 125), (10001, 3, 50), (1001, 3, 167), (7, 8, 4), (0, 3, 16), (50000, 6, 125), (200, 3, 50)):
        b = [shard_bounds(n, g, r, align=a) for r in range(g)]
        assert b[0][0] == 0 and b[-1][1] == n and all(b[i][1] == b[i + 1][0] for i in range(g - 1))
        assert all(lo % a == 0 or lo == n for lo, _ in b)
        sizes = [hi - lo for lo, hi in b]
        assert max(sizes) - min(sizes) <= a
    monkeypatch.setenv("MCD_SHARD_ALIGN", "50")
    assert [shard_bounds(201, 3, r) for r in range(3)] == [(0, 100), (100, 200), (200, 201)]
    monkeypatch.delenv("MCD_SHARD_ALIGN")
    assert [shard_bounds(201, 3, r) for r in range(3)] == [(0, 67), (67, 134), (134, 201)]


def test_two_ranks_with_cross_shard_ties(mcd):
    """Equal activations on both shards: the merge must keep the lower GLOBAL image index first."""
    import cpu_ops
    v = torch.tensor([[1.0, 1.0, 0.5, 1.0]])                      # one neuron, two ranks x two candidates
    vals, pos = cpu_ops.col_topk(v, 3, neuron_major=True)
    assert pos[0].tolist() == [0, 1, 3]


def test_native_row_formatting_hypothesis(mcd):
    """Property test (hypothesis) of the native float32 / int64 row formatters against str(ndarray): arbitrary finite
    and non-finite float32 bit patterns, any row length up to 64 -- rows outside the native regime must fall back to
    numpy's own text, rows inside it must reproduce it character for character."""
    hyp = pytest.importorskip("hypothesis")
    from hypothesis import given, settings, strategies as st
    from hypothesis.extra import numpy as hnp
    from mammo_clip_dissect_amd.pipeline import format_f32_rows, format_i64_rows

    rows_f32 = hnp.arrays(np.float32, st.tuples(st.integers(1, 6), st.integers(1, 24)),
                          elements=st.floats(width=32, allow_nan=True, allow_infinity=True))
    near = st.floats(min_value=float(np.float32(2.0 ** -13)), max_value=float(np.float32(2.0 ** 23)), width=32)
    rows_near = hnp.arrays(np.float32, st.tuples(st.integers(1, 6), st.integers(1, 24)), elements=near)

    @settings(max_examples=300, deadline=None)
    @given(rows_f32)
    def any_bits(a):
        assert format_f32_rows(a) == [str(r) for r in a]

    @settings(max_examples=400, deadline=None)
    @given(rows_near, st.floats(min_value=0.5, max_value=2.0), st.booleans())
    def native_regime(a, scale, neg):
        # squeeze every row into one order of magnitude so that the native path (max/min <= 999) is the one under test
        a = (np.float32(scale) * (np.float32(1.0) + (a % np.float32(7.0)))).astype(np.float32)
        if neg:
            a = -a
        assert format_f32_rows(a) == [str(r) for r in a]

    @settings(max_examples=200, deadline=None)
    @given(hnp.arrays(np.int64, st.tuples(st.integers(1, 6), st.integers(1, 12)),
                      elements=st.integers(min_value=-2 ** 62, max_value=2 ** 62)))
    def ints(a):
        assert format_i64_rows(a) == [str(r) for r in a]

    any_bits()
    native_regime()
    ints()


def test_packaged_gemm_picks_file(mcd):
    """tuning.py: the packaged TunableOp table names this image's stack (so PyTorch's validator accepts it on the GPU
    box), lists only fp32 GEMM entries, and enabling it without a GPU is a harmless no-op."""
    from mammo_clip_dissect_amd import tuning
    rows = [l.strip().split(",") for l in open(tuning.RESULTS) if l.strip()]
    val = {r[1]: r[2] for r in rows if r[0] == "Validator"}
    assert val["PT_VERSION"] == ".".join(torch.__version__.split("+")[0].split(".")[:3])
    assert val["GCN_ARCH_NAME"].startswith("gfx950")
    ops = [r for r in rows if r[0] != "Validator"]
    assert len(ops) >= 8 and all(r[0].startswith("Gemm") and "_float_" in r[0] for r in ops)
    assert tuning.enable_gemm_tuning() in (True, False)      # never raises; True only where the stack matches


def test_vit_tower_host_structure(mcd):
    """The tower keeps nn.Module semantics around the encoder-side kernels: same state_dict keys as a plain
    Conv2d / LayerNorm / Linear tower, and on CPU tensors (no kernel can run) it IS that plain tower -- embed() equals
    conv + cat + add, a block equals its textbook form, attention equals heads_out followed by proj."""
    import torch
    import torch.nn.functional as F
    from mammo_clip_dissect_amd.concept_vit import data_utils
    torch.manual_seed(0)
    t = data_utils.ViTTower(image_size=32, depth=2, dim=64, heads=1, mlp=128).eval()
    keys = set(t.state_dict())
    for k in ("patch_embed.weight", "patch_embed.bias", "cls_token", "pos_embed", "layernorm.weight", "layernorm.bias",
              "encoder.layer.0.norm1.weight", "encoder.layer.0.attn.qkv.weight", "encoder.layer.0.attn.proj.bias",
              "encoder.layer.1.fc1.weight", "encoder.layer.1.fc2.bias", "encoder.layer.1.norm2.bias"):
        assert k in keys, k
    assert isinstance(t.layernorm, torch.nn.LayerNorm)
    x = torch.randn(2, 3, 32, 32)
    with torch.no_grad():
        e = t.embed(x)
        ref = t.patch_embed(x).flatten(2).transpose(1, 2)
        ref = torch.cat([t.cls_token.expand(2, -1, -1), ref], dim=1) + t.pos_embed
        assert torch.equal(e, ref)
        blk = t.encoder.layer[0]
        a = blk.attn
        assert torch.equal(a(e), a.proj(a.heads_out(e)))
        x1 = e + a(blk.norm1(e))
        want = x1 + blk.fc2(F.gelu(blk.fc1(blk.norm2(x1))))
        assert torch.equal(blk(e), want)
        assert torch.equal(t(x), t.layernorm(t.encoder(e)))


def _fuzz_worker(rank, world, port, n_cases, seed, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    groups = {g: dist.new_group(list(range(g))) for g in range(1, world + 1)}   # every rank creates every group
    rng = np.random.default_rng(seed)                                           # same stream on every rank
    bad = []
    for c in range(n_cases):
        g = int(rng.integers(2, world + 1))
        K = int(rng.choice([1, 2, 5, 17, 50, 100]))
        N = int(rng.integers(K, K + 300)) if rng.random() < 0.8 else K            # N == K: every image is selected
        widths = [int(w) for w in rng.integers(1, 20, size=int(rng.integers(1, 4)))]
        C = int(rng.choice([3, 31, 37, 64, 100]))
        case = (N, widths, C, 16, K, int(rng.integers(0, 1 << 30)))
        out = _run_dissect(g, rank, *case, group=groups[g]) if rank < g else None
        if rank == 0:
            single = _run_dissect(1, 0, *case, group=groups[1])
            if not all(torch.equal(a, b) for a, b in zip(single, out)):
                bad.append((g,) + case)
    if rank == 0:
        q.put(bad)
    dist.barrier()
    dist.destroy_process_group()


def test_ranks_fuzz(mcd):
    """Random probe-set sizes, rank counts (2..4, as sub-groups of one 4-process gloo world), layer widths, concept counts and
    top_k -- shards smaller than top_k, ranks without images or without neurons included: the sharded result equals the
    one-rank result bit for bit in every case."""
    world = 4
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_fuzz_worker, args=(r, world, port, 30, 5, q)) for r in range(world)]
    for p in procs:
        p.start()
    bad = q.get(timeout=600)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert bad == []


def test_bench_refuses_to_run_fewer_ranks_than_asked():
    """`bench.py --gpus N` must never measure one rank under the name of N.  With the self-launch turned off,
    or with a WORLD_SIZE that disagrees with --gpus, it exits non-zero before any GPU call and prints no JSON line."""
    import subprocess
    bench = os.path.join(ROOT, "bench.py")
    base = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, bench, "--gpus", "2"], env=dict(base, MCD_BENCH_NO_LAUNCH="1"), capture_output=True,
                       text=True, timeout=300)
    assert r.returncode != 0 and "{" not in r.stdout and "refusing" in r.stderr
    r = subprocess.run([sys.executable, bench, "--gpus", "4"], env=dict(base, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0"),
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "{" not in r.stdout and "must agree" in r.stderr
    r = subprocess.run([sys.executable, bench, "--gpus", "0"], env=base, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0


def test_bench_cfg2_preset_pairs_the_one_gpu_and_the_eight_gpu_job(monkeypatch):
    """`bench.py --config cfg2` is configs[2] as ONE job at every rank count -- 10 000 images in all, encoder batches
    of 1 250, shard boundaries on batch multiples -- so that at 1, 2, 4 and 8 ranks the SAME eight batches of the global image order
    are encoded, each whole and by exactly one rank (what encoder-inclusive bit-identity of the CSV needs: every fp32 hipBLASLt
    solution is stream-K, its summation order depends on the batch's row count)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    from mammo_clip_dissect_amd.pipeline import shard_bounds
    for g in (1, 2, 4, 8):
        monkeypatch.setattr(sys, "argv", ["bench.py", "--config", "cfg2", "--gpus", str(g)])
        a = bench.parse()
        assert (a.config, a.preset, a.global_images, a.batch, a.align_shards) == ("headline", "cfg2", 10000, 1250, True)
        batches = []
        for r in range(g):
            lo, hi = shard_bounds(a.global_images, g, r, align=a.batch)
            assert lo % a.batch == 0 and (hi % a.batch == 0 or hi == a.global_images) and hi > lo
            batches += [(b, min(b + a.batch, hi)) for b in range(lo, hi, a.batch)]
        assert batches == [(b, b + 1250) for b in range(0, 10000, 1250)]          # the same eight batches at every rank count
    monkeypatch.setattr(sys, "argv", ["bench.py"])
    a = bench.parse()
    assert a.preset is None and a.global_images is None and not a.align_shards     # the headline default is untouched
