"""GPU: the fused all-layer pipeline, the utils mirrors and the three drivers, end to end through the C ABI,
against the oracle run on the very same activations/embeddings (copied to the host)."""
import ast
import glob
import os
import re

import numpy as np
import pandas as pd
import pytest
import torch

import util

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CONCEPTS = os.path.join(ROOT, "mammo-clip-dissect_amd", "Concepts", "Specific_concepts_sorted.txt")


def _problem(dev, N, widths, C, D, seed):
    g = torch.Generator().manual_seed(seed)
    At = torch.randn(sum(widths), N, generator=g)
    E_img = torch.randn(N, D, generator=g)
    E_txt = torch.randn(C, D, generator=g)
    return At, E_img, E_txt


def _fused(dev, At, E_img, E_txt, widths, K):
    from mammo_clip_dissect_amd.pipeline import Dissector
    N = At.shape[1]
    dis = Dissector(N, ["l%d" % i for i in range(len(widths))], widths, E_txt.shape[0], E_txt.shape[1], dev, top_k=K)
    dis.At[:, :N] = At.to(dev)
    dis.E_img[:] = E_img.to(dev)
    dis.cursor = N
    return dis, dis.finish(E_txt.to(dev))


def test_fused_equals_per_layer_api(mcd, dev):
    """One launch per kernel over all layers == the per-layer drop-in calls, bit for bit."""
    from mammo_clip_dissect_amd import core
    from mammo_clip_dissect_amd.concept_vit import similarity
    widths, N, C, D, K = [64, 33, 7, 130], 600, 763, 512, 100
    At, E_img, E_txt = _problem(dev, N, widths, C, D, 3)
    dis, res = _fused(dev, At, E_img, E_txt, widths, K)
    P = core.embed_gemm(core.normalize_rows(E_img.to(dev)), core.normalize_rows(E_txt.to(dev)))
    o = 0
    for w in widths:
        A = At[o:o + w].t().contiguous().to(dev)
        sim = similarity.soft_wpmi(P, A, top_k=K, device=str(dev))
        assert torch.equal(sim, res.sim[o:o + w])
        v, i = core.row_topk(sim, 10)
        assert torch.equal(v, res.vals[o:o + w]) and torch.equal(i, res.ids[o:o + w])
        _, t5 = core.col_topk(A, 5)
        assert torch.equal(t5, res.top_ids[o:o + w])
        o += w


def test_finish_as_one_hipgraph(mcd, dev):
    """The scoring side captured in a hipGraph and replayed (new text embeddings, new activations in place):
    the same bits as the eager launches."""
    widths, N, C, D, K = [64, 33, 130], 700, 763, 512, 100
    At, E_img, E_txt = _problem(dev, N, widths, C, D, 8)
    dis, res = _fused(dev, At, E_img, E_txt, widths, K)
    eager = [t.clone() for t in (res.sim, res.vals, res.ids, res.top_ids)]
    g1 = dis.finish_graphed(E_txt.to(dev))
    for a, b in zip(eager, (g1.sim, g1.vals, g1.ids, g1.top_ids)):
        assert torch.equal(a, b)
    # replay on other inputs: the graph reads the dissector's own buffers and the captured text-embedding buffer
    At2, E_img2, E_txt2 = _problem(dev, N, widths, C, D, 9)
    dis.At[:, :N] = At2.to(dev)
    dis.E_img[:] = E_img2.to(dev)
    g2 = dis.finish_graphed(E_txt2.to(dev))
    got = [t.clone() for t in (g2.sim, g2.vals, g2.ids, g2.top_ids)]
    ref = dis.finish(E_txt2.to(dev))
    for a, b in zip(got, (ref.sim, ref.vals, ref.ids, ref.top_ids)):
        assert torch.equal(a, b)


def test_fused_against_oracle(mcd, dev, oracle):
    widths, N, C, D, K = [48, 80], 1000, 763, 512, 100
    At, E_img, E_txt = _problem(dev, N, widths, C, D, 4)
    dis, res = _fused(dev, At, E_img, E_txt, widths, K)
    o = 0
    for w in widths:
        ref = oracle.dissect_layer(E_img.numpy(), E_txt.numpy(), At[o:o + w].t().contiguous().numpy(), top_k=K)
        util.assert_sim_close(res.sim[o:o + w].cpu().numpy(), ref["sim"], "fused layer")
        assert np.array_equal(res.top_ids[o:o + w].cpu().numpy().T, ref["top_ids"])          # integer: exact
        util.assert_topk_ids(res.ids[o:o + w].cpu().numpy(), None, ref["ids"], ref["sim"], 10)
        o += w


def test_get_activation_hook_matches_reference_semantics(mcd, dev):
    """reference utils.py:27-52: 4-D -> mean/amax over H,W; 3-D -> token 0; 2-D -> as is; tuple unwrapped."""
    from mammo_clip_dissect_amd.concept_vit import utils
    x4 = torch.randn(6, 10, 5, 7, device=dev)
    x3 = torch.randn(6, 9, 12, device=dev)
    x2 = torch.randn(6, 11, device=dev)
    for mode, f4 in (("avg", lambda t: t.mean(dim=[2, 3])), ("max", lambda t: t.amax(dim=[2, 3]))):
        outs = []
        h = utils.get_activation(outs, mode)
        h(None, None, x4); h(None, None, x3); h(None, None, x2)
        if mode == "avg":
            h(None, None, (x3, "aux"))
        assert torch.allclose(outs[0], f4(x4), atol=1e-6)
        assert torch.equal(outs[1], x3[:, 0]) and torch.equal(outs[2], x2)
        if mode == "avg":
            assert torch.equal(outs[3], x3[:, 0])


def _parse_list(s):
    return [float(v) for v in re.sub(r"[\[\]\n]", " ", s).split()]


def _check_csv_against_oracle(csv_path, act_dir_glob, layers, oracle, variant, top_k, words):
    """Every cell of the driver's CSV against the oracle run on the driver's own cache files: images exact (integer);
    ALL similarity values within the north_star tolerance (1e-4); ALL description ranks identical wherever the
    oracle's ranking is decided (gap to both neighbours > ARGMAX_GAP; the tie rule for the rest: two implementations
    whose scores differ in the last ulp may swap neighbours closer than that, and the reference itself leaves ties
    unspecified)."""
    df = pd.read_csv(csv_path)
    assert list(df.columns) == ["layer", "unit", "description", "similarity", "images"]
    files = glob.glob(act_dir_glob, recursive=True)
    clip_f = [f for f in files if f.endswith("_ViT-B16.pt") and "Specific_concepts" not in f][0]
    text_f = [f for f in files if "Specific_concepts" in f][0]
    E_img = torch.load(clip_f, weights_only=True).numpy()
    E_txt = torch.load(text_f, weights_only=True).numpy()
    n_decided = n_ranks = 0
    k = 10 if variant == "og" else 1
    for layer in layers:
        tf = [f for f in files if f.endswith("_%s.pt" % layer)][0]
        A = torch.load(tf, weights_only=True).numpy()
        assert A.ndim == 2 and A.shape[0] == E_img.shape[0]                    # cache format [N, U_layer]
        ref = oracle.dissect_layer(E_img, E_txt, A, top_k=top_k, k_desc=k, blas=False)   # P in the fixture host's order, as K1
        sub = df[df.layer == layer].reset_index(drop=True)
        assert len(sub) == A.shape[1] and sub.unit.tolist() == list(range(A.shape[1]))
        # images column: integer, exact, numpy's own formatting
        want_imgs = [str(r) for r in ref["top_ids"].T.astype(np.int64)]
        assert sub.images.tolist() == want_imgs
        if variant == "og":
            got_ids = np.array([[words.index(w) for w in ast.literal_eval(d)] for d in sub.description])
            got_sim = np.array([_parse_list(t) for t in sub.similarity], np.float32)
        else:
            got_ids = np.array([[words.index(w)] for w in sub.description])
            got_sim = np.array([[float(t)] for t in sub.similarity], np.float32)
        assert got_ids.shape == (A.shape[1], k) and got_sim.shape == (A.shape[1], k)
        srt = np.sort(ref["sim"], axis=1)[:, ::-1][:, :k]
        assert np.abs(got_sim.astype(np.float64) - srt).max() <= util.SIM_ATOL                 # all ranks, 1e-4
        frac = util.assert_topk_ids(got_ids, None, ref["ids"], ref["sim"], k, "%s %s" % (variant, layer))
        n_decided += frac * got_ids.size
        n_ranks += got_ids.size
    assert n_decided > 0.5 * n_ranks, (n_decided, n_ranks)


def test_describe_broad_neurons_end_to_end(mcd, dev, oracle, tmp_path):
    """M-Mammo-CLIP Dissect on a small synthetic probe set: driver -> cache files -> CSV, all on the GPU path."""
    from mammo_clip_dissect_amd.concept_vit import describe_broad_neurons as drv
    layers = ["image_encoder.encoder.layer[0]", "image_encoder.encoder.layer[6]", "image_encoder.encoder.layer[11]"]
    act, res = str(tmp_path / "acts"), str(tmp_path / "results")
    out = drv.main(["--target_model", "breastclip_vit", "--target_layers", ",".join(layers), "--d_probe",
                    "synthetic_300_224", "--concept_set", CONCEPTS, "--batch_size", "100", "--device", str(dev),
                    "--activation_dir", act, "--result_dir", res, "--top_k", "100"])
    csvs = glob.glob(os.path.join(out, "*.csv"))
    assert len(csvs) == 1 and os.path.basename(csvs[0]) == "synthetic_300_224_not_mammo_pretrained_breast_clip_descriptions.csv"
    assert len(glob.glob(os.path.join(out, "*_args.txt"))) == 1
    words = open(CONCEPTS).read().split("\n")
    _check_csv_against_oracle(csvs[0], act + "/**/*.pt", layers, oracle, "og", 100, words)
    fused_bytes = open(csvs[0], "rb").read()
    # second run reuses the activation cache (reference utils.py:128,162,318: skip when the files exist) and therefore
    # takes the per-layer route (the reference's loop over the cache files): same bytes as the fused route
    mt = {f: os.path.getmtime(f) for f in glob.glob(act + "/**/*.pt", recursive=True)}
    assert len(mt) == len(layers) + 2
    out2 = drv.main(["--target_model", "breastclip_vit", "--target_layers", ",".join(layers), "--d_probe",
                     "synthetic_300_224", "--concept_set", CONCEPTS, "--batch_size", "100", "--device", str(dev),
                     "--activation_dir", act, "--result_dir", res + "2", "--top_k", "100"])
    assert mt == {f: os.path.getmtime(f) for f in mt}
    assert open(glob.glob(os.path.join(out2, "*.csv"))[0], "rb").read() == fused_bytes


def test_host_loader_route_equals_device_probe_shapes(mcd, dev, oracle, tmp_path, monkeypatch):
    """MCD_PROBE_ON_HOST=1: the probe set comes from the host dataset through a DataLoader (the reference's way,
    utils.py:489-490) instead of the device-resident generator; same driver, same checks."""
    from mammo_clip_dissect_amd.concept_vit import describe_clip_neurons as drv
    monkeypatch.setenv("MCD_PROBE_ON_HOST", "1")
    layers = ["layer1", "layer4"]
    act, res = str(tmp_path / "acts"), str(tmp_path / "results")
    out = drv.main(["--target_model", "resnet50", "--target_layers", ",".join(layers), "--d_probe", "synthetic_128_224",
                    "--concept_set", CONCEPTS, "--batch_size", "64", "--device", str(dev), "--activation_dir", act,
                    "--result_dir", res])
    words = open(CONCEPTS).read().split("\n")
    _check_csv_against_oracle(os.path.join(out, "descriptions.csv"), act + "/*.pt", layers, oracle, "clip", 100, words)


def test_describe_clip_neurons_resnet50(mcd, dev, oracle, tmp_path):
    """BASELINE configs[0] plumbing: describe_clip_neurons.py, 256 random 224x224 images, ResNet-50 target,
    conv1 + layer1..4 (64/256/512/1024/2048 channels, avg pooled), 763 concepts, top-1 descriptions."""
    from mammo_clip_dissect_amd.concept_vit import describe_clip_neurons as drv
    layers = ["conv1", "layer1", "layer2", "layer3", "layer4"]
    act, res = str(tmp_path / "acts"), str(tmp_path / "results")
    out = drv.main(["--target_model", "resnet50", "--target_layers", ",".join(layers), "--d_probe", "synthetic_256_224",
                    "--concept_set", CONCEPTS, "--batch_size", "64", "--device", str(dev), "--activation_dir", act,
                    "--result_dir", res])
    df = pd.read_csv(os.path.join(out, "descriptions.csv"))
    assert [int((df.layer == l).sum()) for l in layers] == [64, 256, 512, 1024, 2048]
    words = open(CONCEPTS).read().split("\n")
    _check_csv_against_oracle(os.path.join(out, "descriptions.csv"), act + "/*.pt", layers, oracle, "clip", 100, words)


def test_describe_broad_neurons_efficientnet_b5_all_blocks(mcd, dev, oracle, tmp_path):
    """The reference's own launch line (run_clipdissect.sh:6-9): Mammo-CLIP EfficientNet-B5 target + dissector, all 39
    `image_encoder._blocks[i]` hooks (4-D outputs, avg pooled: 6992 neurons), on a small synthetic probe set."""
    from mammo_clip_dissect_amd.concept_vit import describe_broad_neurons as drv
    layers = ["image_encoder._blocks[%d]" % i for i in range(39)]
    act, res = str(tmp_path / "acts"), str(tmp_path / "results")
    out = drv.main(["--target_model", "breastclip", "--target_layers", ", ".join(layers), "--d_probe", "synthetic_160_224",
                    "--concept_set", CONCEPTS, "--batch_size", "40", "--device", str(dev), "--activation_dir", act,
                    "--result_dir", res, "--top_k", "100"])
    csvs = glob.glob(os.path.join(out, "*.csv"))
    df = pd.read_csv(csvs[0])
    widths = [24] * 3 + [40] * 5 + [64] * 5 + [128] * 7 + [176] * 7 + [304] * 9 + [512] * 3
    assert [int((df.layer == l).sum()) for l in layers] == widths and len(df) == 6992
    words = open(CONCEPTS).read().split("\n")
    _check_csv_against_oracle(csvs[0], act + "/**/*.pt", [layers[0], layers[20], layers[38]], oracle, "og", 100, words)


def test_packaged_gemm_picks_are_accepted(mcd, dev):
    """On the GPU box PyTorch's TunableOp validator must accept tunableop_gfx950.csv (same image as the one it was
    tuned on); a linear layer of the ViT's shape then still computes the fp32 result."""
    from mammo_clip_dissect_amd import tuning
    assert tuning.enable_gemm_tuning() is True
    g = torch.Generator(device=dev).manual_seed(0)
    x = torch.randn(1000, 768, device=dev, generator=g)
    w = torch.randn(2304, 768, device=dev, generator=g)
    b = torch.randn(2304, device=dev, generator=g)
    y = torch.nn.functional.linear(x, w, b)
    ref = (x.double() @ w.double().t() + b.double()).float()
    assert float((y - ref).abs().max()) <= 2e-3 * float(ref.abs().max())
    torch.cuda.tunable.enable(False)


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [1, 2, 3, 4, 5])
def test_broad_driver_random_configs(mcd, dev, oracle, tmp_path, seed):
    """Random probe-set sizes, batch sizes that do not divide them, layer subsets and top_k through the drop-in driver: the CSV
    of the fused route against the oracle on the run's own cache files, and the same bytes from the cache route."""
    from mammo_clip_dissect_amd.concept_vit import describe_broad_neurons as drv
    rng = np.random.default_rng(seed)
    top_k = int(rng.choice([28, 64, 100]))
    n = int(rng.integers(top_k, top_k + 260))
    batch = int(rng.choice([37, 64, 100, 333]))
    ids = sorted(rng.choice(12, size=int(rng.integers(1, 4)), replace=False).tolist())
    layers = ["image_encoder.encoder.layer[%d]" % i for i in ids]
    act, res = str(tmp_path / "acts"), str(tmp_path / "results")
    argv = ["--target_model", "breastclip_vit", "--target_layers", ",".join(layers), "--d_probe", "synthetic_%d_224" % n,
            "--concept_set", CONCEPTS, "--batch_size", str(batch), "--device", str(dev), "--activation_dir", act,
            "--top_k", str(top_k)]
    out = drv.main(argv + ["--result_dir", res])
    csvs = glob.glob(os.path.join(out, "*.csv"))
    assert len(csvs) == 1
    words = open(CONCEPTS).read().split("\n")
    _check_csv_against_oracle(csvs[0], act + "/**/*.pt", layers, oracle, "og", top_k, words)
    out2 = drv.main(argv + ["--result_dir", res + "2"])          # the cache files exist now: the per-layer route
    csv2 = glob.glob(os.path.join(out2, "*.csv"))
    assert open(csv2[0], "rb").read() == open(csvs[0], "rb").read()


def test_bench_stress_line_carries_its_evidence(dev):
    """`bench.py --config stress` (a reduced shape here): ONE JSON line whose roofline names K4s with gathered bytes next to the
    algorithmic ones, and whose gemm_stress carries the fraction of the bf16 MFMA peak AND the vendor library's plain GEMM of
    the same shape, timed live (the yardstick DESIGN.md section 7 quotes)."""
    import json
    import subprocess
    import sys
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--config", "stress", "--steps", "1", "--warmup", "1",
           "--stress-images", "3000", "--stress-concepts", "1536"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=dict(os.environ, MCD_BENCH_NO_LAUNCH="1"))
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["rccl_ranks"] == 1 and d["dtype"] == "bf16" and d["config"]["core_only"]
    roof = d["roofline"]
    assert "wpmi_bf16_kernel" in roof["kernel"] and roof["bound"] == "hbm" and roof["traffic"] is None   # no PMC file of this shape
    assert roof["gathered_bytes"] == 2.0 * 1536 * 9216 * 100 and roof["gathered_over_algorithmic"] >= 1.0
    g = d["gemm_stress"]
    assert g["shape"] == [3000, 1536, 512] and 0.0 < g["frac_of_peak"] < 1.0
    assert 0.0 < g["kernel_ms_in_call"] <= g["ms"] and 0.0 < g["kernel_ms"] and g["ms_in_pass"] > 0 and "pmc_kernel_ms" not in g    # PMC figures only at their own shape
    y = g["library_yardstick"]
    assert "error" not in y, y
    assert y["ms_images_by_concepts"] > 0 and y["ms_concepts_by_images"] > 0 and 0.0 < y["frac_of_peak"] < 1.0
    assert set(d["stage_ms"]) == {"gemm", "softmax", "topk", "wpmi", "logsumexp", "row_topk"}
    rc = d["roofline_core"]          # the stress chain has no K2 launch (fused into K1s)
    assert {"K1", "K3", "K4", "K5", "K6"} <= set(rc) and "K2" not in rc and rc["K1"]["peak_tflops"] == 2500.0


def test_bench_headline_line_carries_the_stress_gemm_and_true_kernel_figures(dev):
    """The driver-run headline line (a reduced probe set here) carries north_star's GEMM figure measured in the same run
    (`gemm_stress`: the whole call and the kernel alone, at the full 25 000 x 10 000 x 512), `gemm` timed around the K1 kernel
    itself (not the host-gapped stage), and a K4 roofline whose `frac` is against the paper floor of instructions per log."""
    import json
    import subprocess
    import sys
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "1", "--images", "600", "--batch", "300",
           "--no-cpu-baseline"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=dict(os.environ, MCD_BENCH_NO_LAUNCH="1"))
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    gs = d["gemm_stress"]
    assert "error" not in gs, gs
    # (the kernel launched back to back runs in a lower clock regime than inside the call: kernel_ms may exceed the call's mean)
    assert gs["shape"] == [25000, 10000, 512] and 0.0 < gs["kernel_ms_in_call"] <= gs["ms"] and 0.0 < gs["kernel_ms"] <= 1.15 * gs["ms"]
    assert gs["schema"] == 5       #  kernel_* = the kernel alone; frac_of_peak = the whole entry point, as in rounds 2-3
    assert abs(gs["kernel_frac_of_peak"] - 2.0 * 25000 * 10000 * 512 / (gs["kernel_ms"] * 1e-3) / 1e12 / 2500.0) < 2e-3
    assert abs(gs["frac_of_peak"] - 2.0 * 25000 * 10000 * 512 / (gs["ms"] * 1e-3) / 1e12 / 2500.0) < 2e-3
    assert 0.05 < gs["frac_of_peak"] <= gs["kernel_in_call_frac_of_peak"] < 0.75 and 0.05 < gs["kernel_frac_of_peak"] < 0.75
    # every core kernel with its own in-pass bracket and roofline fraction
    rc = d["roofline_core"]
    assert {"K1", "K2", "K3", "K4", "K5", "K6"} <= set(rc)
    assert all(0.0 < rc[k]["ms"] and 0.0 < rc[k]["frac"] < 1.0 for k in ("K1", "K2", "K3", "K4", "K5", "K6"))
    assert rc["K4"]["bound"] == "valu" and rc["K3"]["bound"] == "hbm" and rc["K1"]["bound"] == "mfma"
    assert abs(rc["K4"]["ms"] - d["roofline"]["avg_launch_ms"]) < 1e-3
    gm = d["gemm"]
    assert 0.0 < gm["ms"] <= gm["stage_ms_with_host_gaps"] and gm["tflops"] > 0
    roof = d["roofline"]
    assert roof["bound"] == "valu" and roof["valu_instr_per_log_floor"] == 7.5 and 0.0 < roof["frac"] < 1.0
    assert "valu_instr_per_log" not in roof                    # PMC-derived figures only at the shape they were counted on
