"""GPU: the BASELINE.json configurations and the north_star's CSV acceptance item, on the HIP path.

  * HIP-computed similarities -> write_descriptions_csv -> bytes == the CSV the REFERENCE's driver code wrote for the
    same inputs (tests/golden/descriptions_*.csv, made by make_golden.py from /root/reference), at 1 rank and through the
    2- and 3-rank rehearsal (rank 0's file; uneven shards).
  * configs[0] literally: describe_og_neurons.py -> og_utils, ResNet-50 target, 256 synthetic 224x224 images, 763 concepts.
  * configs[3]'s target: breastclip_classifier with --num_class 4 and 1 (reference run_clipdissect.sh:16-36).
  * configs[4]'s arithmetic: Dissector(gemm_mode="bf16") at 10 000 concepts, property checks (no parity claim in bf16).
  * one reference-made golden at configs[1]'s real layer size (10 000 x 768 x 763).
"""
import glob
import io
import os
import socket
import sys

import numpy as np
import pandas as pd
import pytest
import torch
import torch.multiprocessing as mp

import util

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CONCEPTS = os.path.join(ROOT, "mammo-clip-dissect_amd", "Concepts", "Specific_concepts_sorted.txt")


def _words():
    with open(CONCEPTS) as f:
        return f.read().split("\n")


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


# ---- (a) CSV bytes -------------------------------------------------------------------------------------
def _shared2_csvs(world, rank):
    """The shared2 fixture (one probe set = the main case, two layers) through the fused pipeline; returns the two CSV
    texts (og, clip) as rank 0 writes them, or None on the other ranks."""
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import mammo_clip_dissect_amd  # noqa: F401
    import util as U
    from mammo_clip_dissect_amd.pipeline import Dissector, shard_bounds, write_descriptions_csv
    dev = torch.device("cuda:0")
    zm, z2 = U.golden("main"), U.golden("shared2")
    At = np.concatenate([zm["A"].T, z2["A1"].T])                 # [64 + 24, 256] neuron-major
    N = At.shape[1]
    lo, hi = shard_bounds(N, world, rank)
    dis = Dissector(hi - lo, ["layer_a", "layer_b"], [64, 24], 763, 512, dev, top_k=int(z2["top_k"]),
                    gather=U.host_staged_gather() if world > 1 else None)
    dis.At[:, :hi - lo] = torch.from_numpy(At[:, lo:hi].copy()).to(dev)
    dis.E_img[:] = torch.from_numpy(zm["E_img"][lo:hi].copy()).to(dev)
    dis.cursor = hi - lo
    res = dis.finish(torch.from_numpy(zm["E_txt"]).to(dev))
    torch.cuda.synchronize()
    if rank != 0:
        return None
    out = {}
    words = open(CONCEPTS).read().split("\n")
    for variant in ("og", "clip"):
        buf = io.StringIO()
        write_descriptions_csv(res, words, buf, variant)
        out[variant] = buf.getvalue()
    out["sim"] = res.sim.cpu().numpy()
    return out


def _csv_worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    out = _shared2_csvs(world, rank)
    if rank == 0:
        q.put(out)
    dist.barrier()
    dist.destroy_process_group()


def _diff_report(got, want):
    g, w = got.splitlines(), want.splitlines()
    bad = [i for i, (a, b) in enumerate(zip(g, w)) if a != b]
    return "%d of %d lines differ; first: %r vs %r" % (len(bad) + abs(len(g) - len(w)), len(w),
                                                       g[bad[0]] if bad else None, w[bad[0]] if bad else None)


@pytest.mark.parametrize("world", [1, 2, 3])
def test_hip_csv_bytes_equal_reference_csv(world):
    """north_star: "bit-identical top-concept-per-neuron CSV vs reference at 1 and N GPUs".  Embeddings + activations
    of the reference-made fixture -> K1a/K1/K2/K3/K4/K5/K6 on the MI355X -> the CSV writer; every byte of both driver
    variants must equal what the reference's own code wrote (descriptions_shared2_{og,clip}.csv)."""
    if world == 1:
        out = _shared2_csvs(1, 0)
    else:
        ctx = mp.get_context("spawn")
        q = ctx.Queue()
        port = _free_port()
        procs = [ctx.Process(target=_csv_worker, args=(r, world, port, q)) for r in range(world)]
        for p in procs:
            p.start()
        out = q.get(timeout=600)
        for p in procs:
            p.join(timeout=120)
            assert p.exitcode == 0
    zm, z2 = util.golden("main"), util.golden("shared2")
    ref_sim = np.concatenate([zm["soft_wpmi"], z2["soft_wpmi1"]])
    d = np.abs(out["sim"].astype(np.float64) - ref_sim)
    stats = "sims vs reference: max %.3e, %.5f bit-identical" % (d.max(), (d == 0).mean())
    for variant in ("og", "clip"):
        want = open(os.path.join(util.GOLDEN, "descriptions_shared2_%s.csv" % variant), newline="").read()
        assert out[variant] == want, "%s (%d ranks): %s; %s" % (variant, world, _diff_report(out[variant], want), stats)


def test_hip_csv_bytes_two_probe_sets(dev):
    """The round-1 golden CSVs (two 'layers' that are two different probe sets: the main and relu cases), each scored by
    its own Dissector on the HIP path and written as one file."""
    from mammo_clip_dissect_amd.pipeline import DissectResult, Dissector, write_descriptions_csv
    parts = []
    for name in ("main", "relu"):
        z = util.golden(name)
        N, U = z["A"].shape
        dis = Dissector(N, ["x"], [U], 763, 512, dev, top_k=int(z["top_k"]))
        dis.At[:, :N] = torch.from_numpy(z["A"].T.copy()).to(dev)
        dis.E_img[:] = torch.from_numpy(z["E_img"]).to(dev)
        dis.cursor = N
        parts.append(dis.finish(torch.from_numpy(z["E_txt"]).to(dev)))
    res = DissectResult(["layer_a", "layer_b"], [p.sim.shape[0] for p in parts], *[
        torch.cat([getattr(p, f) for p in parts]) for f in ("sim", "vals", "ids", "top_ids", "top_vals")], None)
    for variant in ("og", "clip"):
        buf = io.StringIO()
        write_descriptions_csv(res, _words(), buf, variant)
        want = open(os.path.join(util.GOLDEN, "descriptions_%s.csv" % variant), newline="").read()
        assert buf.getvalue() == want, "%s: %s" % (variant, _diff_report(buf.getvalue(), want))


# ---- (f) the real-size golden ----------------------------------------------------------------------------
def test_real_layer_size_golden_on_hip(dev):
    """configs[1]'s layer shape, reference-made outputs (make_golden.py main_round2): from the embeddings through
    K1a/K1 (P must be the reference's bits: sha256), then soft_wpmi at the boundary tolerance, top-5 images exact, the top
    concept exact on ALL 768 neurons (the reference's smallest top-1 gap is 10 ulp), and every top-10 rank whose gaps to
    its neighbours in the reference's full score matrix exceed one ulp (all but a handful; the count goes to the stats file)."""
    from mammo_clip_dissect_amd import core
    from mammo_clip_dissect_amd.concept_vit import similarity
    g = util.n10k_inputs()
    z = g["z"]
    P = core.embed_gemm(core.normalize_rows(torch.from_numpy(g["E_img"]).to(dev)),
                        core.normalize_rows(torch.from_numpy(g["E_txt"]).to(dev)))
    assert util.sha256(P.cpu().numpy()) == str(z["P_sha256"])
    A = torch.from_numpy(g["A"]).to(dev)
    sim = similarity.soft_wpmi(P, A, top_k=g["K"], device=str(dev))
    util.assert_sim_boundary(sim.cpu().numpy(), z["soft_wpmi"], "n10k soft_wpmi")
    _, t5 = core.col_topk(A, 5)
    assert np.array_equal(t5.cpu().numpy().T, z["top5"])
    v10, i10 = core.row_topk(sim, 10)
    assert np.array_equal(i10.cpu().numpy()[:, 0], z["imax"])           # all 768 rows
    assert np.abs(v10.cpu().numpy() - z["vals10"]).max() <= util.SIM_BOUNDARY_ATOL
    n_dec, n_und = util.assert_topk_against_full_sim(i10.cpu().numpy(), z["soft_wpmi"], 10, "n10k")
    assert n_und <= 40 and n_dec >= 7600                                  # 7 680 ranks in all


# ---- (b) configs[0] -----------------------------------------------------------------------------------------
def test_describe_og_neurons_resnet50_config0(dev, oracle, tmp_path):
    """BASELINE configs[0] as written: describe_og_neurons.py (-> og_utils.save_activations, reference og_utils.py:374-471,
    get_similarity_from_activations :474-518) on 256 random 224x224 images, ResNet-50 target (conv1 + layer1-4), 763 concepts,
    top-10 descriptions.  The reference calls encode_image on every target (og_utils.py:93); the offline ResNet-50 aliases
    it to forward."""
    from test_gpu_pipeline import _check_csv_against_oracle
    from mammo_clip_dissect_amd.concept_vit import describe_og_neurons as drv
    layers = ["conv1", "layer1", "layer2", "layer3", "layer4"]
    act, res = str(tmp_path / "acts"), str(tmp_path / "results")
    out = drv.main(["--target_model", "resnet50", "--target_layers", ",".join(layers), "--d_probe", "synthetic_256_224",
                    "--concept_set", CONCEPTS, "--batch_size", "64", "--device", str(dev), "--activation_dir", act,
                    "--result_dir", res])
    df = pd.read_csv(os.path.join(out, "descriptions.csv"))
    assert [int((df.layer == l).sum()) for l in layers] == [64, 256, 512, 1024, 2048]
    assert os.path.exists(os.path.join(out, "args.txt"))
    # og_utils' save prefix (reference og_utils.py:394-406) in front of the reference's file names
    files = glob.glob(act + "/**/*.pt", recursive=True)
    assert files and all("clip_dissector_resnet50_target_synthetic_256_224_small_not_mammo_pretrained_" in f for f in files)
    _check_csv_against_oracle(os.path.join(out, "descriptions.csv"), act + "/**/*.pt", layers, oracle, "og", 100, _words())


# ---- (c) configs[3]'s target ------------------------------------------------------------------------------
@pytest.mark.parametrize("num_class", [4, 1, 5])
def test_describe_broad_neurons_classifier(dev, oracle, tmp_path, num_class):
    """C-Mammo-CLIP Dissect: the fine-tuned EfficientNet-B5 classifier target with --num_class 4 / 1 / 5 = all the class counts of the four tasks (reference
    run_clipdissect.sh:16-36, data_utils.py:53-61), Mammo-CLIP dissector; a second encoder (target != dissector)."""
    from test_gpu_pipeline import _check_csv_against_oracle
    from mammo_clip_dissect_amd.concept_vit import describe_broad_neurons as drv
    layers = ["image_encoder._blocks[%d]" % i for i in (0, 17, 38)]
    act, res = str(tmp_path / "acts"), str(tmp_path / "results")
    out = drv.main(["--target_model", "breastclip_classifier", "--num_class", str(num_class), "--target_layers",
                    ",".join(layers), "--d_probe", "synthetic_160_224", "--concept_set", CONCEPTS, "--batch_size", "40",
                    "--device", str(dev), "--activation_dir", act, "--result_dir", res, "--top_k", "100"])
    csvs = glob.glob(os.path.join(out, "*.csv"))
    assert len(csvs) == 1
    df = pd.read_csv(csvs[0])
    assert [int((df.layer == l).sum()) for l in layers] == [24, 128, 512]
    _check_csv_against_oracle(csvs[0], act + "/**/*.pt", layers, oracle, "og", 100, _words())


# ---- (d) configs[4]'s arithmetic ---------------------------------------------------------------------------
def test_dissector_bf16_chain_at_10k_concepts(dev):
    """Dissector(gemm_mode="bf16") at C = 10 000 (the stress configuration's concept count; one launch sequence of the
    whole bf16 chain).  No parity claim in bf16; what must hold: P within bf16 rounding of the fp32 P, image sets exact
    (they do not depend on P), scores finite, columns of exp(sim) average to 1 per layer (similarity.py:70-72 with lam = 1:
    sim = pdge - logsumexp + log U), the top-1 concept equal to the fp32 chain's wherever that one is decided by more than
    the bf16 error allows, and every reported value equal to sim at the reported index."""
    from mammo_clip_dissect_amd.pipeline import Dissector
    N, C, D, K = 3000, 10000, 512, 100
    widths = [96, 160]
    g = torch.Generator().manual_seed(5)
    At = torch.randn(sum(widths), N, generator=g)
    E_img, E_txt = torch.randn(N, D, generator=g), torch.randn(C, D, generator=g)
    outs = {}
    for mode in ("f32", "bf16"):
        dis = Dissector(N, ["a", "b"], widths, C, D, dev, top_k=K, gemm_mode=mode)
        dis.At[:, :N] = At.to(dev)
        dis.E_img[:] = E_img.to(dev)
        dis.cursor = N
        outs[mode] = dis.finish(E_txt.to(dev))
    f, b = outs["f32"], outs["bf16"]
    assert torch.equal(f.top_ids, b.top_ids)
    sim = b.sim
    assert bool(torch.isfinite(sim).all())
    for name, sl in b.layer_slices():
        m = torch.exp(sim[sl].double()).mean(dim=0)
        assert float((m - 1).abs().max()) < 1e-3, name
    assert torch.equal(torch.gather(sim, 1, b.ids.long()), b.vals)
    assert bool((b.vals[:, :-1] >= b.vals[:, 1:]).all())
    d = (b.sim - f.sim).abs()
    assert float(d.max()) < 0.5 and float(d.mean()) < 0.05, (float(d.max()), float(d.mean()))
    top2 = torch.topk(f.sim, 2, dim=1).values
    decided = (top2[:, 0] - top2[:, 1]) > 4 * float(d.max())
    assert torch.equal(b.ids[decided, 0], f.ids[decided, 0])


@pytest.mark.parametrize("size,n,batch", [(448, 120, 24), (1024, 104, 8)])
def test_vit_tower_at_another_resolution(dev, oracle, tmp_path, size, n, batch):
    """configs[4] asks for 1024 x 1024 probes; the offline ViT towers take their resolution as a parameter
    ('breastclip_vit_<size>').  448 x 448 (785 tokens) and configs[4]'s own 1024 x 1024 (4 097 tokens) -- both beyond K9's
    256-token limit, so the attention is PyTorch's SDPA -- through the drop-in driver, CSV checked against the oracle like
    every other driver run."""
    from test_gpu_pipeline import _check_csv_against_oracle
    from mammo_clip_dissect_amd.concept_vit import describe_broad_neurons as drv
    layers = ["image_encoder.encoder.layer[0]", "image_encoder.encoder.layer[11]"]
    act, res = str(tmp_path / "acts"), str(tmp_path / "results")
    out = drv.main(["--target_model", "breastclip_vit_%d" % size, "--target_layers", ",".join(layers), "--d_probe",
                    "synthetic_%d_%d" % (n, size), "--concept_set", CONCEPTS, "--batch_size", str(batch), "--device", str(dev),
                    "--activation_dir", act, "--result_dir", res, "--top_k", "100"])
    csvs = glob.glob(os.path.join(out, "*.csv"))
    assert len(csvs) == 1
    _check_csv_against_oracle(csvs[0], act + "/**/*.pt", layers, oracle, "og", 100, _words())
