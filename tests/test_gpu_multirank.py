"""GPU: the image-sharded multi-rank path with the REAL HIP kernels.  gpurun gives one GPU, and RCCL cannot put two
ranks on one device, so the ranks rendezvous over gloo and the TEST injects a host-staged all-gather
(tests/util.py: host_staged_gather; the package itself only ships the RCCL transport) while every kernel runs on cuda:0.  What is checked is what SURVEY 8e promises: the 2- and
3-rank results equal the 1-rank result bit for bit, and bench.py's multi-rank protocol produces one valid JSON line."""
import glob
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _problem(N, widths, C, D, seed):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(sum(widths), N, generator=g), torch.randn(N, D, generator=g), torch.randn(C, D, generator=g)


def _run(world, rank, N, widths, C, D, K, seed, gemm_mode="f32"):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import mammo_clip_dissect_amd  # noqa: F401
    import util
    from mammo_clip_dissect_amd.pipeline import Dissector, shard_bounds
    dev = torch.device("cuda:0")
    At, E_img, E_txt = _problem(N, widths, C, D, seed)
    lo, hi = shard_bounds(N, world, rank)
    n_l = hi - lo
    dis = Dissector(n_l, ["l%d" % i for i in range(len(widths))], widths, C, D, dev, top_k=K, gemm_mode=gemm_mode,
                    gather=util.host_staged_gather() if world > 1 else None)
    dis.At[:, :n_l] = At[:, lo:hi].to(dev)
    dis.E_img[:] = E_img[lo:hi].to(dev)
    dis.cursor = n_l
    r = dis.finish(E_txt.to(dev), k_desc=10, k_img=5)
    torch.cuda.synchronize()
    return [t.cpu().numpy() for t in (r.sim, r.vals, r.ids, r.top_ids, r.top_vals)]


def _worker(rank, world, port, case, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    out = _run(world, rank, *case)
    q.put((rank, out))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,case", [(2, (1200, [96, 40, 7], 763, 512, 100, 21)), (3, (900, [64, 130], 763, 512, 100, 22)),
                                        (3, (1001, [64, 33], 763, 512, 100, 23)),    # 334 + 334 + 333 images
                                        (4, (250, [40], 763, 512, 100, 24)),         # 63+63+62+62: every shard < top_k
                                        # the stress chain (bf16 E + reciprocal row sums gathered instead of fp32 S): an
                                        # image's row of E and its row sum do not depend on which rank, tile or launch
                                        # computes them, so this chain is bit-identical across rank counts too
                                        (3, (1001, [64, 33], 1500, 512, 100, 25, "bf16"))])
def test_ranks_on_hip_bit_identical_to_one(world, case):
    single = _run(1, 0, *case)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, case, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=600) for _ in range(world))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    for r in range(world):
        for a, b in zip(single, got[r]):
            assert np.array_equal(a, b)


def test_bench_two_ranks_protocol():
    """bench.py under torch.distributed.run with 2 ranks (gloo rendezvous, both on cuda:0): one JSON line from rank 0,
    whole-job value = images of both ranks / max-over-ranks time."""
    env = dict(os.environ, MCD_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1",
           "--images", "500", "--batch", "250"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["config"]["global_images"] == 1000 and j["scaling"] == "weak"
    assert abs(j["value"] - 1000 * j["steps"] / (j["ms_per_step"] * j["steps"] / 1000.0)) / j["value"] < 1e-3
    assert j["roofline"]["frac"] > 0 and "cpu_baseline" not in j


def test_bench_plain_invocation_starts_its_own_ranks():
    """`python bench.py --gpus 2` with no torch.distributed.run around it starts its own two ranks as child
    processes (here rehearsed over gloo on the one GPU) and relays rank 0's single JSON line with n_gpus == 2."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(MCD_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--images", "500",
           "--batch", "250"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout[-2000:]
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["rccl_ranks"] == 2 and j["dist_backend"] == "gloo"
    assert j["config"]["global_images"] == 1000 and j["config"]["images_per_gpu"] == [500, 500]
    # with RCCL asked for (the default) two ranks on a one-GPU box must fail loudly, never fall back to one rank
    env.pop("MCD_DIST_BACKEND")
    if torch.cuda.device_count() < 2:
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
        assert r.returncode != 0 and not [l for l in r.stdout.splitlines() if l.startswith("{")]
        assert "RCCL ranks need" in r.stderr


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs: the RCCL transport with more than one rank")
def test_bench_two_gpus_over_rccl():
    """Only where the box has two GPUs (the driver's 8-GPU node; gpurun's boxes have one): `python bench.py --gpus 2` started
    plainly -- two ranks, one per GPU, backend nccl = RCCL over xGMI: the three all-gathers of SURVEY 8(e) on device tensors,
    the object broadcast of the encoder-GEMM picks, one JSON line with rccl_ranks == 2."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MCD_DIST_BACKEND")}
    env.update(HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1", "--images", "500",
           "--batch", "250", "--no-cpu-baseline"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, r.stderr[-3000:]
    j = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert j["n_gpus"] == 2 and j["rccl_ranks"] == 2 and j["dist_backend"] == "nccl" and j["config"]["global_images"] == 1000
    # and the strong-scaling mode on uneven shards (501 + 500), CSV written by rank 0
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--global-images", "1001",
           "--batch", "250", "--no-cpu-baseline"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, r.stderr[-3000:]
    j = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert j["scaling"] == "strong" and j["config"]["images_per_gpu"] == [501, 500]


def test_bench_strong_scaling_uneven_shards():
    """configs[2]'s mode: ONE probe set sharded over the ranks (`--global-images`), here 1001 images over 3 ranks
    (334 + 334 + 333), through the drop-in driver."""
    env = dict(os.environ, MCD_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "3", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "3", "--steps", "1", "--warmup", "0",
           "--global-images", "1001", "--batch", "167"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    j = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert j["n_gpus"] == 3 and j["scaling"] == "strong" and j["config"]["global_images"] == 1001
    assert j["config"]["images_per_gpu"] == [334, 334, 333]
    assert "describe_broad_neurons.main" in j["config"]["entry_point"] and j["roofline"]["frac"] > 0


def _driver_csv(world, rank, tmp, n_images, batch):
    """describe_broad_neurons.main on this rank's shard (prebuilt models + resident probes + injected gather)."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import mammo_clip_dissect_amd  # noqa: F401
    import util
    from mammo_clip_dissect_amd.concept_vit import data_utils, describe_broad_neurons, utils
    from mammo_clip_dissect_amd.pipeline import shard_bounds
    dev = torch.device("cuda:0")
    clip_model, target_model = utils.build_mammo_models("breastclip_vit", dev)
    lo, hi = shard_bounds(n_images, world, rank)
    d_probe = "synthetic_%d_224" % n_images
    data = data_utils.get_data(d_probe, None, dev, lo, hi)
    layers = ["image_encoder.encoder.layer[%d]" % i for i in (0, 5, 11)]
    concepts = os.path.join(ROOT, "mammo-clip-dissect_amd", "Concepts", "Specific_concepts_sorted.txt")
    out = describe_broad_neurons.main(
        ["--target_model", "breastclip_vit", "--target_layers", ",".join(layers), "--d_probe", d_probe, "--concept_set",
         concepts, "--batch_size", str(batch), "--device", "cuda:0", "--activation_dir", os.path.join(tmp, "acts%d_%d" % (world, rank)),
         "--result_dir", os.path.join(tmp, "res%d" % world), "--top_k", "100"],
        prebuilt={"clip_model": clip_model, "target_model": target_model, "data": data,
                  "gather": util.host_staged_gather() if world > 1 else None})
    torch.cuda.synchronize()
    return out


def _driver_worker(rank, world, port, tmp, n_images, batch, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    os.environ["WORLD_SIZE"] = str(world); os.environ["RANK"] = str(rank); os.environ["LOCAL_RANK"] = "0"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    out = _driver_csv(world, rank, tmp, n_images, batch)
    q.put((rank, out))
    dist.barrier()
    dist.destroy_process_group()


def test_driver_encoder_to_csv_bytes_one_vs_two_ranks(tmp_path, monkeypatch):
    """The WHOLE job -- encoder forwards included -- at 1 rank and at 2 ranks: the CSV rank 0 writes must be the same
    bytes.  The encoder GEMMs run through hipBLASLt, whose algorithm this build normally picks by timing (per process);
    MCD_BLASLT_PICK=heuristic pins the first heuristic candidate, and equal batch shapes on every rank (100 images per
    rank, batches of 50) then give the same summation order whichever rank encodes an image."""
    monkeypatch.setenv("MCD_BLASLT_PICK", "heuristic")
    tmp = str(tmp_path)
    one = _driver_csv(1, 0, tmp, 200, 50)
    csv1 = open(glob.glob(os.path.join(one, "*.csv"))[0], "rb").read()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_driver_worker, args=(r, 2, port, tmp, 200, 50, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=900) for _ in range(2))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    csv2 = open(glob.glob(os.path.join(got[0], "*.csv"))[0], "rb").read()
    assert csv1 == csv2 and len(csv1) > 100000


def test_rccl_backend_initialises_and_gathers():
    """The transport the real multi-GPU run uses (backend "nccl" = RCCL), as far as one GPU allows: a 1-rank process
    group created the way bench.py creates it, then the three collectives the sharded path issues
    (all_gather_into_tensor on float32 rows, barrier, all_reduce MAX on the elapsed time) and the object broadcast that carries
    rank 0's encoder-GEMM picks."""
    code = r"""
import os, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "%d")
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
x = torch.arange(12, dtype=torch.float32, device=dev).view(3, 4)
out = torch.empty(3, 4, device=dev)
dist.all_gather_into_tensor(out, x)
assert torch.equal(out, x)
dist.barrier()
t = torch.tensor([1.5], dtype=torch.float64, device=dev)
dist.all_reduce(t, op=dist.ReduceOp.MAX)
assert float(t.item()) == 1.5
# what pipeline.sync_encoder_gemm_picks sends: a pickled dict through broadcast_object_list on the RCCL backend (device-staged)
obj = [{"qkv": 3, "fc1": 17}]
dist.broadcast_object_list(obj, src=0)
assert obj[0] == {"qkv": 3, "fc1": 17}
torch.cuda.synchronize()
dist.destroy_process_group()
print("rccl ok")
""" % _free_port()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "rccl ok" in r.stdout, r.stderr[-2000:]


def _fuzz_worker(rank, world, port, n_cases, seed, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import mammo_clip_dissect_amd  # noqa: F401
    import util
    from mammo_clip_dissect_amd.pipeline import Dissector, shard_bounds
    dev = torch.device("cuda:0")
    groups = {g: dist.new_group(list(range(g))) for g in range(1, world + 1)}
    rng = np.random.default_rng(seed)                    # the same stream on every rank
    bad = []

    def run(g, r, N, widths, C, D, K, sd, mode):
        At, E_img, E_txt = _problem(N, widths, C, D, sd)
        lo, hi = shard_bounds(N, g, r)
        dis = Dissector(hi - lo, ["l%d" % i for i in range(len(widths))], widths, C, D, dev, top_k=K, gemm_mode=mode,
                        group=groups[g], gather=util.host_staged_gather(groups[g]) if g > 1 else None)
        dis.At[:, :hi - lo] = At[:, lo:hi].to(dev)
        dis.E_img[:] = E_img[lo:hi].to(dev)
        dis.cursor = hi - lo
        res = dis.finish(E_txt.to(dev), k_desc=min(10, C), k_img=min(5, N))
        torch.cuda.synchronize()
        return [t.cpu() for t in (res.sim, res.vals, res.ids, res.top_ids, res.top_vals)]

    for c in range(n_cases):
        g = int(rng.integers(2, world + 1))
        K = int(rng.choice([1, 4, 28, 100]))
        N = int(rng.integers(K, K + 900))
        widths = [int(w) for w in rng.integers(1, 140, size=int(rng.integers(1, 4)))]
        C = int(rng.choice([40, 763, 1000]))
        mode = "bf16" if rng.random() < 0.3 else "f32"
        case = (N, widths, C, 512, K, int(rng.integers(0, 1 << 30)), mode)
        out = run(g, rank, *case) if rank < g else None
        if rank == 0:
            single = run(1, 0, *case)
            if not all(torch.equal(a, b) for a, b in zip(single, out)):
                bad.append((g,) + case)
    if rank == 0:
        q.put(bad)
    dist.barrier()
    dist.destroy_process_group()


def test_ranks_on_hip_fuzz():
    """Random probe-set sizes, 2..4 ranks (sub-groups of one 4-process world sharing the GPU), layer widths, concept counts,
    top_k, fp32 and bf16 chains: the sharded HIP result equals the one-rank HIP result bit for bit."""
    world = 4
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_fuzz_worker, args=(r, world, port, 16, 9, q)) for r in range(world)]
    for p in procs:
        p.start()
    bad = q.get(timeout=900)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert bad == []


def test_driver_encoder_to_csv_bytes_one_vs_three_ranks_uneven(tmp_path, monkeypatch):
    """Encoder-inclusive identity with UNEVEN shards and a probe-set size that is no batch multiple: 260 images
    in batches of 50 -- one rank encodes 50, 50, 50, 50, 50, 10; three ranks with MCD_SHARD_ALIGN=50 hold 100 / 100 / 60 images,
    i.e. the same six batches of the global image order, two each.  No hipBLASLt fp32 solution is free of stream-K
    (profiles/r04_blaslt_algos.txt), so identical bits need identical batches; with them, rank 0's CSV is the same bytes."""
    monkeypatch.setenv("MCD_BLASLT_PICK", "heuristic")
    monkeypatch.setenv("MCD_SHARD_ALIGN", "50")
    tmp = str(tmp_path)
    one = _driver_csv(1, 0, tmp, 260, 50)
    csv1 = open(glob.glob(os.path.join(one, "*.csv"))[0], "rb").read()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_driver_worker, args=(r, 3, port, tmp, 260, 50, q)) for r in range(3)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=900) for _ in range(3))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    csv3 = open(glob.glob(os.path.join(got[0], "*.csv"))[0], "rb").read()
    assert csv1 == csv3


def test_driver_encoder_to_csv_bytes_one_vs_four_ranks_cfg2_preset(tmp_path, monkeypatch):
    """bench.py --config cfg2 in miniature: ONE probe set, a batch size that divides it into more batches than ranks,
    shard boundaries on batch multiples (MCD_SHARD_ALIGN = the batch) -- at 1 rank and at 4 ranks every batch of the global image
    order is encoded whole by exactly one rank, so the encoders' bits, and with them rank 0's CSV bytes, are the same (10 batches of
    25 images over 4 ranks: 3 + 3 + 2 + 2 batches)."""
    monkeypatch.setenv("MCD_BLASLT_PICK", "heuristic")
    monkeypatch.setenv("MCD_SHARD_ALIGN", "25")
    tmp = str(tmp_path)
    one = _driver_csv(1, 0, tmp, 250, 25)
    csv1 = open(glob.glob(os.path.join(one, "*.csv"))[0], "rb").read()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_driver_worker, args=(r, 4, port, tmp, 250, 25, q)) for r in range(4)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=900) for _ in range(4))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    csv4 = open(glob.glob(os.path.join(got[0], "*.csv"))[0], "rb").read()
    assert csv1 == csv4

