#!/usr/bin/env python3
"""bench.py -- probe images/sec dissected (763 concepts, all layers), BASELINE.json configs[1]:
M-Mammo-CLIP Dissect, Mammo-CLIP ViT-B/16 target + dissector, 10k synthetic 224x224 mammograms per GPU,
763 concepts, all 12 transformer blocks (12 x 768 neurons), soft-WPMI.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One step = ONE CALL OF THE DROP-IN DRIVER, `describe_broad_neurons.main(argv, prebuilt=...)` -- the entry point the
reference launches (concept_vit/describe_broad_neurons.py:51-175, run_clipdissect.sh:6-9) -- over a probe set that is
resident in HBM: encoder forward over every image with the K0 hooks writing the activation matrix (PyTorch-ROCm fp32,
single pass: target == dissector), text tower over the 763 concepts, the HIP core once for all layers (K1 GEMM, K2
softmax, K3 top-K images, K4 soft-WPMI, K5 logsumexp, K6 top-10), the reference-format CSV + args file written by rank
0, and the reference-format activation-cache files (a side output, written by a background thread; --no-activation-cache
drops them).  Only model construction and the generation of the synthetic probe set are outside the step.

Scaling modes: weak (default; `--images` per GPU, config.workload says "weak") and strong (`--global-images N`: ONE probe
set of N images sharded over the ranks, uneven shards allowed -- BASELINE configs[2]).  The collectives are the three
all-gathers of SURVEY.md 8e over RCCL.

Other workloads (never the headline): `--config stress` = the bf16 core at one rank's share of configs[4] (25 000 images x
10 000 concepts x 12 x 768 neurons) with a `gemm_stress` object (image-embedding x text-embedding GEMM: TFLOP/s and
fraction of the 2.5 PFLOP/s bf16 MFMA peak); `--config core` = the fp32 core alone on random activations (dev / PMC runs).

The JSON line carries `roofline` for the dominant kernel of the dissection core (K4, HBM roofline),
`roofline_encoder` for the hand-written encoder kernel with the most GPU time (K9 attention, fp32 MFMA), both timed live
with HIP events on the launch stream inside the timed region, `library_gemm_share` (hipBLASLt time / step) and
`cpu_baseline` (rank 0, N=1 only): the like-for-like CPU rate of the same job on this box's host cores.
"""
import argparse
import json
import os
import shutil
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch
import torch.distributed as dist

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8 TB/s
F32_MFMA_PEAK_TF = 157.3   # v_mfma_f32_32x32x2_f32
BF16_MFMA_PEAK_TF = 2500.0  # dense bf16 MFMA
# VALU issue peak: one wave-instruction (64 lanes) per 4 cycles and SIMD (what SQ_ACTIVE_INST_VALU counts per SQ_INSTS_VALU on
# gfx950: profiles/r03_k4_pmc_sq.txt), 256 CUs x 4 SIMDs at 2.4 GHz
VALU_PEAK_LANE_INSTR_S = 256 * 4 * 2.4e9 / 4 * 64
# K4 (parity mode), the implementation-independent floor of its vector work: lane-instructions per log with every fp32 operation
# of the reference rounded on its own and a correctly rounded log -- packed fp32 (2 logs per instruction):
K4_FLOOR_PER_LOG = {"term (g - 1, * p, + 1, + min_prob: 4 packed per pair)": 2.0,
                    "log (exact-r table method: 10 packed per pair)": 5.0,
                    "sum over the K rows (1 packed add per pair)": 0.5}
K4_FLOOR = sum(K4_FLOOR_PER_LOG.values())      # 7.5; what the kernel adds on top (addressing, table indices, loop) is measured, below
K4_PMC = "r05_k4_pmc.json"                 # SQ_INSTS_VALU etc. of this tree at configs[1]'s shape (scripts/r05_pmc.sh)
GEXP_PMC = "r05_gemm_stress_pmc.json"      # K1s' counters at 25 000 x 10 000 x 512 (scripts/r05_pmc.sh)
TRAFFIC_CORE = "r05_pmc_traffic.json"
TRAFFIC_STRESS = "r05_stress_pmc_traffic.json"
CONCEPTS = os.path.join(ROOT, "mammo-clip-dissect_amd", "Concepts", "Specific_concepts_sorted.txt")
KERNEL_NAMES = {"gemm": "K1 normalize+embed_gemm", "softmax": "K2 row_softmax",
                "topk": "K3 col_topk (neuron_topk_fast_kernel)",
                "wpmi": "K4 wpmi_score (wpmi_slice_kernel<soft, accurate log, S_IS_PROB>)",
                "logsumexp": "K5 logsumexp_sub", "row_topk": "K6 row_topk"}
STAGES = ["gemm", "softmax", "topk", "wpmi", "logsumexp", "row_topk"]


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--images", type=int, default=10000, help="probe images PER GPU (weak scaling)")
    ap.add_argument("--global-images", type=int, default=None,
                    help="strong scaling: ONE probe set of this many images sharded over the ranks (configs[2]: 10000)")
    ap.add_argument("--batch", type=int, default=None,
                    help="images per encoder forward (the reference hard-codes 20 / 50, utils.py:84,:297).  Default: 2500 for "
                         "the ViT target, 125 for the EfficientNet one (its activations are 20x larger)")
    ap.add_argument("--image-size", type=int, default=224)
    ap.add_argument("--target", default="breastclip_vit")
    ap.add_argument("--top-k", type=int, default=100)
    ap.add_argument("--config", default="headline", choices=["headline", "cfg2", "stress", "core"],
                    help="cfg2: BASELINE configs[2] as ONE job at every rank count -- 10 000 images in all, encoder batches of 1 250, "
                         "shard boundaries on batch multiples: the same batches are encoded whole at --gpus 1 / 2 / 4 / 8, so the "
                         "CSV bytes are the same by construction (north_star: bit-identical CSV at 1 and 8 GPUs)")
    ap.add_argument("--core-only", action="store_true", help="alias of --config core")
    ap.add_argument("--no-activation-cache", action="store_true",
                    help="do not write the reference-format activation cache files (the driver's side output)")
    ap.add_argument("--align-shards", action="store_true",
                    help="shard boundaries on encoder-batch multiples (MCD_SHARD_ALIGN = --batch): what encoder-inclusive "
                         "bit-identity of the CSV across rank counts needs (pipeline.shard_align)")
    ap.add_argument("--cpu-baseline-layers", type=int, default=12, help="layers the CPU oracle is timed on")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-tunableop", action="store_true", help="encoder GEMMs on the libraries' default solutions")
    ap.add_argument("--tune", action="store_true", help="let TunableOp tune unseen GEMM shapes")
    ap.add_argument("--stress-images", type=int, default=25000, help="--config stress: images of one rank's share")
    ap.add_argument("--stress-concepts", type=int, default=10000)
    a = ap.parse_args()
    if a.core_only:
        a.config = "core"
    a.preset = None
    if a.config == "cfg2":
        # the paired 1-GPU / 8-GPU preset: what differs between rank counts is only who encodes which batch
        a.preset, a.config = "cfg2", "headline"
        a.global_images = a.global_images or 10000
        a.batch = a.batch or 1250
        a.align_shards = True
    return a


def host_cpu_share():
    """CPUs this process may actually use: min(affinity, cgroup cpu.max quota)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, n)


def algorithmic_work(stage, N_total, N_local, C, D, widths, K, world, s_bytes=4):
    """ALGORITHMIC bytes (or flops) of one launch of each core kernel on one rank.
    s_bytes: bytes per element of the similarity matrix K4 gathers (4: fp32 S; 2: the stress chain's bf16 E)."""
    U = sum(widths)
    U_rank = (U + world - 1) // world
    if stage == "gemm":      # K1a + K1: 2*N*C*D flop; 4(ND + CD) + s_bytes*NC bytes
        return dict(flops=2.0 * N_local * C * D, bytes=4.0 * (N_local * D + C * D) + s_bytes * N_local * C)
    if stage == "softmax":   # K2: read P, write S
        return dict(bytes=8.0 * N_local * C)
    if stage == "topk":      # K3: one read of the activations + (value,index) out
        return dict(bytes=4.0 * N_local * U + 8.0 * K * U)
    if stage == "wpmi":      # K4: every touched row of S once per layer + indices + output
        share = U_rank / float(U)
        rows = sum(min(N_total, w * K) for w in widths) * share
        return dict(bytes=s_bytes * C * rows + 4.0 * K * U_rank + 4.0 * U_rank * C)
    if stage == "logsumexp":  # K5: read pdge, write sim
        return dict(bytes=8.0 * U * C)
    if stage == "row_topk":  # K6
        return dict(bytes=4.0 * U * C)
    raise KeyError(stage)


class StageTimer:
    """HIP events between the stages of Dissector.finish (pipeline.STAGE_MARK), recorded on torch's current stream =
    the stream libmcd_hip.so launches on."""

    def __init__(self):
        self.runs, self.cur, self.on = [], None, False

    def mark(self, name):
        if not self.on:
            return
        if name == "start":
            self.cur = []
            self.runs.append(self.cur)
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        self.cur.append((name, e))

    def stage_ms(self):
        """Stage = from the previous mark to the stage's own (host launch gaps between its kernels included)."""
        out = {s: 0.0 for s in STAGES}
        for marks in self.runs:
            prev = None
            for name, e in marks:
                if prev is not None and name in out:
                    out[name] += prev.elapsed_time(e)
                if not name.endswith(":begin") and not name.endswith(":end"):
                    prev = e
        n = max(len(self.runs), 1)
        return {k: v / n for k, v in out.items()}

    def kernel_ms(self, name):
        """A single launch bracketed by its own "<name>:begin" / "<name>:end" marks (None when the marks are missing)."""
        tot, n = 0.0, 0
        for marks in self.runs:
            d = dict(marks)
            if name + ":begin" in d and name + ":end" in d:
                tot += d[name + ":begin"].elapsed_time(d[name + ":end"])
                n += 1
        return tot / n if n else None


def init_dist(args):
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    backend = os.environ.get("MCD_DIST_BACKEND", "nccl")   # nccl == RCCL; "gloo" only to rehearse ranks on one GPU
    if world != args.gpus:
        # never benchmark fewer ranks than asked for under the asked-for name
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d -- the two must agree (plain `python bench.py --gpus N` starts "
                         "its own N ranks; under torch.distributed.run pass --nproc-per-node N)" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the product path has no CPU fallback)")
    n_dev = torch.cuda.device_count()
    if world > 1 and backend == "nccl" and n_dev < world:
        raise SystemExit("bench.py: %d RCCL ranks need %d GPUs, this node shows %d (MCD_DIST_BACKEND=gloo rehearses ranks on "
                         "fewer GPUs, tests only)" % (world, world, n_dev))
    # one process per GPU; the modulo only matters when ranks are rehearsed on fewer GPUs than ranks (gloo, tests)
    local_rank %= max(1, n_dev)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
    gather = None
    if world > 1 and backend != "nccl":
        # rehearsal of several ranks on ONE GPU (tests): RCCL cannot put two ranks on one device, so the ranks meet over
        # gloo and the payload is staged through the host.  The package itself only ships the RCCL transport.
        def gather(t):
            host = torch.empty((world * t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype)
            dist.all_gather_into_tensor(host, t.contiguous().cpu())
            return host.to(t.device)
    return world, rank, dev, backend, gather


def make_barrier(world):
    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
    return barrier


def max_over_ranks(elapsed, world, dev, backend):
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    return elapsed


STRESS_KERNEL_NAMES = dict(KERNEL_NAMES, gemm="K1s normalize + bf16 conversion + gemm_nt_bf16_exp_v6_kernel + rowsum_finish",
                           softmax="(fused into K1s)", wpmi="K4s wpmi_score_bf16 (wpmi_bf16_kernel<soft>, v_log_f32)")


def core_roofline(stage_ms, N_total, N_l, C, widths, K, world, traffic_file, traffic_ok, s_bytes=4, note=None, names=KERNEL_NAMES,
                  k4_ms=None, valu_bound=False):
    """the dominant kernel of the dissection core is K4 (the only stage that is ONE kernel launch and the
    one with the most GPU time at every shape measured); its duration comes from HIP events placed directly around its
    launch.  The other stages' times (stage_ms) span several launches plus the host gaps between them.

    valu_bound (the fp32 parity chain): K4's HBM traffic equals its algorithmic bytes and removing its LDS bank conflicts buys
    1.7 % (profiles/r03_k4_lds_ablation.txt) -- what it runs on is vector-instruction issue for U*K*C correctly rounded logs.
    `frac` is then measured against THAT roofline: logs per second at the FLOOR of K4_FLOOR lane-instructions per log (a paper
    count, independent of the kernel) and one wave-instruction per 4 cycles and SIMD; the HBM figures stay as hbm_*."""
    dom = "wpmi" if k4_ms else max(stage_ms, key=lambda s: stage_ms[s])
    w = algorithmic_work(dom, N_total, N_l, C, 512, widths, K, world, s_bytes)
    ms = k4_ms if k4_ms else stage_ms[dom]
    achieved = w["bytes"] / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
    traffic, src = None, None
    if traffic_ok:
        try:
            pmc = json.load(open(os.path.join(ROOT, "profiles", traffic_file)))
            traffic = pmc.get(dom, {}).get("hbm_bytes")
            src = "profiles/" + traffic_file + " (PMC passes FETCH_SIZE x2 + WRITE_SIZE of this shape on this tree; not counted live)"
        except (OSError, ValueError):
            pass
    r = {"kernel": names[dom], "bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": src,
         "algorithmic_bytes": w["bytes"], "avg_launch_ms": round(ms, 4)}
    if valu_bound and dom == "wpmi" and ms > 0:
        # `frac` against a bound that does not depend on the kernel: the VALU issue rate over the FLOOR of lane-instructions per
        # log; how many the kernel really issues, and how busy that keeps the issue ports, comes from the PMC file of
        # this tree and is attached only at the shape it was counted on
        logs = float(sum(widths)) * K * C / max(world, 1)
        peak = VALU_PEAK_LANE_INSTR_S / K4_FLOOR / 1e9
        ach = logs / (ms * 1e-3) / 1e9
        r.update({"bound": "valu", "achieved": round(ach, 1), "peak": round(peak, 1), "unit": "Glog/s", "frac": round(ach / peak, 4),
                  "algorithmic_logs": logs, "valu_instr_per_log_floor": K4_FLOOR, "floor_breakdown": K4_FLOOR_PER_LOG,
                  "valu_peak": "256 CUs x 4 SIMDs x 2.4 GHz / 4 cycles per wave-instruction x 64 lanes = %.3g lane-instructions/s" % VALU_PEAK_LANE_INSTR_S,
                  "hbm_achieved_gbs": round(achieved, 1), "hbm_frac": round(achieved / HBM_PEAK_GBS, 4)})
        if traffic_ok:
            try:
                pmc = json.load(open(os.path.join(ROOT, "profiles", K4_PMC)))
                if pmc.get("logs") == logs:
                    per_log = pmc["SQ_INSTS_VALU"] * 64.0 / logs
                    r.update({"valu_instr_per_log": round(per_log, 2), "valu_issue_util": round(ach * 1e9 * per_log / VALU_PEAK_LANE_INSTR_S, 4),
                              "pmc_source": "profiles/" + K4_PMC + " (SQ_INSTS_VALU per launch, rocprofv3 --pmc on this tree)"})
            except (OSError, ValueError, KeyError):
                pass
    if note:
        r["note"] = note
    return r


def roofline_core(timer, stage_ms, N_total, N_l, C, widths, K, world, s_bytes=4, stress=False):
    """Every kernel of the dissection core as it ran IN THE TIMED PASSES: HIP events directly around each launch
    (pipeline.Dissector.finish's "<stage>:begin" / ":end" marks, mean over the steps), the algorithmic work of that launch
     and the fraction of the roofline that bounds it.  These are the figures DESIGN.md's table leads with;
    isolated warm re-runs of a kernel (caches hot, nothing in front) read better and are labelled as such where quoted."""
    out = {}
    for key, stage in (("K1", "gemm"), ("K2", "softmax"), ("K3", "topk"), ("K4", "wpmi"), ("K5", "logsumexp"), ("K6", "row_topk")):
        ms = timer.kernel_ms(stage)
        if not ms or ms <= 0:
            continue
        w = algorithmic_work(stage, N_total, N_l, C, 512, widths, K, world, s_bytes)
        e = {"ms": round(ms, 4)}
        if stage == "gemm":
            peak = BF16_MFMA_PEAK_TF if stress else F32_MFMA_PEAK_TF
            tf = w["flops"] / (ms * 1e-3) / 1e12
            e.update({"algorithmic_flops": w["flops"], "bound": "mfma", "achieved_tflops": round(tf, 1), "peak_tflops": peak,
                      "frac": round(tf / peak, 4)})
            if stress:
                e["what"] = "the whole mcd_embed_gemm_exp call (conversion + GEMM kernel + row-sum finish); the kernel alone: gemm_stress"
        else:
            gbs = w["bytes"] / (ms * 1e-3) / 1e9
            e.update({"algorithmic_bytes": w["bytes"], "bound": "hbm", "achieved_gbs": round(gbs, 1), "peak_gbs": HBM_PEAK_GBS,
                      "frac": round(gbs / HBM_PEAK_GBS, 4)})
            if stage == "wpmi" and not stress:
                logs = float(sum(widths)) * K * C / max(world, 1)
                peak = VALU_PEAK_LANE_INSTR_S / K4_FLOOR / 1e9
                e.update({"bound": "valu", "algorithmic_logs": logs, "achieved_glogs": round(logs / (ms * 1e-3) / 1e9, 1),
                          "peak_glogs": round(peak, 1), "frac": round(logs / (ms * 1e-3) / 1e9 / peak, 4), "hbm_frac": round(gbs / HBM_PEAK_GBS, 4)})
        out[key] = e
    out["how"] = ("one launch each, between HIP events recorded on the launch stream inside the timed passes (mean over the steps); "
                  "frac = algorithmic work / ms / the peak of the named bound (hbm: 8 TB/s; K1: the fp32 (parity chain) or bf16 (stress "
                  "chain) MFMA peak; K4 of the parity chain: the VALU floor of `roofline`)")
    return out


# ======================================================================================================================
# headline: the drop-in driver
# ======================================================================================================================
def run_headline(args):
    world, rank, dev, backend, gather = init_dist(args)
    import mammo_clip_dissect_amd  # noqa: F401  (raises if libmcd_hip.so is missing)
    from mammo_clip_dissect_amd import core, pipeline
    from mammo_clip_dissect_amd.concept_vit import data_utils, describe_broad_neurons, utils
    from mammo_clip_dissect_amd.pipeline import shard_bounds, sync_encoder_gemm_picks

    torch.backends.cuda.matmul.allow_tf32 = False
    if not args.no_tunableop:
        from mammo_clip_dissect_amd.tuning import enable_gemm_tuning
        enable_gemm_tuning(tune=args.tune)
    else:
        os.environ["MCD_NO_TUNABLEOP"] = "1"
    if args.no_activation_cache:
        os.environ["MCD_ACTIVATION_CACHE"] = "0"
    strong = args.global_images is not None
    N_total = args.global_images if strong else args.images * world
    B = args.batch or (125 if args.target == "breastclip" else 2500)
    if args.align_shards:
        os.environ["MCD_SHARD_ALIGN"] = str(B)
    lo, hi = shard_bounds(N_total, world, rank)
    N_l = hi - lo
    with open(CONCEPTS) as f:
        words = f.read().split("\n")
    C = len(words)

    # ---- built once, outside the step: models (random init, seed 0: no checkpoints offline) and the resident probe set
    clip_model, target_model = utils.build_mammo_models(args.target, dev)
    if args.image_size != 224:
        raise SystemExit("the offline towers are built for 224 x 224 inputs")
    if args.target == "breastclip":      # not the headline: the EfficientNet-B5 shape of configs[3] (39 MBConv blocks)
        blocks = target_model.image_encoder._blocks
        layer_names = ["image_encoder._blocks[%d]" % i for i in range(len(blocks))]
        widths = []
        hs = [b.register_forward_hook(lambda m, i, o: widths.append(int(o.shape[1]))) for b in blocks]
        with torch.no_grad():
            target_model.encode_image(torch.zeros(1, 3, args.image_size, args.image_size, device=dev))
        for h in hs:
            h.remove()
    else:
        blocks = target_model.image_encoder.encoder.layer
        layer_names = ["image_encoder.encoder.layer[%d]" % i for i in range(len(blocks))]
        widths = [768] * len(blocks)
    d_probe = "synthetic_%d_%d" % (N_total, args.image_size)
    data = data_utils.get_data(d_probe, None, dev, lo, hi)
    images = data.images()                        # generated on the device, resident in HBM from here on
    prebuilt = {"clip_model": clip_model, "target_model": target_model, "data": data, "gather": gather}

    # One-time setup outside any step (like building the model): the first call of every GEMM shape makes
    # libmcd_blaslt.so time its hipBLASLt candidates, so push one batch (and the shorter last one) and the concept set
    # through the towers here; then every rank takes rank 0's picks (same algorithm => same bits on every rank).
    with torch.no_grad():
        if N_l > 0:
            clip_model.image_projection(clip_model.encode_image(images[:B]))
            if N_l % B:
                clip_model.encode_image(images[:N_l % B])
        tok = {k: v.to(dev) for k, v in clip_model.tokenize(words).items()}
        for i in range(0, C, B):
            clip_model.text_projection(clip_model.encode_text({k: v[i:i + B] for k, v in tok.items()}))
    torch.cuda.synchronize()
    sync_encoder_gemm_picks()

    work = tempfile.mkdtemp(prefix="mcd_bench_")
    timer = StageTimer()
    pipeline.STAGE_MARK = timer.mark
    step_no = [0]

    def one_step():
        i = step_no[0]
        step_no[0] += 1
        argv = ["--target_model", args.target, "--target_layers", ",".join(layer_names), "--d_probe", d_probe,
                "--concept_set", CONCEPTS, "--batch_size", str(B), "--device", str(dev),
                "--activation_dir", os.path.join(work, "acts_%d_r%d" % (i, rank)), "--result_dir", os.path.join(work, "results"),
                "--top_k", str(args.top_k)]
        return describe_broad_neurons.main(argv, prebuilt=prebuilt)

    barrier = make_barrier(world)
    for _ in range(args.warmup):
        one_step()
    data_utils.ATTENTION_EVENTS = attn_events = []   # K9 launches of the timed steps (every 8th is bracketed)
    core.LINEAR_EVENTS = lin_events = []             # every hipBLASLt call of the timed steps
    timer.on = True
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out_dir = one_step()
    barrier()
    elapsed = max_over_ranks(time.perf_counter() - t0, world, dev, backend)
    timer.on = False
    data_utils.ATTENTION_EVENTS = None
    core.LINEAR_EVENTS = None

    stage_ms = timer.stage_ms()
    core_ms = sum(stage_ms.values())
    value = N_total * args.steps / elapsed
    mode = "strong: ONE probe set of %d images sharded over %d rank(s)" % (N_total, world) if strong else \
        "weak: %d images per GPU" % args.images
    out = {
        "metric": "probe images/sec dissected (763 concepts, all layers)",
        "value": round(value, 2), "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(1000.0 * elapsed / args.steps, 3), "higher_is_better": True,
        "scaling": "strong" if strong else "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "rccl_ranks": dist.get_world_size() if world > 1 else 1, "dist_backend": dist.get_backend() if world > 1 else None,
        "config": {"workload": ("configs[%d]: M-Mammo-CLIP Dissect through the drop-in driver describe_broad_neurons.main(): "
                                "Mammo-CLIP ViT-B/16 target+dissector (random init), %s, synthetic %dx%d images resident in HBM, "
                                "%d concepts, %d layers x 768 neurons, soft_wpmi top_k=%d; CSV + args.txt%s written inside the step"
                                % (2 if strong else 1, mode, args.image_size, args.image_size, C, len(widths), args.top_k,
                                   "" if args.no_activation_cache else " + activation cache files"))
                   if args.target == "breastclip_vit" else
                   "NOT the headline workload: target %s, %s, %d concepts, %d layers / %d neurons"
                   % (args.target, mode, C, len(widths), sum(widths)),
                   "entry_point": "mammo_clip_dissect_amd.concept_vit.describe_broad_neurons.main",
                   "images_per_gpu": N_l if world == 1 else [b - a for a, b in (shard_bounds(N_total, world, r) for r in range(world))],
                   "global_images": N_total, "batch": B, "parallelism": "image-sharded dp%d" % world,
                   "preset": args.preset,
                   "shard_align": int(os.environ.get("MCD_SHARD_ALIGN", "1")),
                   "encoder_bit_identity_across_rank_counts": ("yes: shard boundaries on batch multiples" if int(os.environ.get("MCD_SHARD_ALIGN", "1")) == B
                                                               else "only for equal batch shapes: hipBLASLt's fp32 kernels are all stream-K "
                                                                    "(profiles/r04_blaslt_algos.txt); --align-shards makes it unconditional"),
                   "activation_cache_written": not args.no_activation_cache and world == 1,
                   "encoder_gemm": ("fp32 hipBLASLt; the ViT blocks' four GEMMs and the patch embedding with rank 0's best-of-32 pick "
                                    "per shape, broadcast to every rank (libmcd_blaslt.so); other nn.Linear calls: "
                                    + ("library defaults" if args.no_tunableop else "TunableOp picks (tunableop_gfx950.csv)"))},
        "core_ms": round(core_ms, 4), "core_images_per_s": round(N_total / (core_ms / 1000.0), 1) if core_ms > 0 else None,
        "stage_ms": {k: round(v, 4) for k, v in stage_ms.items()},
    }
    if rank == 0:
        k4_note = ("VALU-bound: U*K*C = %.3g correctly rounded logs per launch; the peak is the VALU issue rate over the floor of %.1f "
                   "lane-instructions per log (floor_breakdown); with conflict-free table lookups the kernel is 1.7 %% faster "
                   "(profiles/r03_k4_lds_ablation.txt); HBM traffic = algorithmic bytes"
                   % (float(sum(widths)) * args.top_k * C / max(world, 1), K4_FLOOR))
        out["roofline"] = core_roofline(stage_ms, N_total, N_l, C, widths, args.top_k, world, TRAFFIC_CORE,
                                        world == 1 and N_l == 10000 and args.target == "breastclip_vit",
                                        note=k4_note, k4_ms=timer.kernel_ms("wpmi"), valu_bound=True)
        out["roofline_core"] = roofline_core(timer, stage_ms, N_total, N_l, C, widths, args.top_k, world)
        launches_per_step = len(blocks) * ((N_l + B - 1) // B)
        if attn_events:
            # K9: algorithmic flops = 4 * T^2 * 64 per head and image (QK^T and PV), per launch B * heads of them
            a_ms = sum(e0.elapsed_time(e1) for e0, e1, _, _, _ in attn_events) / len(attn_events)
            _, _, Ba, Ta, Ha = attn_events[0]
            a_flops = 4.0 * Ba * Ha * Ta * Ta * 64
            k9_traffic = None
            try:
                k9 = json.load(open(os.path.join(ROOT, "profiles", "r01_v10_pmc_traffic_k9.json")))
                if (Ba, Ta, Ha) == (k9.get("B"), 197, 12):
                    k9_traffic = k9.get("hbm_bytes")
            except (OSError, ValueError):
                pass
            tf = a_flops / (a_ms * 1e-3) / 1e12
            out["roofline_encoder"] = {
                "kernel": "K9 vit_attention (vit_attention_kernel, fp32 MFMA, %d images x %d heads x %d tokens per launch, "
                          "%d launches per step)" % (Ba, Ha, Ta, launches_per_step),
                "bound": "mfma", "achieved": round(tf, 2), "peak": F32_MFMA_PEAK_TF, "unit": "TFLOP/s",
                "frac": round(tf / F32_MFMA_PEAK_TF, 4), "traffic": k9_traffic,
                "traffic_source": "profiles/r01_v10_pmc_traffic_k9.json (PMC passes; not counted live)" if k9_traffic else None,
                "algorithmic_flops": a_flops, "avg_launch_ms": round(a_ms, 4), "timed_launches": len(attn_events),
                "share_of_step": round(a_ms * launches_per_step / (1000.0 * elapsed / args.steps), 4),
                "note": "outside SURVEY 8(d)'s K0-K6; the kernel executes (224/197)^2 = 1.29x the algorithmic flops"}
        if lin_events:
            lib_ms = sum(e0.elapsed_time(e1) for e0, e1, _, _, _ in lin_events) / args.steps
            gf = sum(2.0 * M * N * K for _, _, M, N, K in lin_events) / args.steps
            out["library_gemm_share"] = {
                "share_of_step": round(lib_ms / (1000.0 * elapsed / args.steps), 4), "ms_per_step": round(lib_ms, 2),
                "tflops": round(gf / (lib_ms * 1e-3) / 1e12, 1), "peak_f32_mfma": F32_MFMA_PEAK_TF,
                "calls_per_step": len(lin_events) // args.steps,
                "what": "fp32 hipBLASLt GEMMs of the towers issued through libmcd_blaslt.so (qkv, proj+residual, fc1, fc2+residual, "
                        "patch embedding), HIP events around every call inside the timed region: the headline is a library-GEMM "
                        "number; in-tree kernels are the rest"}
        wg = algorithmic_work("gemm", N_total, N_l, C, 512, widths, args.top_k, world)
        # K1 alone: 20 launches back to back between two HIP events on operands of this run's shape (a pair of events around ONE
        # launch of an 80 us kernel also measures the host's launch latency: it read 0.12 ms)
        k1_ms = k1_kernel_ms(core, dev, N_l, C) if N_l > 0 else None
        if k1_ms:
            out["gemm"] = {"kernel": "K1 gemm_nt_f32_dma_kernel (fp32 MFMA, K-tiles by LDS-DMA, MKL's K-block order)", "ms": round(k1_ms, 4),
                           "tflops": round(wg["flops"] / (k1_ms * 1e-3) / 1e12, 2), "peak_f32_mfma": F32_MFMA_PEAK_TF,
                           "frac_of_peak": round(wg["flops"] / (k1_ms * 1e-3) / 1e12 / F32_MFMA_PEAK_TF, 4),
                           "stage_ms_with_host_gaps": round(stage_ms["gemm"], 4),
                           "single_launch_between_events_ms": round(timer.kernel_ms("gemm") or 0.0, 4),
                           "note": "ms / tflops: the GEMM kernel alone (20 launches back to back after the timed region, same shape); "
                                   "single_launch_between_events_ms: one launch inside the timed step, host launch latency included; "
                                   "stage_ms_with_host_gaps: K1a normalize x2 + K1 + the host gaps between the three launches of the "
                                   "driver's un-graphed scoring pass"}
        # north_star's one kernel target, measured in THIS run: the image x text bf16 GEMM with the exp epilogue at
        # one rank's share of configs[4], a few launches behind the timed region, buffers freed afterwards
        if world == 1:
            try:
                out["gemm_stress"] = gemm_stress_probe(dev)
            except Exception as e:        # a report, never a reason to lose the bench line
                out["gemm_stress"] = {"error": str(e)}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args, out_dir, work, clip_model, images, words, widths, N_l)
        print(json.dumps(out), flush=True)
    pipeline.STAGE_MARK = None
    shutil.rmtree(work, ignore_errors=True)
    if world > 1:
        dist.destroy_process_group()


def k1_kernel_ms(core, dev, N, C, D=512, reps=20):
    """mcd_embed_gemm (fp32 MFMA, parity mode) at [N, C, D]: mean of `reps` back-to-back launches into a preallocated P."""
    g = torch.Generator(device=dev).manual_seed(77)
    I = core.normalize_rows(torch.randn(N, D, device=dev, generator=g))
    T = core.normalize_rows(torch.randn(C, D, device=dev, generator=g))
    P = core.embed_gemm(I, T)
    for _ in range(3):
        core.embed_gemm(I, T, out=P)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        core.embed_gemm(I, T, out=P)
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps


def mfma_ceiling(tflops):
    """What a bare v_mfma_f32_16x16x32_bf16 loop on random operands (operands in registers, one wave per SIMD) sustains on this part
    -- scripts/micro/mfma_fill.hip D, measured on this tree's round (profiles/r05_mfma_ceiling.json) -- and `tflops` as a fraction of it."""
    try:
        c = json.load(open(os.path.join(ROOT, "profiles", "r05_mfma_ceiling.json")))
        lo, hi = min(c["tflops_16x16x32_random"]), max(c["tflops_16x16x32_random"])
        return {"measured_mfma_ceiling_tflops": [lo, hi], "frac_of_measured_mfma_ceiling": [round(tflops / hi, 4), round(tflops / lo, 4)],
                "ceiling_source": "profiles/r05_mfma_ceiling.json (scripts/micro/mfma_fill.hip D: bare v_mfma_f32_16x16x32_bf16 loop, random operands)"}
    except (OSError, ValueError, KeyError):
        return {}


def gemm_stress_probe(dev, N=25000, C=10000, D=512, a=10.0, reps=8):
    """mcd_embed_gemm_exp at [N, C, D] on random embeddings: `ms` = the whole call (normalise + bf16 conversion, GEMM kernel, row-sum
    finish) by HIP events on the launch stream, mean over `reps` calls after 2 warm-up calls; `kernel_ms` = the GEMM kernel alone:
    the library launches it `reps` times back to back between its own event pair (mcd_embed_gemm_exp_time_kernel), per launch."""
    from mammo_clip_dissect_amd import core, _lib
    L = _lib.load()
    g = torch.Generator(device=dev).manual_seed(4242)
    I = torch.randn(N, D, device=dev, generator=g)
    T = torch.randn(C, D, device=dev, generator=g)
    for _ in range(2):
        core.embed_gemm_exp(I, T, a, normalize=True)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        core.embed_gemm_exp(I, T, a, normalize=True)
    e.record()
    torch.cuda.synchronize()
    ms = s.elapsed_time(e) / reps
    L.mcd_embed_gemm_exp_time_kernel(reps)         # the GEMM kernel `reps` times back to back between the library's own event pair
    try:
        k_ms = []
        for _ in range(3):
            core.embed_gemm_exp(I, T, a, normalize=True)
            k_ms.append(float(L.mcd_embed_gemm_exp_kernel_ms()))
    finally:
        L.mcd_embed_gemm_exp_time_kernel(0)
    kernel_ms = sorted(k_ms)[1]                    # median of three timed calls
    # ... and the kernel as it runs INSIDE the call (one launch between the library's events, behind the conversion kernel, in a
    # sequence of `reps` whole calls; the last call's pair is read).  Back to back the chip holds a lower clock than with the 15 us
    # of conversion and row-sum kernels between two launches (profiles/r05_gexp_v6.txt (g)): both figures are reported
    L.mcd_embed_gemm_exp_time_kernel(1)
    try:
        kc = []
        for _ in range(3):
            for _ in range(reps):
                core.embed_gemm_exp(I, T, a, normalize=True)
            kc.append(float(L.mcd_embed_gemm_exp_kernel_ms()))
    finally:
        L.mcd_embed_gemm_exp_time_kernel(0)
    kernel_in_call_ms = sorted(kc)[1]
    # the same kernel on a chip that has been idle: the clock it holds under this kernel's sustained load is lower than the one it
    # starts with (profiles/r04_gexp_v4.txt (h), r05_gexp_v6.txt (g)) -- 4 back-to-back launches after 1.5 s without GPU work, reported beside the warm figure
    torch.cuda.synchronize()
    time.sleep(1.5)
    L.mcd_embed_gemm_exp_time_kernel(4)
    try:
        core.embed_gemm_exp(I, T, a, normalize=True)
        idle_ms = float(L.mcd_embed_gemm_exp_kernel_ms())
    finally:
        L.mcd_embed_gemm_exp_time_kernel(0)
    flops = 2.0 * N * C * D
    del I, T
    torch.cuda.empty_cache()
    return {"schema": 5,
            "kernel": "K1s: normalise + bf16 conversion (fragment-major), gemm_nt_bf16_exp_v6_kernel (one wave per SIMD, "
                      "v_mfma_f32_16x16x32_bf16 on named registers, the epilogue inside the next tile's first two k-steps, bf16 out + "
                      "row sums by MFMA), row-sum finish",
            "shape": [N, C, D], "ms": round(ms, 4), "kernel_ms": round(kernel_ms, 4),
            "tflops": round(flops / (ms * 1e-3) / 1e12, 1), "kernel_tflops": round(flops / (kernel_ms * 1e-3) / 1e12, 1),
            "peak": BF16_MFMA_PEAK_TF, "frac_of_peak": round(flops / (ms * 1e-3) / 1e12 / BF16_MFMA_PEAK_TF, 4),
            "kernel_frac_of_peak": round(flops / (kernel_ms * 1e-3) / 1e12 / BF16_MFMA_PEAK_TF, 4),
            "kernel_ms_in_call": round(kernel_in_call_ms, 4),
            "kernel_in_call_frac_of_peak": round(flops / (kernel_in_call_ms * 1e-3) / 1e12 / BF16_MFMA_PEAK_TF, 4),
            "kernel_ms_after_idle": round(idle_ms, 4),
            "kernel_frac_of_peak_after_idle": round(flops / (idle_ms * 1e-3) / 1e12 / BF16_MFMA_PEAK_TF, 4),
            **mfma_ceiling(flops / (kernel_ms * 1e-3) / 1e12),
            "reps": reps, "how": "schema 5 (ADVICE r4): frac_of_peak is the WHOLE entry point's again, as in rounds 2-3 (round 4 had "
                                 "put the kernel's figure under that name and the call's under call_frac_of_peak); kernel_* = the GEMM "
                                 "kernel alone (HIP events inside the library, `reps` launches back to back, warm chip); kernel_ms_in_call: "
                                 "one launch between the library's events inside a sequence of whole calls; *_after_idle: 4 "
                                 "back-to-back launches after 1.5 s of an idle GPU (the clock before it settles under the load), not "
                                 "the headline figure"}


def cpu_baseline(args, out_dir, work, model, images, words, widths, N_l):
    """The same job on this box's host cores, like for like (rank 0, N = 1): the encoder forward of the same tower in
    PyTorch's own CPU kernels (what the reference's CPU run does) on a small sample, plus the CPU oracle's similarity path
    (reference utils.py:566-612 + similarity.py + describe_broad_neurons.py:101-102, P recomputed per layer as the reference
    does) on the activations of the last timed step; value = the end-to-end rate."""
    import glob
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle as O
    ncpu = host_cpu_share()
    O.lib()
    O.set_num_threads(ncpu)          # OpenMP loops of the C oracle
    torch.set_num_threads(ncpu)
    try:
        from threadpoolctl import threadpool_limits
        threadpool_limits(ncpu)      # numpy's BLAS (the reference's torch.matmul is a BLAS call too)
    except Exception:
        pass
    res = {"unit": "images/s", "cores": O.num_threads(), "kind": "port"}
    # ---- similarity path on the last step's cache files (or random stand-ins when the cache was not written) ----
    nl = max(1, min(args.cpu_baseline_layers, len(widths)))
    files = sorted(glob.glob(os.path.join(work, "acts_*", "**", "*.pt"), recursive=True), key=os.path.getmtime)
    g = torch.Generator().manual_seed(0)
    layer_files = [f for f in files if "encoder.layer" in f or "_blocks" in f][-len(widths):][:nl]
    if len(layer_files) == nl:
        A_h = [torch.load(f, weights_only=True).numpy() for f in layer_files]
        E_img_h = torch.load([f for f in files if f.endswith("_ViT-B16.pt") and "Specific" not in f][-1], weights_only=True).numpy()
        E_txt_h = torch.load([f for f in files if "Specific" in f][-1], weights_only=True).numpy()
        src = "the activation cache files of the last timed step"
    else:
        A_h = [torch.randn(N_l, w, generator=g).numpy() for w in widths[:nl]]
        E_img_h, E_txt_h = torch.randn(N_l, 512, generator=g).numpy(), torch.randn(len(words), 512, generator=g).numpy()
        src = "random activations of the same shape (no cache files were written)"
    tc = time.perf_counter()
    for i in range(nl):
        O.dissect_layer(E_img_h, E_txt_h, A_h[i], top_k=args.top_k)
    sim_s = (time.perf_counter() - tc) * len(widths) / nl
    res["similarity_only_images_per_s"] = round(N_l / sim_s, 1)
    # ---- encoder forward on the CPU, small sample ----
    try:
        import copy
        ns = 32
        m_cpu = copy.deepcopy(model).to("cpu").eval()
        for mod in m_cpu.modules():
            mod._forward_hooks.clear()
        x_cpu = images[:ns].cpu()
        with torch.no_grad():
            m_cpu.encode_image(x_cpu[:4])
            tc = time.perf_counter()
            m_cpu.image_projection(m_cpu.encode_image(x_cpu))
            enc_s = time.perf_counter() - tc
        enc_rate = ns / enc_s
        res["encoder_images_per_s"] = round(enc_rate, 1)
        res["value"] = round(1.0 / (1.0 / enc_rate + sim_s / N_l), 1)
        res["sample"] = ("end to end, like for like with `value`: %d images through the same tower on PyTorch CPU (%d threads) + the "
                         "oracle's similarity path (normalise + I.T^T per layer, softmax, top-%d, soft-WPMI, logsumexp, top-10/top-5) on "
                         "%d of %d layers x 768 neurons at N=%d from %s, time scaled x%d/%d; no CSV"
                         % (ns, ncpu, args.top_k, nl, len(widths), N_l, src, len(widths), nl))
        del m_cpu
    except Exception as e:   # the baseline is a report, never a reason to lose the bench line
        res["value"] = res["similarity_only_images_per_s"]
        res["sample"] = "similarity path only (encoder sample skipped: %s)" % (e,)
    return res


# ======================================================================================================================
# --config core / stress: the HIP core alone on random activations (never the headline)
# ======================================================================================================================
def run_core(args):
    world, rank, dev, backend, gather = init_dist(args)
    import mammo_clip_dissect_amd  # noqa: F401
    from mammo_clip_dissect_amd.pipeline import Dissector
    stress = args.config == "stress"
    L, UL = 12, 768
    widths = [UL] * L
    if stress:
        N_l, C, mode, s_bytes = args.stress_images, args.stress_concepts, "bf16", 2
    else:
        with open(CONCEPTS) as f:
            C = len(f.read().split("\n"))
        N_l, mode, s_bytes = args.images, "f32", 4
    N_total = N_l * world
    dis = Dissector(N_l, ["l%d" % i for i in range(L)], widths, C, 512, dev, top_k=args.top_k, gemm_mode=mode, gather=gather)
    g = torch.Generator(device=dev).manual_seed(1234 + rank)
    dis.At.normal_(generator=g)
    dis.E_img.normal_(generator=g)
    dis.cursor = N_l
    E_txt = torch.randn(C, 512, device=dev, generator=g)
    timer = StageTimer()
    barrier = make_barrier(world)
    for _ in range(max(args.warmup, 1)):
        res = dis.finish(E_txt)
    timer.on = True
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        res = dis.finish(E_txt, marks=timer.mark)
    barrier()
    elapsed = max_over_ranks(time.perf_counter() - t0, world, dev, backend)
    assert bool(torch.isfinite(res.sim).all())
    stage_ms = timer.stage_ms()
    core_ms = sum(stage_ms.values())
    out = {
        "metric": "probe images/sec dissected (%d concepts, all layers), CORE ONLY" % C,
        "value": round(N_total * args.steps / elapsed, 1), "unit": "images/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(1000.0 * elapsed / args.steps, 4), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "bf16" if stress else "f32", "data": "synthetic",
        "rccl_ranks": dist.get_world_size() if world > 1 else 1, "dist_backend": dist.get_backend() if world > 1 else None,
        "config": {"workload": ("NOT the headline: the dissection core alone (no encoder forwards, no CSV) on random activations and "
                                "embeddings, %s: %d images per GPU x %d concepts x %d layers x %d neurons, soft_wpmi top_k=%d, %s chain"
                                % ("one rank's share of configs[4] (200 000 images / 8 GPUs)" if stress else "configs[1]'s shape",
                                   N_l, C, L, UL, args.top_k,
                                   "bf16 (MFMA GEMM with the exp fused into its epilogue, bf16 similarity matrix, v_log_f32 log; no "
                                   "parity claim)" if stress else "fp32 parity")),
                   "images_per_gpu": N_l, "global_images": N_total, "parallelism": "image-sharded dp%d" % world, "core_only": True},
        "core_ms": round(core_ms, 4), "stage_ms": {k: round(v, 4) for k, v in stage_ms.items()},
    }
    if rank == 0:
        out["roofline"] = core_roofline(stage_ms, N_total, N_l, C, widths, args.top_k, world,
                                        TRAFFIC_STRESS if stress else TRAFFIC_CORE,
                                        world == 1 and ((N_l == 25000 and C == 10000) if stress else N_l == 10000), s_bytes,
                                        note=("algorithmic bytes count every touched row of E once per layer; the kernel gathers U*K rows "
                                              "of %d bytes out of L2 (63 %% of the gathers' lines) and the Infinity Cache; the gather pattern alone, "
                                              "without arithmetic, runs at 18 TB/s (profiles/r03_k4s_pmc.txt, "
                                              "profiles/r03_gather_path.txt)" % (2 * C))
                                        if stress else None, names=STRESS_KERNEL_NAMES if stress else KERNEL_NAMES,
                                        k4_ms=timer.kernel_ms("wpmi"), valu_bound=not stress)
        out["roofline_core"] = roofline_core(timer, stage_ms, N_total, N_l, C, widths, args.top_k, world, s_bytes, stress)
        if stress and timer.kernel_ms("wpmi"):
            gathered = 2.0 * C * sum(widths) * args.top_k / max(world, 1)     # U*K rows of C bf16 values
            out["roofline"]["gathered_bytes"] = gathered
            out["roofline"]["gathered_over_algorithmic"] = round(gathered / out["roofline"]["algorithmic_bytes"], 2)
            out["roofline"]["gathered_gbs"] = round(gathered / (timer.kernel_ms("wpmi") * 1e-3) / 1e9, 1)
        wg = algorithmic_work("gemm", N_total, N_l, C, 512, widths, args.top_k, world, s_bytes)
        g_ms = timer.kernel_ms("gemm") or stage_ms["gemm"]       # the call alone (marks around it) / the stage
        if g_ms > 0:
            tf = wg["flops"] / (g_ms * 1e-3) / 1e12
            if stress:
                probe = gemm_stress_probe(dev, N_l, C)           # same call, plus the GEMM kernel alone (library event pair)
                probe["ms_in_pass"] = round(g_ms, 4)
                probe["algorithmic_flops"], probe["algorithmic_bytes"] = wg["flops"], wg["bytes"]
                if world == 1 and N_l == 25000 and C == 10000:   # counters of this tree at exactly this shape
                    try:
                        probe.update(json.load(open(os.path.join(ROOT, "profiles", GEXP_PMC))))
                    except (OSError, ValueError):
                        pass
                probe["library_yardstick"] = library_bf16_gemm_yardstick(N_l, C, dev, 2.0 * N_l * C * 512)
                out["gemm_stress"] = probe
            else:
                out["gemm"] = {"kernel": "K1 gemm_nt_f32_dma_kernel (fp32 MFMA)", "shape": [N_l, C, 512], "ms": round(g_ms, 4),
                               "tflops": round(tf, 1), "peak": F32_MFMA_PEAK_TF, "frac_of_peak": round(tf / F32_MFMA_PEAK_TF, 4),
                               "stage_ms_with_host_gaps": round(stage_ms["gemm"], 4),
                               "algorithmic_flops": wg["flops"], "algorithmic_bytes": wg["bytes"]}
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


def library_bf16_gemm_yardstick(N, C, dev, flops, reps=10):
    """The vendor library's PLAIN bf16 GEMM of the same shape, timed live beside the stress line (outside the timed steps):
    bf16 operands already converted, bf16 result, no exp, no row sums -- torch.matmul = hipBLASLt / rocBLAS, both operand
    orders, plus the same output at 8 x the reduction depth (where the library's epilogue no longer weighs).  A yardstick for
    `frac_of_peak`, not a product path: nothing in the package calls it."""
    def timed(fn):
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(reps):
            fn()
        e.record()
        torch.cuda.synchronize()
        return s.elapsed_time(e) / reps
    try:
        g = torch.Generator(device=dev).manual_seed(99)
        Ib = torch.nn.functional.normalize(torch.randn(N, 512, device=dev, generator=g)).to(torch.bfloat16)
        Tb = torch.nn.functional.normalize(torch.randn(C, 512, device=dev, generator=g)).to(torch.bfloat16)
        o1 = torch.empty(N, C, device=dev, dtype=torch.bfloat16)
        o2 = torch.empty(C, N, device=dev, dtype=torch.bfloat16)
        t1 = timed(lambda: torch.matmul(Ib, Tb.t(), out=o1))
        t2 = timed(lambda: torch.matmul(Tb, Ib.t(), out=o2))
        Ik = torch.randn(N, 4096, device=dev, dtype=torch.bfloat16, generator=g)
        Tk = torch.randn(C, 4096, device=dev, dtype=torch.bfloat16, generator=g)
        t3 = timed(lambda: torch.matmul(Ik, Tk.t(), out=o1))
        best = min(t1, t2)
        return {"what": "torch.matmul (hipBLASLt / rocBLAS) bf16 x bf16 -> bf16 at the same [N, C, 512]: the plain GEMM only",
                "ms_images_by_concepts": round(t1, 4), "ms_concepts_by_images": round(t2, 4),
                "tflops": round(flops / (best * 1e-3) / 1e12, 1),
                "frac_of_peak": round(flops / (best * 1e-3) / 1e12 / BF16_MFMA_PEAK_TF, 4),
                "k4096_ms": round(t3, 4), "k4096_tflops": round(8 * flops / (t3 * 1e-3) / 1e12, 1),
                "k4096_frac_of_peak": round(8 * flops / (t3 * 1e-3) / 1e12 / BF16_MFMA_PEAK_TF, 4)}
    except Exception as e:   # a yardstick, never a reason to lose the bench line
        return {"error": str(e)}


def launch_ranks(args, argv):
    """`python bench.py --gpus N` started plainly (no WORLD_SIZE in the environment): start the N ranks as CHILD processes
    -- `python -m torch.distributed.run --nproc-per-node N bench.py <same flags>` -- before this process has made any GPU
    call (importing torch does not initialise HIP), relay rank 0's ONE JSON line and exit with the launcher's code.
    MCD_BENCH_NO_LAUNCH=1 turns the launch off; the call then fails instead of measuring one GPU under the name of N."""
    import socket
    import subprocess
    if os.environ.get("MCD_BENCH_NO_LAUNCH") == "1":
        sys.stderr.write("bench.py: --gpus %d without a torch.distributed.run environment and MCD_BENCH_NO_LAUNCH=1: refusing to "
                         "run one rank under the name of %d\n" % (args.gpus, args.gpus))
        return 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"), MCD_BENCH_CHILD="1")
    env.setdefault("OMP_NUM_THREADS", str(max(1, host_cpu_share() // args.gpus)))
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)   # stderr passes through
    lines = []
    for line in proc.stdout:
        if line.startswith("{") and '"metric"' in line:
            lines.append(line.rstrip("\n"))
        else:
            sys.stderr.write(line)
    rc = proc.wait()
    if rc != 0:
        sys.stderr.write("bench.py: the %d-rank launch exited with code %d\n" % (args.gpus, rc))
        return rc
    if len(lines) != 1:
        sys.stderr.write("bench.py: expected ONE JSON line from rank 0, got %d\n" % len(lines))
        return 3
    if json.loads(lines[0]).get("n_gpus") != args.gpus:
        sys.stderr.write("bench.py: the ranks report n_gpus=%r, asked for %d\n" % (json.loads(lines[0]).get("n_gpus"), args.gpus))
        return 4
    print(lines[0], flush=True)
    return 0


if __name__ == "__main__":
    _a = parse()
    if _a.gpus < 1:
        raise SystemExit("bench.py: --gpus must be >= 1")
    if _a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(_a, sys.argv[1:]))
    if _a.config == "headline":
        run_headline(_a)
    else:
        run_core(_a)
