#!/usr/bin/env python3
"""bench.py -- probe images/sec dissected (763 concepts, all layers), BASELINE.json configs[1]:
M-Mammo-CLIP Dissect, Mammo-CLIP ViT-B/16 target + dissector, 10k synthetic 224x224 mammograms per GPU,
763 concepts, all 12 transformer blocks (12 x 768 neurons), soft-WPMI.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One step = one full pass of the hot path over the probe set, inputs resident in HBM:
  encoder forward over every image with the K0 hooks writing the activation matrix (PyTorch-ROCm fp32,
  single pass: target == dissector), text tower over the 763 concepts, then the HIP core
  (K1 GEMM, K2 softmax, K3 top-K images, K4 soft-WPMI, K5 logsumexp, K6 top-10), then rank 0 writes
  the reference-format CSV.  Weak scaling: every rank holds `--images` images (default 10000); the
  collectives are the three all-gathers of SURVEY.md 8e.

The JSON line carries `roofline` for the hand-written kernel with the most GPU time in the timed region -- K9, the
encoder's fp32 attention (MFMA-bound), unless the core's slowest kernel outweighs it (--core-only) --
`roofline_core` for the slowest kernel of the dissection core (K4, HBM roofline), both timed live with HIP events on
the launch stream inside the timed region, and `cpu_baseline` (the CPU oracle's similarity path on this box's host
cores, rank 0, N=1 only).
"""
import argparse
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch
import torch.distributed as dist

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8 TB/s
F32_MFMA_PEAK_TF = 157.3   # v_mfma_f32_32x32x2_f32


def core_lr_available():
    from mammo_clip_dissect_amd import core
    return core.linear_residual_available()


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--images", type=int, default=10000, help="probe images PER GPU")
    ap.add_argument("--batch", type=int, default=None,
                    help="images per encoder forward (the reference hard-codes 20 / 50, utils.py:84,:297); with the per-shape "
                         "hipBLASLt picks larger batches run the GEMMs faster: 250 -> 3700, 1000 -> 3785, 2500 -> 3820 images/s.  "
                         "Default: 2500 for the ViT target, 125 for the EfficientNet one (its activations are 20x larger)")
    ap.add_argument("--image-size", type=int, default=224)
    ap.add_argument("--target", default="breastclip_vit")
    ap.add_argument("--top-k", type=int, default=100)
    ap.add_argument("--cpu-baseline-layers", type=int, default=12, help="layers the CPU oracle is timed on (all 12: ~1.5 s on 16 cores)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--core-only", action="store_true", help="dev: skip forwards/CSV, time the HIP core alone")
    ap.add_argument("--no-tunableop", action="store_true", help="encoder GEMMs on the libraries' default solutions")
    ap.add_argument("--tune", action="store_true", help="let TunableOp tune unseen GEMM shapes (writes tunableop_results*.csv)")
    return ap.parse_args()


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


def algorithmic_work(stage, N_total, N_local, C, D, widths, K, world):
    """ALGORITHMIC bytes (or flops) of one launch of each core kernel on one rank."""
    U = sum(widths)
    U_rank = (U + world - 1) // world
    if stage == "gemm":      # K1a + K1: 2*N*C*D flop; 4(ND + CD + NC) bytes
        return dict(flops=2.0 * N_local * C * D, bytes=4.0 * (N_local * D + C * D + N_local * C))
    if stage == "softmax":   # K2: read P, write S
        return dict(bytes=8.0 * N_local * C)
    if stage == "topk":      # K3: one read of the activations + (value,index) out
        return dict(bytes=4.0 * N_local * U + 8.0 * K * U)
    if stage == "wpmi":      # K4: every touched row of S once per layer + indices + output
        share = U_rank / float(U)
        rows = sum(min(N_total, w * K) for w in widths) * share
        return dict(bytes=4.0 * C * rows + 4.0 * K * U_rank + 4.0 * U_rank * C)
    if stage == "logsumexp":  # K5: read pdge, write sim
        return dict(bytes=8.0 * U * C)
    if stage == "row_topk":  # K6
        return dict(bytes=4.0 * U * C)
    raise KeyError(stage)


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the product path has no CPU fallback)")
    # one process per GPU; the modulo only matters when ranks are rehearsed on fewer GPUs than ranks (tests)
    local_rank %= max(1, torch.cuda.device_count())
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    backend = os.environ.get("MCD_DIST_BACKEND", "nccl")   # nccl == RCCL; "gloo" only to rehearse ranks on one GPU
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
    assert world == args.gpus or world == 1, "launch with torch.distributed.run for --gpus > 1"

    import mammo_clip_dissect_amd  # noqa: F401  (raises if libmcd_hip.so is missing)
    from mammo_clip_dissect_amd.concept_vit import data_utils
    from mammo_clip_dissect_amd.pipeline import Dissector, write_descriptions_csv

    torch.backends.cuda.matmul.allow_tf32 = False
    if not args.no_tunableop:
        from mammo_clip_dissect_amd.tuning import enable_gemm_tuning
        enable_gemm_tuning(tune=args.tune)
    N_l, B = args.images, args.batch or (125 if args.target == "breastclip" else 2500)
    with open(os.path.join(ROOT, "mammo-clip-dissect_amd", "Concepts", "Specific_concepts_sorted.txt")) as f:
        words = f.read().split("\n")
    C = len(words)

    # ---- model (random init, seed 0: no checkpoints offline), hooks, resident inputs ----------------
    model, _ = data_utils.get_target_model(args.target, dev, seed=0)
    if args.target == "breastclip":      # not the headline: the EfficientNet-B5 shape of configs[3] (39 MBConv blocks)
        blocks = model.image_encoder._blocks
        layer_names = ["image_encoder._blocks[%d]" % i for i in range(len(blocks))]
        widths = []
        hs = [b.register_forward_hook(lambda m, i, o: widths.append(int(o.shape[1]))) for b in blocks]
        with torch.no_grad():
            model.encode_image(torch.zeros(1, 3, args.image_size, args.image_size, device=dev))
        for h in hs:
            h.remove()
    else:
        blocks = model.image_encoder.encoder.layer
        layer_names = ["image_encoder.encoder.layer[%d]" % i for i in range(len(blocks))]
        widths = [768] * len(blocks)
    gather = None
    if world > 1 and backend != "nccl":
        # rehearsal of several ranks on ONE GPU (tests): RCCL cannot put two ranks on one device, so the ranks meet over
        # gloo and the payload is staged through the host.  The package itself only ships the RCCL transport.
        def gather(t):
            host = torch.empty((world * t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype)
            dist.all_gather_into_tensor(host, t.contiguous().cpu())
            return host.to(t.device)
    dis = Dissector(N_l, layer_names, widths, C, 512, dev, top_k=args.top_k, gather=gather)
    handles = [blk.register_forward_hook(dis.hook(i)) for i, blk in enumerate(blocks)]
    tokens = {k: v.to(dev) for k, v in model.tokenize(words).items()}
    g = torch.Generator(device=dev).manual_seed(1234 + rank)
    images = None
    if not args.core_only:
        images = torch.empty((N_l, 3, args.image_size, args.image_size), dtype=torch.float32, device=dev)
        for i in range(0, N_l, 1000):
            images[i:i + 1000].normal_(generator=g)
    else:
        dis.At.normal_(generator=g)
        dis.E_img.normal_(generator=g)
    out_dir = tempfile.mkdtemp(prefix="mcd_bench_")

    stage_names = ["gemm", "softmax", "topk", "wpmi", "logsumexp", "row_topk"]
    events = []   # per timed step: list of (name, event)

    def one_step(record):
        marks = []

        def mark(name):
            if record:
                e = torch.cuda.Event(enable_timing=True)
                e.record()   # torch's current stream = the stream libmcd_hip.so launches on
                marks.append((name, e))
        t0 = time.perf_counter()
        with torch.no_grad():
            if not args.core_only:
                dis.reset()
                for i in range(0, N_l, B):
                    x = images[i:i + B]
                    feats = model.encode_image(x)                      # hooks fire: K0 -> At
                    dis.add_image_features(model.image_projection(feats))
                    dis.advance(x.shape[0])
                E_txt = model.text_projection(model.encode_text(tokens))
            else:
                dis.cursor = N_l
                E_txt = torch.randn(C, 512, device=dev, generator=g)
            res = dis.finish(E_txt, marks=mark)
        csv_s = 0.0
        if rank == 0 and not args.core_only:
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            write_descriptions_csv(res, words, os.path.join(out_dir, "descriptions.csv"), "og")
            csv_s = time.perf_counter() - t1
        if record:
            events.append(marks)
        return res, E_txt, time.perf_counter() - t0, csv_s

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # One-time setup outside any step (like building the model): the first call of every GEMM shape makes
    # libmcd_blaslt.so time its hipBLASLt candidates, so push one batch and the concept set through the towers here --
    # with --warmup 0 that selection would otherwise land in the timed region.
    if not args.core_only:
        with torch.no_grad():
            dis.reset()
            model.image_projection(model.encode_image(images[:B]))
            model.text_projection(model.encode_text(tokens))
            if N_l % B:
                model.encode_image(images[:N_l % B])       # the last, shorter batch is a GEMM shape of its own
            dis.reset()
        torch.cuda.synchronize()
    for _ in range(args.warmup):
        one_step(False)
    data_utils.ATTENTION_EVENTS = attn_events = []   # K9 launches of the timed steps (every 8th is bracketed)
    barrier()
    t0 = time.perf_counter()
    csv_total = 0.0
    for _ in range(args.steps):
        res, E_txt, _, csv_s = one_step(True)
        csv_total += csv_s
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- per-kernel durations of the core (HIP events recorded inside the timed region) --------------
    stage_ms = {s: 0.0 for s in stage_names}
    for marks in events:
        prev = None
        for name, e in marks:
            if prev is not None and name in stage_ms:
                stage_ms[name] += prev.elapsed_time(e)
            prev = e
    for s in stage_ms:
        stage_ms[s] /= max(len(events), 1)
    core_ms = sum(stage_ms.values())

    N_total = N_l * world
    value = N_total * args.steps / elapsed
    out = {
        "metric": "probe images/sec dissected (763 concepts, all layers)",
        "value": round(value, 2), "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(1000.0 * elapsed / args.steps, 3), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "configs[1]: M-Mammo-CLIP Dissect, Mammo-CLIP ViT-B/16 target+dissector (random init), "
                               "%d synthetic %dx%d images per GPU, %d concepts, %d layers x 768 neurons, soft_wpmi top_k=%d"
                               % (N_l, args.image_size, args.image_size, C, len(widths), args.top_k)
                               if args.target == "breastclip_vit" else
                               "NOT the headline workload: target %s, %d images per GPU, %d concepts, %d layers / %d neurons"
                               % (args.target, N_l, C, len(widths), sum(widths)),
                   "images_per_gpu": N_l, "global_images": N_total, "batch": B, "parallelism": "image-sharded dp%d" % world,
                   "encoder_gemm": ("fp32 hipBLASLt; the ViT blocks' four GEMMs and the patch embedding with this process's best-of-32 pick per "
                                    "shape (libmcd_blaslt.so); other nn.Linear calls: " + ("library defaults" if args.no_tunableop else "TunableOp picks (tunableop_gfx950.csv)")),
                   "core_only": bool(args.core_only)},
        "core_ms": round(core_ms, 4), "core_images_per_s": round(N_total / (core_ms / 1000.0), 1) if core_ms > 0 else None,
        "csv_ms": round(1000.0 * csv_total / args.steps, 2),
        "stage_ms": {k: round(v, 4) for k, v in stage_ms.items()},
    }
    # the two residual GEMMs of a ViT block (x + proj(.), x + fc2(.)): fused hipBLASLt calls or PyTorch's linear + add
    out["config"]["encoder_residual"] = "nn.Linear + add (PyTorch)"
    if args.target == "breastclip_vit" and not args.core_only and data_utils.FUSED_RESIDUAL and core_lr_available():
        import ctypes
        from mammo_clip_dissect_amd import _lib as _l
        info = {}
        for name, (n_, k_) in {"qkv": (2304, 768), "proj": (768, 768), "fc1": (3072, 768), "fc2": (768, 3072)}.items():
            ms_, tried_ = ctypes.c_float(0), ctypes.c_int(0)
            _l.load_blaslt().mcd_linear_residual_plan_info(B * 197, n_, k_, ctypes.byref(ms_), ctypes.byref(tried_))
            info[name] = "%.3f ms (best of %d hipBLASLt candidates)" % (ms_.value, tried_.value)
        out["config"]["encoder_residual"] = ("one hipBLASLt GEMM with bias + beta*C epilogue (libmcd_blaslt.so): proj %s, fc2 %s; "
                                             "qkv %s and fc1 %s through the same library (bias epilogue only)"
                                             % (info["proj"], info["fc2"], info["qkv"], info["fc1"]))
    if rank == 0:
        # roofline of the slowest hand-written kernel
        dom = max(stage_ms, key=lambda s: stage_ms[s])
        w = algorithmic_work(dom, N_total, N_l, C, 512, widths, args.top_k, world)
        ms = stage_ms[dom]
        achieved = w["bytes"] / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
        traffic = None   # HBM bytes per launch from the PMC passes committed under profiles/ (config-2 shape only)
        try:
            pmc = json.load(open(os.path.join(ROOT, "profiles", "r01_v9_pmc_traffic.json")))
            if world == 1 and N_l == 10000 and not args.core_only or args.core_only and N_l == 10000:
                traffic = pmc.get(dom, {}).get("hbm_bytes")
        except (OSError, ValueError):
            pass
        core_roofline = {"kernel": {"gemm": "K1 normalize+embed_gemm", "softmax": "K2 row_softmax",
                                      "topk": "K3 col_topk (neuron_topk_fast_kernel)", "wpmi": "K4 wpmi_score (wpmi_slice_kernel<soft, accurate log, S_IS_PROB>)",
                                      "logsumexp": "K5 logsumexp_sub", "row_topk": "K6 row_topk"}[dom],
                           "bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                           "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                           "algorithmic_bytes": w["bytes"], "avg_launch_ms": round(ms, 4)}
        if dom == "wpmi":
            # what actually bounds K4 (PMC, profiles/r01_v5_k4_pmc_sq.txt): the correctly rounded logs, not bytes
            core_roofline["note"] = ("VALU/LDS-bound: U*K*C = %.3g accurate logs per launch, 53 VALU instructions per 6; "
                                       "PMC at this shape: VALU issue 68 %% and LDS 64 %% of the %.2f ms at ~2.0 GHz, HBM traffic = "
                                       "algorithmic bytes" % (float(sum(widths)) * args.top_k * C / max(world, 1), ms))
        out["roofline"] = core_roofline
        if attn_events:
            # K9: algorithmic flops = 4 * T^2 * 64 per head and image (QK^T and PV), per launch B * heads of them
            a_ms = sum(e0.elapsed_time(e1) for e0, e1, _, _, _ in attn_events) / len(attn_events)
            _, _, Ba, Ta, Ha = attn_events[0]
            a_flops = 4.0 * Ba * Ha * Ta * Ta * 64
            launches_per_step = len(blocks) * ((N_l + B - 1) // B)
            if a_ms * launches_per_step > ms:   # more GPU time in the step than the core's slowest kernel
                k9_traffic = None
                try:
                    k9 = json.load(open(os.path.join(ROOT, "profiles", "r01_v10_pmc_traffic_k9.json")))
                    if (Ba, Ta, Ha) == (k9.get("B"), 197, 12):
                        k9_traffic = k9.get("hbm_bytes")
                except (OSError, ValueError):
                    pass
                tf = a_flops / (a_ms * 1e-3) / 1e12
                out["roofline"] = {"kernel": "K9 vit_attention (vit_attention_kernel, fp32 MFMA, %d images x %d heads x %d tokens "
                                             "per launch, %d launches per step)" % (Ba, Ha, Ta, launches_per_step),
                                   "bound": "mfma", "achieved": round(tf, 2), "peak": F32_MFMA_PEAK_TF, "unit": "TFLOP/s",
                                   "frac": round(tf / F32_MFMA_PEAK_TF, 4), "traffic": k9_traffic,
                                   "algorithmic_flops": a_flops, "algorithmic_bytes": 16.0 * Ba * Ta * Ha * 64,
                                   "avg_launch_ms": round(a_ms, 4), "timed_launches": len(attn_events),
                                   "note": "dtype f32: peak = dense v_mfma_f32_32x32x2_f32 rate; the kernel executes "
                                           "(224/197)^2 = 1.29x the algorithmic flops (32-wide tiles)"}
                out["roofline_core"] = core_roofline
        wg = algorithmic_work("gemm", N_total, N_l, C, 512, widths, args.top_k, world)
        if stage_ms["gemm"] > 0:
            out["gemm"] = {"tflops": round(wg["flops"] / (stage_ms["gemm"] * 1e-3) / 1e12, 2), "peak_f32_mfma": F32_MFMA_PEAK_TF,
                           "ms": round(stage_ms["gemm"], 4), "note": "K1a normalize x2 + K1 fp32-MFMA GEMM"}
        # ---- CPU baseline: the oracle's similarity path (reference utils.py:566-612 + similarity.py +
        #      describe_broad_neurons.py:101-102, P recomputed per layer as the reference does) -------------
        if world == 1 and not args.no_cpu_baseline:
            sys.path.insert(0, os.path.join(ROOT, "oracle"))
            import oracle as O
            nl = max(1, min(args.cpu_baseline_layers, len(widths)))
            E_img_h = dis.E_img.cpu().numpy()
            E_txt_h = E_txt.float().cpu().numpy()
            A_h = [dis.At[dis.offsets[i]:dis.offsets[i + 1], :N_l].t().contiguous().cpu().numpy() for i in range(nl)]
            O.lib()
            ncpu = host_cpu_share()
            O.set_num_threads(ncpu)          # OpenMP loops of the C oracle
            torch.set_num_threads(ncpu)
            try:
                from threadpoolctl import threadpool_limits
                threadpool_limits(ncpu)      # numpy's BLAS (the reference's torch.matmul is a BLAS call too)
            except Exception:
                pass
            tc = time.perf_counter()
            for i in range(nl):
                O.dissect_layer(E_img_h, E_txt_h, A_h[i], top_k=args.top_k)
            cpu_s = (time.perf_counter() - tc) * len(widths) / nl
            out["cpu_baseline"] = {"value": round(N_l / cpu_s, 1), "unit": "images/s", "cores": O.num_threads(),
                                   "kind": "port",
                                   "sample": "oracle similarity path only (normalise + I.T^T per layer, softmax, top-%d, "
                                             "soft-WPMI, logsumexp, top-10/top-5; NO encoder forwards, NO CSV) on %d of %d "
                                             "layers x 768 neurons at N=%d, time scaled x%d/%d; compare with "
                                             "core_images_per_s, not value" % (args.top_k, nl, len(widths), N_l, len(widths), nl)}
            # The other half of the job on the same host cores, for scale: the encoder forward of the same tower in
            # PyTorch's own CPU kernels (what the reference's CPU run does), on a small sample.  Not the oracle.
            if not args.core_only:
                try:
                    import copy
                    ns = 32
                    m_cpu = copy.deepcopy(model).to("cpu").eval()
                    for mod in m_cpu.modules():
                        mod._forward_hooks.clear()   # the copies of the K0 hooks want device tensors
                    x_cpu = images[:ns].cpu()
                    with torch.no_grad():
                        m_cpu.encode_image(x_cpu[:4])
                        tc = time.perf_counter()
                        m_cpu.image_projection(m_cpu.encode_image(x_cpu))
                        enc_s = time.perf_counter() - tc
                    enc_rate = ns / enc_s
                    out["cpu_baseline"]["encoder_images_per_s"] = round(enc_rate, 1)
                    out["cpu_baseline"]["end_to_end_images_per_s"] = round(1.0 / (1.0 / enc_rate + cpu_s / N_l), 1)
                    out["cpu_baseline"]["encoder_sample"] = ("%d images through the same tower on PyTorch CPU (%d threads); "
                                                              "end_to_end = encoder + similarity path per image: the "
                                                              "number to hold against value" % (ns, ncpu))
                    del m_cpu
                except Exception as e:   # the baseline is a report, never a reason to lose the bench line
                    out["cpu_baseline"]["encoder_sample"] = "skipped: %s" % (e,)
        print(json.dumps(out), flush=True)
    for h in handles:
        h.remove()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
