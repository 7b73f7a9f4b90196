"""Shared body of the three describe_*_neurons.py drivers (reference mains: describe_clip_neurons.py:37-93,
describe_og_neurons.py:49-152, describe_broad_neurons.py:51-175).

Two routes to the same CSV:
  * fused (the fast path): when save_activations has just run the extraction, its result is still resident in HBM
    (utils.Extraction) and every layer is scored in ONE pass -- P and S once (the reference recomputes them per layer,
    utils.py:570-594), one launch per kernel over all layers (pipeline.Dissector.finish) -- and the CSV is written by
    pipeline.write_descriptions_csv.  Taken for soft_wpmi / wpmi.
  * per layer (the reference's loop, line for line): when the activation cache already existed (nothing was
    extracted), or for the other similarity functions: get_similarity_from_activations per layer from the cache files,
    the `outputs` dict, DataFrame.to_csv.  The torch.max / torch.topk calls are the K6 / K3 HIP kernels.
Both routes give the same bytes (tests/test_gpu_pipeline.py: test_fused_driver_equals_cache_driver)."""
import datetime
import json
import os

import pandas as pd
import torch

from .. import core


def setup_device(args):
    """Before any model runs: packaged per-shape GEMM picks for the encoder forwards (tuning.py) and, under
    torch.distributed.run (WORLD_SIZE > 1), one process per GPU over RCCL with this rank's device."""
    if str(getattr(args, "device", "cuda")).startswith("cuda") and os.environ.get("MCD_NO_TUNABLEOP", "0") != "1":
        from ..tuning import enable_gemm_tuning
        enable_gemm_tuning()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        import torch.distributed as dist
        local = int(os.environ.get("LOCAL_RANK", "0")) % max(1, torch.cuda.device_count())
        torch.cuda.set_device(local)
        args.device = "cuda:%d" % local
        if not dist.is_initialized():
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            dist.init_process_group("nccl", device_id=torch.device(args.device))


def describe_layers(args, utils_mod, names_for, variant, pass_top_k, pass_d_probe, live=None):
    """variant 'clip': top-1 description (describe_clip_neurons.py:64); 'og': top-10 (describe_og_neurons.py:99).
    live: the utils.Extraction save_activations returned (None: score from the cache files).
    Returns a pipeline.DissectResult (fused route) or a pandas DataFrame (per-layer route)."""
    if live is not None and args.similarity_fn in ("soft_wpmi", "wpmi") and live.target_layers == list(args.target_layers) \
            and os.environ.get("MCD_DRIVER_PER_LAYER", "0") != "1":
        # reference: utils.py:602 passes top_k (describe_broad_neurons.py:94-96); og_utils.py:508 / CLIP_og_utils.py:165
        # do not, so the similarity function's own default applies (100 / 28)
        live.dis.set_scoring(args.similarity_fn, args.top_k if pass_top_k else None)
        res = live.dis.finish(live.E_txt, k_desc=1 if variant == "clip" else 10, k_img=5)
        live.start_writer()      # the cache files: device -> host copies behind the scoring kernels, beside the CSV writing
        return res
    if live is not None:
        live.wait()       # the per-layer route reads the cache files the writer thread is producing
    from . import similarity
    similarity_fn = getattr(similarity, args.similarity_fn)   # reference: eval("similarity.{}".format(...))
    outputs = {"layer": [], "unit": [], "description": [], "similarity": [], "images": []}
    with open(args.concept_set, 'r') as f:
        words = (f.read()).split('\n')
    for target_layer in args.target_layers:
        target_save_name, clip_save_name, text_save_name = names_for(target_layer)
        kw = {}
        if pass_d_probe:
            kw["d_probe"] = args.d_probe
        if pass_top_k:
            kw["top_k"] = args.top_k
        similarities, target_feats = utils_mod.get_similarity_from_activations(
            target_save_name, clip_save_name, text_save_name, similarity_fn, return_target_feats=True,
            device=args.device, **kw)
        tf = target_feats.to(args.device)
        if variant == "clip":
            vals, ids = core.row_topk(similarities, 1)                    # torch.max(similarities, dim=1)
            vals, ids = vals[:, 0], ids[:, 0]
        else:
            vals, ids = core.row_topk(similarities, 10)                   # torch.topk(similarities, k=10, dim=1)
        _, top_ids = core.col_topk(tf, 5, want_vals=False)                # torch.topk(target_feats, k=5, dim=0)
        top_ids = top_ids.long().t()                                       # [5, U] int64 like torch.topk
        del similarities, target_feats
        if variant == "clip":
            descriptions = [words[int(idx)] for idx in ids]
        else:
            descriptions = []
            for id in ids:
                descriptions.append([words[int(idx)] for idx in id])
        outputs["unit"].extend([i for i in range(len(vals))])
        outputs["layer"].extend([target_layer] * len(vals))
        outputs["description"].extend(descriptions)
        outputs["similarity"].extend(vals.cpu().numpy())
        outputs["images"].extend(top_ids.T.cpu().numpy())
        del top_ids, vals, ids
    return pd.DataFrame(outputs)


def write_results(df, args, csv_name="descriptions.csv", txt_name="args.txt", variant="og", live=None):
    """reference describe_clip_neurons.py:85-91 / describe_broad_neurons.py:122-172: {result_dir}/{target}_{time}/ with the
    CSV and the args.  `df`: DataFrame (per-layer route) or DissectResult (fused route).  In a multi-rank run rank 0
    writes.  Waits for the activation-cache writer before returning."""
    import torch.distributed as dist
    from ..pipeline import DissectResult, write_descriptions_csv
    rank = dist.get_rank() if (dist.is_available() and dist.is_initialized()) else 0
    result_dir = args.result_dir if args.result_dir != "" else "."   # reference default "" breaks os.mkdir
    save_path = "{}/{}_{}".format(result_dir, args.target_model,
                                  datetime.datetime.now().strftime("%y_%m_%d_%H_%M"))
    if rank == 0:
        if not os.path.exists(result_dir):
            os.mkdir(result_dir)
        os.makedirs(save_path, exist_ok=True)
        if isinstance(df, DissectResult):
            with open(args.concept_set, 'r') as f:
                words = (f.read()).split('\n')
            write_descriptions_csv(df, words, os.path.join(save_path, csv_name), variant)
        else:
            df.to_csv(os.path.join(save_path, csv_name), index=False)
        with open(os.path.join(save_path, txt_name), 'w') as f:
            json.dump(args.__dict__, f, indent=2)
    if live is not None:
        live.wait()
    return save_path


def broad_file_names(args):
    """The output file-name ladder of describe_broad_neurons.py:128-169."""
    d = args.d_probe
    if args.Breast_clip_chkpt is not None:
        if args.finetuned_img_classifier_chkpt is not None:
            stem = "NEW_vindr_cancer_finetuned_breast_clip_classifier_descriptions"
        else:
            stem = {"vindr": "NEW_vindr_mammo_pretrained_breast_clip_classifier_descriptions",
                    "imagenet_subsets": "imagenet_subsets_spec_small_mammo_pretrained_breast_clip_classifier_descriptions"
                    }.get(d, "%s_mammo_pretrained_breast_clip_classifier_descriptions" % d)
        return stem + ".csv", stem + "_args.txt"
    stem = {"vindr": "NEW_vindr_not_mammo_pretrained_breast_clip_descriptions",
            "imagenet_subsets": "imagenet_subsets_spec_small_not_mammo_pretrained_breast_clip_descriptions"
            }.get(d, "%s_not_mammo_pretrained_breast_clip_descriptions" % d)
    txt = "imagenet_subsets_not_spec_small_mammo_pretrained_breast_clip_descriptions_args.txt" \
        if d == "imagenet_subsets" else stem + "_args.txt"
    return stem + ".csv", txt
