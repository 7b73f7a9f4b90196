"""Shared body of the three describe_*_neurons.py drivers (reference mains: describe_clip_neurons.py:37-93,
describe_og_neurons.py:49-152, describe_broad_neurons.py:51-175).  The per-layer loop, the `outputs` dict and
the DataFrame -> CSV step follow the reference line for line; the torch.max / torch.topk calls are the K6 / K3
HIP kernels."""
import datetime
import json
import os

import pandas as pd
import torch

from .. import core


def describe_layers(args, utils_mod, names_for, variant, pass_top_k, pass_d_probe):
    """variant 'clip': top-1 description (describe_clip_neurons.py:64); 'og': top-10 (describe_og_neurons.py:99)."""
    if str(getattr(args, "device", "cuda")).startswith("cuda") and os.environ.get("MCD_NO_TUNABLEOP", "0") != "1":
        from ..tuning import enable_gemm_tuning
        enable_gemm_tuning()        # packaged per-shape GEMM picks for the encoder forwards (tuning.py)
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


def write_results(df, args, csv_name="descriptions.csv", txt_name="args.txt"):
    result_dir = args.result_dir if args.result_dir != "" else "."   # reference default "" breaks os.mkdir
    if not os.path.exists(result_dir):
        os.mkdir(result_dir)
    save_path = "{}/{}_{}".format(result_dir, args.target_model,
                                  datetime.datetime.now().strftime("%y_%m_%d_%H_%M"))
    os.makedirs(save_path, exist_ok=True)
    df.to_csv(os.path.join(save_path, csv_name), index=False)
    with open(os.path.join(save_path, txt_name), 'w') as f:
        json.dump(args.__dict__, f, indent=2)
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
