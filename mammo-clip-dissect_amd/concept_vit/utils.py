"""Mirror of the reference's concept_vit/utils.py (Mammo-CLIP dissector orchestration) for the hot path.

Kept names and contracts (reference concept_vit/utils.py):
    get_activation(outputs, mode)                                   :27-52
    get_save_names(clip_name, target_name, target_layer, d_probe, concept_set, pool_mode, save_dir)  :54-62
    save_activations(clip_name, target_name, target_layers, d_probe, concept_set, batch_size, device,
                     pool_mode, save_dir, breast_clip_ckh=None, fine_tuned_ckh=None, args=None)       :430-564
    get_similarity_from_activations(target_save_name, clip_save_name, text_save_name, similarity_fn,
                                    return_target_feats=True, device="cuda", d_probe="vindr", top_k=100)  :566-612
    get_clip_feats(clip_save_name, text_save_name, device="cuda", d_probe="vindr")                    :670-702
    _all_saved / _make_save_dir                                                                       :648-668

What differs, deliberately:
  * models and probe data come from the offline factory (data_utils): nothing is downloaded;
  * hooks pool with the K0 HIP kernel straight into ONE activation matrix (no list + torch.cat), and
    when the target IS the dissector the images are encoded once, not twice (reference :550-552);
  * the per-layer P = I_hat @ T_hat^T of get_similarity_from_activations is computed by the HIP GEMM and
    cached for the run (the reference recomputes it for each of the 12-39 layers);
  * failures raise (the reference swallows them with `except Exception: print`, :336-337).
The activation-cache files keep the reference's names and format: torch.save of float32 tensors
[N, U_layer] / [N, 512] / [C, 512], so caches are interchangeable.
"""
import os
import re

import torch
from torch.utils.data import DataLoader

from .. import core
from ..pipeline import Dissector
from . import data_utils

PM_SUFFIX = {"max": "_max", "avg": ""}
_P_CACHE = {}


# ---- hooks (reference :27-52) ---------------------------------------------------------------------
def get_activation(outputs, mode):
    '''
    mode: how to pool activations: one of avg, max
    for fc or ViT neurons does no pooling
    (same semantics as the reference; the pooling itself is the K0 HIP kernel)
    '''
    if mode not in ("avg", "max"):
        raise ValueError("pool mode must be 'avg' or 'max'")

    def hook(model, input, output):
        if mode == 'avg' and type(output) is tuple:   # the reference unwraps tuples only in 'avg' mode
            output = output[0]
        out = output.detach()
        if out.dim() not in (2, 3, 4):
            return
        width = out.shape[1] if out.dim() in (2, 4) else out.shape[2]
        dst = torch.empty((out.shape[0], width), dtype=torch.float32, device=out.device)
        core.hook_pool(out, mode, dst, 0, 0, False)
        outputs.append(dst)
    return hook


def get_save_names(clip_name, target_name, target_layer, d_probe, concept_set, pool_mode, save_dir):
    target_save_name = "{}/{}_{}_{}{}.pt".format(save_dir, d_probe, target_name, target_layer,
                                                 PM_SUFFIX[pool_mode])
    clip_save_name = "{}/{}_{}.pt".format(save_dir, d_probe, clip_name.replace('/', ''))
    concept_set_name = (concept_set.split("/")[-1]).split(".")[0]
    text_save_name = "{}/{}_{}.pt".format(save_dir, concept_set_name, clip_name.replace('/', ''))
    return target_save_name, clip_save_name, text_save_name


def save_prefix(d_probe, breast_clip_ckh=None, fine_tuned_ckh=None):
    """The prefix the reference's writer puts in front of the save names (:456-468)."""
    if breast_clip_ckh is not None:
        if fine_tuned_ckh is not None:
            return "/newest_{}_cancer_finetuned_".format(d_probe)
        return "/latest_{}_mammo_pretrained_".format(d_probe)
    return "/Latest_{}_not_mammo_pretrained_".format(d_probe)


def resolve_layer(model, name):
    """'image_encoder._blocks[3]' -> the module (the reference evals the string, :135-136)."""
    obj = model
    for part in re.findall(r"[A-Za-z_][A-Za-z_0-9]*|\[\d+\]", name.strip()):
        obj = obj[int(part[1:-1])] if part.startswith("[") else getattr(obj, part)
    return obj


def _all_saved(save_names):
    """
    save_names: {layer_name:save_path} dict
    Returns True if there is a file corresponding to each one of the values in save_names,
    else Returns False
    """
    for save_name in save_names.values():
        if not os.path.exists(save_name):
            return False
    return True


def _make_save_dir(save_name):
    """
    creates save directory if one does not exist
    save_name: full save path
    """
    save_dir = save_name[:save_name.rfind("/")]
    if save_dir and not os.path.exists(save_dir):
        os.makedirs(save_dir)
    return


def _read_concepts(concept_set):
    with open(concept_set, 'r') as f:
        words = (f.read()).split('\n')
    return [i for i in words if i != ""]   # reference :495-498


def _layer_width(model, layer, sample, forward):
    """Neurons a hook on `layer` yields (one tiny forward)."""
    got = []
    h = layer.register_forward_hook(lambda m, i, o: got.append(o[0] if type(o) is tuple else o))
    with torch.no_grad():
        forward(sample)
    h.remove()
    o = got[0]
    return o.shape[1] if o.dim() in (2, 4) else o.shape[2]


def extract_and_save(clip_model, target_model, encode_target, target_layers, dataset, words, tokenize, batch_size,
                     device, pool_mode, target_tmpl, clip_save_name, text_save_name):
    """Shared body of the three save_activations variants: one pass over D_probe with the K0 hooks writing the
    activation matrix, dissector image/text embeddings, then the reference-format cache files."""
    layer_files = {l: target_tmpl.format(l) for l in target_layers}
    need_target = not _all_saved(layer_files)
    need_clip = not os.path.exists(clip_save_name)
    need_text = not os.path.exists(text_save_name)
    for n in list(layer_files.values()) + [clip_save_name, text_save_name]:
        _make_save_dir(n)

    with torch.no_grad():
        if need_text:   # reference :390-414
            tok = tokenize(["{}".format(word) for word in words])
            tok = {k: v.to(device) for k, v in tok.items()} if isinstance(tok, dict) else tok.to(device)
            feats = []
            keys = list(tok.keys()) if isinstance(tok, dict) else None
            n_txt = tok[keys[0]].shape[0] if keys else tok.shape[0]
            for i in range(0, n_txt, batch_size):
                chunk = {k: v[i:i + batch_size] for k, v in tok.items()} if keys else tok[i:i + batch_size]
                t = clip_model.encode_text(chunk)
                if getattr(clip_model, "projection", False):
                    t = clip_model.text_projection(t)
                feats.append(t.float())
            torch.save(torch.cat(feats).cpu(), text_save_name)
        if not (need_target or need_clip):
            return
        N = len(dataset)
        loader = DataLoader(dataset, batch_size=batch_size, shuffle=False, num_workers=0)
        first = dataset[0][0].unsqueeze(0).to(device)
        layers = [resolve_layer(target_model, l) for l in target_layers]
        widths = [_layer_width(target_model, m, first, encode_target) for m in layers]
        same = target_model is clip_model
        dis = Dissector(N, list(target_layers), widths, len(words), data_utils.PROJ_DIM, device,
                        pool_mode=pool_mode)
        handles = [m.register_forward_hook(dis.hook(i)) for i, m in enumerate(layers)] if need_target else []
        try:
            for batch in loader:
                images = (batch[0] if isinstance(batch, (list, tuple)) else batch["images"]).to(device)
                if need_target:
                    out = encode_target(images)                    # hooks fire (reference :174-181)
                if need_clip:
                    f = out if (same and need_target) else clip_model.encode_image(images)   # one pass when same
                    if getattr(clip_model, "projection", False):
                        f = clip_model.image_projection(f)
                    dis.add_image_features(f.float())
                dis.advance(images.shape[0])
        finally:
            for h in handles:
                h.remove()
        if need_clip:
            torch.save(dis.E_img.cpu(), clip_save_name)
        if need_target:
            for i, l in enumerate(target_layers):   # cache format: [N, U_layer] float32 (reference :188-196)
                torch.save(dis.At[dis.offsets[i]:dis.offsets[i + 1], :N].t().contiguous().cpu(), layer_files[l])


def save_activations(clip_name, target_name, target_layers, d_probe,
                     concept_set, batch_size, device, pool_mode, save_dir, breast_clip_ckh=None, fine_tuned_ckh=None,
                     args=None):
    """Mammo-CLIP dissector + target (reference :430-564).  Dissector = BreastClip; the target is built by
    data_utils.get_target_model.  `clip_name` only names files, as in the reference."""
    finetuned = fine_tuned_ckh
    tower = "vit" if target_name == "breastclip_vit" else "cnn"
    clip_model, _ = data_utils.get_target_model("breastclip_vit" if tower == "vit" else "breastclip", device,
                                                ckpt=breast_clip_ckh)
    if target_name in ("breastclip", "breastclip_vit") and finetuned is None:
        target_model = clip_model          # same weights: encode the probe set once
    elif target_name == "breastclip_classifier":
        target_model, _ = data_utils.get_target_model(target_name, device, args=args, ckpt=breast_clip_ckh,
                                                      n_class=getattr(args, "num_class", 1), finetuned_ckpt=finetuned)
    else:
        target_model, _ = data_utils.get_target_model(target_name, device, ckpt=breast_clip_ckh)
    data = data_utils.get_data(d_probe, None)
    words = _read_concepts(concept_set)
    t_name, c_name, x_name = get_save_names(clip_name=clip_name, target_name=target_name, target_layer='{}',
                                            d_probe=d_probe, concept_set=concept_set, pool_mode=pool_mode,
                                            save_dir=save_dir)
    pre = save_dir + save_prefix(d_probe, breast_clip_ckh, fine_tuned_ckh)   # reference :508-516
    extract_and_save(clip_model, target_model, target_model.encode_image, target_layers, data, words,
                     clip_model.tokenize, batch_size, device, pool_mode, pre + t_name, pre + c_name, pre + x_name)
    return


def _load_feats(path, device):
    return torch.load(path, map_location='cpu', weights_only=True).float().to(device)


def get_clip_feats(clip_save_name, text_save_name, device="cuda", d_probe="vindr"):
    """P = I_hat @ T_hat^T from the cached embeddings (reference :670-702), on the GPU, cached per run."""
    key = (os.path.abspath(clip_save_name), os.path.abspath(text_save_name), str(device),
           os.path.getmtime(clip_save_name), os.path.getmtime(text_save_name))
    if key not in _P_CACHE:
        _P_CACHE.clear()
        with torch.no_grad():
            image_features = _load_feats(clip_save_name, device)
            text_features = _load_feats(text_save_name, device)
            core.normalize_rows(image_features, out=image_features)   # :577
            core.normalize_rows(text_features, out=text_features)     # :578
            _P_CACHE[key] = core.embed_gemm(image_features, text_features)   # :594
    return _P_CACHE[key]


def get_similarity_from_activations(target_save_name, clip_save_name, text_save_name, similarity_fn,
                                    return_target_feats=True, device="cuda", d_probe="vindr", top_k=100):
    clip_feats = get_clip_feats(clip_save_name, text_save_name, device=device, d_probe=d_probe)
    target_feats = torch.load(target_save_name, map_location='cpu', weights_only=True).to(device)
    similarity = similarity_fn(clip_feats, target_feats, device=device, top_k=top_k)   # reference :602
    if return_target_feats:
        return similarity, target_feats
    return similarity
