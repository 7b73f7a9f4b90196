"""Mirror of the reference's concept_vit/og_utils.py: OpenAI-CLIP dissector, Mammo-CLIP (or any) target.

Kept contracts (reference og_utils.py): get_activation :31-56, get_save_names :58-66,
save_activations :374-471, get_similarity_from_activations :474-518 (no `top_k` argument: the similarity
function runs with its own default, as in the reference).  The CLIP dissector is the offline ClipViT of
data_utils (reference: clip.load(clip_name), a URL download); text goes through the same offline tokenizer.
"""
import torch

from . import data_utils
from . import utils as _u
from .utils import get_activation, get_save_names, _all_saved, _make_save_dir  # noqa: F401

PM_SUFFIX = _u.PM_SUFFIX


def save_prefix(target_name, d_probe, breast_clip_ckh=None, fine_tuned_ckh=None):
    """reference og_utils.py:394-406"""
    if breast_clip_ckh is not None:
        if fine_tuned_ckh is not None:
            return "/Latest_vindr_calc_finetuned_"
        return "/clip_dissector_{}_target_{}_small_mammo_pretrained_".format(target_name, d_probe)
    return "/clip_dissector_{}_target_{}_small_not_mammo_pretrained_".format(target_name, d_probe)


def _clip_dissector(device):
    model, _ = data_utils.get_target_model("clip", device)
    tok = data_utils.HashTokenizer()
    return model, (lambda texts: tok(texts, max_length=77))


def save_activations(clip_name, target_name, target_layers, d_probe,
                     concept_set, batch_size, device, pool_mode, save_dir, breast_clip_ckh=None, fine_tuned_ckh=None,
                     args=None):
    """reference og_utils.py:374-471.  Returns the Extraction left on the device (None when every cache file existed)."""
    clip_model, tokenize = _clip_dissector(device)
    if target_name == "clip":
        target_model = clip_model
        encode_target = target_model                      # reference :464-469 -> target_model(images), :127
    elif target_name == "breastclip_classifier":
        target_model, _ = data_utils.get_target_model(target_name, device, args=args, ckpt=breast_clip_ckh,
                                                      n_class=getattr(args, "num_class", 1),
                                                      finetuned_ckpt=fine_tuned_ckh)
        encode_target = target_model.encode_image
    else:
        target_model, _ = data_utils.get_target_model(target_name, device, ckpt=breast_clip_ckh)
        encode_target = target_model.encode_image         # reference :93
    data = _u._probe_data(d_probe, device)
    words = _u._read_concepts(concept_set)
    t_name, c_name, x_name = get_save_names(clip_name=clip_name, target_name=target_name, target_layer='{}',
                                            d_probe=d_probe, concept_set=concept_set, pool_mode=pool_mode,
                                            save_dir=save_dir)
    pre = save_dir + save_prefix(target_name, d_probe, breast_clip_ckh, fine_tuned_ckh)
    return _u.extract_and_save(clip_model, target_model, encode_target, target_layers, data, words, tokenize, batch_size,
                               device, pool_mode, pre + t_name, pre + c_name, pre + x_name)


def get_similarity_from_activations(target_save_name, clip_save_name, text_save_name, similarity_fn,
                                    return_target_feats=True, device="cuda", d_probe="vindr"):
    clip_feats = _u.get_clip_feats(clip_save_name, text_save_name, device=device, d_probe=d_probe)
    target_feats = torch.load(target_save_name, map_location='cpu', weights_only=True).to(device)
    similarity = similarity_fn(clip_feats, target_feats, device=device)   # reference :508
    if return_target_feats:
        return similarity, target_feats
    return similarity
