"""Mirror of the reference's concept_vit/CLIP_og_utils.py (the original CLIP-Dissect utilities): CLIP
dissector, torchvision-style target called as target_model(images), plain save names (no prefix).

Kept contracts (reference CLIP_og_utils.py): get_activation :13-36, get_save_names :38-46,
save_activations :126-150, get_similarity_from_activations :153-175.
"""
import torch

from . import data_utils
from . import utils as _u
from .og_utils import _clip_dissector
from .utils import get_activation, get_save_names, _all_saved, _make_save_dir  # noqa: F401

PM_SUFFIX = _u.PM_SUFFIX


def save_activations(clip_name, target_name, target_layers, d_probe,
                     concept_set, batch_size, device, pool_mode, save_dir):
    clip_model, tokenize = _clip_dissector(device)
    target_model = clip_model if target_name == "clip" else data_utils.get_target_model(target_name, device)[0]
    data = _u._probe_data(d_probe, device)
    words = _u._read_concepts(concept_set)
    t_name, c_name, x_name = get_save_names(clip_name=clip_name, target_name=target_name, target_layer='{}',
                                            d_probe=d_probe, concept_set=concept_set, pool_mode=pool_mode,
                                            save_dir=save_dir)
    return _u.extract_and_save(clip_model, target_model, target_model, target_layers, data, words, tokenize, batch_size,
                               device, pool_mode, t_name, c_name, x_name)   # reference :70-72: target_model(images)


def get_similarity_from_activations(target_save_name, clip_save_name, text_save_name, similarity_fn,
                                    return_target_feats=True, device="cuda"):
    clip_feats = _u.get_clip_feats(clip_save_name, text_save_name, device=device)
    target_feats = torch.load(target_save_name, map_location='cpu', weights_only=True)   # CPU, as the reference
    similarity = similarity_fn(clip_feats, target_feats, device=device)                  # reference :165
    if return_target_feats:
        return similarity, target_feats
    return similarity
