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
import threading

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
    """The three cache-file names of a (dissector, target, layer, probe set, concept set) combination.  The format
    strings ARE the contract (reference utils.py:54-62): caches written by either implementation must be found by the
    other."""
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
    """True when every cache file named in the {layer: path} dict is already on disk -- the reference's test for skipping
    the target-model pass (utils.py:648-657)."""
    return all(os.path.exists(path) for path in save_names.values())


def _make_save_dir(save_name):
    """Create the directory part of a cache-file path if it is missing (utils.py:659-668)."""
    save_dir = os.path.dirname(save_name)
    if save_dir:
        os.makedirs(save_dir, exist_ok=True)


def _read_concepts(concept_set):
    with open(concept_set, 'r') as f:
        words = (f.read()).split('\n')
    return [i for i in words if i != ""]   # reference :495-498


def _layer_widths(model, layers, sample, forward):
    """Neurons a hook on each of `layers` yields: ONE tiny forward with a probe on every layer, cached on the model
    (the widths are a property of the architecture)."""
    cache = model.__dict__.setdefault("_mcd_layer_widths", {})
    missing = [m for m in layers if id(m) not in cache]
    if missing:
        got = {}
        hs = [m.register_forward_hook(lambda mod, i, o, k=id(m): got.__setitem__(k, o[0] if type(o) is tuple else o))
              for m in missing]
        with torch.no_grad():
            forward(sample)
        for h in hs:
            h.remove()
        for m in missing:
            o = got[id(m)]
            cache[id(m)] = int(o.shape[1] if o.dim() in (2, 4) else o.shape[2])
    return [cache[id(m)] for m in layers]


class Extraction:
    """What one extraction pass leaves resident in HBM: the Dissector (activation matrix of all target layers + the
    dissector's image embeddings) and the text embeddings.  The drivers score it in ONE fused pass
    (Dissector.finish) instead of re-loading the cache files layer by layer; the cache files are still written, by a
    background thread, as the reference's side output."""

    def __init__(self, dis, E_txt, target_layers, writer=None):
        self.dis, self.E_txt, self.target_layers, self.writer = dis, E_txt, list(target_layers), writer

    def start_writer(self):
        """Start the cache-file writer (idempotent).  The drivers call it once the scoring kernels are queued, so the
        writer's device -> host copies run beside the CSV writing on the host, not beside the scoring kernels."""
        if self.writer is not None and not self.writer.is_alive() and not self.writer.started:
            self.writer.begin()

    def wait(self):
        if self.writer is not None:
            self.start_writer()
            self.writer.join()
            if self.writer.error is not None:
                raise self.writer.error
            self.writer = None


class _CacheWriter(threading.Thread):
    """torch.save of the reference-format cache files ([N, U_layer] per layer, [N, 512] image embeddings) off the main
    thread, while the main thread scores and writes the CSV: device -> PINNED host copies on a stream of its own (370 MB
    at config 2: 7 ms pinned against 100 ms pageable), then the file writes on two threads (torch.save releases the GIL
    for part of its work: 96 ms serial, 46 ms on two threads, no better on more)."""

    def __init__(self, device, jobs):
        super().__init__(daemon=True)
        self.device, self.jobs, self.error, self.started = device, jobs, None, False
        self.ready = None

    def begin(self):
        self.started = True
        if torch.device(self.device).type == "cuda":
            with torch.cuda.device(self.device):
                self.ready = torch.cuda.Event()
                self.ready.record()           # everything queued so far (extraction, scoring) precedes the copies
        self.start()

    def _save_some(self, items):
        try:
            for t, path in items:
                torch.save(t, path)
        except Exception as e:
            self.error = e

    def run(self):
        try:
            staged = []
            if self.ready is not None:
                with torch.cuda.device(self.device):
                    side = torch.cuda.Stream(device=self.device)
                    side.wait_event(self.ready)
                    with torch.cuda.stream(side):
                        for make, path in self.jobs:
                            src = make()
                            dst = torch.empty(src.shape, dtype=src.dtype, pin_memory=True)
                            dst.copy_(src, non_blocking=True)
                            staged.append((dst, path))
                    side.synchronize()
            else:
                staged = [(make().cpu(), path) for make, path in self.jobs]
            helper = threading.Thread(target=self._save_some, args=(staged[1::2],), daemon=True)
            helper.start()
            self._save_some(staged[0::2])
            helper.join()
        except Exception as e:   # surfaced by Extraction.wait(): a failed save must not pass silently (reference :336-337 does)
            self.error = e


def _dist_info():
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        return dist.get_world_size(), dist.get_rank()
    return 1, 0


def extract_and_save(clip_model, target_model, encode_target, target_layers, dataset, words, tokenize, batch_size,
                     device, pool_mode, target_tmpl, clip_save_name, text_save_name, write_caches=None, gather=None):
    """Shared body of the three save_activations variants: one pass over D_probe with the K0 hooks writing the
    activation matrix, dissector image/text embeddings, then the reference-format cache files.
    Returns an Extraction (everything still resident on the device) when the pass ran, None when every cache file
    already existed (reference: skip, utils.py:128,162,318).
    Multi-rank (torch.distributed initialised): `dataset` is this rank's shard; the cache files are a single-process
    feature and are not written."""
    world, rank = _dist_info()
    if write_caches is None:
        write_caches = world == 1 and os.environ.get("MCD_ACTIVATION_CACHE", "1") != "0"
    layer_files = {l: target_tmpl.format(l) for l in target_layers}
    need_target = not _all_saved(layer_files) or world > 1
    need_clip = not os.path.exists(clip_save_name) or world > 1
    need_text = not os.path.exists(text_save_name) or world > 1
    if write_caches:
        for n in list(layer_files.values()) + [clip_save_name, text_save_name]:
            _make_save_dir(n)

    with torch.no_grad():
        E_txt = None
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
            E_txt = torch.cat(feats)
            if write_caches:
                torch.save(E_txt.cpu(), text_save_name)
        if not (need_target or need_clip):
            return None
        if E_txt is None:
            E_txt = _load_feats(text_save_name, device)
        N = len(dataset)
        on_device = hasattr(dataset, "device_batches")
        if on_device:
            batches = dataset.device_batches(batch_size)
            first = dataset.images()[:1] if N > 0 else None
        else:
            batches = DataLoader(dataset, batch_size=batch_size, shuffle=False, num_workers=0)
            first = dataset[0][0].unsqueeze(0).to(device) if N > 0 else None
        if first is None:   # a rank without images still needs the layer widths
            first = torch.zeros((1, 3, 224, 224), dtype=torch.float32, device=device)
        layers = [resolve_layer(target_model, l) for l in target_layers]
        widths = _layer_widths(target_model, layers, first, encode_target)
        same = target_model is clip_model
        dis = Dissector(N, list(target_layers), widths, len(words), data_utils.PROJ_DIM, device,
                        pool_mode=pool_mode, gather=gather)
        handles = [m.register_forward_hook(dis.hook(i)) for i, m in enumerate(layers)] if need_target else []
        try:
            for batch in batches:
                if on_device:
                    images = batch
                else:
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
        if not need_clip:
            dis.E_img.copy_(_load_feats(clip_save_name, device))
        if not need_target:   # the layers' cache files exist, the dissector embeddings did not: load the activations
            for i, l in enumerate(target_layers):
                A = torch.load(layer_files[l], map_location='cpu', weights_only=True).float().to(device)
                core.transpose(A, out=dis.At[dis.offsets[i]:dis.offsets[i + 1], :N])
        writer = None
        if write_caches:
            jobs = []
            if need_clip:
                jobs.append((lambda: dis.E_img, clip_save_name))
            if need_target:   # cache format: [N, U_layer] float32 (reference :188-196)
                for i, l in enumerate(target_layers):
                    jobs.append((lambda i=i: dis.At[dis.offsets[i]:dis.offsets[i + 1], :N].t(), layer_files[l]))
            writer = _CacheWriter(device, jobs)
        return Extraction(dis, E_txt, target_layers, writer)


def _probe_data(d_probe, device):
    """D_probe for the extraction loop: generated on the device and resident there when `device` is a GPU (this rank's
    shard of it in a multi-rank run); MCD_PROBE_ON_HOST=1 keeps the host dataset + DataLoader path."""
    from ..pipeline import shard_bounds
    world, rank = _dist_info()
    n = len(data_utils.get_data(d_probe, None))
    lo, hi = shard_bounds(n, world, rank)
    if os.environ.get("MCD_PROBE_ON_HOST", "0") == "1":
        return data_utils.get_data(d_probe, None, None, lo, hi)
    return data_utils.get_data(d_probe, None, device, lo, hi)


def build_mammo_models(target_name, device, breast_clip_ckh=None, fine_tuned_ckh=None, args=None):
    """The model side of save_activations (reference :443-483): (dissector, target).  Separate so that a caller that
    dissects repeatedly (bench.py) builds the models once."""
    finetuned = fine_tuned_ckh
    tower = "vit" if target_name.startswith("breastclip_vit") else "cnn"
    clip_model, _ = data_utils.get_target_model(target_name if tower == "vit" else "breastclip", device,
                                                ckpt=breast_clip_ckh)
    if (target_name == "breastclip" or tower == "vit") and finetuned is None:
        target_model = clip_model          # same weights: encode the probe set once
    elif target_name == "breastclip_classifier":
        target_model, _ = data_utils.get_target_model(target_name, device, args=args, ckpt=breast_clip_ckh,
                                                      n_class=getattr(args, "num_class", 1), finetuned_ckpt=finetuned)
    else:
        target_model, _ = data_utils.get_target_model(target_name, device, ckpt=breast_clip_ckh)
    return clip_model, target_model


def save_activations(clip_name, target_name, target_layers, d_probe,
                     concept_set, batch_size, device, pool_mode, save_dir, breast_clip_ckh=None, fine_tuned_ckh=None,
                     args=None, prebuilt=None):
    """Mammo-CLIP dissector + target (reference :430-564).  Dissector = BreastClip; the target is built by
    data_utils.get_target_model.  `clip_name` only names files, as in the reference.
    prebuilt: optional dict(clip_model=, target_model=, data=) from an earlier call (models and the resident probe set
    are then reused).  Returns the Extraction (None when every cache file existed)."""
    if prebuilt is not None:
        clip_model, target_model, data = prebuilt["clip_model"], prebuilt["target_model"], prebuilt["data"]
    else:
        clip_model, target_model = build_mammo_models(target_name, device, breast_clip_ckh, fine_tuned_ckh, args)
        data = _probe_data(d_probe, device)
    words = _read_concepts(concept_set)
    t_name, c_name, x_name = get_save_names(clip_name=clip_name, target_name=target_name, target_layer='{}',
                                            d_probe=d_probe, concept_set=concept_set, pool_mode=pool_mode,
                                            save_dir=save_dir)
    pre = save_dir + save_prefix(d_probe, breast_clip_ckh, fine_tuned_ckh)   # reference :508-516
    return extract_and_save(clip_model, target_model, target_model.encode_image, target_layers, data, words,
                            clip_model.tokenize, batch_size, device, pool_mode, pre + t_name, pre + c_name, pre + x_name,
                            gather=(prebuilt or {}).get("gather"))


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
            image_features = core.normalize_rows(image_features.contiguous())   # :577
            text_features = core.normalize_rows(text_features.contiguous())     # :578
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
