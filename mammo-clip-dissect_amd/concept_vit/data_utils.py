"""Offline target-model and probe-data factory: mirror of the reference's concept_vit/data_utils.py.

Same entry points -- get_target_model(target_name, device, ...) -> (model.eval(), preprocess) and
get_data(dataset_name, preprocess) -- (reference data_utils.py:38-93, :102-311), but every reference
loader fetches weights or datasets from the network, which this build never does.
Here the architectures are built locally with the hook-point names the reference's launch scripts use
(run_clipdissect.sh, run_og_clip.sh) and get random-init weights under a fixed seed, or weights from a
LOCAL checkpoint path.  These modules are host-side PyTorch plumbing (the encoder forwards); the
dissection core they feed is the HIP library.

    target_name            hook points (target_layers)                 neurons
    breastclip             image_encoder._blocks[0..38]                EfficientNet-B5: 6992
    breastclip_vit         image_encoder.encoder.layer[0..11]          ViT-B/16: 12 x 768
    breastclip_classifier  image_encoder._blocks[0..38]                + linear head (n_class)
    clip                   vision_model.encoder.layers[0..11]          CLIP ViT-B/16: 12 x 768
    resnet50               conv1, layer1..layer4                       64/256/512/1024/2048
"""
import math
import os
import zlib

import torch
import torch.nn as nn
import torch.nn.functional as F

PROJ_DIM = 512
# The image tower's mask-free fp32 attention runs on the HIP kernel K9; MCD_NO_HIP_ATTENTION=1 (or setting this to
# False) keeps PyTorch's SDPA, e.g. to time one against the other.  Masked (text tower), autograd or non-fp32 calls
# always take SDPA.
HIP_ATTENTION = os.environ.get("MCD_NO_HIP_ATTENTION", "0") != "1"
# bench.py sets this to a list to time K9 inside the forwards: every 8th call appends (start, end, B, T, heads) with
# two HIP events recorded on the launch stream around the kernel.
ATTENTION_EVENTS = None
_attention_calls = 0
# The two residual updates of a block, x + proj(.) and x + fc2(.), as one hipBLASLt GEMM each (bias epilogue + beta*C,
# core.linear_residual) instead of nn.Linear + an elementwise add over the residual stream.  Inference-time fp32 only;
# MCD_NO_FUSED_RESIDUAL=1 (or False here, or a missing libmcd_blaslt.so) keeps PyTorch's two kernels.
FUSED_RESIDUAL = os.environ.get("MCD_NO_FUSED_RESIDUAL", "0") != "1"


# nn.LayerNorm of the towers on the HIP kernel K10 (csrc/k_ln.hip); MCD_NO_HIP_LAYER_NORM=1 keeps ATen's.
HIP_LAYER_NORM = os.environ.get("MCD_NO_HIP_LAYER_NORM", "0") != "1"


class _LayerNorm(nn.LayerNorm):
    """nn.LayerNorm (same parameters / state_dict keys); inference-time fp32 CUDA inputs take K10."""

    def forward(self, x):
        D = x.shape[-1]
        if (HIP_LAYER_NORM and x.is_cuda and x.dtype == torch.float32 and x.is_contiguous() and not torch.is_grad_enabled()
                and len(self.normalized_shape) == 1 and D % 4 == 0 and D <= 2048 and self.weight is not None
                and self.bias is not None):
            from .. import core
            return core.layer_norm(x, self.weight, self.bias, self.eps)
        return super().forward(x)


def _linear(mod, x):
    """mod(x) for an nn.Linear; on the fused path through libmcd_blaslt.so as well (res = None: same hipBLASLt GEMM with
    the bias epilogue as PyTorch's, but with this process's own best-of-32 pick instead of the library default)."""
    if _fused_residual_ok(x) and x.is_contiguous() and mod.bias is not None:
        from .. import core
        return core.linear_residual(None, x, mod.weight, mod.bias)
    return mod(x)


def _fused_residual_ok(x):
    if not (FUSED_RESIDUAL and x.is_cuda and x.dtype == torch.float32 and not torch.is_grad_enabled()):
        return False
    from .. import core
    return core.linear_residual_available()


# ------------------------------------------------------------------------------------------------------
# ViT-B/16 tower (module names follow HF ViTModel: embeddings / encoder.layer[i] / layernorm)
# ------------------------------------------------------------------------------------------------------
class _Attention(nn.Module):
    def __init__(self, dim, heads):
        super().__init__()
        self.heads = heads
        self.qkv = nn.Linear(dim, 3 * dim)
        self.proj = nn.Linear(dim, dim)

    def forward(self, x, mask=None):
        return self.proj(self.heads_out(x, mask))

    def heads_out(self, x, mask=None):
        """The concatenated head outputs [B, T, D], i.e. attention before the output projection."""
        B, T, D = x.shape
        qkv = _linear(self.qkv, x)
        if (HIP_ATTENTION and mask is None and qkv.is_cuda and qkv.dtype == torch.float32 and D == 64 * self.heads
                and T <= 256 and not (torch.is_grad_enabled() and qkv.requires_grad)):
            # K9 (csrc/k_attn.hip): one launch, reads the fused projection's layout, writes the proj input's
            from .. import core
            global _attention_calls
            _attention_calls += 1
            if ATTENTION_EVENTS is not None and _attention_calls % 8 == 0:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                o = core.vit_attention(qkv, self.heads)
                e1.record()
                ATTENTION_EVENTS.append((e0, e1, B, T, self.heads))
                return o
            return core.vit_attention(qkv, self.heads)
        q, k, v = qkv.view(B, T, 3, self.heads, D // self.heads).permute(2, 0, 3, 1, 4)
        o = F.scaled_dot_product_attention(q, k, v, attn_mask=mask)
        return o.transpose(1, 2).reshape(B, T, D)


class _Block(nn.Module):
    def __init__(self, dim, heads, mlp):
        super().__init__()
        self.norm1 = _LayerNorm(dim, eps=1e-12)
        self.attn = _Attention(dim, heads)
        self.norm2 = _LayerNorm(dim, eps=1e-12)
        self.fc1 = nn.Linear(dim, mlp)
        self.fc2 = nn.Linear(mlp, dim)

    def forward(self, x, mask=None):
        if _fused_residual_ok(x):
            from .. import core
            x = x.contiguous()
            # x1 is a new tensor (the block's input is left alone); the second update is in place on x1
            x1 = core.linear_residual(x, self.attn.heads_out(self.norm1(x), mask).contiguous(), self.attn.proj.weight,
                                      self.attn.proj.bias)
            h = F.gelu(_linear(self.fc1, self.norm2(x1)))
            return core.linear_residual(x1, h, self.fc2.weight, self.fc2.bias, out=x1)
        x = x + self.attn(self.norm1(x), mask)
        return x + self.fc2(F.gelu(self.fc1(self.norm2(x))))


class _Encoder(nn.Module):
    def __init__(self, depth, dim, heads, mlp, list_name):
        super().__init__()
        setattr(self, list_name, nn.ModuleList([_Block(dim, heads, mlp) for _ in range(depth)]))
        self._list_name = list_name

    def forward(self, x, mask=None):
        for blk in getattr(self, self._list_name):
            x = blk(x, mask)
        return x


class ViTTower(nn.Module):
    """[B,3,H,W] -> token sequence [B, 1+(H/16)*(W/16), 768]; hook points encoder.<list_name>[i]."""

    def __init__(self, image_size=224, patch=16, dim=768, depth=12, heads=12, mlp=3072, list_name="layer"):
        super().__init__()
        self.out_dim = dim
        self.patch_embed = nn.Conv2d(3, dim, patch, patch)
        n = (image_size // patch) ** 2
        self.cls_token = nn.Parameter(torch.zeros(1, 1, dim))
        self.pos_embed = nn.Parameter(torch.zeros(1, n + 1, dim))
        self.encoder = _Encoder(depth, dim, heads, mlp, list_name)
        self.layernorm = _LayerNorm(dim, eps=1e-12)

    def forward(self, x):
        return self.layernorm(self.encoder(self.embed(x)))

    def embed(self, x):
        """Patch embedding + class token + position embedding -> [B, 1 + n, dim]."""
        P = self.patch_embed.kernel_size[0]
        if (_fused_residual_ok(x) and x.is_contiguous() and x.dim() == 4 and self.patch_embed.bias is not None
                and P % 4 == 0 and x.shape[2] % P == 0 and x.shape[3] % P == 0
                and 1 + (x.shape[2] // P) * (x.shape[3] // P) == self.pos_embed.shape[1]):
            # the convolution as ONE GEMM that writes the token sequence directly (K11 + libmcd_blaslt.so): rows of
            # patch pixels (a zero row in every image's class-token slot) times the conv weight, plus the bias, plus a
            # residual operand that holds the position embedding (and cls + pos[0] - bias in the class-token rows).
            # No MIOpen call (its choice of algorithm varied between 0.5 and 1.2 ms from box to box), no cat, no add.
            from .. import core
            B = x.shape[0]
            return core.linear_residual(self._embed_residual(B), core.patchify(x, P),
                                        self.patch_embed.weight.view(self.patch_embed.out_channels, -1), self.patch_embed.bias)
        x = self.patch_embed(x).flatten(2).transpose(1, 2)
        return torch.cat([self.cls_token.expand(x.shape[0], -1, -1), x], dim=1) + self.pos_embed

    def _embed_residual(self, B):
        """[B, 1 + n, dim]: pos_embed, with cls_token + pos_embed[0] - bias in row 0 (the GEMM adds the bias back)."""
        key = (B, self.pos_embed._version, self.cls_token._version, self.patch_embed.bias._version, self.pos_embed.device)
        cache = self.__dict__.setdefault("_embed_res_cache", {})
        if cache.get("key") != key:
            r = self.pos_embed.detach().expand(B, -1, -1).contiguous()
            r[:, 0] = self.cls_token.detach()[0, 0] + self.pos_embed.detach()[0, 0] - self.patch_embed.bias.detach()
            cache["key"], cache["res"] = key, r
        return cache["res"]


# ------------------------------------------------------------------------------------------------------
# EfficientNet-B5 tower (parameter names follow the public EfficientNet-PyTorch layout so a local
# Mammo-CLIP checkpoint's image_encoder.* keys line up: _conv_stem, _bn0, _blocks[i]._expand_conv ...)
# ------------------------------------------------------------------------------------------------------
class _SameConv(nn.Conv2d):
    """TensorFlow 'SAME' padding (asymmetric for stride 2), as the b5 'tf_' weights expect."""

    def forward(self, x):
        ih, iw = x.shape[-2:]
        kh, kw = self.kernel_size
        sh, sw = self.stride
        ph = max((math.ceil(ih / sh) - 1) * sh + kh - ih, 0)
        pw = max((math.ceil(iw / sw) - 1) * sw + kw - iw, 0)
        if ph or pw:
            x = F.pad(x, [pw // 2, pw - pw // 2, ph // 2, ph - ph // 2])
        return F.conv2d(x, self.weight, self.bias, self.stride, 0, self.dilation, self.groups)


class _MBConv(nn.Module):
    def __init__(self, cin, cout, k, s, expand, se_ratio=0.25):
        super().__init__()
        mid = cin * expand
        self.expand = expand != 1
        if self.expand:
            self._expand_conv = _SameConv(cin, mid, 1, bias=False)
            self._bn0 = nn.BatchNorm2d(mid, momentum=0.01, eps=1e-3)
        self._depthwise_conv = _SameConv(mid, mid, k, s, groups=mid, bias=False)
        self._bn1 = nn.BatchNorm2d(mid, momentum=0.01, eps=1e-3)
        sq = max(1, int(cin * se_ratio))
        self._se_reduce = _SameConv(mid, sq, 1)
        self._se_expand = _SameConv(sq, mid, 1)
        self._project_conv = _SameConv(mid, cout, 1, bias=False)
        self._bn2 = nn.BatchNorm2d(cout, momentum=0.01, eps=1e-3)
        self.skip = s == 1 and cin == cout

    def forward(self, x):
        y = x
        if self.expand:
            y = F.silu(self._bn0(self._expand_conv(y)))
        y = F.silu(self._bn1(self._depthwise_conv(y)))
        se = self._se_expand(F.silu(self._se_reduce(y.mean(dim=[2, 3], keepdim=True))))
        y = self._bn2(self._project_conv(torch.sigmoid(se) * y))
        return x + y if self.skip else y


def _round_filters(f, width, divisor=8):
    f *= width
    nf = max(divisor, int(f + divisor / 2) // divisor * divisor)
    if nf < 0.9 * f:
        nf += divisor
    return int(nf)


class EfficientNetB5Tower(nn.Module):
    """EfficientNet-B5 (width 1.6, depth 2.2): 39 MBConv blocks, output [B, 2048] pooled features."""
    # B0 stage table: repeats, kernel, stride, expand, out channels
    STAGES = [(1, 3, 1, 1, 16), (2, 3, 2, 6, 24), (2, 5, 2, 6, 40), (3, 3, 2, 6, 80), (3, 5, 1, 6, 112),
              (4, 5, 2, 6, 192), (1, 3, 1, 6, 320)]

    def __init__(self, width=1.6, depth=2.2, num_classes=1):
        super().__init__()
        stem = _round_filters(32, width)
        self._conv_stem = _SameConv(3, stem, 3, 2, bias=False)
        self._bn0 = nn.BatchNorm2d(stem, momentum=0.01, eps=1e-3)
        blocks, cin = [], stem
        for r, k, s, e, o in self.STAGES:
            cout = _round_filters(o, width)
            for i in range(int(math.ceil(depth * r))):
                blocks.append(_MBConv(cin, cout, k, s if i == 0 else 1, e))
                cin = cout
        self._blocks = nn.ModuleList(blocks)
        self.out_dim = _round_filters(1280, width)
        self._conv_head = _SameConv(cin, self.out_dim, 1, bias=False)
        self._bn1 = nn.BatchNorm2d(self.out_dim, momentum=0.01, eps=1e-3)
        self._fc = nn.Linear(self.out_dim, num_classes)

    def forward(self, x):
        x = F.silu(self._bn0(self._conv_stem(x)))
        for b in self._blocks:
            x = b(x)
        x = F.silu(self._bn1(self._conv_head(x)))
        return x.mean(dim=[2, 3])


# ------------------------------------------------------------------------------------------------------
# text tower (BERT-base shaped) + offline tokenizer
# ------------------------------------------------------------------------------------------------------
class HashTokenizer:
    """Deterministic offline stand-in for BertTokenizerFast('emilyalsentzer/Bio_ClinicalBERT') (whose vocab
    is not in the container): lower-cased whitespace/punctuation split, ids = 1000 + crc32(word) % 27000,
    [CLS]=101 ... [SEP]=102, padding 0.  Returns the same dict the reference's tokenize() returns."""
    vocab_size = 28996

    def __call__(self, texts, max_length=256, padding=True, truncation=True, return_tensors="pt"):
        rows = []
        for t in texts:
            words = "".join(ch if ch.isalnum() else " " for ch in t.lower()).split()
            ids = [101] + [1000 + zlib.crc32(w.encode()) % 27000 for w in words]
            ids = ids[:max_length - 1] + [102]
            rows.append(ids)
        L = max(len(r) for r in rows)
        input_ids = torch.zeros(len(rows), L, dtype=torch.long)
        mask = torch.zeros(len(rows), L, dtype=torch.long)
        for i, r in enumerate(rows):
            input_ids[i, :len(r)] = torch.tensor(r)
            mask[i, :len(r)] = 1
        return {"input_ids": input_ids, "attention_mask": mask, "token_type_ids": torch.zeros_like(input_ids)}


class TextTower(nn.Module):
    def __init__(self, vocab=28996, dim=768, depth=12, heads=12, mlp=3072, max_pos=512):
        super().__init__()
        self.out_dim = dim
        self.word = nn.Embedding(vocab, dim)
        self.pos = nn.Embedding(max_pos, dim)
        self.norm = nn.LayerNorm(dim, eps=1e-12)
        self.encoder = _Encoder(depth, dim, heads, mlp, "layer")

    def forward(self, tokens):
        ids, mask = tokens["input_ids"], tokens["attention_mask"]
        x = self.norm(self.word(ids) + self.pos(torch.arange(ids.shape[1], device=ids.device))[None])
        attn = mask[:, None, None, :].bool()
        return self.encoder(x, attn)


class LinearProjectionHead(nn.Module):
    def __init__(self, in_dim, proj_dim):
        super().__init__()
        self.projection = nn.Linear(in_dim, proj_dim, bias=False)

    def forward(self, x):
        return self.projection(x)


# ------------------------------------------------------------------------------------------------------
# Mammo-CLIP ("BreastClip") shaped model: same public surface as reference model/clip.py:12-137
# ------------------------------------------------------------------------------------------------------
class BreastClip(nn.Module):
    """encode_image / encode_text / tokenize / image_projection / text_projection / projection, as the
    reference's utils.py:315-414 uses them.  image tower: 'cnn' = EfficientNet-B5 (the shipped Mammo-CLIP),
    'vit' = ViT-B/16 (reference model/modules/image_encoder.py:14-52, CLS token model/clip.py:49-52)."""

    def __init__(self, image_tower="cnn", image_size=224, text_depth=12):
        super().__init__()
        self.model_type = image_tower
        if image_tower == "cnn":
            self.image_encoder = EfficientNetB5Tower()
        else:
            self.image_encoder = ViTTower(image_size=image_size)
        self.text_encoder = TextTower(depth=text_depth)
        self.text_pooling = "eos"
        self.projection = True
        self.image_projection = LinearProjectionHead(self.image_encoder.out_dim, PROJ_DIM)
        self.text_projection = LinearProjectionHead(self.text_encoder.out_dim, PROJ_DIM)
        self.tokenizer = HashTokenizer()

    def encode_image(self, image):
        f = self.image_encoder(image)
        return f if self.model_type == "cnn" else f[:, 0]

    def encode_text(self, text_tokens):
        if not isinstance(text_tokens, dict):
            raise ValueError("Text tokens must be a dictionary")
        f = self.text_encoder(text_tokens)
        eos = text_tokens["attention_mask"].sum(dim=-1) - 1  # 'eos' pooling, model/clip.py:66-69
        return f[torch.arange(f.shape[0], device=f.device), eos]

    def tokenize(self, texts, max_length=256, padding=True, truncation=True):
        if isinstance(texts, str):
            texts = [texts]
        return self.tokenizer(texts, max_length=max_length, padding=padding, truncation=truncation)


class BreastClipClassifier(nn.Module):
    """Fine-tuned classifier target (reference Classifiers/models/breast_clip_classifier.py:6-81):
    EfficientNet-B5 image encoder + linear head; forward(images) -> logits [B, n_class]."""

    def __init__(self, n_class=1):
        super().__init__()
        self.image_encoder = EfficientNetB5Tower()
        self.classifier = nn.Linear(self.image_encoder.out_dim, n_class)

    def encode_image(self, image):
        return self.image_encoder(image)

    def forward(self, images):
        return self.classifier(self.image_encoder(images))


class ClipViT(nn.Module):
    """OpenAI-CLIP ViT-B/16 shaped dissector/target: hook points vision_model.encoder.layers[i]."""

    def __init__(self, image_size=224, text_depth=12):
        super().__init__()
        self.vision_model = ViTTower(image_size=image_size, list_name="layers")
        self.visual_projection = nn.Linear(768, PROJ_DIM, bias=False)
        self.text_model = TextTower(vocab=49408, dim=512, heads=8, mlp=2048, depth=text_depth, max_pos=77)
        self.text_projection = nn.Linear(512, PROJ_DIM, bias=False)

    def encode_image(self, image):
        return self.visual_projection(self.vision_model(image)[:, 0])

    def encode_text(self, tokens):
        f = self.text_model(tokens)
        eos = tokens["attention_mask"].sum(dim=-1) - 1
        return self.text_projection(f[torch.arange(f.shape[0], device=f.device), eos])

    def forward(self, image):
        return self.encode_image(image)


# ------------------------------------------------------------------------------------------------------
# ResNet-50 (torchvision layout: conv1, bn1, layer1..4, fc)
# ------------------------------------------------------------------------------------------------------
class _Bottleneck(nn.Module):
    def __init__(self, cin, width, stride):
        super().__init__()
        cout = width * 4
        self.conv1 = nn.Conv2d(cin, width, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(width)
        self.conv2 = nn.Conv2d(width, width, 3, stride, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(width)
        self.conv3 = nn.Conv2d(width, cout, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(cout)
        self.downsample = None
        if stride != 1 or cin != cout:
            self.downsample = nn.Sequential(nn.Conv2d(cin, cout, 1, stride, bias=False), nn.BatchNorm2d(cout))

    def forward(self, x):
        y = F.relu(self.bn1(self.conv1(x)))
        y = F.relu(self.bn2(self.conv2(y)))
        y = self.bn3(self.conv3(y))
        return F.relu(y + (x if self.downsample is None else self.downsample(x)))


class ResNet50(nn.Module):
    def __init__(self, num_classes=1000):
        super().__init__()
        self.conv1 = nn.Conv2d(3, 64, 7, 2, 3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        cin = 64
        for i, (w, n, s) in enumerate([(64, 3, 1), (128, 4, 2), (256, 6, 2), (512, 3, 2)], 1):
            blocks = []
            for j in range(n):
                blocks.append(_Bottleneck(cin, w, s if j == 0 else 1))
                cin = w * 4
            setattr(self, "layer%d" % i, nn.Sequential(*blocks))
        self.fc = nn.Linear(cin, num_classes)

    def forward(self, x):
        x = F.max_pool2d(F.relu(self.bn1(self.conv1(x))), 3, 2, 1)
        x = self.layer4(self.layer3(self.layer2(self.layer1(x))))
        return self.fc(x.mean(dim=[2, 3]))

    encode_image = forward  # describe_og_neurons.py calls encode_image on every target


# ------------------------------------------------------------------------------------------------------
# factories
# ------------------------------------------------------------------------------------------------------
def _load_local(model, ckpt):
    """ckpt: None, a state dict, {'model': state dict} (reference utils.py:451-455), or a LOCAL path
    (loaded with weights_only=True: nothing from the file is executed)."""
    if ckpt is None:
        return model
    if isinstance(ckpt, str):
        ckpt = torch.load(ckpt, map_location="cpu", weights_only=True)
    sd = ckpt["model"] if isinstance(ckpt, dict) and "model" in ckpt else ckpt
    model.load_state_dict(sd, strict=False)
    return model


def get_target_model(target_name, device, args=None, ckpt=None, n_class=None, finetuned_ckpt=None, seed=0, image_size=224):
    """Returns (target model in eval mode, preprocess) -- reference data_utils.py:38-93.  Weights are
    random-init under `seed` unless a local checkpoint is given; nothing is downloaded.
    image_size: input resolution of the ViT towers (position embeddings are sized for it; the reference's HF ViT is
    built for its checkpoint's resolution the same way); also accepted as a suffix, 'breastclip_vit_1024'.  Beyond
    256 tokens the attention takes PyTorch's SDPA (K9 covers T <= 256)."""
    if target_name.startswith("breastclip_vit_") and target_name[len("breastclip_vit_"):].isdigit():
        image_size = int(target_name[len("breastclip_vit_"):])
        target_name = "breastclip_vit"
    with torch.random.fork_rng(devices=[]):
        torch.manual_seed(seed)
        if target_name == "breastclip":
            model = BreastClip("cnn")
        elif target_name == "breastclip_vit":
            model = BreastClip("vit", image_size=image_size)
        elif target_name == "breastclip_classifier":
            if n_class is None:
                raise ValueError("Arguments `args`, `ckpt`, and `n_class` must be provided for BreastClipClassifier.")
            model = BreastClipClassifier(n_class=n_class)
        elif target_name == "clip":
            model = ClipViT(image_size=image_size)
        elif target_name == "resnet50":
            model = ResNet50()
        else:
            raise ValueError("unknown target model %r (offline build: breastclip, breastclip_vit, "
                             "breastclip_classifier, clip, resnet50)" % (target_name,))
    _load_local(model, ckpt)
    _load_local(model, finetuned_ckpt)
    return model.to(device).eval(), None


class SyntheticImages(torch.utils.data.Dataset):
    """D_probe stand-in: image i is randn(3,H,W) from (seed, i) -- the same image on every rank/shard.
    Items follow the reference datasets: ((image, label)) tuples (reference data_utils.py:102-311)."""

    def __init__(self, n, size=224, seed=1234):
        self.n, self.size, self.seed = n, size, seed

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        g = torch.Generator().manual_seed(self.seed * 1000003 + i)
        return torch.randn(3, self.size, self.size, generator=g), 0

    def on_device(self, device, lo=0, hi=None):
        """The same kind of probe set generated ON the device and kept resident in HBM (images lo..hi-1)."""
        return DeviceSyntheticImages(self.n, self.size, self.seed, device, lo, self.n if hi is None else hi)


class DeviceSyntheticImages(torch.utils.data.Dataset):
    """Synthetic probe images generated on the GPU and resident in HBM: image i is randn(3,H,W) of a device generator
    seeded with (seed, i), so it is the same image whatever the batch size, the shard or the rank (not the same VALUES
    as the CPU SyntheticImages: the two generators are different algorithms).  The extraction loop slices batches
    straight out of the resident tensor (no DataLoader, no host round trip) -- 6 GB for 10 000 images of 224 x 224, of
    the 288 GB of an MI355X.  Holds images lo..hi-1 of the n (this rank's shard)."""

    def __init__(self, n, size, seed, device, lo=0, hi=None):
        self.n_total, self.size, self.seed = int(n), int(size), int(seed)
        self.lo, self.hi = int(lo), int(self.n_total if hi is None else hi)
        self.device = torch.device(device)
        self._images = None

    def __len__(self):
        return self.hi - self.lo

    def images(self):
        if self._images is None:
            g = torch.Generator(device=self.device)
            x = torch.empty((len(self), 3, self.size, self.size), dtype=torch.float32, device=self.device)
            for j in range(len(self)):
                g.manual_seed(self.seed * 1000003 + self.lo + j)
                x[j].normal_(generator=g)
            self._images = x
        return self._images

    def __getitem__(self, i):
        return self.images()[i], 0

    def device_batches(self, batch_size):
        x = self.images()
        for i in range(0, x.shape[0], batch_size):
            yield x[i:i + batch_size]


def get_data(dataset_name, preprocess=None, device=None, lo=0, hi=None):
    """'synthetic_<N>' or 'synthetic_<N>_<size>' (e.g. synthetic_10000_224).  Real datasets are not in the
    container (reference data_utils.py:102-311 reads VinDr/CSAW/EMBED/ImageNet paths).  With a CUDA `device` the
    probe set is generated on the device and stays resident there (DeviceSyntheticImages); lo/hi select a rank's
    shard of it."""
    if dataset_name.startswith("synthetic"):
        parts = dataset_name.split("_")
        n = int(parts[1]) if len(parts) > 1 else 256
        size = int(parts[2]) if len(parts) > 2 else 224
        ds = SyntheticImages(n, size)
        if device is not None and torch.device(device).type == "cuda":
            return ds.on_device(device, lo, hi)
        if lo != 0 or (hi is not None and hi != n):
            return torch.utils.data.Subset(ds, range(lo, n if hi is None else hi))
        return ds
    raise ValueError("dataset %r is not available offline; use synthetic_<N>[_<size>]" % (dataset_name,))
