"""Model configurations and parameter containers for the two frozen networks on the hot path.

No checkpoints exist offline (SURVEY.md §7 "No weights, no network"), so `synthetic_*` build
seeded random parameters with the REAL architectures' shapes (FLOPs and bytes identical to the
reference's `imagenet_f16_16384` VQGAN, vqgan.py:86, and OpenAI CLIP ViT-B/32).  Parameter names
and orders are the upstream state-dict ones, so a real checkpoint's state dict drops in
(`decoder.*` / `post_quant_conv.*` / `quantize.embedding.weight`; `visual.*`).
"""
from __future__ import annotations

import math
from collections import OrderedDict
from dataclasses import dataclass
from typing import Tuple

import torch


# --------------------------------------------------------------------------- VQGAN (taming) config
@dataclass
class VqganConfig:
    ch: int = 128
    ch_mult: Tuple[int, ...] = (1, 1, 2, 2, 4)
    num_res_blocks: int = 2
    attn_resolutions: Tuple[int, ...] = (16,)
    resolution: int = 256
    z_channels: int = 256
    embed_dim: int = 256
    n_embed: int = 16384
    out_ch: int = 3

    @property
    def num_resolutions(self) -> int:  # DrawingInterface.get_num_resolutions (vqgan.py:187-188)
        return len(self.ch_mult)

    def oracle_cfg(self) -> dict:
        return dict(ch=self.ch, ch_mult=self.ch_mult, num_res_blocks=self.num_res_blocks,
                    attn_resolutions=self.attn_resolutions, resolution=self.resolution,
                    z_channels=self.z_channels, out_ch=self.out_ch)


VQGAN_CONFIGS = {
    "imagenet_f16_16384": VqganConfig(),
    # reduced graph with the same operator mix, for fast parity tests
    "tiny_f4": VqganConfig(ch=128, ch_mult=(1, 2, 2), num_res_blocks=1, attn_resolutions=(16,), resolution=64,
                           z_channels=128, embed_dim=128, n_embed=512),
}


def vqgan_param_shapes(cfg: VqganConfig) -> "OrderedDict[str, Tuple[int, ...]]":
    """Ordered (name -> shape) list in the order the C ABI expects (include/prx.h, prx_vqgan_create)."""
    sh: "OrderedDict[str, Tuple[int, ...]]" = OrderedDict()

    def conv(name, cout, cin, k):
        sh[name + ".weight"] = (cout, cin, k, k)
        sh[name + ".bias"] = (cout,)

    def norm(name, c):
        sh[name + ".weight"] = (c,)
        sh[name + ".bias"] = (c,)

    def res(name, cin, cout):
        norm(name + ".norm1", cin); conv(name + ".conv1", cout, cin, 3)
        norm(name + ".norm2", cout); conv(name + ".conv2", cout, cout, 3)
        if cin != cout:
            conv(name + ".nin_shortcut", cout, cin, 1)

    def attn(name, c):
        norm(name + ".norm", c)
        for t in ("q", "k", "v", "proj_out"):
            conv(name + "." + t, c, c, 1)

    sh["quantize.embedding.weight"] = (cfg.n_embed, cfg.embed_dim)
    conv("post_quant_conv", cfg.z_channels, cfg.embed_dim, 1)
    nres = len(cfg.ch_mult)
    block_in = cfg.ch * cfg.ch_mult[-1]
    curr_res = cfg.resolution // 2 ** (nres - 1)
    conv("decoder.conv_in", block_in, cfg.z_channels, 3)
    res("decoder.mid.block_1", block_in, block_in)
    attn("decoder.mid.attn_1", block_in)
    res("decoder.mid.block_2", block_in, block_in)
    for lvl in reversed(range(nres)):
        block_out = cfg.ch * cfg.ch_mult[lvl]
        for b in range(cfg.num_res_blocks + 1):
            res(f"decoder.up.{lvl}.block.{b}", block_in, block_out)
            block_in = block_out
            if curr_res in cfg.attn_resolutions:
                attn(f"decoder.up.{lvl}.attn.{b}", block_in)
        if lvl != 0:
            conv(f"decoder.up.{lvl}.upsample.conv", block_in, block_in, 3)
            curr_res *= 2
    norm("decoder.norm_out", block_in)
    conv("decoder.conv_out", cfg.out_ch, block_in, 3)
    return sh


def vqgan_encoder_param_shapes(cfg: VqganConfig, in_channels: int = 3) -> "OrderedDict[str, Tuple[int, ...]]":
    """Ordered (name -> shape) list of taming's Encoder + quant_conv (+ the codebook), in the order the C ABI expects
    (include/prx.h, prx_vqgan_enc_create).  Names are the taming state-dict keys (`encoder.*`, `quant_conv.*`)."""
    sh: "OrderedDict[str, Tuple[int, ...]]" = OrderedDict()

    def conv(name, cout, cin, k):
        sh[name + ".weight"] = (cout, cin, k, k)
        sh[name + ".bias"] = (cout,)

    def norm(name, c):
        sh[name + ".weight"] = (c,)
        sh[name + ".bias"] = (c,)

    def res(name, cin, cout):
        norm(name + ".norm1", cin); conv(name + ".conv1", cout, cin, 3)
        norm(name + ".norm2", cout); conv(name + ".conv2", cout, cout, 3)
        if cin != cout:
            conv(name + ".nin_shortcut", cout, cin, 1)

    def attn(name, c):
        norm(name + ".norm", c)
        for t in ("q", "k", "v", "proj_out"):
            conv(name + "." + t, c, c, 1)

    sh["quantize.embedding.weight"] = (cfg.n_embed, cfg.embed_dim)
    conv("encoder.conv_in", cfg.ch, in_channels, 3)
    nres = len(cfg.ch_mult)
    block_in = cfg.ch
    curr_res = cfg.resolution
    for lvl in range(nres):
        block_out = cfg.ch * cfg.ch_mult[lvl]
        for b in range(cfg.num_res_blocks):
            res(f"encoder.down.{lvl}.block.{b}", block_in, block_out)
            block_in = block_out
            if curr_res in cfg.attn_resolutions:
                attn(f"encoder.down.{lvl}.attn.{b}", block_in)
        if lvl != nres - 1:
            conv(f"encoder.down.{lvl}.downsample.conv", block_in, block_in, 3)
            curr_res //= 2
    res("encoder.mid.block_1", block_in, block_in)
    attn("encoder.mid.attn_1", block_in)
    res("encoder.mid.block_2", block_in, block_in)
    norm("encoder.norm_out", block_in)
    conv("encoder.conv_out", cfg.z_channels, block_in, 3)
    conv("quant_conv", cfg.embed_dim, cfg.z_channels, 1)
    return sh


def _init_conv_like(name, shape, g):
    if name == "quantize.embedding.weight":
        return torch.randn(shape, generator=g)
    if name.endswith(".weight") and len(shape) == 1:      # GroupNorm gamma
        return 1.0 + 0.1 * torch.randn(shape, generator=g)
    if name.endswith(".bias") and (".norm" in name or "norm_out" in name):
        return 0.05 * torch.randn(shape, generator=g)
    if name.endswith(".bias"):
        return 0.02 * torch.randn(shape, generator=g)
    fan_in = shape[1] * shape[2] * shape[3]
    gain = 1.0
    if name.endswith("conv2.weight") or name.endswith("proj_out.weight"):
        gain = 0.5       # residual branches: keep the stream O(1) through 20 blocks
    if name.endswith("conv_out.weight"):
        gain = 0.7
    return torch.randn(shape, generator=g) * (gain / math.sqrt(fan_in))


def synthetic_vqgan_encoder_params(cfg: VqganConfig, seed: int = 0, codebook: torch.Tensor = None,
                                   in_channels: int = 3) -> "OrderedDict[str, torch.Tensor]":
    """Seeded random Encoder/quant_conv weights.  `codebook`: reuse the decoder's `quantize.embedding.weight` (a taming
    checkpoint has ONE codebook shared by encode and decode)."""
    g = torch.Generator().manual_seed(seed + 7919)
    out: "OrderedDict[str, torch.Tensor]" = OrderedDict()
    for name, shape in vqgan_encoder_param_shapes(cfg, in_channels).items():
        if name == "quantize.embedding.weight" and codebook is not None:
            out[name] = codebook
            continue
        out[name] = _init_conv_like(name, shape, g)
    return out


def synthetic_vqgan_params(cfg: VqganConfig, seed: int = 0) -> "OrderedDict[str, torch.Tensor]":
    g = torch.Generator().manual_seed(seed)
    out: "OrderedDict[str, torch.Tensor]" = OrderedDict()
    for name, shape in vqgan_param_shapes(cfg).items():
        if name == "quantize.embedding.weight":
            t = torch.randn(shape, generator=g)
        elif name.endswith(".weight") and len(shape) == 1:      # GroupNorm gamma
            t = 1.0 + 0.1 * torch.randn(shape, generator=g)
        elif name.endswith(".bias") and (".norm" in name or "norm_out" in name):
            t = 0.05 * torch.randn(shape, generator=g)
        elif name.endswith(".bias"):
            t = 0.02 * torch.randn(shape, generator=g)
        else:
            fan_in = shape[1] * shape[2] * shape[3]
            gain = 1.0
            if name.endswith("conv2.weight") or name.endswith("proj_out.weight"):
                gain = 0.5       # residual branches: keep the stream O(1) through 20 blocks
            if name.endswith("conv_out.weight"):
                gain = 0.7
            t = torch.randn(shape, generator=g) * (gain / math.sqrt(fan_in))
        out[name] = t
    return out


# --------------------------------------------------------------------------- CLIP ViT config
@dataclass
class ClipVitConfig:
    name: str = "ViT-B/32"
    input_resolution: int = 224
    patch_size: int = 32
    width: int = 768
    layers: int = 12
    heads: int = 12
    output_dim: int = 512

    @property
    def tokens(self) -> int:
        return (self.input_resolution // self.patch_size) ** 2 + 1


CLIP_CONFIGS = {
    "ViT-B/32": ClipVitConfig(),
    "ViT-B/16": ClipVitConfig("ViT-B/16", 224, 16, 768, 12, 12, 512),
    "ViT-L/14": ClipVitConfig("ViT-L/14", 224, 14, 1024, 24, 16, 768),
    # reduced tower (same operators, 2 layers) for fast parity tests
    "tiny-B/32": ClipVitConfig("tiny-B/32", 224, 32, 256, 2, 4, 128),
}


def clip_vit_param_shapes(cfg: ClipVitConfig) -> "OrderedDict[str, Tuple[int, ...]]":
    w = cfg.width
    sh: "OrderedDict[str, Tuple[int, ...]]" = OrderedDict()
    sh["conv1.weight"] = (w, 3, cfg.patch_size, cfg.patch_size)
    sh["class_embedding"] = (w,)
    sh["positional_embedding"] = (cfg.tokens, w)
    sh["ln_pre.weight"] = (w,); sh["ln_pre.bias"] = (w,)
    for i in range(cfg.layers):
        p = f"transformer.resblocks.{i}."
        sh[p + "ln_1.weight"] = (w,); sh[p + "ln_1.bias"] = (w,)
        sh[p + "attn.in_proj_weight"] = (3 * w, w); sh[p + "attn.in_proj_bias"] = (3 * w,)
        sh[p + "attn.out_proj.weight"] = (w, w); sh[p + "attn.out_proj.bias"] = (w,)
        sh[p + "ln_2.weight"] = (w,); sh[p + "ln_2.bias"] = (w,)
        sh[p + "mlp.c_fc.weight"] = (4 * w, w); sh[p + "mlp.c_fc.bias"] = (4 * w,)
        sh[p + "mlp.c_proj.weight"] = (w, 4 * w); sh[p + "mlp.c_proj.bias"] = (w,)
    sh["ln_post.weight"] = (w,); sh["ln_post.bias"] = (w,)
    sh["proj"] = (w, cfg.output_dim)
    return sh


def synthetic_clip_vit_params(cfg: ClipVitConfig, seed: int = 0) -> "OrderedDict[str, torch.Tensor]":
    """OpenAI's initialisation scheme (clip/model.py CLIP.initialize_parameters) with seeded draws."""
    g = torch.Generator().manual_seed(seed)
    w = cfg.width
    proj_std = (w ** -0.5) * ((2 * cfg.layers) ** -0.5)
    attn_std = w ** -0.5
    fc_std = (2 * w) ** -0.5
    out: "OrderedDict[str, torch.Tensor]" = OrderedDict()
    for name, shape in clip_vit_param_shapes(cfg).items():
        if name == "conv1.weight":
            t = torch.randn(shape, generator=g) / math.sqrt(3 * cfg.patch_size ** 2)
        elif name in ("class_embedding", "positional_embedding", "proj"):
            t = torch.randn(shape, generator=g) * (w ** -0.5)
        elif name.endswith("in_proj_weight"):
            t = torch.randn(shape, generator=g) * attn_std
        elif name.endswith("out_proj.weight") or name.endswith("c_proj.weight"):
            t = torch.randn(shape, generator=g) * proj_std
        elif name.endswith("c_fc.weight"):
            t = torch.randn(shape, generator=g) * fc_std
        elif name.endswith(".weight"):       # LayerNorm gamma
            t = 1.0 + 0.05 * torch.randn(shape, generator=g)
        else:                                # biases / LayerNorm beta
            t = 0.02 * torch.randn(shape, generator=g)
        out[name] = t
    return out


# --------------------------------------------------------------------------- CLIP text tower config
@dataclass
class ClipTextConfig:
    name: str = "ViT-B/32"
    vocab_size: int = 49408
    context_length: int = 77
    width: int = 512
    layers: int = 12
    heads: int = 8
    output_dim: int = 512


# text-side hyper-parameters of the OpenAI checkpoints (clip/model.py `build_model`: heads = width // 64)
CLIP_TEXT_CONFIGS = {
    "ViT-B/32": ClipTextConfig(),
    "ViT-B/16": ClipTextConfig("ViT-B/16"),
    "ViT-L/14": ClipTextConfig("ViT-L/14", width=768, heads=12, output_dim=768),
    "tiny-B/32": ClipTextConfig("tiny-B/32", vocab_size=1000, context_length=77, width=256, layers=2, heads=4, output_dim=128),
    "RN50x4": ClipTextConfig("RN50x4", width=640, heads=10, output_dim=640),
    "RN50": ClipTextConfig("RN50", width=512, heads=8, output_dim=1024),
    "RN101": ClipTextConfig("RN101", width=512, heads=8, output_dim=512),
}


def clip_text_param_shapes(cfg: ClipTextConfig) -> "OrderedDict[str, Tuple[int, ...]]":
    """OpenAI state-dict names of the text side, in the order the C ABI expects (include/prx.h, prx_clip_text_create)."""
    w = cfg.width
    sh: "OrderedDict[str, Tuple[int, ...]]" = OrderedDict()
    sh["token_embedding.weight"] = (cfg.vocab_size, w)
    sh["positional_embedding"] = (cfg.context_length, w)
    for i in range(cfg.layers):
        p = f"transformer.resblocks.{i}."
        sh[p + "ln_1.weight"] = (w,); sh[p + "ln_1.bias"] = (w,)
        sh[p + "attn.in_proj_weight"] = (3 * w, w); sh[p + "attn.in_proj_bias"] = (3 * w,)
        sh[p + "attn.out_proj.weight"] = (w, w); sh[p + "attn.out_proj.bias"] = (w,)
        sh[p + "ln_2.weight"] = (w,); sh[p + "ln_2.bias"] = (w,)
        sh[p + "mlp.c_fc.weight"] = (4 * w, w); sh[p + "mlp.c_fc.bias"] = (4 * w,)
        sh[p + "mlp.c_proj.weight"] = (w, 4 * w); sh[p + "mlp.c_proj.bias"] = (w,)
    sh["ln_final.weight"] = (w,); sh["ln_final.bias"] = (w,)
    sh["text_projection"] = (w, cfg.output_dim)
    return sh


def synthetic_clip_text_params(cfg: ClipTextConfig, seed: int = 0) -> "OrderedDict[str, torch.Tensor]":
    """OpenAI's initialisation scheme (clip/model.py CLIP.initialize_parameters) with seeded draws."""
    g = torch.Generator().manual_seed(seed + 104729)
    w = cfg.width
    proj_std = (w ** -0.5) * ((2 * cfg.layers) ** -0.5)
    attn_std = w ** -0.5
    fc_std = (2 * w) ** -0.5
    out: "OrderedDict[str, torch.Tensor]" = OrderedDict()
    for name, shape in clip_text_param_shapes(cfg).items():
        if name == "token_embedding.weight":
            t = torch.randn(shape, generator=g) * 0.02
        elif name == "positional_embedding":
            t = torch.randn(shape, generator=g) * 0.01
        elif name == "text_projection":
            t = torch.randn(shape, generator=g) * (w ** -0.5)
        elif name.endswith("in_proj_weight"):
            t = torch.randn(shape, generator=g) * attn_std
        elif name.endswith("out_proj.weight") or name.endswith("c_proj.weight"):
            t = torch.randn(shape, generator=g) * proj_std
        elif name.endswith("c_fc.weight"):
            t = torch.randn(shape, generator=g) * fc_std
        elif name.endswith(".weight"):       # LayerNorm gamma
            t = 1.0 + 0.05 * torch.randn(shape, generator=g)
        else:                                # biases / LayerNorm beta
            t = 0.02 * torch.randn(shape, generator=g)
        out[name] = t
    return out


# --------------------------------------------------------------------------- CLIP ModifiedResNet config (RN50x4, ...)
@dataclass
class ClipResNetConfig:
    name: str = "RN50x4"
    input_resolution: int = 288
    width: int = 80
    layers: Tuple[int, ...] = (4, 6, 10, 6)
    heads: int = 40                 # width * 32 // 64
    output_dim: int = 640

    @property
    def embed_dim(self) -> int:     # channels entering the attention pool
        return self.width * 32

    @property
    def final_grid(self) -> int:
        return self.input_resolution // 32


CLIP_RESNET_CONFIGS = {
    "RN50x4": ClipResNetConfig(),
    "RN50": ClipResNetConfig("RN50", 224, 64, (3, 4, 6, 3), 32, 1024),
    "RN101": ClipResNetConfig("RN101", 224, 64, (3, 4, 23, 3), 32, 512),      # quality `supreme` names it (pixray.py:1830)
    # reduced tower with the same operator mix (stem, stride-1 and stride-2 bottlenecks, attention pool)
    "tiny-RN": ClipResNetConfig("tiny-RN", 96, 16, (1, 2, 1, 1), 8, 64),
}


def clip_resnet_param_shapes(cfg: ClipResNetConfig) -> "OrderedDict[str, Tuple[int, ...]]":
    """OpenAI `visual.*` state-dict names of a ModifiedResNet (prefix stripped), in definition order."""
    sh: "OrderedDict[str, Tuple[int, ...]]" = OrderedDict()
    w = cfg.width

    def bn(name, c):
        for k in ("weight", "bias", "running_mean", "running_var"):
            sh[f"{name}.{k}"] = (c,)

    sh["conv1.weight"] = (w // 2, 3, 3, 3); bn("bn1", w // 2)
    sh["conv2.weight"] = (w // 2, w // 2, 3, 3); bn("bn2", w // 2)
    sh["conv3.weight"] = (w, w // 2, 3, 3); bn("bn3", w)
    inplanes = w
    for li, nblocks in enumerate(cfg.layers):
        planes = w * 2 ** li
        for b in range(nblocks):
            stride = 2 if (li > 0 and b == 0) else 1
            pre = f"layer{li + 1}.{b}"
            sh[pre + ".conv1.weight"] = (planes, inplanes, 1, 1); bn(pre + ".bn1", planes)
            sh[pre + ".conv2.weight"] = (planes, planes, 3, 3); bn(pre + ".bn2", planes)
            sh[pre + ".conv3.weight"] = (planes * 4, planes, 1, 1); bn(pre + ".bn3", planes * 4)
            if stride > 1 or inplanes != planes * 4:
                sh[pre + ".downsample.0.weight"] = (planes * 4, inplanes, 1, 1); bn(pre + ".downsample.1", planes * 4)
            inplanes = planes * 4
    C = cfg.embed_dim
    sh["attnpool.positional_embedding"] = (cfg.final_grid ** 2 + 1, C)
    for n_ in ("k_proj", "q_proj", "v_proj"):
        sh[f"attnpool.{n_}.weight"] = (C, C); sh[f"attnpool.{n_}.bias"] = (C,)
    sh["attnpool.c_proj.weight"] = (cfg.output_dim, C); sh["attnpool.c_proj.bias"] = (cfg.output_dim,)
    return sh


def synthetic_clip_resnet_params(cfg: ClipResNetConfig, seed: int = 0) -> "OrderedDict[str, torch.Tensor]":
    """Seeded random weights; BatchNorm running statistics are non-trivial so that the fold is exercised."""
    g = torch.Generator().manual_seed(seed + 31337)
    out: "OrderedDict[str, torch.Tensor]" = OrderedDict()
    for name, shape in clip_resnet_param_shapes(cfg).items():
        if name.endswith("running_mean"):
            t = 0.1 * torch.randn(shape, generator=g)
        elif name.endswith("running_var"):
            t = 0.5 + torch.rand(shape, generator=g)
        elif ".bn" in name or name.startswith("bn") or "downsample.1" in name:
            t = (1.0 + 0.1 * torch.randn(shape, generator=g)) if name.endswith("weight") else 0.05 * torch.randn(shape, generator=g)
            if name.endswith("bn3.weight") and name.startswith("layer"):
                # residual-branch gain 0.2: with 26 bottlenecks the variance of the residual stream then grows 1.04^26 = 2.8x,
                # as in a BatchNorm-trained tower.  (Round 1 used 0.5: 1.25^26 = 330x, which saturated the attention pool's
                # softmax and made the random tower ill-conditioned -- rounding its WEIGHTS to bf16 alone moved the embedding by
                # 1.7-3 % depending on the seed, against 0.4 % now.)
                t = t * 0.2
        elif name == "attnpool.positional_embedding":
            t = torch.randn(shape, generator=g) * (shape[1] ** -0.5)
        elif name.endswith(".bias"):
            t = 0.02 * torch.randn(shape, generator=g)
        elif len(shape) == 4:
            fan_in = shape[1] * shape[2] * shape[3]
            t = torch.randn(shape, generator=g) * math.sqrt(2.0 / fan_in)
        else:
            t = torch.randn(shape, generator=g) * (shape[1] ** -0.5)
        out[name] = t
    return out


def fold_clip_resnet_params(cfg: ClipResNetConfig, p) -> "OrderedDict[str, torch.Tensor]":
    """Frozen eval-mode BatchNorm folded into the preceding bias-free conv (w' = w * gamma / sqrt(var + eps), b' = beta -
    mean * gamma / sqrt(var + eps)); q/k/v projections of the attention pool concatenated.  Order = what
    prx_clip_resnet_create expects (include/prx.h)."""
    out: "OrderedDict[str, torch.Tensor]" = OrderedDict()

    def fold(conv, bn, key):
        w = p[conv + ".weight"].double()
        s = p[bn + ".weight"].double() / torch.sqrt(p[bn + ".running_var"].double() + 1e-5)
        out[key + ".weight"] = (w * s.view(-1, 1, 1, 1)).float()
        out[key + ".bias"] = (p[bn + ".bias"].double() - p[bn + ".running_mean"].double() * s).float()

    fold("conv1", "bn1", "stem1"); fold("conv2", "bn2", "stem2"); fold("conv3", "bn3", "stem3")
    inplanes = cfg.width
    for li, nblocks in enumerate(cfg.layers):
        planes = cfg.width * 2 ** li
        for b in range(nblocks):
            stride = 2 if (li > 0 and b == 0) else 1
            pre = f"layer{li + 1}.{b}"
            fold(pre + ".conv1", pre + ".bn1", pre + ".c1")
            fold(pre + ".conv2", pre + ".bn2", pre + ".c2")
            fold(pre + ".conv3", pre + ".bn3", pre + ".c3")
            if stride > 1 or inplanes != planes * 4:
                fold(pre + ".downsample.0", pre + ".downsample.1", pre + ".ds")
            inplanes = planes * 4
    out["attnpool.positional_embedding"] = p["attnpool.positional_embedding"].float()
    out["attnpool.in_proj_weight"] = torch.cat([p[f"attnpool.{n_}.weight"] for n_ in ("q_proj", "k_proj", "v_proj")], 0).float()
    out["attnpool.in_proj_bias"] = torch.cat([p[f"attnpool.{n_}.bias"] for n_ in ("q_proj", "k_proj", "v_proj")], 0).float()
    out["attnpool.c_proj.weight"] = p["attnpool.c_proj.weight"].float()
    out["attnpool.c_proj.bias"] = p["attnpool.c_proj.bias"].float()
    return out


# ------------------------------------------------------------------------------------------ VGG16 (StyleLoss plugin)
VGG16_CFG = (64, 64, "M", 128, 128, "M", 256, 256, 256, "M", 512, 512, 512, "M", 512, 512, 512)   # torchvision cfg "D" to relu5_3


def vgg16_param_shapes() -> "OrderedDict[str, tuple]":
    """torchvision `vgg16().features` state-dict names and shapes (conv layers 0,2,5,...,28)."""
    sh: "OrderedDict[str, tuple]" = OrderedDict()
    idx, cin = 0, 3
    for v in VGG16_CFG:
        if v == "M":
            idx += 1
            continue
        sh[f"features.{idx}.weight"] = (v, cin, 3, 3)
        sh[f"features.{idx}.bias"] = (v,)
        cin = v
        idx += 2
    return sh


def synthetic_vgg16_params(seed: int = 0) -> "OrderedDict[str, torch.Tensor]":
    """Seeded He-initialised weights (no checkpoint exists offline): activations stay O(1) through the 13 ReLU layers."""
    g = torch.Generator().manual_seed(seed + 1616)
    out: "OrderedDict[str, torch.Tensor]" = OrderedDict()
    for name, shape in vgg16_param_shapes().items():
        if name.endswith("bias"):
            out[name] = 0.05 * torch.randn(shape, generator=g)
        else:
            out[name] = torch.randn(shape, generator=g) * math.sqrt(2.0 / (shape[1] * 9))
    return out
