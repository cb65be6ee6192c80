"""State-dict adapters for real checkpoints.  No checkpoint can be downloaded offline, so these
are exercised in tests with randomly initialised upstream-format state dicts; the formats are:

* OpenAI CLIP (`clip.load(name)` state dict, slip.py:175): keys `visual.conv1.weight`, `visual.class_embedding`,
  `visual.positional_embedding`, `visual.ln_pre.*`, `visual.transformer.resblocks.{i}.*`, `visual.ln_post.*`,
  `visual.proj` -> strip the `visual.` prefix (weights may be fp16 on CUDA: cast to fp32).
* HF `CLIPVisionModelWithProjection` -> the same names (q/k/v projections are concatenated into `in_proj_*`).
* taming-transformers VQGAN Lightning checkpoint (`vqgan.py:124-140`): `state_dict` with `decoder.*`,
  `post_quant_conv.*`, `quantize.embedding.weight` for the decoder runner, plus `encoder.*` / `quant_conv.*` for the
  encoder runner when the checkpoint has them (`loss.*` is dropped, as `del model.loss` does at vqgan.py:139).
* torchvision `vgg16` (`models.vgg16(pretrained=True)`, Losses/StyleLoss.py:27): `features.{0,2,5,...,28}.{weight,bias}`;
  the `classifier.*` tensors are dropped (the StyleLoss extractor stops at relu5_3).
"""
from collections import OrderedDict
from typing import Dict

import torch

from .weights import (ClipTextConfig, ClipVitConfig, VqganConfig, clip_text_param_shapes, clip_vit_param_shapes,
                      vgg16_param_shapes, vqgan_encoder_param_shapes, vqgan_param_shapes)


def _check(params: Dict[str, torch.Tensor], shapes) -> "OrderedDict[str, torch.Tensor]":
    out = OrderedDict()
    for name, shape in shapes.items():
        if name not in params:
            raise KeyError(f"checkpoint is missing {name}")
        t = params[name].detach().float().contiguous()
        if tuple(t.shape) != tuple(shape):
            raise ValueError(f"{name}: checkpoint has shape {tuple(t.shape)}, the configuration needs {tuple(shape)}")
        out[name] = t
    return out


def clip_visual_from_openai(state_dict: Dict[str, torch.Tensor], cfg: ClipVitConfig):
    params = {k[len("visual."):]: v for k, v in state_dict.items() if k.startswith("visual.")}
    return _check(params, clip_vit_param_shapes(cfg))


def clip_visual_from_hf(state_dict: Dict[str, torch.Tensor], cfg: ClipVitConfig):
    sd = state_dict
    p = {"class_embedding": sd["vision_model.embeddings.class_embedding"],
         "conv1.weight": sd["vision_model.embeddings.patch_embedding.weight"],
         "positional_embedding": sd["vision_model.embeddings.position_embedding.weight"],
         "ln_pre.weight": sd["vision_model.pre_layrnorm.weight"], "ln_pre.bias": sd["vision_model.pre_layrnorm.bias"],
         "ln_post.weight": sd["vision_model.post_layernorm.weight"], "ln_post.bias": sd["vision_model.post_layernorm.bias"],
         "proj": sd["visual_projection.weight"].T}
    for i in range(cfg.layers):
        a, b = f"transformer.resblocks.{i}.", f"vision_model.encoder.layers.{i}."
        p[a + "attn.in_proj_weight"] = torch.cat([sd[b + f"self_attn.{n}_proj.weight"] for n in "qkv"], 0)
        p[a + "attn.in_proj_bias"] = torch.cat([sd[b + f"self_attn.{n}_proj.bias"] for n in "qkv"], 0)
        p[a + "attn.out_proj.weight"] = sd[b + "self_attn.out_proj.weight"]
        p[a + "attn.out_proj.bias"] = sd[b + "self_attn.out_proj.bias"]
        for (x, y) in (("ln_1", "layer_norm1"), ("ln_2", "layer_norm2"), ("mlp.c_fc", "mlp.fc1"), ("mlp.c_proj", "mlp.fc2")):
            p[a + x + ".weight"] = sd[b + y + ".weight"]
            p[a + x + ".bias"] = sd[b + y + ".bias"]
    return _check(p, clip_vit_param_shapes(cfg))


def clip_text_from_openai(state_dict: Dict[str, torch.Tensor], cfg: ClipTextConfig):
    """text side of an OpenAI CLIP state dict: the un-prefixed entries (`token_embedding.weight`, `positional_embedding`,
    `transformer.resblocks.*`, `ln_final.*`, `text_projection`)"""
    return _check(dict(state_dict), clip_text_param_shapes(cfg))


def clip_text_from_hf(state_dict: Dict[str, torch.Tensor], cfg: ClipTextConfig):
    sd = state_dict
    p = {"token_embedding.weight": sd["text_model.embeddings.token_embedding.weight"],
         "positional_embedding": sd["text_model.embeddings.position_embedding.weight"],
         "ln_final.weight": sd["text_model.final_layer_norm.weight"], "ln_final.bias": sd["text_model.final_layer_norm.bias"],
         "text_projection": sd["text_projection.weight"].T}
    for i in range(cfg.layers):
        a, b = f"transformer.resblocks.{i}.", f"text_model.encoder.layers.{i}."
        p[a + "attn.in_proj_weight"] = torch.cat([sd[b + f"self_attn.{n}_proj.weight"] for n in "qkv"], 0)
        p[a + "attn.in_proj_bias"] = torch.cat([sd[b + f"self_attn.{n}_proj.bias"] for n in "qkv"], 0)
        p[a + "attn.out_proj.weight"] = sd[b + "self_attn.out_proj.weight"]
        p[a + "attn.out_proj.bias"] = sd[b + "self_attn.out_proj.bias"]
        for (x, y) in (("ln_1", "layer_norm1"), ("ln_2", "layer_norm2"), ("mlp.c_fc", "mlp.fc1"), ("mlp.c_proj", "mlp.fc2")):
            p[a + x + ".weight"] = sd[b + y + ".weight"]
            p[a + x + ".bias"] = sd[b + y + ".bias"]
    return _check(p, clip_text_param_shapes(cfg))


def vqgan_from_taming(state_dict: Dict[str, torch.Tensor], cfg: VqganConfig, with_encoder: bool = None):
    """-> one ordered dict usable as `settings.vqgan_state_dict`: the decoder entries, and the encoder entries when the
    checkpoint has them (`with_encoder=None`: if present; True: required; False: dropped)."""
    sd = dict(state_dict.get("state_dict", state_dict))
    out = _check(sd, vqgan_param_shapes(cfg))
    if with_encoder is None:
        with_encoder = any(k.startswith("encoder.") for k in sd)
    if with_encoder:
        out.update(_check(sd, vqgan_encoder_param_shapes(cfg)))
    return out


TAMING_TARGETS = {"taming.models.vqgan.VQModel": ("", False), "taming.models.vqgan.GumbelVQ": ("", True),
                  "taming.models.cond_transformer.Net2NetTransformer": ("first_stage_model.", False)}


def vqgan_config_from_taming_yaml(path_or_dict):
    """taming's model yaml (`model.target` + `model.params.{embed_dim, n_embed, ddconfig}`; what `OmegaConf.load` reads at
    vqgan.py:121) -> (VqganConfig, key prefix of the first-stage model inside the checkpoint, gumbel flag).  For a
    Net2NetTransformer config the first stage is `params.first_stage_config` (vqgan.py:131-135)."""
    import yaml
    doc = path_or_dict
    if not isinstance(doc, dict):
        with open(path_or_dict) as f:
            doc = yaml.safe_load(f)
    model = doc["model"]
    target = model["target"]
    if target not in TAMING_TARGETS:
        raise ValueError(f"unknown model type: {target}")                     # vqgan.py:136-137
    prefix, gumbel = TAMING_TARGETS[target]
    prm = model["params"]
    if prefix:                                                                # the transformer wraps a VQModel config
        prm = prm["first_stage_config"]["params"]
    dd = prm["ddconfig"]
    if dd.get("double_z", False):
        raise ValueError("ddconfig.double_z=True is a KL autoencoder, not a VQGAN")
    cfg = VqganConfig(ch=int(dd["ch"]), ch_mult=tuple(int(m) for m in dd["ch_mult"]), num_res_blocks=int(dd["num_res_blocks"]),
                      attn_resolutions=tuple(int(r) for r in dd["attn_resolutions"]), resolution=int(dd["resolution"]),
                      z_channels=int(dd["z_channels"]), embed_dim=int(prm["embed_dim"]), n_embed=int(prm["n_embed"]),
                      out_ch=int(dd.get("out_ch", 3)))
    if int(dd.get("in_channels", 3)) != 3 or cfg.out_ch != 3:
        raise ValueError("only RGB VQGANs are supported (ddconfig.in_channels / out_ch must be 3)")
    return cfg, prefix, gumbel


def load_taming(config_path: str, checkpoint_path: str, with_encoder: bool = None):
    """The reference's `load_model` (vqgan.py:96-140) without the download and without taming: yaml ddconfig -> VqganConfig,
    Lightning `.ckpt` (`{"state_dict": ...}` with `loss.*` discriminator entries that `del model.loss` drops) -> the ordered
    parameter dict of the HIP runner.  GumbelVQ keeps its codebook under `quantize.embed.weight` (vqgan.py:193): renamed to
    the runner's `quantize.embedding.weight`.  -> (cfg, params, gumbel)"""
    cfg, prefix, gumbel = vqgan_config_from_taming_yaml(config_path)
    blob = torch.load(checkpoint_path, map_location="cpu", weights_only=False)
    sd = blob.get("state_dict", blob)
    if prefix:
        sd = {k[len(prefix):]: v for k, v in sd.items() if k.startswith(prefix)}
    sd = {k: v for k, v in sd.items() if not k.startswith("loss.")}
    if gumbel and "quantize.embed.weight" in sd:
        sd["quantize.embedding.weight"] = sd.pop("quantize.embed.weight")
    out = vqgan_from_taming(sd, cfg, with_encoder)
    if gumbel:
        # GumbelQuantize's logits projection (a 1x1 convolution z_channels -> n_embed): not on the decode path; kept so that
        # VqganDrawer can ENCODE an image the way GumbelVQ.encode does (init_image / overlay / z labels, vqgan.py:174-185)
        for k in ("quantize.proj.weight", "quantize.proj.bias"):
            if k in sd:
                out[k] = sd[k].detach().float().contiguous()
    return cfg, out, gumbel


def clip_state_dict_from_archive(path: str) -> Dict[str, torch.Tensor]:
    """What `clip.load(name, download_root="models")` reads (slip.py:175): OpenAI publishes every CLIP model as a TorchScript
    archive (`ViT-B-32.pt`, ...), and `clip.load` takes `torch.jit.load(path).state_dict()` from it; a plain pickled state dict
    (or `{"state_dict": ...}`) is accepted as well, as `clip.load` does when the file is not an archive.  The result feeds
    `clip_visual_from_openai` / `clip_text_from_openai` (fp16 tensors are cast to fp32 there)."""
    try:
        module = torch.jit.load(path, map_location="cpu")
        sd = module.state_dict()
    except RuntimeError:
        blob = torch.load(path, map_location="cpu", weights_only=False)
        sd = blob.get("state_dict", blob) if isinstance(blob, dict) else blob.state_dict()
    return {k: v.detach() for k, v in sd.items()}


def clip_config_from_state_dict(sd: Dict[str, torch.Tensor], name: str = "checkpoint") -> ClipVitConfig:
    """`clip.model.build_model` reads the ViT geometry off the tensors' shapes; so does this (visual tower of the ViT family)."""
    if "visual.proj" not in sd:
        raise ValueError("not a CLIP ViT state dict (no visual.proj): ModifiedResNet towers are configured by name")
    width = sd["visual.conv1.weight"].shape[0]
    patch = sd["visual.conv1.weight"].shape[-1]
    layers = len({k.split(".")[3] for k in sd if k.startswith("visual.transformer.resblocks.")})
    grid = round((sd["visual.positional_embedding"].shape[0] - 1) ** 0.5)
    return ClipVitConfig(name, patch * grid, patch, width, layers, width // 64, sd["visual.proj"].shape[1])


def vgg16_from_torchvision(state_dict: Dict[str, torch.Tensor]):
    """torchvision VGG16 state dict (optionally wrapped as {"state_dict": ...} or prefixed `module.`) -> the 26 tensors the
    StyleLoss extractor needs (`pixray_amd.style_loss.Vgg16Extractor(params=...)`)"""
    sd = state_dict.get("state_dict", state_dict)
    params = {(k[len("module."):] if k.startswith("module.") else k): v for k, v in sd.items()}
    return _check(params, vgg16_param_shapes())
