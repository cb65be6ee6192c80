"""ORACLE (test infrastructure only -- never imported by the product package).

CPU fp32 restatement of the CLIP visual tower that pixray's perceptor wrapper runs
(`CLIP_Base.encode_image`, /root/reference/slip.py:62-66, preprocessing slip.py:21-42,52-60).
The tower itself lives in the un-vendored dependency openai/CLIP (`clip/model.py`,
unpinned in /root/reference/requirements.txt:29); its published algorithm
(`VisionTransformer`, `ResidualAttentionBlock`, `QuickGELU`, `LayerNorm`) is restated here
from SURVEY.md Appendix A.1.  Parity status: **unpinned** by the reference's own tests (they
hold no numeric fixture for this path); pinned here against an independent implementation
(HF `CLIPVisionModelWithProjection`, tests/test_oracle_cross.py) and golden vectors made
from it (tests/golden/).

Parameters use OpenAI's state-dict names under `visual.`.
"""
from __future__ import annotations

import math
from typing import Dict

import torch
import torch.nn.functional as F

CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)  # slip.py:55
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)


def normalize_minmax(img: torch.Tensor) -> torch.Tensor:
    """slip.py:21-36 with input_range=None (forced by slip.py:64): batch-global min/max renorm."""
    minv = img.min()
    img = img - minv
    maxv = img.max()
    if maxv != 0:
        img = img / maxv
    return img


def preprocess(imgs: torch.Tensor) -> torch.Tensor:
    """slip.py:58-60: adjust_range(imgs,[0,1]) then Resize/CenterCrop (no-ops at the tower's own
    resolution, which is what MakeCutouts produces: pixray.py:643-649) then Normalize(mean,std)."""
    x = normalize_minmax(imgs)
    x = x * (1.0 - 0.0) + 0.0
    mean = torch.tensor(CLIP_MEAN, dtype=x.dtype, device=x.device).view(1, 3, 1, 1)
    std = torch.tensor(CLIP_STD, dtype=x.dtype, device=x.device).view(1, 3, 1, 1)
    return (x - mean) / std


def quick_gelu(x):
    return x * torch.sigmoid(1.702 * x)


def vit_forward(p: Dict[str, torch.Tensor], x: torch.Tensor, *, patch: int, heads: int, layers: int) -> torch.Tensor:
    """clip.model.VisionTransformer.forward [UPSTREAM]; x is the preprocessed [N,3,R,R] batch."""
    w = p["conv1.weight"]
    width = w.shape[0]
    x = F.conv2d(x, w, stride=patch)                       # [N, width, g, g]
    x = x.reshape(x.shape[0], width, -1).permute(0, 2, 1)  # [N, g*g, width]
    cls = p["class_embedding"].to(x.dtype) + torch.zeros(x.shape[0], 1, width, dtype=x.dtype, device=x.device)
    x = torch.cat([cls, x], dim=1)
    x = x + p["positional_embedding"]
    x = F.layer_norm(x, (width,), p["ln_pre.weight"], p["ln_pre.bias"], 1e-5)
    N, T, _ = x.shape
    hd = width // heads
    for i in range(layers):
        pre = f"transformer.resblocks.{i}."
        h = F.layer_norm(x, (width,), p[pre + "ln_1.weight"], p[pre + "ln_1.bias"], 1e-5)
        qkv = F.linear(h, p[pre + "attn.in_proj_weight"], p[pre + "attn.in_proj_bias"])
        q, k, v = qkv.split(width, dim=-1)
        q = q.reshape(N, T, heads, hd).permute(0, 2, 1, 3)
        k = k.reshape(N, T, heads, hd).permute(0, 2, 1, 3)
        v = v.reshape(N, T, heads, hd).permute(0, 2, 1, 3)
        att = torch.softmax((q @ k.transpose(-1, -2)) / math.sqrt(hd), dim=-1)
        o = (att @ v).permute(0, 2, 1, 3).reshape(N, T, width)
        x = x + F.linear(o, p[pre + "attn.out_proj.weight"], p[pre + "attn.out_proj.bias"])
        h = F.layer_norm(x, (width,), p[pre + "ln_2.weight"], p[pre + "ln_2.bias"], 1e-5)
        h = quick_gelu(F.linear(h, p[pre + "mlp.c_fc.weight"], p[pre + "mlp.c_fc.bias"]))
        x = x + F.linear(h, p[pre + "mlp.c_proj.weight"], p[pre + "mlp.c_proj.bias"])
    x = F.layer_norm(x[:, 0, :], (width,), p["ln_post.weight"], p["ln_post.bias"], 1e-5)
    return x @ p["proj"]


def encode_image(p, cutouts, *, patch, heads, layers, apply_preprocess=True):
    """CLIP_Base.encode_image (slip.py:62-66): preprocess -> tower -> divide by the L2 norm."""
    x = preprocess(cutouts) if apply_preprocess else cutouts
    e = vit_forward(p, x, patch=patch, heads=heads, layers=layers)
    return e / e.norm(dim=-1, keepdim=True)
