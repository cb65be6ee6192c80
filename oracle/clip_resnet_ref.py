"""CPU fp32 restatement of CLIP's ModifiedResNet visual tower (RN50x4 of BASELINE.json configs[2]) -- TEST INFRASTRUCTURE
ONLY (never imported by pixray_amd/).

Call site: `CLIP_Base.encode_image` (/root/reference/slip.py:62-66) -> `model.encode_image` -> `visual(image)`; the tower
lives in openai/CLIP (`clip/model.py`: ModifiedResNet, Bottleneck, AttentionPool2d), un-vendored and unpinned
(requirements.txt:29) [UPSTREAM].  No second implementation of this tower exists offline (HF transformers only has the
ViT variant), so this file is restated from the published architecture: **parity unpinned** (DESIGN.md §1c).

    stem:  conv3x3(3->w/2, stride 2) bn relu, conv3x3(w/2->w/2) bn relu, conv3x3(w/2->w) bn relu, avgpool 2
    layer1..4 of Bottleneck(inplanes, planes, stride) (stride 1, 2, 2, 2; expansion 4):
        out = relu(bn1(conv1x1(x)));  out = relu(bn2(conv3x3(out)));  out = avgpool(stride)(out);  out = bn3(conv1x1(out))
        identity = bn(conv1x1(avgpool(stride)(x))) when stride > 1 or inplanes != planes*4
        return relu(out + identity)
    attnpool: tokens = [mean(x), x_1..x_HW] + positional_embedding; multi-head attention with the mean token as the only
              query (separate q/k/v projections, c_proj output projection)
BatchNorm in eval mode (running statistics, eps 1e-5): the model is frozen (slip.py:176)."""
import math
from typing import Dict, Sequence

import torch
import torch.nn.functional as F


def _bn(p, pre, x):
    return F.batch_norm(x, p[pre + ".running_mean"], p[pre + ".running_var"], p[pre + ".weight"], p[pre + ".bias"], False, 0.0, 1e-5)


def _bottleneck(p, pre, x, stride):
    out = F.relu(_bn(p, pre + ".bn1", F.conv2d(x, p[pre + ".conv1.weight"])))
    out = F.relu(_bn(p, pre + ".bn2", F.conv2d(out, p[pre + ".conv2.weight"], padding=1)))
    if stride > 1:
        out = F.avg_pool2d(out, stride)
    out = _bn(p, pre + ".bn3", F.conv2d(out, p[pre + ".conv3.weight"]))
    identity = x
    if (pre + ".downsample.0.weight") in p:
        identity = F.avg_pool2d(x, stride) if stride > 1 else x
        identity = _bn(p, pre + ".downsample.1", F.conv2d(identity, p[pre + ".downsample.0.weight"]))
    return F.relu(out + identity)


def attention_pool(p, x, heads):
    N, C, H, W = x.shape
    t = x.flatten(2).permute(2, 0, 1)                                   # [HW, N, C]
    t = torch.cat([t.mean(dim=0, keepdim=True), t], dim=0)               # [HW+1, N, C]
    t = t + p["attnpool.positional_embedding"][:, None, :]
    q = F.linear(t[:1], p["attnpool.q_proj.weight"], p["attnpool.q_proj.bias"])          # [1, N, C]
    k = F.linear(t, p["attnpool.k_proj.weight"], p["attnpool.k_proj.bias"])
    v = F.linear(t, p["attnpool.v_proj.weight"], p["attnpool.v_proj.bias"])
    hd = C // heads
    T = t.shape[0]
    q = q.reshape(1, N, heads, hd).permute(1, 2, 0, 3)                   # [N, heads, 1, hd]
    k = k.reshape(T, N, heads, hd).permute(1, 2, 0, 3)
    v = v.reshape(T, N, heads, hd).permute(1, 2, 0, 3)
    att = torch.softmax((q @ k.transpose(-1, -2)) / math.sqrt(hd), dim=-1)
    o = (att @ v).permute(2, 0, 1, 3).reshape(1, N, C)
    return F.linear(o, p["attnpool.c_proj.weight"], p["attnpool.c_proj.bias"])[0]       # [N, output_dim]


def resnet_forward(p: Dict[str, torch.Tensor], x: torch.Tensor, *, layers: Sequence[int], heads: int) -> torch.Tensor:
    """clip.model.ModifiedResNet.forward; x is the preprocessed [N,3,R,R] batch -> [N, output_dim]."""
    x = F.relu(_bn(p, "bn1", F.conv2d(x, p["conv1.weight"], stride=2, padding=1)))
    x = F.relu(_bn(p, "bn2", F.conv2d(x, p["conv2.weight"], padding=1)))
    x = F.relu(_bn(p, "bn3", F.conv2d(x, p["conv3.weight"], padding=1)))
    x = F.avg_pool2d(x, 2)
    for li, nblocks in enumerate(layers):
        for b in range(nblocks):
            x = _bottleneck(p, f"layer{li + 1}.{b}", x, 2 if (li > 0 and b == 0) else 1)
    return attention_pool(p, x, heads)


def encode_image(p, cutouts, *, layers, heads, apply_preprocess=True):
    """CLIP_Base.encode_image (slip.py:62-66): preprocess -> tower -> divide by the L2 norm."""
    from . import clip_vit_ref
    x = clip_vit_ref.preprocess(cutouts) if apply_preprocess else cutouts
    e = resnet_forward(p, x, layers=layers, heads=heads)
    return e / e.norm(dim=-1, keepdim=True)
