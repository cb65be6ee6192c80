"""CPU fp32 restatement of CLIP's text side -- TEST INFRASTRUCTURE ONLY (never imported by pixray_amd/).

Follows `CLIP_Base.encode_text` (/root/reference/slip.py:68-70): `clip.tokenize(text)` then `model.encode_text(tokens)`.
`model.encode_text` lives in openai/CLIP (`clip/model.py`, un-vendored and unpinned: requirements.txt:29) [UPSTREAM]:

    x = token_embedding(text) + positional_embedding
    x = transformer(x)            # ResidualAttentionBlocks with the additive causal mask (-inf above the diagonal)
    x = ln_final(x)
    x = x[arange(n), text.argmax(dim=-1)] @ text_projection

Pinned against an independent implementation: HF `CLIPTextModelWithProjection` (hidden_act="quick_gelu",
eos_token_id=2 so that HF pools at `argmax(input_ids)` exactly as OpenAI does) -- tests/golden/clip_text_golden.npz.
"""
import math
from typing import Dict

import torch
import torch.nn.functional as F


def quick_gelu(x):
    return x * torch.sigmoid(1.702 * x)


def text_forward(p: Dict[str, torch.Tensor], tokens: torch.Tensor, *, heads: int, layers: int) -> torch.Tensor:
    """tokens: int64 [n, ctx] -> [n, output_dim] (not normalised)."""
    x = p["token_embedding.weight"][tokens] + p["positional_embedding"]
    N, T, width = x.shape
    hd = width // heads
    mask = torch.full((T, T), float("-inf")).triu_(1)
    for i in range(layers):
        pre = f"transformer.resblocks.{i}."
        h = F.layer_norm(x, (width,), p[pre + "ln_1.weight"], p[pre + "ln_1.bias"], 1e-5)
        qkv = F.linear(h, p[pre + "attn.in_proj_weight"], p[pre + "attn.in_proj_bias"])
        q, k, v = qkv.split(width, dim=-1)
        q = q.reshape(N, T, heads, hd).permute(0, 2, 1, 3)
        k = k.reshape(N, T, heads, hd).permute(0, 2, 1, 3)
        v = v.reshape(N, T, heads, hd).permute(0, 2, 1, 3)
        att = torch.softmax((q @ k.transpose(-1, -2)) / math.sqrt(hd) + mask, dim=-1)
        o = (att @ v).permute(0, 2, 1, 3).reshape(N, T, width)
        x = x + F.linear(o, p[pre + "attn.out_proj.weight"], p[pre + "attn.out_proj.bias"])
        h = F.layer_norm(x, (width,), p[pre + "ln_2.weight"], p[pre + "ln_2.bias"], 1e-5)
        h = quick_gelu(F.linear(h, p[pre + "mlp.c_fc.weight"], p[pre + "mlp.c_fc.bias"]))
        x = x + F.linear(h, p[pre + "mlp.c_proj.weight"], p[pre + "mlp.c_proj.bias"])
    x = F.layer_norm(x, (width,), p["ln_final.weight"], p["ln_final.bias"], 1e-5)
    return x[torch.arange(N), tokens.argmax(dim=-1)] @ p["text_projection"]
