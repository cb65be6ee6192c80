"""ORACLE (test infrastructure only -- never imported by the product package).

CPU fp32 restatement of `VqganDrawer.synth` (/root/reference/vqgan.py:190-195):
    z_q  = vector_quantize(z.movedim(1,3), codebook).movedim(3,1)      vqgan.py:60-64 (+ReplaceGrad 48-58)
    out  = clamp_with_grad(model.decode(z_q).add(1).div(2), 0, 1)       vqgan.py:66-79,195
and of `clip_z` (vqgan.py:202-204).

`model.decode` = `post_quant_conv` + `Decoder.forward` lives in the un-vendored dependency
taming-transformers (bfirsh fork @7a6e64ee, /root/reference/requirements.txt:28:
taming/models/vqgan.py, taming/modules/diffusionmodules/model.py); its published algorithm
is restated from SURVEY.md Appendix A.2.  Parity status: **unpinned** by the reference's own
tests; pinned here (tests/test_oracle_cross.py) against the reference's own in-repo fragments
(`vector_quantize`, `ReplaceGrad`, `ClampWithGrad`, extracted from /root/reference by AST)
and against an independent implementation of the decoder (HF `JanusVQVAEDecoder`).

Parameter names follow taming's state dict (`decoder.*`, `post_quant_conv.*`,
`quantize.embedding.weight`).
"""
from __future__ import annotations

from typing import Dict, List, Tuple

import torch
import torch.nn.functional as F


# ---- in-repo fragments (restated; checked against the AST-extracted originals in tests) ----
class ReplaceGrad(torch.autograd.Function):  # vqgan.py:48-58
    @staticmethod
    def forward(ctx, x_forward, x_backward):
        ctx.shape = x_backward.shape
        return x_forward

    @staticmethod
    def backward(ctx, grad_in):
        return None, grad_in.sum_to_size(ctx.shape)


replace_grad = ReplaceGrad.apply


def vector_quantize(x, codebook):  # vqgan.py:60-64
    d = x.pow(2).sum(dim=-1, keepdim=True) + codebook.pow(2).sum(dim=1) - 2 * x @ codebook.T
    indices = d.argmin(-1)
    x_q = F.one_hot(indices, codebook.shape[0]).to(d.dtype) @ codebook
    return replace_grad(x_q, x)


def vq_indices(x, codebook):
    d = x.pow(2).sum(dim=-1, keepdim=True) + codebook.pow(2).sum(dim=1) - 2 * x @ codebook.T
    return d.argmin(-1), d


def vq_exactness(x, codebook, idx):
    """The exactness rule of the nearest-code search (integer output, vqgan.py:60-64), stated without reference to any fp32
    implementation: a position is a NEAR-TIE when its two smallest FLOAT64 distances differ by less than the fp32 rounding of
    the distance expression ((|x|^2 + |c|^2) - 2 x.c formed in fp32: 8 ulp of the magnitude the sum is formed at).  Outside
    that set `idx` must be the float64 argmin; inside it, it must be one of the tied codes.
    Returns (violations, near_ties, differs_from_f64_argmin)."""
    xd, cd, idx = x.double(), codebook.double(), idx.long()
    d64 = (xd * xd).sum(1, keepdim=True) + (cd * cd).sum(1)[None] - 2.0 * xd @ cd.t()
    best = d64.argmin(1)
    top2 = d64.topk(2, dim=1, largest=False).values
    tol = 8.0 * 2.0 ** -24 * ((xd * xd).sum(1) + (cd * cd).sum(1)[best])
    near = (top2[:, 1] - top2[:, 0]) < tol
    rows = torch.arange(x.shape[0])
    ok = (idx == best) | (near & ((d64[rows, idx] - d64[rows, best]) < tol))
    return int((~ok).sum()), int(near.sum()), int((idx != best).sum())


class ClampWithGrad(torch.autograd.Function):  # vqgan.py:66-79
    @staticmethod
    def forward(ctx, input, min, max):
        ctx.min = min
        ctx.max = max
        ctx.save_for_backward(input)
        return input.clamp(min, max)

    @staticmethod
    def backward(ctx, grad_in):
        input, = ctx.saved_tensors
        return grad_in * (grad_in * (input - input.clamp(ctx.min, ctx.max)) >= 0), None, None


clamp_with_grad = ClampWithGrad.apply


# ---- taming Decoder [UPSTREAM] ---------------------------------------------------------------
def _norm(p, pre, x):
    return F.group_norm(x, 32, p[pre + ".weight"], p[pre + ".bias"], eps=1e-6)


def _swish(x):
    return x * torch.sigmoid(x)


def _conv(p, pre, x, pad):
    return F.conv2d(x, p[pre + ".weight"], p[pre + ".bias"], padding=pad)


def _resblock(p, pre, x):
    h = _conv(p, pre + ".conv1", _swish(_norm(p, pre + ".norm1", x)), 1)
    h = _conv(p, pre + ".conv2", _swish(_norm(p, pre + ".norm2", h)), 1)
    if (pre + ".nin_shortcut.weight") in p:
        x = _conv(p, pre + ".nin_shortcut", x, 0)
    return x + h


def _attnblock(p, pre, x):
    h = _norm(p, pre + ".norm", x)
    q = _conv(p, pre + ".q", h, 0)
    k = _conv(p, pre + ".k", h, 0)
    v = _conv(p, pre + ".v", h, 0)
    b, c, hh, ww = q.shape
    q = q.reshape(b, c, hh * ww).permute(0, 2, 1)      # b, hw, c
    k = k.reshape(b, c, hh * ww)                       # b, c, hw
    w_ = torch.bmm(q, k) * (int(c) ** (-0.5))          # b, hw(q), hw(k)
    w_ = torch.softmax(w_, dim=2)
    v = v.reshape(b, c, hh * ww)
    h_ = torch.bmm(v, w_.permute(0, 2, 1)).reshape(b, c, hh, ww)
    return x + _conv(p, pre + ".proj_out", h_, 0)


def decoder_layout(ch=128, ch_mult=(1, 1, 2, 2, 4), num_res_blocks=2, attn_resolutions=(16,), resolution=256,
                   z_channels=256, out_ch=3) -> dict:
    """Static description of the Decoder graph (shared with nothing in the product: the product has its
    own copy of this arithmetic in C++).  Returns block_in, per-level block lists, etc."""
    nres = len(ch_mult)
    block_in = ch * ch_mult[nres - 1]
    curr_res = resolution // 2 ** (nres - 1)
    levels = []
    bi = block_in
    for i_level in reversed(range(nres)):
        blocks = []
        block_out = ch * ch_mult[i_level]
        for _ in range(num_res_blocks + 1):
            blocks.append((bi, block_out, curr_res in attn_resolutions))
            bi = block_out
        up = i_level != 0
        levels.append((i_level, blocks, up, bi))
        if up:
            curr_res *= 2
    return dict(block_in=block_in, levels=levels, z_channels=z_channels, out_ch=out_ch, ch=ch)


def decode(p: Dict[str, torch.Tensor], z_q: torch.Tensor, cfg: dict) -> torch.Tensor:
    """taming VQModel.decode: post_quant_conv then Decoder.forward."""
    lay = decoder_layout(**cfg)
    x = _conv(p, "post_quant_conv", z_q, 0)
    h = _conv(p, "decoder.conv_in", x, 1)
    h = _resblock(p, "decoder.mid.block_1", h)
    h = _attnblock(p, "decoder.mid.attn_1", h)
    h = _resblock(p, "decoder.mid.block_2", h)
    for (i_level, blocks, up, _) in lay["levels"]:
        for i_block, (_, _, has_attn) in enumerate(blocks):
            h = _resblock(p, f"decoder.up.{i_level}.block.{i_block}", h)
            if has_attn:
                h = _attnblock(p, f"decoder.up.{i_level}.attn.{i_block}", h)
        if up:
            h = F.interpolate(h, scale_factor=2.0, mode="nearest")
            h = _conv(p, f"decoder.up.{i_level}.upsample.conv", h, 1)
    h = _swish(_norm(p, "decoder.norm_out", h))
    return _conv(p, "decoder.conv_out", h, 1)


# ---- taming Encoder + VQModel.encode [UPSTREAM] ----------------------------------------------
def encoder_forward(p: Dict[str, torch.Tensor], x: torch.Tensor, cfg: dict) -> torch.Tensor:
    """taming `Encoder.forward` (taming/modules/diffusionmodules/model.py, double_z=False): conv_in; per level
    num_res_blocks ResnetBlocks (+AttnBlock where the NOMINAL resolution is in attn_resolutions), Downsample = zero pad
    (0,1,0,1) + 3x3 stride-2 conv except after the last level; mid block_1/attn_1/block_2; GroupNorm+swish; conv_out."""
    ch, ch_mult = cfg["ch"], cfg["ch_mult"]
    nres = len(ch_mult)
    curr_res = cfg["resolution"]
    h = _conv(p, "encoder.conv_in", x, 1)
    for lvl in range(nres):
        for b in range(cfg["num_res_blocks"]):
            h = _resblock(p, f"encoder.down.{lvl}.block.{b}", h)
            if curr_res in cfg["attn_resolutions"]:
                h = _attnblock(p, f"encoder.down.{lvl}.attn.{b}", h)
        if lvl != nres - 1:
            h = F.pad(h, (0, 1, 0, 1), mode="constant", value=0)
            h = F.conv2d(h, p[f"encoder.down.{lvl}.downsample.conv.weight"], p[f"encoder.down.{lvl}.downsample.conv.bias"], stride=2)
            curr_res //= 2
    h = _resblock(p, "encoder.mid.block_1", h)
    h = _attnblock(p, "encoder.mid.attn_1", h)
    h = _resblock(p, "encoder.mid.block_2", h)
    h = _swish(_norm(p, "encoder.norm_out", h))
    return _conv(p, "encoder.conv_out", h, 1)


def encode(p: Dict[str, torch.Tensor], x: torch.Tensor, cfg: dict):
    """`z, *_ = model.encode(x)` as VqganDrawer uses it (vqgan.py:174-185): taming `VQModel.encode` = encoder ->
    quant_conv -> VectorQuantizer2 (argmin of the same distance as vqgan.py:60-64; the returned tensor is numerically
    the code vectors, `z + (z_q - z).detach()`).  Returns (z_q NCHW, indices [h*w], pre-quantisation latent NCHW)."""
    h = _conv(p, "quant_conv", encoder_forward(p, x, cfg), 0)
    cb = p["quantize.embedding.weight"]
    hl = h.movedim(1, 3)
    idx, _ = vq_indices(hl, cb)
    z_q = cb[idx.reshape(-1)].reshape(hl.shape).movedim(3, 1)
    return z_q, idx.reshape(-1), h


def synth(p, z, cfg):
    """VqganDrawer.synth (vqgan.py:190-195), non-gumbel branch."""
    z_q = vector_quantize(z.movedim(1, 3), p["quantize.embedding.weight"]).movedim(3, 1)
    return clamp_with_grad(decode(p, z_q, cfg).add(1).div(2), 0, 1)


def z_bounds(p):
    """vqgan.py:155-158"""
    w = p["quantize.embedding.weight"]
    return w.min(dim=0).values[None, :, None, None], w.max(dim=0).values[None, :, None, None]


def clip_z(z, z_min, z_max):
    """vqgan.py:202-204"""
    return z.maximum(z_min).minimum(z_max)
