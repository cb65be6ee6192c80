"""torch.autograd.Function wrappers over the C ABI (include/prx.h).

Every op here enqueues HIP kernels on torch's current stream and returns real torch tensors with
a `grad_fn`, so unmodified reference plugins (`LossInterface.get_loss`, `FilterInterface.forward`,
custom drawers) compose with them through ordinary autograd.  There is no CPU
path: tensors must live on a ROCm device and the shared library must be built.
"""
from __future__ import annotations

import ctypes
from typing import Sequence

import torch

from . import _lib
from ._lib import PrxError, call, precision_code


def _need_cuda(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise PrxError("pixray_amd ops run on an MI355X only (got a CPU tensor); there is no CPU fallback")


def _stream():
    return _lib.current_stream()


_DEFERRED_DESTROY = []


def _destroy_handle(fn_name: str, h) -> None:
    """`prx_*_destroy` frees device memory (hipFree), which a hipGraph capture in progress does not survive: a handle whose
    last reference drops during a capture (an autograd ctx released by the captured backward, the garbage collector) is
    parked and destroyed by the next handle destruction outside a capture (or at interpreter exit with the process)."""
    try:
        capturing = torch.cuda.is_available() and torch.cuda.is_current_stream_capturing()
    except Exception:
        capturing = False
    if capturing:
        _DEFERRED_DESTROY.append((fn_name, h))
        return
    lib = _lib.load()
    while _DEFERRED_DESTROY:
        n, hh = _DEFERRED_DESTROY.pop()
        getattr(lib, n)(hh)
    getattr(lib, fn_name)(h)


def _weight_array(tensors: Sequence[torch.Tensor]):
    arr = (ctypes.c_void_p * len(tensors))()
    for i, t in enumerate(tensors):
        if t.dtype != torch.float32 or not t.is_contiguous() or not t.is_cuda:
            raise PrxError("weights must be contiguous fp32 device tensors")
        arr[i] = t.data_ptr()
    return arr


# --------------------------------------------------------------------------------------- cutouts
class _MakeCutoutsFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, img, desc, noise, S, base_hw, spot_mask):
        _need_cuda(img, desc, noise, spot_mask)
        assert img.dim() == 4 and img.shape[0] == 1 and img.shape[1] == 3, "MakeCutouts expects [1,3,H,W]"
        img = img.contiguous().float()
        n = desc.shape[0]
        H, W = img.shape[2], img.shape[3]
        Hb, Wb = base_hw
        dev = img.device
        pooled = torch.empty(3, S, S, device=dev)
        argmax = torch.empty(3, S, S, device=dev, dtype=torch.int32)
        base = torch.empty(3, Hb, Wb, device=dev) if (Hb, Wb) != (S, S) else None
        stage_a = torch.empty(n, 3, Hb, Wb, device=dev)
        out = torch.empty(n, 3, S, S, device=dev)
        if spot_mask is not None:
            spot_mask = spot_mask.to(torch.uint8).contiguous()
            assert spot_mask.shape == (3, S, S), f"spot mask must be [3,{S},{S}]"
        call("prx_cutouts_forward", img, H, W, desc, noise, spot_mask, n, S, Hb, Wb, pooled, argmax, base, stage_a, out, _stream())
        ctx.save_for_backward(desc, argmax, stage_a)
        ctx.spot_mask = spot_mask
        ctx.geom = (n, S, Hb, Wb, H, W)
        return out

    @staticmethod
    def backward(ctx, g):
        desc, argmax, stage_a = ctx.saved_tensors
        n, S, Hb, Wb, H, W = ctx.geom
        g = g.contiguous().float()
        dev = g.device
        g_a = torch.empty(n, 3, Hb, Wb, device=dev)
        g_priv = torch.empty(n, 3, Hb, Wb, device=dev)
        uv = torch.empty(n, Hb * Wb, 2, device=dev)
        g_base = torch.empty(3, Hb, Wb, device=dev)
        g_pooled = torch.empty(3, S, S, device=dev)
        g_img = torch.empty(1, 3, H, W, device=dev)
        call("prx_cutouts_backward", g, desc, ctx.spot_mask, n, S, Hb, Wb, H, W, stage_a, argmax, g_a, g_priv, uv, g_base, g_pooled,
             g_img, _stream())
        return g_img, None, None, None, None, None


def make_cutouts(img, desc, noise, S, base_hw=None, spot_mask=None):
    """`base_hw`: size of the aspect-rescaled pooled image ((S, S) on a square canvas; pixray_amd.cutouts.base_size);
    `spot_mask`: bool/uint8 [3,S,S], pooled pixels to blank (spot prompts, pixray.py:453-466)."""
    return _MakeCutoutsFn.apply(img, desc, noise, S, tuple(base_hw) if base_hw is not None else (S, S), spot_mask)


# --------------------------------------------------------------------------------------- CLIP ViT
def _weights_in_abi_order(params, shapes, device, what):
    """the C ABI takes bare device pointers in a documented order (include/prx.h): a missing tensor or one whose shape is not the
    one the configuration implies would be read out of bounds on the device, so both are refused here by name"""
    ws = []
    for name, shape in shapes.items():
        if name not in params:
            raise KeyError(f"{what}: the weights have no tensor named {name!r}")
        t = params[name]
        if tuple(t.shape) != tuple(shape):
            raise ValueError(f"{what}: {name} has shape {tuple(t.shape)}, the configuration implies {tuple(shape)}")
        ws.append(t.to(device=device, dtype=torch.float32).contiguous())
    return ws


class ClipVitHandle:
    """Owns a `prx_clip_vit` (packed weights + activation workspace for `max_batch` cutouts).
    `precision`: "fp16" (default) / "bf16" (fast paths) or "f32" (exact-f32 MFMA parity mode, include/prx.h PRX_PREC_*)."""
    abi = "prx_clip_vit"

    def __init__(self, cfg, params, max_batch: int, device, precision=None):
        from .weights import clip_vit_param_shapes
        ws = _weights_in_abi_order(params, clip_vit_param_shapes(cfg), device, f"CLIP ViT {getattr(cfg, 'name', '')}")
        self.precision = precision_code(precision)
        c = _ClipCfg(cfg.input_resolution, cfg.patch_size, cfg.width, cfg.layers, cfg.heads, cfg.output_dim, max_batch, self.precision)
        h = ctypes.c_void_p()
        call("prx_clip_vit_create", ctypes.addressof(h), ctypes.addressof(c), _keep(self, _weight_array(ws)), len(ws), _stream())
        torch.cuda.synchronize(device)   # weight tensors may now be released
        self.h = h
        self.cfg = cfg
        self.max_batch = max_batch
        self.device = device
        self.generation = 0              # bumped by every forward: the handle keeps ONE forward's activations (_ClipEncodeFn)

    @property
    def gemm_ctx(self):
        return _lib.load().prx_clip_vit_gemm_ctx(self.h)

    def __del__(self):
        h = getattr(self, "h", None)
        if h is not None and h.value:
            try:
                _destroy_handle("prx_clip_vit_destroy", h)
            except Exception:
                pass
            self.h = None


def _keep(obj, arr):
    obj._arr = arr   # keep the ctypes array alive for the duration of the call
    return ctypes.addressof(arr)


class _ClipResNetCfg(ctypes.Structure):
    _fields_ = [("input_resolution", ctypes.c_int), ("width", ctypes.c_int), ("layers", ctypes.c_int * 4), ("heads", ctypes.c_int),
                ("output_dim", ctypes.c_int), ("max_batch", ctypes.c_int), ("precision", ctypes.c_int)]


class ClipResNetHandle:
    """Owns a `prx_clip_resnet` (CLIP ModifiedResNet tower: RN50x4, ...); same protocol as ClipVitHandle.  `params` is
    the OpenAI `visual.*` state dict (BatchNorm un-folded); the fold happens here."""
    abi = "prx_clip_resnet"

    def __init__(self, cfg, params, max_batch: int, device, precision=None):
        from .weights import clip_resnet_param_shapes, fold_clip_resnet_params
        _weights_in_abi_order(params, clip_resnet_param_shapes(cfg), "cpu", f"CLIP ModifiedResNet {getattr(cfg, 'name', '')}")   # names / shapes only
        folded = fold_clip_resnet_params(cfg, params)
        ws = [t.to(device=device, dtype=torch.float32).contiguous() for t in folded.values()]
        self.precision = precision_code(precision)
        c = _ClipResNetCfg()
        c.input_resolution, c.width, c.heads, c.output_dim, c.max_batch = cfg.input_resolution, cfg.width, cfg.heads, cfg.output_dim, max_batch
        c.precision = self.precision
        for i, l in enumerate(cfg.layers):
            c.layers[i] = l
        h = ctypes.c_void_p()
        call("prx_clip_resnet_create", ctypes.addressof(h), ctypes.addressof(c), _keep(self, _weight_array(ws)), len(ws), _stream())
        torch.cuda.synchronize(device)
        self.h = h
        self.cfg = cfg
        self.max_batch = max_batch
        self.device = device
        self.generation = 0

    @property
    def gemm_ctx(self):
        return _lib.load().prx_clip_resnet_gemm_ctx(self.h)

    def __del__(self):
        h = getattr(self, "h", None)
        if h is not None and h.value:
            try:
                _destroy_handle("prx_clip_resnet_destroy", h)
            except Exception:
                pass
            self.h = None


class _ClipCfg(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int) for n in ("input_resolution", "patch_size", "width", "layers", "heads",
                                            "output_dim", "max_batch", "precision")]


class _VqganCfg(ctypes.Structure):
    _fields_ = [("ch", ctypes.c_int), ("ch_mult", ctypes.c_int * 8), ("n_mult", ctypes.c_int),
                ("num_res_blocks", ctypes.c_int), ("attn_resolution", ctypes.c_int), ("resolution", ctypes.c_int),
                ("z_channels", ctypes.c_int), ("embed_dim", ctypes.c_int), ("n_embed", ctypes.c_int),
                ("out_ch", ctypes.c_int), ("latent_h", ctypes.c_int), ("latent_w", ctypes.c_int), ("precision", ctypes.c_int)]


class _ClipEncodeFn(torch.autograd.Function):
    """perceptor.encode_image (slip.py:62-66) on a tower handle.

    The handle holds the activations of ONE forward (288 GB make "keep everything" cheap, but per forward).  pixray calls
    encode_image several times per iteration on the same perceptor (spot prompts pixray.py:1282-1292, image prompts
    1307-1336) before a single backward, so every forward bumps `handle.generation`, and a backward whose generation is
    no longer the handle's re-runs its forward (same cutouts, same min/max -> bit-identical activations) before it
    differentiates.  Nothing is ever differentiated through another call's activations."""

    @staticmethod
    def forward(ctx, cutouts, handle, group, comm=None, fixed_range=None):
        _need_cuda(cutouts)
        cutouts = cutouts.contiguous().float()
        n = cutouts.shape[0]
        R = handle.cfg.input_resolution
        assert cutouts.shape[1:] == (3, R, R), f"perceptor expects [n,3,{R},{R}] cutouts"
        dev = cutouts.device
        ctx.fixed_range = fixed_range is not None
        if fixed_range is not None:
            # a caller-given input range (slip.py:21-36 with input_range, or images that are already in [0, 1]): no batch min / max,
            # nothing to exchange between ranks, and no gradient through the range in the backward
            mm = torch.tensor([float(fixed_range[0]), float(fixed_range[1])], device=dev)
        else:
            mm = torch.empty(2, device=dev)
            call(handle.abi + "_minmax", handle.h, cutouts, n, mm, _stream())
        if fixed_range is None and (group is not None or comm is not None):
            # batch-global renorm couples every cutout (slip.py:21-36): min/max over all ranks -- on the C-ABI one-shot exchange
            # (csrc/comm.hip) when the session has one, else torch.distributed (RCCL)
            mm[0].neg_()
            if comm is not None:
                comm.all_reduce_(mm, "max")
            else:
                import torch.distributed as dist
                dist.all_reduce(mm, op=dist.ReduceOp.MAX, group=group)
            mm[0].neg_()
        emb = torch.empty(n, handle.cfg.output_dim, device=dev)
        call(handle.abi + "_encode", handle.h, cutouts, n, mm, emb, _stream())
        handle.generation += 1
        ctx.generation = handle.generation
        ctx.save_for_backward(cutouts, mm)
        ctx.handle = handle
        ctx.group = group
        ctx.comm = comm
        return emb

    @staticmethod
    def backward(ctx, g):
        cutouts, mm = ctx.saved_tensors
        handle = ctx.handle
        g = g.contiguous().float()
        dev = g.device
        if handle.generation != ctx.generation:
            # another forward ran on this handle since ours: restore our activations (deterministic kernels, same inputs)
            scratch = torch.empty(cutouts.shape[0], handle.cfg.output_dim, device=dev)
            call(handle.abi + "_encode", handle.h, cutouts, cutouts.shape[0], mm, scratch, _stream())
            handle.generation += 1
            ctx.generation = handle.generation
        acc = torch.empty(4, device=dev, dtype=torch.float64)
        call(handle.abi + "_backward_reduce", handle.h, cutouts, mm, g, acc, _stream())
        if ctx.fixed_range:
            acc.zero_()                       # the range is a constant of the call: no d/dmin, d/dmax terms
        elif ctx.comm is not None:
            ctx.comm.all_reduce_(acc, "sum")
        elif ctx.group is not None:
            import torch.distributed as dist
            dist.all_reduce(acc, op=dist.ReduceOp.SUM, group=ctx.group)
        gc = torch.empty_like(cutouts)
        call(handle.abi + "_backward_finish", handle.h, cutouts, mm, acc, gc, _stream())
        return gc, None, None, None, None


def clip_encode_image(cutouts, handle: ClipVitHandle, group=None, comm=None, fixed_range=None):
    """`fixed_range` = (lo, hi): renormalise with this range instead of the batch's min / max (slip.py:21-36 `input_range`)"""
    return _ClipEncodeFn.apply(cutouts, handle, group, comm, fixed_range)


class _ClipTextCfg(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int) for n in ("vocab_size", "context_length", "width", "layers", "heads", "output_dim",
                                            "max_batch")]


class ClipTextHandle:
    """Owns a `prx_clip_text` (CLIP text transformer, forward only: CLIP_Base.encode_text, slip.py:68-70)."""

    def __init__(self, cfg, params, max_batch: int, device):
        from .weights import clip_text_param_shapes
        ws = _weights_in_abi_order(params, clip_text_param_shapes(cfg), device, f"CLIP text tower {getattr(cfg, 'name', '')}")
        c = _ClipTextCfg(cfg.vocab_size, cfg.context_length, cfg.width, cfg.layers, cfg.heads, cfg.output_dim, max_batch)
        h = ctypes.c_void_p()
        call("prx_clip_text_create", ctypes.addressof(h), ctypes.addressof(c), _keep(self, _weight_array(ws)), len(ws), _stream())
        torch.cuda.synchronize(device)
        self.h = h
        self.cfg = cfg
        self.max_batch = max_batch
        self.device = device

    def __del__(self):
        h = getattr(self, "h", None)
        if h is not None and h.value:
            try:
                _destroy_handle("prx_clip_text_destroy", h)
            except Exception:
                pass
            self.h = None


@torch.no_grad()
def clip_encode_text_tokens(tokens, handle: ClipTextHandle):
    """tokens: integer tensor [n, context_length] as `clip.tokenize` returns -> fp32 [n, output_dim] (not normalised)."""
    if tokens.dim() != 2 or tokens.shape[1] != handle.cfg.context_length:
        raise ValueError(f"tokens must be [n, {handle.cfg.context_length}], got {tuple(tokens.shape)}")
    n = tokens.shape[0]
    if n < 1 or n > handle.max_batch:
        raise ValueError(f"batch {n} outside the text handle capacity 1..{handle.max_batch}")
    tk = tokens.detach().to("cpu")
    if int(tk.min()) < 0 or int(tk.max()) >= handle.cfg.vocab_size:
        raise ValueError(f"token ids must lie in [0, {handle.cfg.vocab_size})")
    tk = tk.to(device=handle.device, dtype=torch.int32).contiguous()
    out = torch.empty(n, handle.cfg.output_dim, device=handle.device)
    call("prx_clip_text_encode", handle.h, tk, n, out, _stream())
    return out


# --------------------------------------------------------------------------------------- VQGAN
def _single_attn_resolution(cfg) -> int:
    """taming places AttnBlocks at every level whose NOMINAL resolution (config `resolution` halved per level) is listed in
    `attn_resolutions`; the C ABI carries one such resolution (every published VQGAN config has one: [16], or [32]).  A config in
    which two listed resolutions are actually visited is refused here rather than built with attention missing."""
    visited = {cfg.resolution >> k for k in range(len(cfg.ch_mult))}
    hits = sorted(set(int(r) for r in cfg.attn_resolutions) & visited)
    if len(hits) > 1:
        raise ValueError(f"VQGAN config with attention at {len(hits)} resolutions {hits}: the runner supports one")
    return hits[0] if hits else -1


class VqganHandle:
    """Owns a `prx_vqgan` (codebook, weight packs, activations of one forward).  `precision`: "fp16" (default) | "bf16" | "f32"."""

    def __init__(self, cfg, params, latent_hw, device, precision=None):
        from .weights import vqgan_param_shapes
        ws = _weights_in_abi_order(params, vqgan_param_shapes(cfg), device, "VQGAN decoder")
        c = _VqganCfg()
        c.ch = cfg.ch
        for i, m in enumerate(cfg.ch_mult):
            c.ch_mult[i] = m
        c.n_mult = len(cfg.ch_mult)
        c.num_res_blocks = cfg.num_res_blocks
        c.attn_resolution = _single_attn_resolution(cfg)
        c.resolution = cfg.resolution
        c.z_channels = cfg.z_channels
        c.embed_dim = cfg.embed_dim
        c.n_embed = cfg.n_embed
        c.out_ch = cfg.out_ch
        c.latent_h, c.latent_w = latent_hw
        self.precision = precision_code(precision)
        c.precision = self.precision
        h = ctypes.c_void_p()
        call("prx_vqgan_create", ctypes.addressof(h), ctypes.addressof(c), _keep(self, _weight_array(ws)), len(ws), _stream())
        torch.cuda.synchronize(device)
        self.h = h
        self.cfg = cfg
        self.latent_hw = tuple(latent_hw)
        self.f = 2 ** (len(cfg.ch_mult) - 1)
        self.device = device
        self.last_indices = None
        self.generation = 0              # see _ClipEncodeFn: one forward's activations per handle

    @property
    def gemm_ctx(self):
        return _lib.load().prx_vqgan_gemm_ctx(self.h)

    def z_bounds(self):
        zmin = torch.empty(self.cfg.embed_dim, device=self.device)
        zmax = torch.empty(self.cfg.embed_dim, device=self.device)
        call("prx_vqgan_z_bounds", self.h, zmin, zmax, _stream())
        return zmin, zmax

    def __del__(self):
        h = getattr(self, "h", None)
        if h is not None and h.value:
            try:
                _destroy_handle("prx_vqgan_destroy", h)
            except Exception:
                pass
            self.h = None


class _VqganSynthFn(torch.autograd.Function):
    """VqganDrawer.synth (vqgan.py:190-195).  Same one-forward-per-handle rule as `_ClipEncodeFn`: a `to_image()` or a
    second `synth()` between this forward and its backward bumps the handle's generation, and the backward then re-runs
    the forward from the saved z before differentiating."""

    @staticmethod
    def forward(ctx, z, handle, quantize):
        _need_cuda(z)
        z = z.contiguous().float()
        hh, ww = handle.latent_hw
        assert z.shape == (1, handle.cfg.z_channels, hh, ww), f"z must be [1,{handle.cfg.z_channels},{hh},{ww}]"
        dev = z.device
        img = torch.empty(1, handle.cfg.out_ch, hh * handle.f, ww * handle.f, device=dev)
        idx = torch.empty(hh * ww, device=dev, dtype=torch.int32)
        call("prx_vqgan_synth", handle.h, z, img, idx, int(quantize), _stream())
        handle.last_indices = idx
        handle.generation += 1
        ctx.generation = handle.generation
        ctx.handle = handle
        ctx.quantize = int(quantize)
        ctx.save_for_backward(z)
        return img

    @staticmethod
    def backward(ctx, g):
        handle = ctx.handle
        (z,) = ctx.saved_tensors
        g = g.contiguous().float()
        hh, ww = handle.latent_hw
        if handle.generation != ctx.generation:
            scratch = torch.empty(1, handle.cfg.out_ch, hh * handle.f, ww * handle.f, device=g.device)
            call("prx_vqgan_synth", handle.h, z, scratch, None, ctx.quantize, _stream())
            handle.generation += 1
            ctx.generation = handle.generation
        dz = torch.empty(1, handle.cfg.z_channels, hh, ww, device=g.device)
        call("prx_vqgan_synth_backward", handle.h, g, dz, _stream())
        return dz, None, None


def vqgan_synth(z, handle: VqganHandle, quantize: bool = True):
    return _VqganSynthFn.apply(z, handle, quantize)


class VqganEncHandle:
    """Owns a `prx_vqgan_enc` (taming Encoder + quant_conv + codebook) for one image size; forward only
    (VqganDrawer.init_from_tensor / reapply_from_tensor / get_z_from_tensor, vqgan.py:174-185)."""

    def __init__(self, cfg, params, image_hw, device, in_channels: int = 3):
        from .weights import vqgan_encoder_param_shapes
        ws = _weights_in_abi_order(params, vqgan_encoder_param_shapes(cfg, in_channels), device, "VQGAN encoder")
        c = _VqganCfg()
        c.ch = cfg.ch
        for i, m in enumerate(cfg.ch_mult):
            c.ch_mult[i] = m
        c.n_mult = len(cfg.ch_mult)
        c.num_res_blocks = cfg.num_res_blocks
        c.attn_resolution = _single_attn_resolution(cfg)
        c.resolution = cfg.resolution
        c.z_channels = cfg.z_channels
        c.embed_dim = cfg.embed_dim
        c.n_embed = cfg.n_embed
        c.out_ch = cfg.out_ch
        self.f = 2 ** (len(cfg.ch_mult) - 1)
        H, W = image_hw
        c.latent_h, c.latent_w = H // self.f, W // self.f
        h = ctypes.c_void_p()
        call("prx_vqgan_enc_create", ctypes.addressof(h), ctypes.addressof(c), in_channels, H, W, _keep(self, _weight_array(ws)),
             len(ws), _stream())
        torch.cuda.synchronize(device)
        self.h = h
        self.cfg = cfg
        self.image_hw = (H, W)
        self.in_channels = in_channels
        self.device = device

    def __del__(self):
        h = getattr(self, "h", None)
        if h is not None and h.value:
            try:
                _destroy_handle("prx_vqgan_enc_destroy", h)
            except Exception:
                pass
            self.h = None


@torch.no_grad()
def vqgan_encode(img, handle: VqganEncHandle, return_pre: bool = False):
    """img [1,C,H,W] in [-1,1] -> z [1,embed_dim,H/f,W/f] (the selected code vectors), int32 indices
    [, the latent before quantisation]."""
    _need_cuda(img)
    img = img.contiguous().float()
    H, W = handle.image_hw
    assert img.shape == (1, handle.in_channels, H, W), f"image must be [1,{handle.in_channels},{H},{W}], got {tuple(img.shape)}"
    h0, w0 = H // handle.f, W // handle.f
    z = torch.empty(1, handle.cfg.embed_dim, h0, w0, device=img.device)
    idx = torch.empty(h0 * w0, device=img.device, dtype=torch.int32)
    pre = torch.empty_like(z) if return_pre else None
    call("prx_vqgan_encode", handle.h, img, z, pre, idx, _stream())
    return (z, idx, pre) if return_pre else (z, idx)


# --------------------------------------------------------------------------------------- Prompt loss
_TICKETS = {}


def _ticket(device):
    """the prompt kernel's "last workgroup adds up" counter: a wrapping ticket (atomicInc modulo the grid), zero whenever no
    launch is in flight -- one resident word per (device, stream); a process uses a handful of streams, so the table stays small"""
    key = (str(device), _stream())
    t = _TICKETS.get(key)
    if t is None:
        t = _TICKETS[key] = torch.zeros(1, dtype=torch.int32, device=device)
    return t


class _PromptLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, input, embed, weight, stop, denom):
        _need_cuda(input, embed)
        x = input.contiguous().float()
        e = embed.contiguous().float()
        n, D = x.shape
        m = e.shape[0]
        rowloss = torch.empty(n + 1, device=x.device)          # [n] row values, then the scalar |w| * sum / denom from the same launch
        grad = torch.empty_like(x)
        den = float(denom) if denom is not None else float(n * m)
        call("prx_prompt_loss_fwd_bwd", x, e, n, m, D, float(weight), float(stop), den, rowloss, grad, rowloss[n:], _ticket(x.device), _stream())
        ctx.save_for_backward(grad)
        return rowloss[n]

    @staticmethod
    def backward(ctx, g):
        (grad,) = ctx.saved_tensors
        return grad * g, None, None, None, None


def prompt_loss(input, embed, weight=1.0, stop=float("-inf"), denom=None):
    return _PromptLossFn.apply(input, embed, weight, stop, denom)


# --------------------------------------------------------------------------------------- optimiser
def adam_clamp_step_dev(z, exp_avg, exp_avg_sq, grad, zmin, zmax, hyper, betas=(0.9, 0.999), eps=1e-8):
    """Same, with {lr / bias_correction1, sqrt(bias_correction2)} read from the device tensor `hyper` (graph replay)."""
    _need_cuda(z, grad, hyper)
    hw = z.shape[-1] * z.shape[-2]
    call("prx_adam_clamp_step_dev", z, exp_avg, exp_avg_sq, grad, zmin, zmax, hw, z.numel(), hyper, float(betas[0]),
         float(betas[1]), float(eps), _stream())


def adam_clamp_step(z, exp_avg, exp_avg_sq, grad, zmin, zmax, lr, step, betas=(0.9, 0.999), eps=1e-8):
    """In-place Adam step on z fused with the per-channel clip_z clamp."""
    _need_cuda(z, grad)
    assert z.is_contiguous() and grad.is_contiguous() and z.dtype == torch.float32
    hw = z.shape[-1] * z.shape[-2]
    call("prx_adam_clamp_step", z, exp_avg, exp_avg_sq, grad, zmin, zmax, hw, z.numel(), float(lr), float(betas[0]),
         float(betas[1]), float(eps), int(step), _stream())


# --------------------------------------------------------------------------------------- VGG16 features (StyleLoss plugin)
VGG16_CONV_INDICES = (0, 2, 5, 7, 10, 12, 14, 17, 19, 21, 24, 26, 28)       # torchvision vgg16().features conv layers
VGG16_CAPTURE_LAYERS = (1, 3, 6, 8, 11, 13, 15, 22, 29)                     # Losses/StyleLoss.py:31


class Vgg16Handle:
    """Owns a `prx_vgg16` (torchvision VGG16 `features` up to relu5_3, frozen) for inputs up to `max_hw`.
    `params`: {"features.N.weight", "features.N.bias"} (torchvision state-dict names)."""

    def __init__(self, params, max_hw, device, precision=None):
        self.precision = precision_code(precision)
        from .weights import vgg16_param_shapes
        ws = _weights_in_abi_order(params, vgg16_param_shapes(), device, "VGG16")
        h = ctypes.c_void_p()
        call("prx_vgg16_create", ctypes.addressof(h), _keep(self, _weight_array(ws)), len(ws), int(max_hw[0]), int(max_hw[1]),
             self.precision, _stream())
        torch.cuda.synchronize(device)
        self.h = h
        self.max_hw = (int(max_hw[0]), int(max_hw[1]))
        self.device = device

    @property
    def gemm_ctx(self):
        return _lib.load().prx_vgg16_gemm_ctx(self.h)

    def feature_shape(self, H, W, k):
        h, w, c = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        call("prx_vgg16_feature_shape", int(H), int(W), int(k), ctypes.addressof(h), ctypes.addressof(w), ctypes.addressof(c))
        return h.value, w.value, c.value

    def __del__(self):
        h = getattr(self, "h", None)
        if h is not None and h.value:
            try:
                _destroy_handle("prx_vgg16_destroy", h)
            except Exception:
                pass
            self.h = None


class _Vgg16Fn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, handle):
        _need_cuda(x)
        assert x.dim() == 4 and x.shape[0] == 1 and x.shape[1] == 3, "the VGG16 extractor takes one [1,3,H,W] image"
        H, W = int(x.shape[2]), int(x.shape[3])
        x = x.detach().to(torch.float32).contiguous()
        nbytes = call("prx_vgg16_workspace_bytes", H, W, handle.precision)
        if nbytes <= 0:
            raise PrxError(f"VGG16 extractor: input {H}x{W} is too small")
        work = torch.empty(int(nbytes), dtype=torch.uint8, device=x.device)
        feats = []
        for k in range(len(VGG16_CAPTURE_LAYERS)):
            h, w, c = handle.feature_shape(H, W, k)
            feats.append(torch.empty(1, h, w, c, dtype=torch.float32, device=x.device))
        arr = (ctypes.c_void_p * len(feats))(*[f.data_ptr() for f in feats])
        call("prx_vgg16_forward", handle.h, x, H, W, work, ctypes.addressof(arr), _stream())
        ctx.handle, ctx.work, ctx.hw = handle, work, (H, W)
        return tuple(feats)

    @staticmethod
    def backward(ctx, *gs):
        H, W = ctx.hw
        keep = [None if g is None else g.to(torch.float32).contiguous() for g in gs]
        arr = (ctypes.c_void_p * len(keep))(*[None if g is None else g.data_ptr() for g in keep])
        gx = torch.empty(1, 3, H, W, dtype=torch.float32, device=ctx.work.device)
        call("prx_vgg16_backward", ctx.handle.h, H, W, ctx.work, ctypes.addressof(arr), gx, _stream())
        return gx, None


class _HypercolumnsFn(torch.autograd.Function):
    """`spatial_feature_extract` (Losses/StyleLoss.py:169-223) over all feature maps in one gather launch (and one scatter
    launch backward); rows int64 [L,4,n], wts fp32 [4L+2,n] (tap weights, then the two coordinate channels)"""

    @staticmethod
    def forward(ctx, rows, wts, *feats):
        L, n = len(feats), int(rows.shape[2])
        for f in feats:
            _need_cuda(f)
            if f.dtype != torch.float32 or not f.is_contiguous() or f.dim() != 4 or f.shape[0] != 1:
                raise PrxError("hypercolumns: feature maps must be contiguous fp32 [1,h,w,C] device tensors")
        if rows.dtype != torch.int64 or not rows.is_contiguous() or tuple(rows.shape) != (L, 4, n):
            raise PrxError("hypercolumns: rows must be a contiguous int64 [L,4,n] tensor")
        if wts.dtype != torch.float32 or not wts.is_contiguous() or tuple(wts.shape) != (4 * L + 2, n):
            raise PrxError("hypercolumns: weights must be a contiguous fp32 [4L+2,n] tensor")
        chans = (ctypes.c_int * L)(*[int(f.shape[3]) for f in feats])
        ptrs = (ctypes.c_void_p * L)(*[f.data_ptr() for f in feats])
        ctot = sum(chans)
        out = torch.empty(n, ctot + 2, dtype=torch.float32, device=feats[0].device)
        call("prx_hypercolumns_fwd", ctypes.addressof(ptrs), ctypes.addressof(chans), L, rows, wts, n, out, ctot + 2, _stream())
        ctx.save_for_backward(rows, wts)
        ctx.shapes = [tuple(f.shape) for f in feats]
        return out

    @staticmethod
    def backward(ctx, gout):
        rows, wts = ctx.saved_tensors
        L, n = len(ctx.shapes), int(rows.shape[2])
        gout = gout.to(torch.float32).contiguous()
        grads = [torch.zeros(sh, dtype=torch.float32, device=gout.device) if ctx.needs_input_grad[2 + l] else None
                 for l, sh in enumerate(ctx.shapes)]
        chans = (ctypes.c_int * L)(*[int(sh[3]) for sh in ctx.shapes])
        ptrs = (ctypes.c_void_p * L)(*[g.data_ptr() if g is not None else None for g in grads])
        call("prx_hypercolumns_bwd", ctypes.addressof(ptrs), ctypes.addressof(chans), L, rows, wts, n, gout, int(gout.shape[1]), _stream())
        return (None, None) + tuple(grads)


def hypercolumns(feats, rows, wts):
    """feats: list of [1,h,w,C] fp32 NHWC maps -> [n, sum(C)+2] sampled columns (+ the two coordinate channels)"""
    return _HypercolumnsFn.apply(rows, wts, *[f.contiguous() for f in feats])


def _f32c(t, what):
    _need_cuda(t)
    if t.dtype != torch.float32 or t.dim() != 2:
        raise PrxError(f"{what}: expected a 2-D fp32 device tensor, got {tuple(t.shape)} {t.dtype}")
    return t.contiguous()


def _row_sumsq(X):
    """|x_i|^2 as `pairwise_distances_cos` takes it ((x ** 2).sum(1), Losses/StyleLoss.py:227): the same reduction, so that the
    distance matrix -- and with it every arg-minimum -- is bit for bit the composed expression's"""
    return (X * X).sum(1)


class _RemdFn(torch.autograd.Function):
    """`style_loss` (Losses/StyleLoss.py:272-293) on columns X [n, d] (differentiated) and Y [m, d] (the style image's: no
    gradient): max of the mean row minimum and the mean column minimum of the cosine (+ L2 when `l2`) distance matrix.  The
    product X Y^T is a library GEMM; the distance / minima pass and the backward over the n + m selected pairs are
    csrc/strotss.hip.  `ys`: |y_j|^2, the caller's (the same style columns serve three evaluations)."""

    @staticmethod
    def forward(ctx, X, Y, ys, l2):
        X, Y = _f32c(X, "strotss_remd X"), _f32c(Y, "strotss_remd Y")
        n, d = int(X.shape[0]), int(X.shape[1])
        m = int(Y.shape[0])
        if int(Y.shape[1]) != d or tuple(ys.shape) != (m,):
            raise PrxError(f"strotss_remd: X {tuple(X.shape)}, Y {tuple(Y.shape)}, ys {tuple(ys.shape)} do not match")
        G = torch.mm(X, Y.t())
        xs = _row_sumsq(X)
        packs = torch.empty(n + m, dtype=torch.int64, device=X.device)
        stats = torch.empty(4, dtype=torch.float32, device=X.device)
        call("prx_strotss_remd_fwd", G, m, xs, ys, n, m, int(bool(l2)), d, packs, packs[n:], stats, _stream())
        ctx.save_for_backward(G, X, Y, xs, ys, packs, stats)
        ctx.l2 = int(bool(l2))
        return stats[0].clone()

    @staticmethod
    def backward(ctx, g):
        G, X, Y, xs, ys, packs, stats = ctx.saved_tensors
        n, d, m = int(X.shape[0]), int(X.shape[1]), int(Y.shape[0])
        g = g.to(torch.float32).reshape(1).contiguous()
        dX = torch.empty_like(X)
        nbytes = int(_lib.load().prx_strotss_remd_bwd_workspace_bytes(n, m, d))
        work = torch.empty(nbytes, dtype=torch.uint8, device=X.device)
        call("prx_strotss_remd_bwd", G, m, X, d, Y, d, d, xs, ys, packs, packs[n:], n, m, ctx.l2, stats, g, work, nbytes, dX, d, _stream())
        return dX, None, None, None


def strotss_remd(X, Y, ys=None, l2=False):
    if Y.requires_grad:
        raise PrxError("strotss_remd: the style columns take no gradient")
    if ys is None:
        ys = _row_sumsq(Y.detach())
    return _RemdFn.apply(X, Y, ys, l2)


class _SelfSimFn(torch.autograd.Function):
    """`content_loss` (Losses/StyleLoss.py:246-265): mean |cosine self-distance matrix of X - that of Y|, both [n, d] and both
    differentiated.  Products: library GEMMs; the distance passes: csrc/strotss.hip (the backward returns the symmetrised
    d/dG, so each operand costs one product)."""

    @staticmethod
    def forward(ctx, X, Y):
        X, Y = _f32c(X, "strotss_selfsim X"), _f32c(Y, "strotss_selfsim Y")
        if X.shape != Y.shape:
            raise PrxError(f"strotss_selfsim: X {tuple(X.shape)} and Y {tuple(Y.shape)} differ")
        n = int(X.shape[0])
        Gx, Gy = torch.mm(X, X.t()), torch.mm(Y, Y.t())
        xs, ys = _row_sumsq(X), _row_sumsq(Y)
        partial = torch.empty(n, dtype=torch.float64, device=X.device)
        out = torch.empty(1, dtype=torch.float32, device=X.device)
        call("prx_strotss_selfsim_fwd", Gx, n, xs, Gy, n, ys, n, partial, out, _stream())
        ctx.save_for_backward(Gx, Gy, xs, ys, X, Y)
        return out[0].clone()

    @staticmethod
    def backward(ctx, g):
        Gx, Gy, xs, ys, X, Y = ctx.saved_tensors
        n = int(X.shape[0])
        g = g.to(torch.float32).reshape(1).contiguous()
        S = torch.empty(2, n, n, dtype=torch.float32, device=X.device)
        c = torch.empty(2, n, dtype=torch.float32, device=X.device)
        call("prx_strotss_selfsim_bwd", Gx, n, xs, Gy, n, ys, n, g, S[0], S[1], n, c[0], c[1], _stream())
        dX = torch.addcmul(torch.mm(S[0], X), X, c[0].unsqueeze(1)) if ctx.needs_input_grad[0] else None
        dY = torch.addcmul(torch.mm(S[1], Y), Y, c[1].unsqueeze(1)) if ctx.needs_input_grad[1] else None
        return dX, dY


def strotss_selfsim(X, Y):
    return _SelfSimFn.apply(X, Y)


def vgg16_features(x, handle: Vgg16Handle):
    """x [1,3,H,W] (already normalised for VGG) -> the nine captured feature maps as NHWC fp32 tensors [1,h,w,C]
    (channels-last is the engine's layout; `f.permute(0,3,1,2)` is the reference's NCHW view)."""
    return _Vgg16Fn.apply(x, handle)


# --------------------------------------------------------------------------------------- fft drawer (configs[3])
class FftDrawerHandle:
    """`prx_fft_drawer` (csrc/fft_drawer.hip): spectrum [1,3,H,Wf,2] -> image [1,3,H,W] as exact-f32 GEMMs against twiddle
    matrices, and its backward.  One handle per canvas size."""

    def __init__(self, width: int, height: int, decay: float = 1.5, colors: float = 1.5):
        lib = _lib.load()
        self.h = lib.prx_fft_drawer_create(int(width), int(height), float(decay), float(colors))
        if not self.h:
            raise PrxError("prx_fft_drawer_create failed: " + _lib.last_error())
        self.width, self.height = int(width), int(height)
        self.freq_columns = lib.prx_fft_drawer_freq_columns(self.h)

    def __del__(self):
        h, self.h = getattr(self, "h", None), None
        if h:
            try:
                _destroy_handle("prx_fft_drawer_destroy", h)
            except Exception:
                pass


class _FftSynthFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, params, handle, contrast):
        _need_cuda(params)
        p = params.detach().contiguous().float()
        if tuple(p.shape) != (1, 3, handle.height, handle.freq_columns, 2):
            raise PrxError(f"fft drawer: spectrum {tuple(p.shape)} does not fit a {handle.width} x {handle.height} canvas")
        img = torch.empty(1, 3, handle.height, handle.width, device=p.device)
        call("prx_fft_drawer_synth", handle.h, p, float(contrast), img, _stream())
        ctx.handle, ctx.shape = handle, p.shape
        return img

    @staticmethod
    def backward(ctx, g):
        g = g.contiguous().float()
        gp = torch.empty(ctx.shape, device=g.device)
        call("prx_fft_drawer_backward", ctx.handle.h, g, gp, _stream())
        return gp, None, None


def fft_synth(params: torch.Tensor, handle: FftDrawerHandle, contrast: float = 0.9) -> torch.Tensor:
    """differentiable w.r.t. `params`; a backward differentiates the handle's LAST synth (one synth per backward, as the loop does)"""
    return _FftSynthFn.apply(params, handle, contrast)
