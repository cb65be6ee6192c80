"""Product factory: assembles a `Session` from HIP-backed parts only (no oracle, no CPU fallback)."""
from __future__ import annotations

import json
import os
import types
from typing import Dict, Optional, Sequence

import torch

from . import _lib
from .cutouts import MakeCutouts
from .engine import Session
from .perceptor import get_clip_perceptor
from .prompt import Prompt, parse_prompt
from .vqgan_drawer import VqganDrawer

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def seeded_unit_vectors(n: int, dim: int, seed: int) -> torch.Tensor:
    g = torch.Generator().manual_seed(seed)
    e = torch.randn(n, dim, generator=g)
    return e / e.norm(dim=-1, keepdim=True)


def _maybe_oneshot_comm(group, rank, world_size, size):
    """PRX_ONESHOT_ALLREDUCE=1: carry the per-step all-reduce of dL/d(image) on the C-ABI one-shot direct-write collective
    (`prx_allreduce_grad`, csrc/comm.hip) instead of torch.distributed / RCCL.  Off by default: only one GPU is reachable from
    the build container, so the protocol is tested with two processes on one device and the RCCL path stays the default
    until a node has confirmed it (bench.py prints collectives_ms_per_step for either)."""
    if world_size <= 1 or group is None or os.environ.get("PRX_ONESHOT_ALLREDUCE", "0") != "1":
        return None
    from .comm import OneShotComm
    return OneShotComm(group, rank, world_size, max_bytes=4 * 4 * int(size[0]) * int(size[1]))


VECTORS_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "vectors")


def load_vector_table(f1: str) -> dict:
    """pixray.py:891-904: a vector prompt names a json table {CLIP model name: [[...]]}: a path containing 'json' is opened as
    it is, a bare name is `vectors/<name>.json` -- relative to the working directory as in the reference, else the copy this
    package ships (`textoff`, pixray's DEFAULT --vector_prompts, pixray.py:1732; tools/make_vectors.py)"""
    if "json" in f1:
        path = f1
    else:
        path = os.path.join("vectors", f"{f1}.json")
        if not os.path.exists(path):
            path = os.path.join(VECTORS_DIR, f"{f1}.json")
    with open(path) as f:
        return json.load(f)


def build_vqgan_clip_session(*, size=(256, 256), vqgan_model="imagenet_f16_16384", clip_model="ViT-B/32",
                             num_cuts=64, learning_rate=0.2, iterations=250, prompt_embeds: Optional[torch.Tensor] = None,
                             prompt_weight=1.0, extra_prompts: Sequence = (), seed=0, device="cuda", group=None, rank=0,
                             world_size=1, custom_losses=(), filters=(), learning_rate_drops=(), prompts: Sequence[str] = (),
                             vector_prompts: Sequence[str] = (), init_image: Optional[torch.Tensor] = None, tokenizer=None,
                             clip_text_params=None, image_prompts: Sequence[torch.Tensor] = (), image_prompt_weight=None,
                             image_prompt_shuffle: bool = False, init_weight: float = 0.0, init_weight_dist: float = 0.0,
                             init_weight_pix: float = 0.0, init_weight_cos: float = 0.0, stand_in_prompt: bool = False,
                             overlay_image=None,
                             overlay_every: int = 10, overlay_offset: int = 0, overlay_until: Optional[int] = None,
                             overlay_alpha: Optional[int] = None, precision: Optional[str] = None) -> Session:
    """The headline configuration of BASELINE.json configs[1]: VqganDrawer + one CLIP ViT perceptor + MakeCutouts +
    a text-like Prompt (precomputed embedding; random unit vector when none is given) + Adam on z.

    `prompts`: pixray text prompts "text[:weight[:stop]]" (pixray.py:859-877), encoded by the HIP text tower (needs the
    CLIP merges table, pixray_amd/tokenizer.py); `vector_prompts`: names ("textoff": pixray's default, shipped under
    pixray_amd/vectors/) or json paths of tables {model name: [[...]]} with precomputed CLIP-space vectors,
    "name[:weight[:stop]]", weighted x0.1 as pixray.py:887-915 does; `stand_in_prompt`: keep the seeded unit vector that stands
    in for a text prompt (no CLIP text checkpoint offline) next to the vector prompts; `init_image`:
    [1,3,H,W] in [0,1], encoded to the starting z by the HIP VQGAN encoder (pixray.py:696-718); `image_prompts`: target
    images [1,3,H,W] in [0,1] turned into per-iteration throwaway Prompts through the cached cutout transforms
    (pixray.py:823-835, 1307-1336); `init_weight*`: the z / pixel regularisers of pixray.py:1351-1375 (need `init_image`);
    `overlay_*`: a PIL image (or path) pasted over the current image every `overlay_every` iterations and re-encoded by the HIP
    VQGAN encoder (pixray.py:731-747, 1408-1420); `precision`: "fp16" (the default: IEEE-half MFMA operands, the reference's own GPU arithmetic for CLIP,
    slip.py:175), "bf16" (the same rate, 8 significand bits), "f32" (every
    contraction on the exact-f32 MFMA: the parity mode the bf16 numbers are measured against) or "ref" (the reference's own
    mix on a GPU: fp32 VQGAN decoder, vqgan.py:124-140, + fp16 CLIP towers)."""
    _lib.load()   # fail loudly if the HIP extension is missing
    _lib.require_device()
    dev = torch.device(device)
    drawer_precision, precision = _lib.split_precision(precision)      # "ref": f32 decoder + fp16 towers (the reference's GPU mix)
    settings = types.SimpleNamespace(vqgan_model=vqgan_model, size=tuple(size), weight_seed=seed,
                                     vqgan_config=None, vqgan_checkpoint=None, precision=drawer_precision)
    drawer = VqganDrawer(settings)
    drawer.load_model(settings, dev)
    drawer.init_from_tensor(None if init_image is None else init_image.to(dev) * 2 - 1)
    per_rank = num_cuts // world_size
    clip_models = [clip_model] if isinstance(clip_model, str) else list(clip_model)      # an ensemble shares one decoder pass
    perceptors, cutouts, pms_table = {}, {}, {}
    for mi, name in enumerate(clip_models):
        perceptor = get_clip_perceptor(name, dev, max_batch=per_rank, seed=seed + 1 + 10 * mi, group=group, tokenizer=tokenizer,
                                       text_params=clip_text_params, precision=precision)
        perceptors[name] = perceptor
        if perceptor.input_resolution not in cutouts:                                   # pixray.py:643-649: one table per size
            cutouts[perceptor.input_resolution] = MakeCutouts(perceptor.input_resolution, num_cuts,
                                                              generator=torch.Generator().manual_seed(1000 + seed + mi),
                                                              aspect_width=size[0] / size[1])     # pixray.py:1931
        pms = []
        for prompt in prompts:                                                              # pixray.py:859-877
            txt, weight, stop = parse_prompt(prompt)
            pms.append(Prompt(perceptor.encode_text(txt).float(), weight, stop).to(dev))
        for vp in vector_prompts:                                                           # pixray.py:887-915
            f1, weight, stop = parse_prompt(vp)
            table = load_vector_table(f1)
            if name not in table:                       # pixray.py:906-910: warn and go on without it
                print(f"WARNING: no vector for {name} in {f1}!")
                continue
            pms.append(Prompt(torch.tensor(table[name], dtype=torch.float32), 0.1 * weight, stop).to(dev))
        pe = prompt_embeds[name] if isinstance(prompt_embeds, dict) else (prompt_embeds if mi == 0 else None)
        if pe is None and (stand_in_prompt or not pms):
            pe = seeded_unit_vectors(1, perceptor.output_dim, seed + 2 + mi)
        if pe is not None:
            pms.insert(0, Prompt(pe.to(dev), prompt_weight, float("-inf")).to(dev))
        for (emb, w, stop) in (extra_prompts if mi == 0 else ()):
            pms.append(Prompt(emb.to(dev), w, stop).to(dev))
        pms_table[name] = pms
    return Session(drawer, perceptors, cutouts, pms_table,
                   learning_rate=learning_rate, iterations=iterations, custom_losses=custom_losses, filters=filters,
                   seed=seed, group=group, rank=rank, world_size=world_size, learning_rate_drops=learning_rate_drops,
                   image_prompts={name: [t.to(dev).float() for t in image_prompts] for name in clip_models} if len(image_prompts) else None,
                   image_prompt_weight=image_prompt_weight, image_prompt_shuffle=image_prompt_shuffle,
                   init_weight=init_weight, init_weight_dist=init_weight_dist, init_weight_pix=init_weight_pix,
                   init_weight_cos=init_weight_cos, z_orig=drawer.get_z_copy().detach() if init_image is not None else None,
                   init_image_tensor=None if init_image is None else init_image.to(dev).float(),
                   overlay_image=overlay_image, overlay_every=overlay_every, overlay_offset=overlay_offset,
                   overlay_until=overlay_until, overlay_alpha=overlay_alpha,
                   comm=_maybe_oneshot_comm(group, rank, world_size, size))


def build_fft_clip_session(*, size=(512, 512), clip_model="ViT-L/14", num_cuts=256, iterations=250, seed=0, device="cuda",
                           group=None, rank=0, world_size=1, custom_losses=(), args=None, prompt_embeds=None,
                           precision: Optional[str] = None, fft_lrate: float = 0.3, fft_decay: float = 1.5) -> Session:
    """BASELINE.json configs[3]'s shape: the spectrum drawer plugin (`FftDrawer`, its own Adam, no z) + one CLIP perceptor +
    MakeCutouts + a Prompt + a custom-loss stack handed in by the caller ([{"loss": LossInterface, "weight": w}], e.g.
    StyleLoss + SaturationLoss; `args` is what their `parse_settings` returned)."""
    from .fft_drawer import FftDrawer
    _lib.load()
    _lib.require_device()
    dev = torch.device(device)
    st = types.SimpleNamespace(size=tuple(size), fft_use="fft", fft_decay=fft_decay, fft_lrate=fft_lrate, weight_seed=seed)
    drawer = FftDrawer(st)
    drawer.load_model(st, dev)
    drawer.init_from_tensor(None)
    per_rank = num_cuts // world_size
    precision = _lib.split_precision(precision)[1]        # no decoder here: "ref" = the fp16 tower
    perceptor = get_clip_perceptor(clip_model, dev, max_batch=per_rank, seed=seed + 1, group=group, precision=precision)
    mk = MakeCutouts(perceptor.input_resolution, num_cuts, generator=torch.Generator().manual_seed(1000 + seed),
                     aspect_width=size[0] / size[1])
    pe = prompt_embeds if prompt_embeds is not None else seeded_unit_vectors(1, perceptor.output_dim, seed + 2)
    pms = {clip_model: [Prompt(pe.to(dev), 1.0, float("-inf")).to(dev)]}
    return Session(drawer, {clip_model: perceptor}, {perceptor.input_resolution: mk}, pms, iterations=iterations,
                   custom_losses=list(custom_losses), args=args, seed=seed, group=group, rank=rank, world_size=world_size,
                   comm=_maybe_oneshot_comm(group, rank, world_size, size))


# BASELINE.json `configs` as (builder kwargs); configs[0] is the CPU plumbing case (tests/test_host_logic.py), configs[4]
# (vdiff) has no source in the reference checkout
WORKLOADS = {
    # every configuration carries pixray's default second prompt, `--vector_prompts textoff` at weight x0.1 (pixray.py:887-915,
    # 1732), wherever the reference's table has the tower (it has no ViT-L/14 row: the reference warns and goes on without it)
    "cfg1": dict(kind="vqgan", size=(256, 256), clip_model="ViT-B/32", num_cuts=64, vector_prompts=("textoff",),
                 text="vqgan imagenet_f16_16384 256x256 + CLIP ViT-B/32 + 64 cutouts, 2 prompts (text-prompt stand-in x1.0 + the "
                      "reference's default `textoff` vector prompt x0.1), Adam lr 0.2"),
    "cfg2": dict(kind="vqgan", size=(512, 512), clip_model=["ViT-B/16", "RN50x4"], num_cuts=128, vector_prompts=("textoff",),
                 text="vqgan imagenet_f16_16384 512x512 + CLIP ViT-B/16 + RN50x4 ensemble, 128 cutouts per perceptor, 2 prompts per "
                      "perceptor (stand-in + `textoff` x0.1), Adam lr 0.2"),
    "cfg3": dict(kind="fft", size=(512, 512), clip_model="ViT-L/14", num_cuts=256, vector_prompts=(),
                 text="fft spectrum drawer 512x512 + CLIP ViT-L/14 + 256 cutouts + StyleLoss (VGG16 STROTSS) + SaturationLoss, Adam lr 0.3"),
}


def build_workload(name: str, *, num_cuts=None, precision=None, device="cuda", group=None, rank=0, world_size=1, seed=0,
                   custom_losses=(), args=None) -> Session:
    """A BASELINE.json configuration by name ("cfg1" = configs[1], the headline; "cfg2"; "cfg3"), seeded random weights of
    the real architectures.  `num_cuts` overrides the configuration's cutout count (per-GPU shard sizes)."""
    w = WORKLOADS[name]
    n = int(num_cuts) if num_cuts else w["num_cuts"]
    if w["kind"] == "vqgan":
        return build_vqgan_clip_session(size=w["size"], vqgan_model="imagenet_f16_16384", clip_model=w["clip_model"], num_cuts=n,
                                        learning_rate=0.2, iterations=10 ** 9, seed=seed, device=device, group=group, rank=rank,
                                        world_size=world_size, precision=precision, custom_losses=custom_losses,
                                        vector_prompts=w["vector_prompts"], stand_in_prompt=True)
    sess = build_fft_clip_session(size=w["size"], clip_model=w["clip_model"], num_cuts=n, iterations=10 ** 9, seed=seed,
                                  device=device, group=group, rank=rank, world_size=world_size, custom_losses=custom_losses,
                                  args=args, precision=precision)
    # engine.Session.custom_backward_last (perceptor backward enqueued before the plugins' backward) measured neutral on
    # configs[3]: 366.6 vs 358.0 ms per step; PRX_CUSTOM_BACKWARD_LAST=1 switches it on for an A/B
    sess.custom_backward_last = os.environ.get("PRX_CUSTOM_BACKWARD_LAST", "0") == "1"
    return sess


def session_gemm_contexts(sess: Session):
    """The `prx_gemm_ctx` of every runner handle a session drives (drawer, perceptors, HIP-backed custom losses): bench.py
    and tools/gemm_*.py enable per-launch timing / tile rules on these (the library itself has no global switch)."""
    ctxs = []
    h = getattr(sess.drawer, "handle", None)
    if h is not None and hasattr(h, "gemm_ctx"):
        ctxs.append(h.gemm_ctx)
    for p in sess.perceptors.values():
        h = getattr(p, "handle", None)
        if h is not None and hasattr(h, "gemm_ctx"):
            ctxs.append(h.gemm_ctx)
    for t in sess.custom_losses:
        owner = getattr(t["loss"], "extractor", None) or t["loss"]       # StyleLoss keeps its VGG16 runner in `.extractor`
        h = getattr(owner, "handle", None)
        if h is not None and hasattr(h, "gemm_ctx"):
            ctxs.append(h.gemm_ctx)
    return ctxs


class GemmProfile:
    """Per-launch GEMM timing (HIP events on the launch stream, include/prx.h prx_profile_gemm_*) over every runner
    handle of a session; `tile_rule` applies a per-shape tile / split-K override to all of them (tools/gemm_rules.py)."""

    def __init__(self, sess: Session):
        self.lib = _lib.load()
        self.sess = sess

    @property
    def ctxs(self):
        # resolved at every use: a runner handle can be re-created behind the session's back (Vgg16Extractor.reserve grows its
        # handle when a larger input arrives), and a cached `prx_gemm_ctx*` would then point into freed memory
        return session_gemm_contexts(self.sess)

    def enable(self, on: bool = True):
        for c in self.ctxs:
            self.lib.prx_profile_gemm_enable(c, int(bool(on)))

    def collect(self):
        """-> (total ms, total flop, launches) since the last collect"""
        import ctypes
        tot_ms = tot_fl = 0.0
        tot_n = 0
        for c in self.ctxs:
            ms, fl, n = ctypes.c_double(), ctypes.c_double(), ctypes.c_longlong()
            if self.lib.prx_profile_gemm_collect(c, ctypes.addressof(ms), ctypes.addressof(fl), ctypes.addressof(n)) != 0:
                raise _lib.PrxError("prx_profile_gemm_collect failed: " + _lib.last_error())
            tot_ms += ms.value; tot_fl += fl.value; tot_n += n.value
        return tot_ms, tot_fl, tot_n

    def tile_rule(self, M, N, K, mode, bm, bn, splits):
        for c in self.ctxs:
            self.lib.prx_gemm_tile_rule(c, M, N, K, mode, bm, bn, splits)


def phase_breakdown(sess: Session, first_iteration: int, iters: int = 2) -> Dict[str, float]:
    """Where an iteration's time goes, in ms per iteration: every phase of the forward pass bracketed by device
    synchronisations (so the phases do not overlap and the sum is slower than the free-running loop), the rest of
    `train()` reported as backward + optimiser.  Measurement aid for bench.py (`phase_ms`); the wrappers are removed
    before it returns."""
    import time
    dev = sess.drawer.get_z().device if sess.drawer.get_z() is not None else sess.drawer.params[0].device
    acc: Dict[str, float] = {}
    undo = []

    def wrap(obj, attr, key):
        fn = getattr(obj, attr)

        def timed(*a, **k):
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            r = fn(*a, **k)
            torch.cuda.synchronize(dev)
            acc[key] = acc.get(key, 0.0) + (time.perf_counter() - t0)
            return r
        setattr(obj, attr, timed)
        undo.append((obj, attr, fn))

    wrap(sess.drawer, "synth", "drawer_synth")
    for size, mk in sess.cutoutsTable.items():
        wrap(mk, "forward", f"cutouts_{size}")
    for name, p in sess.perceptors.items():
        wrap(p, "encode_image", f"encode_{name}")
    for t in sess.custom_losses:
        wrap(t["loss"], "get_loss", f"loss_{type(t['loss']).__name__}")
    wrap(sess, "ascend_txt", "forward_total")
    try:
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for i in range(iters):
            sess.train(first_iteration + i)
        torch.cuda.synchronize(dev)
        total = time.perf_counter() - t0
    finally:
        for obj, attr, fn in undo:
            try:
                delattr(obj, attr)            # instance attribute shadowing the class's method
            except AttributeError:
                setattr(obj, attr, fn)
    out = {k: round(1e3 * v / iters, 3) for k, v in acc.items()}
    out["backward_and_step"] = round(1e3 * (total - acc.get("forward_total", 0.0)) / iters, 3)
    out["total_serialised"] = round(1e3 * total / iters, 3)
    return out
