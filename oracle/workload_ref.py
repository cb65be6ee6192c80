"""ORACLE (test infrastructure only): one iteration of each BASELINE.json configuration on the CPU, fp32 torch, built from
the per-operator restatements in this directory -- the checker for `pixray_amd.api.build_workload(...)` at the
configurations' own sizes and the `cpu_baseline` leg of bench.py.  Never imported by the product package.

  cfg1  vqgan 256x256 + ViT-B/32, 64 cutouts                      (configs[1], the headline)
  cfg2  vqgan 512x512 + ViT-B/16 + RN50x4 ensemble, 128 cutouts   (configs[2]; pixray.py:1266-1299: one synth, one
        cutout table per input resolution, one encode_image + Prompt list per perceptor)
  cfg3  fft drawer 512x512 + ViT-L/14, 256 cutouts + StyleLoss + SaturationLoss (configs[3]; custom losses
        pixray.py:1388-1398)

The seeds mirror `pixray_amd.api.build_vqgan_clip_session / build_fft_clip_session`, so the product session and this
oracle hold the same weights, the same start point and (through `fixed_params`) the same augmentation draws.
"""
from __future__ import annotations

import time
from typing import Dict, List, Optional

import torch

from . import clip_resnet_ref, clip_vit_ref, cutouts_ref, fft_ref, prompt_ref, vgg_ref, vqgan_ref


class SaturationLossRef:
    """the reference's SaturationLoss plugin (/root/reference/Losses/SaturationLoss.py:15-30) restated: a colourfulness score
    from std / mean of the opponent colour axes over ALL cutout pixels, one term per cutout table.  Batch-coupled."""
    needs_full_batch = True

    def get_loss(self, cur_cutouts, out, args, globals=None, lossGlobals=None):
        res = []
        for _, cutouts in cur_cutouts.items():
            px = cutouts.permute(0, 2, 3, 1).reshape(-1, 3)
            rg, yb = px[:, 0] - px[:, 1], 0.5 * (px[:, 0] + px[:, 1]) - px[:, 2]
            rg_std, rg_mean = torch.std_mean(rg)
            yb_std, yb_mean = torch.std_mean(yb)
            res.append(-(torch.sqrt(rg_std ** 2 + yb_std ** 2) + 0.3 * torch.sqrt(rg_mean ** 2 + yb_mean ** 2)) / 10.0)
        return res


class OracleVggExtractor:
    """the StyleLoss plugin's extractor surface (`Vgg16_Extractor`, Losses/StyleLoss.py:24-81) on the CPU VGG16 oracle"""

    def __init__(self, params):
        self.params = params

    def __call__(self, x):
        return [f.permute(0, 2, 3, 1).contiguous() for f in vgg_ref.forward(self.params, x, "uniform")]

    def forward_samples_hypercolumn(self, X, samps=100):
        from pixray_amd import style_loss as sl          # the STROTSS arithmetic is pinned to the reference (tests/test_style_loss.py)
        return sl.sample_hypercolumns(self(X), samps)


def towers_of(workload: str, seed: int = 0):
    """[(name, kind, cfg, params)] in the order api.build_* creates the perceptors"""
    from pixray_amd import api, weights
    models = api.WORKLOADS[workload]["clip_model"]
    models = [models] if isinstance(models, str) else list(models)
    kind = api.WORKLOADS[workload]["kind"]
    out = []
    for mi, name in enumerate(models):
        s = seed + 1 + (10 * mi if kind == "vqgan" else 0)
        if name in weights.CLIP_RESNET_CONFIGS:
            cfg = weights.CLIP_RESNET_CONFIGS[name]
            out.append((name, "resnet", cfg, weights.synthetic_clip_resnet_params(cfg, s)))
        else:
            cfg = weights.CLIP_CONFIGS[name]
            out.append((name, "vit", cfg, weights.synthetic_clip_vit_params(cfg, s)))
    return out


def encode(kind, cfg, params, cut):
    if kind == "resnet":
        return clip_resnet_ref.encode_image(params, cut, layers=cfg.layers, heads=cfg.heads)
    return clip_vit_ref.encode_image(params, cut, patch=cfg.patch_size, heads=cfg.heads, layers=cfg.layers)


def draws_for(workload: str, cutn: int, seed: int, iteration: int = 0, with_noise: bool = True) -> Dict[int, dict]:
    """explicit augmentation draws per cutout table (keyed by input resolution), as the parity tests hand them to both sides"""
    from pixray_amd import api
    from pixray_amd import cutouts as pc
    size = api.WORKLOADS[workload]["size"]
    out = {}
    for name, kind, cfg, _ in towers_of(workload, seed):
        S = cfg.input_resolution
        if S in out:
            continue
        g = torch.Generator().manual_seed(5000 + 17 * seed + iteration + 131 * S)
        prm = pc.sample_cutout_params(cutn, S, g, iteration=iteration, aspect=size[0] / size[1])
        if with_noise:
            prm["noise"] = torch.randn(cutn, 3, S, S, generator=g)
        out[S] = prm
    return out


def image_of(workload: str, seed: int, state: Optional[torch.Tensor] = None):
    """-> (leaf tensor the optimiser owns, function leaf -> image [1,3,H,W])"""
    from pixray_amd import api, weights
    w = api.WORKLOADS[workload]
    if w["kind"] == "vqgan":
        cfg = weights.VQGAN_CONFIGS["imagenet_f16_16384"]
        params = weights.synthetic_vqgan_params(cfg, seed)
        f = 2 ** (cfg.num_resolutions - 1)
        if state is None:
            g = torch.Generator().manual_seed(1)          # VqganDrawer.rand_init
            zmin, zmax = vqgan_ref.z_bounds(params)
            state = vqgan_ref.clip_z(torch.randn(1, cfg.z_channels, w["size"][1] // f, w["size"][0] // f, generator=g), zmin, zmax)
        leaf = state.detach().clone().requires_grad_(True)
        return leaf, (lambda z: vqgan_ref.synth(params, z, cfg.oracle_cfg()))
    # the fft drawer (configs[3]): the oracle's own restatement of the spectrum -> image map (explicit DFT sums, oracle/fft_ref.py),
    # not the product's torch.fft class
    size = tuple(w["size"])
    leaf = fft_ref.rand_init(size, seed) if state is None else state.detach().clone().float().requires_grad_(True)
    return leaf, (lambda p: fft_ref.synth(p, size, decay=1.5, contrast=0.9, colors=1.5))


def iteration(workload: str, cutn: int, seed: int = 0, prm: Optional[Dict[int, dict]] = None, state=None, custom=(), args=None,
              cur_iteration: int = 0):
    """one forward + backward of the configuration on the CPU -> dict(losses, grad, img, embeds)"""
    from pixray_amd import api
    prm = prm if prm is not None else draws_for(workload, cutn, seed)
    leaf, synth = image_of(workload, seed, state)
    img = synth(leaf)
    kind = api.WORKLOADS[workload]["kind"]
    losses: List[torch.Tensor] = []
    cuts, emb = {}, None
    for mi, (name, tkind, cfg, params) in enumerate(towers_of(workload, seed)):
        S = cfg.input_resolution
        if S not in cuts:
            cuts[S] = cutouts_ref.make_cutouts(img, prm[S], S)
        emb = encode(tkind, cfg, params, cuts[S])
        e = api.seeded_unit_vectors(1, cfg.output_dim, seed + 2 + (mi if kind == "vqgan" else 0))
        losses.append(prompt_ref.Prompt(e, 1.0, float("-inf"))(emb))
        for vp in api.WORKLOADS[workload]["vector_prompts"]:          # pixray's default `textoff` x0.1 (pixray.py:887-915)
            table = api.load_vector_table(vp)
            if name in table:
                losses.append(prompt_ref.Prompt(torch.tensor(table[name], dtype=torch.float32), 0.1, float("-inf"))(emb))
    for t in custom:
        r = t["loss"].get_loss(cuts, img, args, globals={"cur_iteration": cur_iteration, "embeds": emb}, lossGlobals={})
        losses += [t["weight"] * l for l in (r if isinstance(r, (list, tuple)) else [r])]
    total = sum(losses)
    (grad,) = torch.autograd.grad(total, leaf)
    return dict(losses=[float(l.detach()) for l in losses], grad=grad.detach(), img=img.detach(), embeds=emb.detach(),
                start=leaf.detach().clone())


def _metrics(a, b):
    a, b = a.detach().double().cpu().flatten(), b.detach().double().cpu().flatten()
    return ((a - b).norm() / (b.norm() + 1e-300)).item(), (a @ b / (a.norm() * b.norm() + 1e-300)).item()


def hip_gradient(workload: str, cutn: int, prec: str, prm, seed: int = 0, device: str = "cuda:0", custom_hip=(), args=None, before=None):
    """one iteration of the HIP path (product session, `prec` operand precision) on the explicit draws `prm`
    -> dict(grad, losses, embeds, start), everything on the CPU.  `before()` runs right in front of the iteration (e.g. seeding
    numpy's global generator, which the StyleLoss plugin samples from, the way the oracle side was seeded)"""
    from pixray_amd import api
    sess = api.build_workload(workload, num_cuts=cutn, precision=prec, device=device, seed=seed, custom_losses=custom_hip, args=args)
    for S, mk in sess.cutoutsTable.items():
        mk.fixed_params = prm[S]
    leaf = sess.drawer.get_z() if sess.drawer.get_z() is not None else sess.drawer.params[0]
    start = leaf.detach().cpu().clone()
    if before is not None:
        before()
    losses = sess.ascend_txt()
    sum(losses).backward()
    out = dict(grad=leaf.grad.detach().cpu(), losses=[float(l.detach()) for l in losses], embeds=sess.last_embeds.detach().cpu(),
               start=start)
    del sess
    return out


def compare_with(ref: dict, hip: dict) -> Dict[str, float]:
    """the parity figures of one HIP iteration against an oracle iteration (`iteration(...)`'s dict, or a golden fixture)"""
    rel, cos = _metrics(hip["grad"], ref["grad"])
    return dict(grad_rel_l2=rel, grad_cosine=cos, losses_hip=hip["losses"], losses_ref=list(ref["losses"]),
                loss_abs_err=max(abs(a - b) for a, b in zip(hip["losses"], ref["losses"])),
                embeds_rel_l2=_metrics(hip["embeds"], ref["embeds"])[0])


def compare_workload(workload: str, cutn: int, precisions=("bf16",), seed: int = 0, device: str = "cuda:0", custom_factory=None,
                     custom_ref=(), args=None, ref: Optional[dict] = None, before=None) -> Dict[str, Dict[str, float]]:
    """gradient w.r.t. the optimised tensor (z, or the fft drawer's spectrum) after ONE iteration: HIP path (one session per
    entry of `precisions`) vs this oracle (evaluated once, or handed in as `ref` -- e.g. a committed full-size fixture of
    tools/fullsize_oracle.py), same weights, same start, same explicit augmentation draws and noise.
    `custom_factory(precision) -> [{"loss", "weight"}]` builds the HIP-side custom losses."""
    prm = draws_for(workload, cutn, seed)
    out, grads = {}, {}
    for prec in precisions:
        custom_hip = custom_factory(prec) if custom_factory is not None else ()
        hip = hip_gradient(workload, cutn, prec, prm, seed, device, custom_hip, args, before)
        if ref is None:
            if before is not None:
                before()
            ref = iteration(workload, cutn, seed, prm, state=hip["start"], custom=custom_ref, args=args)
        out[prec] = compare_with(ref, hip)
        grads[prec] = hip["grad"]
    for fast in precisions:
        if fast != "f32" and "f32" in grads:
            r, c = _metrics(grads[fast], grads["f32"])
            out[f"{fast}_vs_f32"] = dict(grad_rel_l2=r, grad_cosine=c)
    return out


def time_workload(workload: str, sample_cutn: int, n_iters: int = 3, warmup: int = 1, seed: int = 0, custom=(), args=None,
                  threads: Optional[int] = None) -> Dict[str, float]:
    """CPU-baseline leg of bench.py: `n_iters` full oracle iterations (forward + backward + the optimiser step is negligible)
    at `sample_cutn` cutouts, plus the time of the part that does not depend on the cutout count, so that the caller can state the sample and extrapolate the
    cutout-proportional part to the configuration's own cutout count."""
    import os
    if threads:
        torch.set_num_threads(threads)
    times, t_img = [], []
    for it in range(warmup + n_iters):
        prm = draws_for(workload, sample_cutn, seed, iteration=it)
        t0 = time.perf_counter()
        iteration(workload, sample_cutn, seed, prm, custom=custom, args=args, cur_iteration=it)
        dt = time.perf_counter() - t0
        # the part that does not scale with the cutout count: drawer forward + backward, plus the custom losses that read
        # only the image (StyleLoss; a batch-coupled loss such as SaturationLoss reads the cutouts and scales with them)
        t1 = time.perf_counter()
        leaf, synth = image_of(workload, seed)
        img = synth(leaf)
        fixed = [img.sum()]
        for t in custom:
            if not getattr(t["loss"], "needs_full_batch", False):
                r = t["loss"].get_loss({}, img, args, globals={"cur_iteration": it, "embeds": None}, lossGlobals={})
                fixed += [t["weight"] * l for l in (r if isinstance(r, (list, tuple)) else [r])]
        torch.autograd.grad(sum(fixed), leaf)
        di = time.perf_counter() - t1
        if it >= warmup:
            times.append(dt); t_img.append(di)
    mean, mean_img = sum(times) / len(times), sum(t_img) / len(t_img)
    return dict(seconds_per_iter=mean, fixed_seconds=mean_img, sample_cutn=sample_cutn, iters=n_iters,
                threads=torch.get_num_threads(), cores=os.cpu_count())
