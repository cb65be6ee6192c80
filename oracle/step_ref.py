"""ORACLE (test infrastructure only -- never imported by the product package).

One full iteration of the reference loop on the CPU in fp32, assembled from the restated pieces:
  train()       /root/reference/pixray.py:1436-1512   zero_grad -> ascend_txt -> backward -> Adam -> clip_z
  ascend_txt()  /root/reference/pixray.py:1243-1406   synth -> MakeCutouts -> encode_image -> Prompt
plus harnesses that run the HIP path and this oracle on identical seeds / explicit augmentation draws and
report the parity metrics (used by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg).
Parity status of the assembled path: **unpinned** by the reference's own tests (see the piece modules).
"""
from __future__ import annotations

import time
from typing import Dict, List, Optional, Sequence, Tuple

import torch

from . import clip_vit_ref, cutouts_ref, prompt_ref, vqgan_ref


class _ShardedMinMaxNorm(torch.autograd.Function):
    """(x - min)/(max - min) with min/max taken over the cutouts of ALL ranks (slip.py:21-36 on a sharded batch).
    Same protocol as the HIP path (ops._ClipEncodeFn): all-reduce MIN/MAX forward; backward all-reduces
    {sum g, sum g*y, #argmin, #argmax} so the gradient through min/max lands on the owning rank's pixels."""

    @staticmethod
    def forward(ctx, x, group):
        import torch.distributed as dist
        mm = torch.stack([-x.min(), x.max()])
        dist.all_reduce(mm, op=dist.ReduceOp.MAX, group=group)
        mn, mx = -mm[0], mm[1]
        rng = mx - mn
        y = (x - mn) / rng if rng != 0 else x - mn
        ctx.save_for_backward(x, y, mn, mx)
        ctx.group = group
        return y

    @staticmethod
    def backward(ctx, g):
        import torch.distributed as dist
        x, y, mn, mx = ctx.saved_tensors
        rng = mx - mn
        if rng == 0:
            return g, None
        acc = torch.stack([g.double().sum(), (g.double() * y.double()).sum(), (x == mn).double().sum(),
                           (x == mx).double().sum()])
        dist.all_reduce(acc, op=dist.ReduceOp.SUM, group=ctx.group)
        gmin = ((acc[1] - acc[0]) / rng / acc[2].clamp(min=1)).to(g.dtype)
        gmax = (-acc[1] / rng / acc[3].clamp(min=1)).to(g.dtype)
        return g / rng + (x == mn) * gmin + (x == mx) * gmax, None


class OraclePerceptor:
    """CLIP_Base-shaped wrapper over the oracle tower (CPU). With `group`, the batch-global min/max renorm spans
    the cutouts of every rank."""

    def __init__(self, cfg, params, group=None):
        self.cfg, self.params, self.group = cfg, params, group
        self.input_resolution, self.output_dim = cfg.input_resolution, cfg.output_dim

    def encode_image(self, imgs, input_range=None, apply_preprocess=True):
        c = self.cfg
        if self.group is None:
            return clip_vit_ref.encode_image(self.params, imgs, patch=c.patch_size, heads=c.heads, layers=c.layers,
                                             apply_preprocess=apply_preprocess)
        x = _ShardedMinMaxNorm.apply(imgs, self.group)
        mean = torch.tensor(clip_vit_ref.CLIP_MEAN, dtype=x.dtype).view(1, 3, 1, 1)
        std = torch.tensor(clip_vit_ref.CLIP_STD, dtype=x.dtype).view(1, 3, 1, 1)
        e = clip_vit_ref.vit_forward(self.params, (x - mean) / std, patch=c.patch_size, heads=c.heads, layers=c.layers)
        return e / e.norm(dim=-1, keepdim=True)


class OracleMakeCutouts(torch.nn.Module):
    """MakeCutouts-shaped wrapper over the oracle cutouts (CPU); draws come from `sampler(iteration)`."""

    def __init__(self, cut_size, cutn, sampler):
        super().__init__()
        self.cut_size, self.cutn, self.sampler = cut_size, cutn, sampler
        self.iteration, self.fill, self.shard, self.transforms = 0, None, None, None
        self.last_params = None

    def forward(self, input, spot=None):
        mask = None
        if spot is not None:                     # pixray.py:453-458; `spot_masks` = (inside, outside) bool [3,S,S]
            mask = self.spot_masks[1] if spot == 0 else self.spot_masks[0]
        if self.transforms is not None:          # cached path (pixray.py:480-486): same geometry, no jitter, no noise here
            out = cutouts_ref.make_cutouts_cached(input, self.last_params, self.cut_size, spot_mask=mask)
            return out if self.shard is None else out[self.shard[0]:self.shard[1]]
        prm = self.sampler(self.iteration, self.fill)
        self.last_params = prm
        self.transforms = cutouts_ref.composed_transforms(prm, self.cut_size)
        out = cutouts_ref.make_cutouts(input, prm, self.cut_size, spot_mask=mask)
        if self.shard is not None:
            out = out[self.shard[0]:self.shard[1]]
        return out


def oracle_losses(vq_params, vq_cfg, clip_params, clip_cfg, z, prm, prompts, S):
    img = vqgan_ref.synth(vq_params, z, vq_cfg.oracle_cfg())
    cut = cutouts_ref.make_cutouts(img, prm, S)
    emb = clip_vit_ref.encode_image(clip_params, cut, patch=clip_cfg.patch_size, heads=clip_cfg.heads,
                                    layers=clip_cfg.layers)
    losses = [prompt_ref.Prompt(e, w, s)(emb) for (e, w, s) in prompts]
    return losses, img, cut, emb


def oracle_iteration(vq_params, vq_cfg, clip_params, clip_cfg, z0, prm, prompts, S):
    z = z0.detach().clone().requires_grad_(True)
    losses, img, cut, emb = oracle_losses(vq_params, vq_cfg, clip_params, clip_cfg, z, prm, prompts, S)
    loss = sum(losses)
    loss.backward()
    return dict(loss=loss.detach(), dz=z.grad.detach(), img=img.detach(), emb=emb.detach())


def _metrics(a: torch.Tensor, b: torch.Tensor) -> Tuple[float, float]:
    # fp64: a cosine accumulated in fp32 over 65k terms is only good to ~1e-6 (it came out as 1.000002 and 0.9999998)
    a, b = a.detach().double().cpu().flatten(), b.detach().double().cpu().flatten()
    rel = ((a - b).norm() / (b.norm() + 1e-300)).item()
    cos = (a @ b / (a.norm() * b.norm() + 1e-300)).item()
    return rel, cos


def _build_hip(vqgan_model, clip_model, size, cutn, seed, device, lr=0.2, precision=None):
    from pixray_amd import api
    return api.build_vqgan_clip_session(size=size, vqgan_model=vqgan_model, clip_model=clip_model, num_cuts=cutn,
                                        seed=seed, device=device, learning_rate=lr, precision=precision,
                                        vector_prompts=("textoff",), stand_in_prompt=True)


def prompt_list(clip_model, clip_cfg, seed):
    """the Prompts of the harness as (embed, weight, stop): the seeded stand-in for a text prompt at weight 1, and pixray's
    DEFAULT second prompt -- `--vector_prompts textoff`, weight 0.1 x 1 (pixray.py:887-915, 1732) -- from the reference's own
    table (the one real CLIP-space vector the reference holds; towers without a row in it, e.g. the tiny test towers, go
    without, as the reference does)"""
    from pixray_amd import api
    out = [(api.seeded_unit_vectors(1, clip_cfg.output_dim, seed + 2), 1.0, float("-inf"))]
    table = api.load_vector_table("textoff")
    if clip_model in table:
        out.append((torch.tensor(table[clip_model], dtype=torch.float32), 0.1, float("-inf")))
    return out


def _oracle_inputs(vqgan_model, clip_model, seed):
    from pixray_amd import weights
    vq_cfg = weights.VQGAN_CONFIGS[vqgan_model]
    clip_cfg = weights.CLIP_CONFIGS[clip_model]
    vq_params = weights.synthetic_vqgan_params(vq_cfg, seed)           # same seeds as api.build_vqgan_clip_session
    clip_params = weights.synthetic_clip_vit_params(clip_cfg, seed + 1)
    return vq_cfg, clip_cfg, vq_params, clip_params


JITTER = True     # tests switch the ColorJitter off to separate its (discontinuous) Jacobian from everything else


def _draws(cutn, S, seed, iteration, with_noise=True, aspect=1.0):
    from pixray_amd import cutouts as pc
    g = torch.Generator().manual_seed(5000 + 17 * seed + iteration)
    prm = pc.sample_cutout_params(cutn, S, g, iteration=iteration, aspect=aspect)
    if with_noise:
        prm["noise"] = torch.randn(cutn, 3, S, S, generator=g)
    if not JITTER:
        prm["z_jit_apply"][:] = False
        prm["w_jit_apply"][:] = False
    return prm


def compare_one_iteration(vqgan_model="imagenet_f16_16384", clip_model="ViT-B/32", size=(256, 256), cutn=64, seed=0,
                          device="cuda:0", precision=None, jitter=True) -> Dict[str, float]:
    """dL/dz (and the intermediate image / embeddings / loss) of the HIP path vs the oracle after ONE iteration.
    `precision`: None / "fp16" (the product default), "bf16", or "f32" (the exact-f32 MFMA parity mode); `jitter=False` draws the same
    augmentations with the ColorJitter switched off."""
    global JITTER
    if not jitter:
        JITTER = False
        try:
            return compare_one_iteration(vqgan_model, clip_model, size, cutn, seed, device, precision)
        finally:
            JITTER = True
    from pixray_amd import api
    sess = _build_hip(vqgan_model, clip_model, size, cutn, seed, device, precision=precision)
    vq_cfg, clip_cfg, vq_params, clip_params = _oracle_inputs(vqgan_model, clip_model, seed)
    S = clip_cfg.input_resolution
    prm = _draws(cutn, S, seed, 0, aspect=size[0] / size[1])      # pixray.py:1931: global_aspect_width
    mk = sess.cutoutsTable[S]
    mk.fixed_params = prm
    z0 = sess.drawer.get_z().detach().cpu().clone()
    prompts = prompt_list(clip_model, clip_cfg, seed)
    losses = sess.ascend_txt()
    loss = sum(losses)
    loss.backward()
    dz = sess.drawer.get_z().grad.detach().cpu()
    img_hip = sess.drawer.synth(0).detach().cpu()
    ref = oracle_iteration(vq_params, vq_cfg, clip_params, clip_cfg, z0, prm, prompts, S)
    idx_ref, _ = vqgan_ref.vq_indices(z0.movedim(1, 3).reshape(-1, z0.shape[1]), vq_params["quantize.embedding.weight"])
    rel, cos = _metrics(dz, ref["dz"])
    erel, _ = _metrics(sess.last_embeds, ref["emb"])
    irel, _ = _metrics(img_hip, ref["img"])
    loss = loss.detach()
    return dict(loss_hip=float(loss), loss_ref=float(ref["loss"]), loss_abs_err=abs(float(loss) - float(ref["loss"])),
                dz_rel_l2=rel, dz_cosine=cos, embeds_rel_l2=erel, image_rel_l2=irel,
                indices_equal=bool(torch.equal(sess.drawer.handle.last_indices.cpu().long(), idx_ref)))


def compare_precisions(vqgan_model="imagenet_f16_16384", clip_model="ViT-B/32", size=(256, 256), cutn=64, seed=0,
                       device="cuda:0", fast="bf16") -> Dict[str, float]:
    """A fast path (`fast`: "fp16" | "bf16") against the exact-f32 MFMA mode ON THE DEVICE, same weights / z / augmentation
    draws / noise: what the 16-bit operand rounding alone costs (no oracle involved; both sides are the product's kernels)."""
    out = {}
    res = {}
    for prec in ("f32", fast):
        sess = _build_hip(vqgan_model, clip_model, size, cutn, seed, device, precision=prec)
        S = next(iter(sess.cutoutsTable))
        sess.cutoutsTable[S].fixed_params = _draws(cutn, S, seed, 0, aspect=size[0] / size[1])
        loss = sum(sess.ascend_txt())
        loss.backward()
        res[prec] = dict(dz=sess.drawer.get_z().grad.detach().cpu(), emb=sess.last_embeds.detach().cpu(),
                         img=sess.drawer.synth(0).detach().cpu(), loss=float(loss.detach()),
                         idx=sess.drawer.handle.last_indices.cpu().long())
        del sess
    out["dz_rel_l2"], out["dz_cosine"] = _metrics(res[fast]["dz"], res["f32"]["dz"])
    out["embeds_rel_l2"], _ = _metrics(res[fast]["emb"], res["f32"]["emb"])
    out["image_rel_l2"], _ = _metrics(res[fast]["img"], res["f32"]["img"])
    out["loss_abs_err"] = abs(res[fast]["loss"] - res["f32"]["loss"])
    out["indices_equal"] = bool(torch.equal(res[fast]["idx"], res["f32"]["idx"]))
    return out


def compare_k_steps(k=10, vqgan_model="imagenet_f16_16384", clip_model="ViT-B/32", size=(256, 256), cutn=64, seed=0,
                    device="cuda:0", lr=0.2, precision=None) -> Dict[str, float]:
    """k optimiser steps (train(): synth .. Adam .. clip_z) of the HIP path vs the oracle.

    The loop is chaotic in the dynamical-systems sense: Adam with lr 0.2 moves every component of z by ~0.2 per step
    and the hard VQ argmin (vqgan.py:60-64) turns a 1% gradient difference into a different code -- and a different
    image -- one step later, so two free-running trajectories (even fp32 CPU vs fp32 CUDA of the reference itself)
    decorrelate within a few steps.  Parity is therefore checked TEACHER-FORCED: both sides start every step from the
    oracle's z (and, for the HIP optimiser, the oracle's Adam moments), the per-step dL/dz and the resulting z are
    compared, and the free-running HIP loss curve is reported next to the oracle's for information.

    The teacher forcing extends to the image.  The reference's dL/d(image) is not a continuous function of the image:
    MakeCutouts' adaptive max pool (pixray.py:443,463) routes a window's gradient to its first maximum, and once the
    optimiser has driven pixels onto the clamp bounds (clamp_with_grad, vqgan.py:195: exactly 0.0 / 1.0 -- ~3 % of the
    image after two steps) whole windows tie, so a pixel that lands 1e-7 inside instead of on the bound re-routes gradient
    spikes that carry most of |dL/d(image)| (measured with the ORACLE ALONE at the oracle's step-2 state: noise of 1e-6 on
    the image changes its dL/d(image) by 0.55-0.63 rel-L2 and its dL/dz by up to 0.17).  Two correct implementations
    whose images agree to rounding therefore disagree on dL/dz at such states by a tie, not by an error.  So every step
    (a) gates the image itself (`image_rel_l2_max`), and (b) evaluates the oracle's cutouts -> CLIP -> loss gradient AT
    THE HIP PATH'S IMAGE (same values, same ties) and chains it through the oracle's own decoder backward at the oracle's
    z: dz_ref = J_decoder_ref(z)^T dL/d(image)_ref(image_hip).  Every stage of the HIP iteration is still compared with the
    oracle's arithmetic; only the amplification of image rounding through the ties is taken out."""
    from pixray_amd import api
    sess = _build_hip(vqgan_model, clip_model, size, cutn, seed, device, lr=lr, precision=precision)
    free = _build_hip(vqgan_model, clip_model, size, cutn, seed, device, lr=lr, precision=precision)      # free-running copy
    vq_cfg, clip_cfg, vq_params, clip_params = _oracle_inputs(vqgan_model, clip_model, seed)
    S = clip_cfg.input_resolution
    mk, mk_free = sess.cutoutsTable[S], free.cutoutsTable[S]
    z_ref = sess.drawer.get_z().detach().cpu().clone().requires_grad_(True)
    opt = torch.optim.Adam([z_ref], lr=lr)
    zmin, zmax = vqgan_ref.z_bounds(vq_params)
    prompts = prompt_list(clip_model, clip_cfg, seed)
    dz_rel, dz_cos, z_err, idx_ok, loss_ref, loss_free, free_idx_ok, img_rel, dz_ind = [], [], [], [], [], [], [], [], []
    vq_bad, vq_near = [], []
    seen = {}
    synth_and_filter = sess.do_synth_and_filter

    def capture(result):
        out, alpha = synth_and_filter(result)
        seen["img"] = out.detach().float().cpu().clone()
        return out, alpha
    sess.do_synth_and_filter = capture
    for it in range(k):
        prm = _draws(cutn, S, seed, it, aspect=size[0] / size[1])
        mk.fixed_params = prm
        mk_free.fixed_params = prm
        # ---- HIP side, started from the oracle's state -------------------------------------------------------------
        sess.drawer.set_z(z_ref.detach().to(device))
        hip_opt = sess.opts[0]
        st_ref = opt.state.get(z_ref, None)
        if st_ref:
            st = hip_opt.state[sess.drawer.get_z()]
            st["step"] = int(st_ref["step"])
            st["exp_avg"] = st_ref["exp_avg"].to(device).clone()
            st["exp_avg_sq"] = st_ref["exp_avg_sq"].to(device).clone()
        sess.train(it)
        dz_hip = sess.drawer.get_z().grad.detach().cpu()
        z_hip = sess.drawer.get_z().detach().cpu()
        idx_hip = sess.drawer.handle.last_indices.cpu().long()
        # ---- oracle step ----------------------------------------------------------------------------------------------
        idx_ref, _ = vqgan_ref.vq_indices(z_ref.detach().movedim(1, 3).reshape(-1, z_ref.shape[1]),
                                          vq_params["quantize.embedding.weight"])
        opt.zero_grad()
        img_ref = vqgan_ref.synth(vq_params, z_ref, vq_cfg.oracle_cfg())
        img_rel.append(_metrics(seen["img"], img_ref)[0])
        img_in = seen["img"].clone().requires_grad_(True)
        cut = cutouts_ref.make_cutouts(img_in, prm, S)
        emb = clip_vit_ref.encode_image(clip_params, cut, patch=clip_cfg.patch_size, heads=clip_cfg.heads, layers=clip_cfg.layers)
        losses = [prompt_ref.Prompt(e, w, s)(emb) for (e, w, s) in prompts]
        g_img, = torch.autograd.grad(sum(losses), img_in)
        z_ref.grad, = torch.autograd.grad(img_ref, z_ref, g_img, retain_graph=True)
        r, c = _metrics(dz_hip, z_ref.grad)
        dz_rel.append(r); dz_cos.append(c)
        # informational: the FULLY independent oracle gradient (its own image into its own cutouts -> CLIP -> loss).  Where pixels
        # sit on a clamp bound the reference's dL/d(image) is discontinuous (tools/oracle_tie_sensitivity.py, tests/test_oracle_pins.py),
        # so this figure can jump by tenths at a step without either side being wrong; it is reported, the gate is on the one above
        img_own = img_ref.detach().clone().requires_grad_(True)
        cut_o = cutouts_ref.make_cutouts(img_own, prm, S)
        emb_o = clip_vit_ref.encode_image(clip_params, cut_o, patch=clip_cfg.patch_size, heads=clip_cfg.heads, layers=clip_cfg.layers)
        g_own, = torch.autograd.grad(sum(prompt_ref.Prompt(e, w, s)(emb_o) for (e, w, s) in prompts), img_own)
        dz_own, = torch.autograd.grad(img_ref, z_ref, g_own)
        dz_ind.append(_metrics(dz_hip, dz_own)[0])
        idx_ok.append(float((idx_hip == idx_ref).float().mean()))
        # integer work is exact: the HIP codes obey the float64 near-tie rule at the (teacher-forced, identical) z of this step --
        # and so do the oracle's own fp32 codes, or the rule would be too tight to be a rule
        x_flat = z_ref.detach().movedim(1, 3).reshape(-1, z_ref.shape[1])
        bad_h, near_h, _ = vqgan_ref.vq_exactness(x_flat, vq_params["quantize.embedding.weight"], idx_hip)
        bad_o, _, _ = vqgan_ref.vq_exactness(x_flat, vq_params["quantize.embedding.weight"], idx_ref)
        vq_bad.append(bad_h + bad_o); vq_near.append(near_h)
        opt.step()
        with torch.no_grad():
            z_ref.copy_(vqgan_ref.clip_z(z_ref, zmin, zmax))
        z_err.append(float((z_hip - z_ref.detach()).abs().max()))
        loss_ref.append(float(sum(l.detach() for l in losses)))
        free.train(it)
        loss_free.append(float(sum(l.detach() for l in free.last_losses)))
        free_idx_ok.append(float((free.drawer.handle.last_indices.cpu().long() == idx_ref).float().mean()))
    z_free = free.drawer.get_z().detach().cpu()
    return dict(steps=k, dz_rel_l2_max=max(dz_rel), dz_cosine_min=min(dz_cos), vq_index_agreement_min=min(idx_ok),
                vq_exactness_violations=sum(vq_bad), vq_near_ties_per_step=vq_near,
                image_rel_l2_max=max(img_rel), dz_rel_l2=dz_rel, dz_rel_l2_independent_oracle=dz_ind,
                z_after_step_max_abs_err=max(z_err), loss_oracle=loss_ref, loss_hip_free_running=loss_free,
                # the FREE-RUNNING copy against the oracle's trajectory ("z after 10 Adam steps"): only
                # meaningful while both sides still select the same codes at every step
                free_running_index_agreement_min=min(free_idx_ok),
                z_free_running_rel_l2=float((z_free - z_ref.detach()).norm() / z_ref.detach().norm()),
                z_free_running_max_abs_err=float((z_free - z_ref.detach()).abs().max()))


def time_oracle_iterations(n_iters=3, warmup=1, vqgan_model="imagenet_f16_16384", clip_model="ViT-B/32",
                           size=(256, 256), cutn=64, seed=0, lr=0.2, threads: Optional[int] = None) -> Dict[str, float]:
    """CPU baseline leg of bench.py: the full oracle iteration (incl. Adam + clip_z) timed on the host cores."""
    import os
    from pixray_amd import api, weights
    if threads:
        torch.set_num_threads(threads)
    vq_cfg, clip_cfg, vq_params, clip_params = _oracle_inputs(vqgan_model, clip_model, seed)
    S = clip_cfg.input_resolution
    f = 2 ** (vq_cfg.num_resolutions - 1)
    g = torch.Generator().manual_seed(1)
    z = torch.randn(1, vq_cfg.z_channels, size[1] // f, size[0] // f, generator=g).requires_grad_(True)
    opt = torch.optim.Adam([z], lr=lr)
    zmin, zmax = vqgan_ref.z_bounds(vq_params)
    prompts = prompt_list(clip_model, clip_cfg, seed)
    times = []
    for it in range(warmup + n_iters):
        t0 = time.perf_counter()
        prm = _draws(cutn, S, seed, it, aspect=size[0] / size[1])
        opt.zero_grad()
        losses, *_ = oracle_losses(vq_params, vq_cfg, clip_params, clip_cfg, z, prm, prompts, S)
        sum(losses).backward()
        opt.step()
        with torch.no_grad():
            z.copy_(vqgan_ref.clip_z(z, zmin, zmax))
        if it >= warmup:
            times.append(time.perf_counter() - t0)
    mean = sum(times) / len(times)
    return dict(seconds_per_iter=mean, iters_per_sec=1.0 / mean, threads=torch.get_num_threads(), cores=os.cpu_count(),
                iters=n_iters)
