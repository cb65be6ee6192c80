"""The experiment behind `oracle/step_ref.compare_k_steps` teacher-forcing the IMAGE: the reference's
dL/d(image) is not a continuous function of the image.  ORACLE ONLY (CPU, fp32) -- no HIP code involved.

`MakeCutouts` pools the canvas with an adaptive average AND an adaptive MAX pool (pixray.py:443,463); the max pool's gradient goes
to the first maximum of a window.  After a couple of Adam steps a few per cent of the pixels sit exactly on the bounds of
`clamp_with_grad` (vqgan.py:66-79,195), so whole windows tie, and a pixel that lands 1e-7 inside the bound instead of on it
re-routes gradient spikes.  Measured here, at the oracle's own state after `--steps` Adam steps:

  * 1e-6 of noise on the image moves the oracle's dL/d(image) by `rel_max` (tenths) and dL/dz by `dz_rel_max`;
  * with the max pool swapped for a second average pool the same noise moves dL/d(image) by `rel_avg` (1e-4 or less).

    python tools/oracle_tie_sensitivity.py                       # the headline configuration (minutes on a few cores)
    python tools/oracle_tie_sensitivity.py --reduced              # the toy graph of smoke() (seconds; what tests/test_oracle_pins.py runs)
"""
import argparse
import json
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def measure(vqgan_model="imagenet_f16_16384", clip_model="ViT-B/32", size=(256, 256), cutn=64, seed=0, steps=2, lr=0.2, eps=1e-6):
    from oracle import clip_vit_ref, cutouts_ref, prompt_ref, step_ref, vqgan_ref
    vq_cfg, clip_cfg, vq_params, clip_params = step_ref._oracle_inputs(vqgan_model, clip_model, seed)
    S = clip_cfg.input_resolution
    f = 2 ** (vq_cfg.num_resolutions - 1)
    g = torch.Generator().manual_seed(seed)
    z = torch.randn(1, vq_cfg.z_channels, size[1] // f, size[0] // f, generator=g).requires_grad_(True)
    prompts = step_ref.prompt_list(clip_model, clip_cfg, seed)
    zmin, zmax = vqgan_ref.z_bounds(vq_params)
    opt = torch.optim.Adam([z], lr=lr)

    def image_grad(img, prm):
        x = img.clone().requires_grad_(True)
        cut = cutouts_ref.make_cutouts(x, prm, S)
        emb = clip_vit_ref.encode_image(clip_params, cut, patch=clip_cfg.patch_size, heads=clip_cfg.heads, layers=clip_cfg.layers)
        loss = sum(prompt_ref.Prompt(e, w, s)(emb) for (e, w, s) in prompts)
        return torch.autograd.grad(loss, x)[0]

    for it in range(steps):                                   # the oracle's own trajectory
        prm = step_ref._draws(cutn, S, seed, it, aspect=size[0] / size[1])
        opt.zero_grad()
        img = vqgan_ref.synth(vq_params, z, vq_cfg.oracle_cfg())
        z.grad, = torch.autograd.grad(img, z, image_grad(img.detach(), prm))
        opt.step()
        with torch.no_grad():
            z.copy_(vqgan_ref.clip_z(z, zmin, zmax))
    prm = step_ref._draws(cutn, S, seed, steps, aspect=size[0] / size[1])
    img = vqgan_ref.synth(vq_params, z, vq_cfg.oracle_cfg())
    base = img.detach()
    noise = torch.randn(base.shape, generator=torch.Generator().manual_seed(seed + 77))
    on_bound = float(((base == 0) | (base == 1)).float().mean())
    rel = lambda a, b: float((a - b).norm() / b.norm())
    g0 = image_grad(base, prm)
    g1 = image_grad(base + eps * noise, prm)
    dz0, = torch.autograd.grad(img, z, g0, retain_graph=True)
    dz1, = torch.autograd.grad(img, z, g1)
    out = dict(config=f"{vqgan_model} {size[0]}x{size[1]} + {clip_model}, {cutn} cutouts, after {steps} oracle Adam steps", noise=eps,
               pixels_on_a_clamp_bound=on_bound, rel_max=rel(g1, g0), dz_rel_max=rel(dz1, dz0))
    pooled = cutouts_ref.pooled_image
    try:                                                       # the same with the max pool swapped for a second average pool
        cutouts_ref.pooled_image = lambda im, s_: F.adaptive_avg_pool2d(im, (s_, s_))
        a0 = image_grad(base, prm)
        a1 = image_grad(base + eps * noise, prm)
    finally:
        cutouts_ref.pooled_image = pooled
    out["rel_avg"] = rel(a1, a0)
    return out


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--reduced", action="store_true")
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--seed", type=int, default=0)
    a = ap.parse_args()
    kw = dict(vqgan_model="tiny_f4", clip_model="tiny-B/32", size=(64, 64), cutn=8) if a.reduced else {}
    print(json.dumps(measure(steps=a.steps, seed=a.seed, **kw)))
