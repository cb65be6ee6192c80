"""Run-to-run reproducibility (GPU).  bf16 activations make the decoder chaotic down to its rounding noise floor: a
1-ulp fp32 difference in one GroupNorm statistic flips a few bf16 roundings and is amplified ~4x per stage to ~3e-3 on
the image (and several % on dL/dz).  So every kernel on the path except the cutout scatter-add (fp32 atomics, the same
as the reference's grid_sampler backward, pixray.py:29) must be bit-reproducible; this pins it."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from pixray_amd import ops, weights, api

DEV = "cuda"


@pytest.mark.parametrize("vq_name,size", [("tiny_f4", 64), ("imagenet_f16_16384", 256)])
def test_vqgan_synth_and_backward_bit_reproducible(vq_name, size):
    cfg = weights.VQGAN_CONFIGS[vq_name]
    f = 2 ** (len(cfg.ch_mult) - 1)
    vh = ops.VqganHandle(cfg, weights.synthetic_vqgan_params(cfg, seed=0), (size // f, size // f), device=DEV)
    torch.manual_seed(3)
    z0 = torch.randn(1, cfg.z_channels, size // f, size // f, device=DEV)
    res = []
    for rep in range(3):
        z = z0.clone().requires_grad_(True)
        img = ops.vqgan_synth(z, vh)
        gi = torch.linspace(-1, 1, img.numel(), device=DEV).reshape(img.shape)
        (dz,) = torch.autograd.grad(img, z, gi)
        res.append((img.detach().clone(), dz.clone()))
    for img, dz in res[1:]:
        assert torch.equal(img, res[0][0]), "decoder forward differs between runs"
        assert torch.equal(dz, res[0][1]), "decoder backward differs between runs"


@pytest.mark.parametrize("clip_name,cutn", [("tiny-B/32", 8), ("ViT-B/32", 64)])
def test_clip_encode_and_backward_bit_reproducible(clip_name, cutn):
    cfg = weights.CLIP_CONFIGS[clip_name]
    ch = ops.ClipVitHandle(cfg, weights.synthetic_clip_vit_params(cfg, seed=0), max_batch=cutn, device=DEV)
    torch.manual_seed(4)
    cut0 = torch.rand(cutn, 3, cfg.input_resolution, cfg.input_resolution, device=DEV)
    res = []
    for rep in range(3):
        cut = cut0.clone().requires_grad_(True)
        emb = ops.clip_encode_image(cut, ch)
        ge = torch.linspace(-1, 1, emb.numel(), device=DEV).reshape(emb.shape)
        (dc,) = torch.autograd.grad(emb, cut, ge)
        res.append((emb.detach().clone(), dc.clone()))
    for emb, dc in res[1:]:
        assert torch.equal(emb, res[0][0]), "CLIP forward differs between runs"
        assert torch.equal(dc, res[0][1]), "CLIP backward differs between runs"


def test_iteration_forward_bit_reproducible_and_grad_close():
    """two sessions with the same seed: identical loss (the whole forward is deterministic); dL/dz differs only through
    the order of the cutout backward's fp32 atomics"""
    kw = dict(size=(64, 64), vqgan_model="tiny_f4", clip_model="tiny-B/32", num_cuts=8, seed=3)
    a = api.build_vqgan_clip_session(**kw)
    b = api.build_vqgan_clip_session(**kw)
    for mk in list(a.cutoutsTable.values()) + list(b.cutoutsTable.values()):
        mk.noise_fac = 0.0            # a and b would draw different device noise
    for s in (a, b):
        s._host_prep(0)
        for opt in s.opts:
            opt.zero_grad(set_to_none=True)
    la = sum(a.ascend_txt()); lb = sum(b.ascend_txt())
    assert torch.equal(la.detach(), lb.detach()), (float(la), float(lb))
    la.backward(); lb.backward()
    ga, gb = a.drawer.get_z().grad, b.drawer.get_z().grad
    rel = ((ga - gb).norm() / ga.norm()).item()
    assert rel < 1e-3, rel
