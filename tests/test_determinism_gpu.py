"""Run-to-run reproducibility (GPU).  bf16 activations make the decoder chaotic down to its rounding noise floor: a
1-ulp fp32 difference in one GroupNorm statistic flips a few bf16 roundings and is amplified ~4x per stage to ~3e-3 on
the image (and several % on dL/dz).  So EVERY kernel on the path must be bit-reproducible -- including the cutout
backward, which the reference runs on grid_sampler_2d_backward's float atomics (the non-determinism pixray.py:29
documents) and this package in gather form (cutouts.hip): the whole iteration is bit-identical from run to run."""
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


@pytest.mark.parametrize("kw", [dict(size=(64, 64), vqgan_model="tiny_f4", clip_model="tiny-B/32", num_cuts=8, seed=3),
                                dict(size=(112, 64), vqgan_model="tiny_f4", clip_model="tiny-B/32", num_cuts=8, seed=5),
                                dict(size=(256, 256), vqgan_model="imagenet_f16_16384", clip_model="ViT-B/32", num_cuts=64, seed=0)],
                         ids=["reduced", "widescreen", "headline"])
def test_whole_iteration_is_bit_reproducible(kw):
    """two sessions with the same seed, two iterations each (even / odd: reflection / border padding): identical loss AND
    bit-identical dL/dz and z after the Adam step -- no kernel on the path sums in a run-dependent order"""
    a = api.build_vqgan_clip_session(**kw)
    b = api.build_vqgan_clip_session(**kw)
    for mk in list(a.cutoutsTable.values()) + list(b.cutoutsTable.values()):
        mk.noise_fac = 0.0            # a and b would draw different device noise
    for it in range(2):
        a.train(it); b.train(it)
        la, lb = sum(l.detach() for l in a.last_losses), sum(l.detach() for l in b.last_losses)
        assert torch.equal(la, lb), (it, float(la), float(lb))
        assert torch.equal(a.drawer.get_z().grad, b.drawer.get_z().grad), (it, "dL/dz differs between runs")
        assert torch.equal(a.drawer.get_z(), b.drawer.get_z()), (it, "z differs after the optimiser step")


@pytest.mark.parametrize("cutn,S,HW,it", [(10, 224, 256, 0), (10, 224, 256, 1), (8, 64, 40, 1)])
def test_cutout_backward_bit_reproducible(cutn, S, HW, it):
    from pixray_amd import cutouts as pc
    g = torch.Generator().manual_seed(9 + it)
    img = torch.rand(1, 3, HW, HW, generator=g).to(DEV)
    prm = pc.sample_cutout_params(cutn, S, g, iteration=it)
    prm["noise"] = torch.randn(cutn, 3, S, S, generator=g)
    gout = torch.randn(cutn, 3, S, S, generator=g).to(DEV)
    grads = []
    for rep in range(3):
        mk = pc.MakeCutouts(S, cutn)
        mk.fixed_params = prm
        x = img.clone().requires_grad_(True)
        (gx,) = torch.autograd.grad(mk(x), x, gout)
        grads.append(gx)
    assert torch.equal(grads[0], grads[1]) and torch.equal(grads[0], grads[2])
