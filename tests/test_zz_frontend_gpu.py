"""The front end (pixray_amd/frontend.py) on the PRODUCT's parts: HIP VQGAN drawer, HIP cutouts, HIP CLIP tower, HIP prompt
loss, fused Adam -- settings dictionary in, PNG with metadata out.  (Named to run after the kernel / parity suites.)"""
import os
import sys

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))

pytestmark = pytest.mark.gpu
DEV = None          # None: the front end's own choice (cuda:0 on a GPU box); tests/test_emu_cpu.py sets "cpu" for the emulated device


def test_settings_to_png_on_the_hip_path(tmp_path):
    from PIL import Image
    from pixray_amd import frontend as fe
    from pixray_amd.engine import HipAdam
    run = fe.Run()
    run.settings = dict(drawer="vqgan", vqgan_model="tiny_f4", clip_models="tiny-B/32", size=[66, 64], num_cuts=8, iterations=6, save_every=3,
                        display_every=4, outdir=str(tmp_path / "out"), seed=3, skip_args=True, init_noise="none", vector_prompts="none",
                        noise_prompt_seeds=[1, 2], noise_prompt_weights=[1.0, 0.5], precision="fp16", learning_rate_drops=[])
    s = fe.apply_settings(run=run)
    sess = fe.do_init(s, run, device=DEV)
    assert sess.drawer.size == (64, 64)                                 # 66 rounded down to a multiple of 2^(resolutions-1)
    if DEV is None:
        assert sess.drawer.get_z().is_cuda and isinstance(sess.opts[0], HipAdam)
    assert len(sess.pmsTable["tiny-B/32"]) == 2
    z0 = sess.drawer.get_z_copy()
    seen = []
    while True:
        done = fe.do_run(s, return_display=True, run=run)
        seen.append(sess.cur_iteration)
        if done:
            break
    assert seen == [4, 6]
    assert float((sess.drawer.get_z() - z0).abs().max()) > 1e-3
    assert all(torch.isfinite(l).all() for l in sess.last_losses)
    img = Image.open(os.path.join(s.outdir, "output.png"))
    assert img.size == (64, 64) and img.text["pixray_seed_used"] == "3" and img.text["pixray_vqgan_model"] == "tiny_f4"
    assert sorted(os.listdir(os.path.join(s.outdir, "steps"))) == ["frame_0000.png", "frame_0003.png", "frame_0006.png"]
    a = np.asarray(Image.open(os.path.join(s.outdir, "steps", "frame_0000.png")), dtype=np.int32)
    b = np.asarray(Image.open(os.path.join(s.outdir, "steps", "frame_0006.png")), dtype=np.int32)
    assert np.abs(a - b).max() > 0                                      # the image moved


@pytest.mark.parametrize("size", [(96, 64), (45, 32), (512, 512)])
def test_fft_drawer_hip_path(size):
    """csrc/fft_drawer.hip (the FftDrawer's map on a GPU) on the device: image and d/d(spectrum) against the explicit-DFT oracle,
    bit-identical when evaluated twice"""
    import types
    from oracle import fft_ref
    from pixray_amd.fft_drawer import FftDrawer
    st = types.SimpleNamespace(size=size, fft_use="fft", fft_decay=1.5, fft_lrate=0.3, weight_seed=3)
    dr = FftDrawer(st)
    dr.load_model(st, "cuda")
    dr.init_from_tensor(None)
    assert dr.hip
    img = dr.synth(0)
    p = fft_ref.rand_init(size, 3)
    ref = fft_ref.synth(p, size)
    assert float((img.detach().cpu() - ref.detach()).abs().max()) < 5e-6
    proj = torch.randn(ref.shape, generator=torch.Generator().manual_seed(1))
    (g,) = torch.autograd.grad((img * proj.cuda()).sum(), dr.params[0])
    (gr,) = torch.autograd.grad((ref * proj).sum(), p)
    assert float((g.cpu() - gr).norm() / gr.norm()) < 2e-5
    img2 = dr.synth(0)                                                    # the std's sums are taken in a fixed order
    (g2,) = torch.autograd.grad((img2 * proj.cuda()).sum(), dr.params[0])
    assert torch.equal(img2, img) and torch.equal(g2, g)
