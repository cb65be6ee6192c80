"""CPU tests (no GPU) of the StyleLoss plugin's STROTSS arithmetic (pixray_amd/style_loss.py) against the golden vectors
produced by the REFERENCE'S OWN code (tests/golden/styleloss_golden.npz, made by tests/golden/make_golden.py from
Losses/StyleLoss.py) -- and, when /root/reference is present, against that code run live.  The VGG16 features come from the
CPU oracle here (test infrastructure); the HIP extractor is tested in test_path_gpu.py / test_e2e_gpu.py."""
import argparse
import os
import sys
import types
import warnings

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)

import _refextract as rx  # noqa: E402
from oracle import vgg_ref  # noqa: E402
from pixray_amd import style_loss as sl  # noqa: E402
from pixray_amd import weights  # noqa: E402

GOLD = np.load(os.path.join(HERE, "golden", "styleloss_golden.npz"))


class OracleExtractor:
    """the plugin's extractor surface on the CPU oracle's VGG16 (channels-last maps, like the HIP extractor's)"""

    def __init__(self, params, space="uniform"):
        self.params, self.space = params, space

    def __call__(self, x):
        return [f.permute(0, 2, 3, 1).contiguous() for f in vgg_ref.forward(self.params, x, self.space)]

    def forward_samples_hypercolumn(self, X, samps=100):
        return sl.sample_hypercolumns(self(X), samps)


@pytest.fixture(scope="module")
def extractor():
    return OracleExtractor(weights.synthetic_vgg16_params(0))


@pytest.mark.parametrize("tag", ["a", "b"])
def test_strotss_loss_and_gradient_match_the_reference_goldens(extractor, tag):
    img = torch.from_numpy(GOLD[f"img_{tag}"]).requires_grad_(True)
    style = torch.from_numpy(GOLD[f"style_{tag}"])
    np.random.seed(int(GOLD[f"seed_{tag}"]))
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        loss = sl.strotss_loss(img, style, float(GOLD[f"cw_{tag}"]), extractor=extractor)
    (g,) = torch.autograd.grad(loss, img)
    assert abs(float(loss.detach()) - float(GOLD[f"loss_{tag}"])) < 1e-5 * max(1.0, abs(float(GOLD[f"loss_{tag}"])))
    gr = torch.from_numpy(GOLD[f"grad_{tag}"])
    assert float((g - gr).norm() / gr.norm()) < 1e-4


def test_hypercolumn_sampling_matches_the_reference_golden(extractor):
    np.random.seed(5)
    cols = extractor.forward_samples_hypercolumn(torch.from_numpy(GOLD["style_a"]), samps=40)
    assert tuple(cols.shape) == (1, sl.N_FEATURE_CHANNELS, 40)
    assert float((cols - torch.from_numpy(GOLD["hyper"])).abs().max()) < 1e-5


@pytest.mark.skipif(not rx.available(), reason="/root/reference not present (GPU box)")
@pytest.mark.parametrize("hw", [(132, 140), (77, 131)], ids=lambda t: f"{t[0]}x{t[1]}")
def test_strotss_against_the_reference_code_run_live(extractor, hw):
    """other sizes and seeds than the fixtures, the reference's functions executed here: three scales on an even canvas, and an
    odd non-square one (every max pool floors an odd side, the sampling grid is ragged, the coarsest maps are 4 x 8) -- the sizes
    `draw_plan` has to predict from the canvas shape alone"""
    ns = rx.styleloss_ns()
    ref_ex = rx.reference_vgg_extractor(ns, weights.synthetic_vgg16_params(0))
    g = torch.Generator().manual_seed(31)
    img = torch.rand(1, 3, *hw, generator=g)
    style = torch.rand(1, 3, *hw, generator=g)
    a = img.clone().requires_grad_(True)
    b = img.clone().requires_grad_(True)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        np.random.seed(77)
        la = ns["strotss_loss"](a, style, 8.0, extractor=ref_ex)
        np.random.seed(77)
        lb = sl.strotss_loss(b, style, 8.0, extractor=extractor)
    (ga,) = torch.autograd.grad(la, a)
    (gb,) = torch.autograd.grad(lb, b)
    assert abs(float(la) - float(lb)) < 1e-5 * max(1.0, abs(float(la)))
    assert float((ga - gb).norm() / ga.norm()) < 1e-4


def test_plugin_settings_skip_schedule_and_style_file(tmp_path, extractor):
    """StyleLoss.py:458-500: argparse settings, the style image read from --style_file and resized (bicubic) to the canvas,
    zero loss before --styleloss_skip and off the --styleloss_every grid, STROTSS otherwise"""
    from PIL import Image
    rng = np.random.RandomState(3)
    Image.fromarray((rng.rand(40, 52, 3) * 255).astype(np.uint8)).save(tmp_path / "style.png")
    parser = sl.StyleLoss.add_settings(argparse.ArgumentParser())
    args = parser.parse_args(["--style_file", str(tmp_path / "style.png"), "--styleloss_skip", "2", "--styleloss_every", "2",
                              "--styleloss_content_weight", "4"])
    loss = sl.StyleLoss(extractor=extractor, device="cpu")
    args = loss.parse_settings(args)
    out = torch.rand(1, 3, 72, 66, requires_grad=True)
    assert float(loss.get_loss({}, out, args, globals={"cur_iteration": 1})) == 0.0
    assert float(loss.get_loss({}, out, args, globals={"cur_iteration": 3})) == 0.0
    assert tuple(loss.resized.shape) == (1, 3, 72, 66)
    np.random.seed(0)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        val = loss.get_loss({}, out, args, globals={"cur_iteration": 4})
    (g,) = torch.autograd.grad(val, out)
    assert torch.isfinite(val) and float(val) > 0 and float(g.abs().sum()) > 0
    with pytest.raises(ValueError):
        sl.StyleLoss(extractor=extractor).get_loss({}, out, args, globals={"cur_iteration": 4})      # no style image


def test_extractor_without_weights_or_gpu_fails_loudly(monkeypatch):
    monkeypatch.delenv("PIXRAY_VGG16_CKPT", raising=False)
    with pytest.raises(RuntimeError, match="VGG16 weights"):
        sl.Vgg16Extractor()
    if not torch.cuda.is_available():
        with pytest.raises(Exception):
            sl.Vgg16Extractor(params=weights.synthetic_vgg16_params(0), device="cpu")      # no CPU fallback


def test_torchvision_vgg16_checkpoint_adapter():
    """a torchvision-format state dict (features.* + classifier.*, optionally DataParallel-prefixed) is reduced to the 26
    feature-extractor tensors; a wrong shape or a missing tensor is an error"""
    from pixray_amd import checkpoints
    sd = {("module." + k): v.half() for k, v in weights.synthetic_vgg16_params(1).items()}
    sd["module.classifier.0.weight"] = torch.zeros(8, 8)
    out = checkpoints.vgg16_from_torchvision({"state_dict": sd})
    assert list(out) == list(weights.vgg16_param_shapes()) and all(t.dtype == torch.float32 for t in out.values())
    bad = dict(sd)
    bad["module.features.5.weight"] = torch.zeros(128, 64, 1, 1)
    with pytest.raises(ValueError):
        checkpoints.vgg16_from_torchvision(bad)
    del sd["module.features.28.bias"]
    with pytest.raises(KeyError):
        checkpoints.vgg16_from_torchvision(sd)


class _FakeRing:
    """cutouts.PinnedRing without pinned memory: one fixed 'device' buffer per table, filled by stage()"""

    def __init__(self, shape, dtype, device, slots=4):
        self.dev = torch.empty(shape, dtype=dtype, device=device)
        self.stages = 0

    def stage(self, value):
        self.dev.copy_(value)
        self.stages += 1
        return self.dev


def test_draw_plan_consumes_numpys_stream_like_the_loss_body(extractor):
    """`draw_plan` (all of an evaluation's numpy draws, made from tensor SHAPES ahead of the device work) leaves numpy's
    global generator exactly where an evaluation that draws as it goes leaves it, and yields the same loss"""
    img = torch.from_numpy(GOLD["img_a"])
    style = torch.from_numpy(GOLD["style_a"])
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        np.random.seed(11)
        la = sl.strotss_loss(img, style, 8.0, extractor=extractor)
        after_a = np.random.randint(1 << 30)
        np.random.seed(11)
        plan = sl.plan_to_device(sl.draw_plan(img.shape[2], img.shape[3], style.shape[2], style.shape[3]), img.device)
        after_b = np.random.randint(1 << 30)
        lb = sl.strotss_loss(img, style, 8.0, extractor=extractor, plan=plan)
    assert after_a == after_b
    assert float(la) == float(lb)
    assert sl.vgg_map_shapes(img.shape[2], img.shape[3]) == [tuple(f.shape[1:3]) for f in extractor(img)]


def test_staged_tables_for_graph_replay_match_drawing_inside_get_loss(extractor, monkeypatch):
    """the hipGraph-replay protocol of the plugin on CPU tensors (rings of pinned memory replaced by plain buffers):
    host_prep() makes the iteration's draws and stages them into FIXED buffers, get_loss() consumes them -- same loss as the
    eager plugin with the same numpy seed, buffers reused from one iteration to the next, and graph_state() reports the
    --styleloss_skip / --styleloss_every schedule a captured iteration bakes in"""
    from pixray_amd import cutouts
    monkeypatch.setattr(cutouts, "PinnedRing", _FakeRing)
    img = torch.from_numpy(GOLD["img_a"])
    style = torch.from_numpy(GOLD["style_a"])
    args = sl.StyleLoss.add_settings(argparse.ArgumentParser()).parse_args(["--styleloss_skip", "2", "--styleloss_every", "2",
                                                                           "--styleloss_content_weight", "8"])
    eager = sl.StyleLoss(extractor=extractor, style_image=style, device="cpu")
    staged = sl.StyleLoss(extractor=extractor, style_image=style, device="cpu")
    assert staged.supports_graph_replay and not staged.graph_capturable
    assert not sl.StyleLoss(extractor=extractor, style_image=style, device="cpu", reference_schedule=True).supports_graph_replay
    staged.enable_static_buffers("cpu")
    assert staged.graph_capturable
    ptrs = None
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for it in range(6):
            np.random.seed(100 + it)
            le = eager.get_loss({}, img, args, globals={"cur_iteration": it})
            np.random.seed(100 + it)
            staged.host_prep(args, it)                  # iteration 0: canvas size not seen yet -> nothing staged, get_loss draws
            used_stage = staged._staged is not None
            ls = staged.get_loss({}, img, args, globals={"cur_iteration": it})
            active = it >= 2 and it % 2 == 0
            assert staged.graph_state(args, it) == (active, (img.shape[2], img.shape[3]))
            assert used_stage == active
            assert (float(le) != 0.0) == active and float(le) == float(ls)
            if active:
                now = [r.dev.data_ptr() for r in staged._rings]
                assert ptrs is None or ptrs == now      # the device work of every iteration reads the same addresses
                ptrs = now
    assert all(r.stages == 2 for r in staged._rings)        # iterations 2 and 4
