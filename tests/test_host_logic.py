"""CPU tests (no GPU): host-side logic of the package.

* the 33 assertions of the reference's own test-suite (tests/test_util.py, tests/test_pixray.py) ported to
  pixray_amd.settings;
* the C-ABI library loads and exports every symbol include/prx.h declares (no compute calls);
* cutout parameter sampling / descriptor construction;
* BASELINE.json configs[0]: the plugin surface end to end on the CPU (pixel-grid drawer, ViT-B/32, cutn=2, 10
  iterations) with the *oracle* supplying the perceptor / cutout / prompt arithmetic -- the Session loop is the product
  code under test, the oracle parts are test infrastructure;
* unmodified reference plugins (Losses/SaturationLoss.py, Losses/SymmetryLoss.py, filters/colorlookup.py loaded from
  /root/reference when present) drop into the loop.
"""
import argparse
from collections import OrderedDict
import ctypes
import importlib.util
import os
import sys
import types

import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _refextract as rx
from oracle import step_ref, prompt_ref
from pixray_amd import _lib, cutouts as pc, weights
from pixray_amd.engine import Session
from pixray_amd.interfaces import DrawingInterface, FilterInterface, LossInterface, install_compat_modules
from pixray_amd.pixel_grid_drawer import PixelGridDrawer
from pixray_amd.settings import apply_overlay, get_file_path, get_learning_rate_drops, parse_unit, split_pipes


# ------------------------------------------------------------------ reference test-suite port (33 assertions)
def test_get_file_path_reference_cases():
    assert get_file_path('/testpath', 'testfile', '.png') == '/testpath/testfile.png'
    assert get_file_path('/testpath/', 'testfile', '.png') == '/testpath/testfile.png'
    with pytest.raises(ValueError):
        get_file_path('/testpath/', '\\test\\filename.png', '.png')
    with pytest.raises(ValueError):
        get_file_path('/testpath/', '/test/filename.png', '.png')
    assert get_file_path('', 'testfile', '.png') == 'testfile.png'
    with pytest.raises(ValueError):
        get_file_path('/testpath/', None, '.png')
    with pytest.raises(ValueError):
        get_file_path('/testpath/', ' ', '.png')
    assert get_file_path('/testpath', 'testfile.png', '.mp4') == '/testpath/testfile.mp4'


def test_parse_unit_reference_cases():
    assert parse_unit('200iterations', 500, 'overlay_until') == 200
    assert parse_unit('200 i', 500, 'overlay_until') == 200
    assert parse_unit('50%', 500, 'overlay_until') == 250
    assert parse_unit('33 percent', 500, 'overlay_until') == 165
    with pytest.raises(ValueError):
        parse_unit(' percent', 500, 'overlay_until')
    assert parse_unit(None, 500, 'overlay_until') is None
    assert parse_unit('200 iterATions    ', 500, 'overlay_until') == 200
    assert parse_unit('50', 500, 'overlay_until') == 250
    assert parse_unit('50', 500, 'overlay_until', 'i') == 50
    assert parse_unit(50, 500, 'overlay_until', 'i') == 50
    assert parse_unit(.6, 500, 'overlay_until', 'i') == 0
    assert parse_unit(.5, 500, 'overlay_until', 'p') == 2
    with pytest.raises(ValueError):
        parse_unit('67.i', 500, 'overlay_until')


def test_split_pipes_reference_cases():
    assert split_pipes(None) is None
    assert split_pipes('test|another') == ['test', 'another']
    assert split_pipes('') == ''
    assert split_pipes('single') == ['single']


def _overlay_args(image, every, offset, until, iterations=250):
    a = types.SimpleNamespace(overlay_image=image, iterations=iterations)
    a.overlay_offset = parse_unit(offset, iterations, "overlay_offset")
    a.overlay_until = parse_unit(until, iterations, "overlay_until")
    a.overlay_every = parse_unit(every, iterations, "overlay_every")
    return a


def test_apply_overlay_reference_cases():
    assert apply_overlay(_overlay_args('image.png', '1i', '0i', '100i'), 10) is True
    assert apply_overlay(_overlay_args(None, '1i', '0i', '100i'), 10) is False
    assert apply_overlay(_overlay_args('image.png', '5i', '10i', '100i'), 10) is False
    assert apply_overlay(_overlay_args('image.png', '5i', '10i', None), 10) is False
    assert apply_overlay(_overlay_args('image.png', '1i', '0i', '5i'), 10) is False


def test_get_learning_rate_drops_reference_cases():
    assert get_learning_rate_drops(None, 300) == []
    assert get_learning_rate_drops([75], 300) == [224]
    assert get_learning_rate_drops([50, 22.5], 300) == [149, 67]


# ------------------------------------------------------------------ C ABI
def test_library_loads_and_exports_every_declared_symbol():
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__ as ge
        ge.build()
    protos = _lib.parse_header()
    assert len(protos) >= 35
    lib = ctypes.CDLL(_lib.LIB_PATH) if False else _lib.load()
    for name in protos:
        assert hasattr(lib, name), f"{name} declared in include/prx.h but not exported"
    assert lib.prx_abi_version() == 3
    for must in ("prx_vqgan_synth", "prx_vqgan_synth_backward", "prx_cutouts_forward", "prx_cutouts_backward",
                 "prx_clip_vit_encode", "prx_clip_vit_backward_reduce", "prx_clip_vit_backward_finish",
                 "prx_prompt_loss_fwd_bwd", "prx_adam_clamp_step", "prx_last_error"):
        assert must in protos


def test_ops_fail_loudly_without_a_gpu():
    from pixray_amd import ops
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(_lib.PrxError):
        ops.prompt_loss(torch.randn(4, 64), torch.randn(1, 64))
    with pytest.raises(_lib.PrxError):
        ops.make_cutouts(torch.rand(1, 3, 32, 32), torch.zeros(2, 36, dtype=torch.float64), None, 16)
    with pytest.raises(_lib.PrxError):          # the StyleLoss hyper-column op has no CPU route either
        ops.hypercolumns([torch.randn(1, 4, 4, 3)], torch.zeros(1, 4, 2, dtype=torch.int64), torch.zeros(6, 2))
    from pixray_amd import api
    with pytest.raises(_lib.PrxError):
        api.build_vqgan_clip_session()


# ------------------------------------------------------------------ cutout parameters
def test_cutout_params_follow_the_reference_distributions():
    g = torch.Generator().manual_seed(0)
    p = pc.sample_cutout_params(64, 224, g, iteration=3)
    assert int(0.6 * 64) == 38 and p["z_persp_rand"].shape == (38, 4, 2) and p["w_trans"].shape == (26, 2)
    assert int(p["reflect"]) == 0                       # odd iteration -> border (pixray.py:1250-1253)
    crop = p["z_crop"]
    area = crop[:, 2] * crop[:, 3] / (224 * 224)
    assert ((area > 0.2) & (area < 1.0)).all()
    assert (crop[:, 0] + crop[:, 2] <= 224).all() and (crop[:, 1] + crop[:, 3] <= 224).all()
    assert (p["z_sat"] >= 0.9).all() and (p["z_sat"] <= 1.1).all() and (p["w_hue"].abs() <= 0.1).all()
    assert (p["w_trans"].abs() <= 0.025 * 224).all() and (p["noise_fac"] <= 0.1).all()
    d = pc.build_descriptors(p, 224)
    assert d.shape == (64, pc.DESC_WORDS) and d.dtype == torch.float64 and torch.isfinite(d).all()
    # same seed -> same draws (ranks must agree when the batch is sharded)
    p2 = pc.sample_cutout_params(64, 224, torch.Generator().manual_seed(0), iteration=3)
    assert torch.equal(pc.build_descriptors(p2, 224), d)


def test_descriptor_geometry_matches_oracle_grids():
    """the pixel positions implied by a descriptor equal the oracle's kornia-style sampling grid"""
    from oracle import cutouts_ref as cr
    S = 32
    g = torch.Generator().manual_seed(4)
    p = pc.sample_cutout_params(10, S, g, iteration=0)
    d = pc.build_descriptors(p, S)
    nz = 6
    # zoom stage A (perspective, GRID_MESH): compare against kornia-style normalised grid
    Mp = cr._persp_matrix(p["z_persp_rand"], 0.4, S)
    sn_dn = torch.linalg.inv(cr.normalize_homography(Mp, (S, S), (S, S)))
    for i in range(nz):
        if bool(p["z_persp_apply"][i]):
            assert torch.allclose(d[i, 0:9].reshape(3, 3), sn_dn[i], atol=1e-9)
        else:
            assert int(d[i, 18]) == pc.MODE_IDENT


# ------------------------------------------------------------------ configs[0]: plumbing on the CPU
class _Settings(types.SimpleNamespace):
    pass


def _cpu_session(cutn=2, iters=10, custom_losses=(), filters=(), world=1, rank=0, group=None, seed=0, drawer=None, **extra):
    cfg = weights.CLIP_CONFIGS["ViT-B/32"]
    params = weights.synthetic_clip_vit_params(cfg, 1)
    g = torch.Generator().manual_seed(7)
    if drawer is None:
        st = _Settings(size=(256, 256), pixel_size=(16, 16), pixel_scale=None)
        drawer = PixelGridDrawer(st)
        drawer.load_model(st, "cpu")
        drawer.init_from_tensor(torch.rand(1, 3, 256, 256, generator=g) * 2 - 1)
    else:
        torch.rand(1, 3, 256, 256, generator=g)
    perceptor = step_ref.OraclePerceptor(cfg, params)

    def sampler(iteration, fill):
        gg = torch.Generator().manual_seed(1000 + iteration)
        prm = pc.sample_cutout_params(cutn, 224, gg, iteration=iteration, fill=fill)
        prm["noise"] = torch.randn(cutn, 3, 224, 224, generator=gg)
        return prm
    mk = step_ref.OracleMakeCutouts(224, cutn, sampler)
    e = torch.randn(1, 512, generator=g)
    pm = prompt_ref.Prompt(e / e.norm(), 1.0, float("-inf"))
    pm.denom = None
    args = _Settings(saturation_weight=1.0, symmetry_weight=1.0)
    return Session(drawer, {"ViT-B/32": perceptor}, {224: mk}, {"ViT-B/32": [pm]}, learning_rate=0.03, iterations=iters,
                   custom_losses=custom_losses, filters=filters, args=args, seed=seed, world_size=world, rank=rank,
                   group=group, **extra)


def test_config0_pixel_grid_vit_b32_cutn2_10_iterations_cpu():
    sess = _cpu_session()
    z0 = sess.drawer.get_z_copy()
    losses = []
    for it in range(10):
        assert sess.train(it)
        losses.append(float(sum(sess.last_losses)))
        assert all(torch.isfinite(l).all() for l in sess.last_losses)
    assert isinstance(sess.opts[0], torch.optim.Adam)        # drawer.get_opts() is None -> Adam([z], lr) (pixray.py:537-553)
    assert (sess.drawer.get_z() - z0).abs().max() > 1e-3     # z moved
    assert min(losses[5:]) < losses[0]                       # and the prompt loss went down
    assert sess.drawer.get_z().min() >= 0 and sess.drawer.get_z().max() <= 1      # clip_z
    img = sess.drawer.to_image()
    assert img.size == (256, 256)


def test_fft_drawer_plugin_runs_with_its_own_optimiser_cpu():
    """BASELINE.json configs[3]'s drawer as a plugin (fftdrawer.py:13-110): get_opts() returns its own Adam over the
    spectrum (pixray.py:525-527), get_z() is None, clip_z() is a no-op -- the loop must cope with all of that"""
    from pixray_amd.fft_drawer import FftDrawer, rfft2d_freqs
    st = _Settings(size=(96, 64), fft_use="fft", fft_decay=1.5, fft_lrate=0.3)
    dr = FftDrawer(st)
    dr.load_model(st, "cpu")
    dr.init_from_tensor(None)
    assert dr.params[0].shape == (1, 3, 64, 96 // 2 + 1, 2) and rfft2d_freqs(64, 96).shape == (64, 49)
    img = dr.synth(0)
    assert img.shape == (1, 3, 64, 96) and 0.0 <= float(img.min()) and float(img.max()) <= 1.0
    sess = _cpu_session(cutn=2, drawer=dr)
    assert sess.opts is dr.opts and sess.opts[0].param_groups[0]["lr"] == 0.3
    p0 = dr.params[0].detach().clone()
    for it in range(3):
        assert sess.train(it)
        assert all(torch.isfinite(l).all() for l in sess.last_losses)
    assert (dr.params[0].detach() - p0).abs().max() > 1e-3
    assert dr.get_z() is None and dr.get_num_resolutions() is None
    assert dr.to_image().size == (96, 64)
    # starting from an image reproduces it (up to the contrast normalisation synth applies)
    g = torch.Generator().manual_seed(0)
    t = torch.rand(1, 3, 64, 96, generator=g)
    dr.reapply_from_tensor(t * 2 - 1)
    with torch.no_grad():
        spec = torch.view_as_complex((dr._scale * dr.params[0]).contiguous())
        rec = torch.sigmoid(torch.einsum("nchw,cd->ndhw", torch.fft.irfftn(spec, s=(64, 96), dim=(-2, -1), norm="ortho"), dr._colors))
    assert (rec - t.clamp(1e-3, 1 - 1e-3)).abs().max() < 1e-4
    with pytest.raises(ValueError):
        bad = FftDrawer(_Settings(size=(32, 32), fft_use="dwt")); bad.load_model(None, "cpu"); bad.init_from_tensor(None)


def _load_reference_plugin(relpath, clsname):
    install_compat_modules()
    spec = importlib.util.spec_from_file_location("refplugin_" + clsname, os.path.join(rx.REF, relpath))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return getattr(mod, clsname)


@pytest.mark.skipif(not rx.available(), reason="/root/reference not present (GPU box)")
def test_unmodified_reference_plugins_drop_in():
    Sat = _load_reference_plugin("Losses/SaturationLoss.py", "SaturationLoss")
    Sym = _load_reference_plugin("Losses/SymmetryLoss.py", "SymmetryLoss")
    assert issubclass(Sat, LossInterface) and issubclass(Sym, LossInterface)      # bound to OUR interface module
    parser = Sat.add_settings(Sym.add_settings(argparse.ArgumentParser()))
    ns = parser.parse_args([])
    assert ns.saturation_weight == 1 and ns.symmetry_weight == 1
    losses = [{"loss": Sat(device="cpu"), "weight": 0.5}, {"loss": Sym(device="cpu"), "weight": 2.0}]
    sess = _cpu_session(custom_losses=losses, iters=2)
    out = sess.ascend_txt()
    assert len(out) == 3 and all(o.requires_grad for o in out)       # prompt + saturation (list of 1) + symmetry
    sum(out).backward()
    assert sess.drawer.get_z().grad.abs().sum() > 0


class _HalfBrightFilter(FilterInterface):
    def forward(self, img):
        return img * 0.5, img.mean() * 0.1


class _CustomDrawer(DrawingInterface):
    """a third-party drawer written only against the duck-typed API"""

    def __init__(self, settings):
        self.z = None

    def load_model(self, settings, device):
        pass

    def get_opts(self, decay_divisor):
        return [torch.optim.SGD([self.z], lr=0.5 / decay_divisor)]

    def init_from_tensor(self, t):
        self.z = torch.full((1, 3, 64, 64), 0.3, requires_grad=True)

    def synth(self, cur_iteration):
        return torch.sigmoid(self.z)

    def clip_z(self):
        pass

    def get_z(self):
        return self.z

    def get_z_copy(self):
        return self.z.clone()


def test_filters_and_custom_drawers_drop_in():
    sess = _cpu_session(filters=[{"filter": _HalfBrightFilter(None), "weight": 1.0}], iters=1)
    out = sess.ascend_txt()
    assert len(out) == 2
    d = _CustomDrawer(None)
    d.init_from_tensor(None)
    sess.drawer = d
    sess.opts = sess.rebuild_optimisers()
    assert isinstance(sess.opts[0], torch.optim.SGD)         # drawer-provided optimisers win (pixray.py:525)
    z0 = d.get_z_copy()
    sess.train(0)
    assert not torch.equal(d.get_z(), z0)


def test_learning_rate_drop_rebuilds_optimisers():
    sess = _cpu_session(iters=4)
    sess.learning_rate_drops = [1]
    sess.train(0)
    assert sess.opts[0].param_groups[0]["lr"] == pytest.approx(0.03)
    sess.train(1)
    assert sess.num_loss_drop == 1 and sess.opts[0].param_groups[0]["lr"] == pytest.approx(0.003)


def test_overlay_image_is_pasted_and_re_encoded_on_schedule():
    """pixray.py:731-747, 1408-1420, 1431-1434, 1457-1461: every `overlay_every` iterations (offset, until) the current image
    gets the RGBA overlay pasted through its alpha and goes back into the drawer via reapply_from_tensor"""
    import numpy as np
    from PIL import Image
    rgba = np.zeros((64, 64, 4), dtype=np.uint8)
    rgba[:, :32] = (255, 0, 0, 255)                       # opaque red left half, transparent right half
    sess = _cpu_session(iters=10, overlay_image=Image.fromarray(rgba, "RGBA"), overlay_every=3, overlay_offset=1, overlay_until=6)
    assert sess.overlay_image_rgba.size == (256, 256)     # resized to the canvas (LANCZOS), like pixray.py:742
    assert [it for it in range(10) if sess.apply_overlay(it)] == [1, 4]
    calls = []
    orig = sess.drawer.reapply_from_tensor
    sess.drawer.reapply_from_tensor = lambda t: (calls.append(t.clone()), orig(t))[1]
    before = np.asarray(sess.drawer.to_image()).astype(np.int32)
    sess.train(0)
    assert not calls
    cur = np.asarray(sess.drawer.to_image()).astype(np.int32)
    sess.train(1)                                         # overlay first, then the optimiser step
    assert len(calls) == 1 and tuple(calls[0].shape) == (1, 3, 256, 256)
    t = calls[0]
    assert float(t.min()) >= -1.0 and float(t.max()) <= 1.0
    left, right = ((t[0, :, :, :100] + 1) / 2 * 255).round(), ((t[0, :, :, 160:] + 1) / 2 * 255).round()
    assert torch.equal(left[0], torch.full_like(left[0], 255)) and float(left[1:].abs().max()) == 0      # red where opaque
    assert np.abs(right.permute(1, 2, 0).numpy() - cur[:, 160:]).max() <= 1                              # untouched where transparent
    for it in range(2, 10):
        sess.train(it)
    assert len(calls) == 2                                # iterations 1 and 4 only (until = 6)


def test_do_run_outer_loop_and_display_returns():
    """pixray.py:1614-1631: run to `iterations`; with return_display the loop hands control back every display_every
    iterations and resumes where it stopped"""
    sess = _cpu_session(iters=5)
    calls = []
    orig = sess.train
    sess.train = lambda it=None: (calls.append(it), orig(it))[1]
    assert sess.do_run() is True
    assert calls == [0, 1, 2, 3, 4, 5] and sess.cur_iteration == 5
    sess2 = _cpu_session(iters=5)
    steps = []
    while not sess2.do_run(return_display=True, display_every=2):
        steps.append(sess2.cur_iteration)
    assert steps == [2, 4] and sess2.cur_iteration == 5
    boom = _cpu_session(iters=3)
    boom.train = lambda it=None: (_ for _ in ()).throw(RuntimeError("out of memory"))
    with pytest.raises(RuntimeError):
        boom.do_run()


class _RgbaDrawer(DrawingInterface):
    """a drawer plugin that returns RGBA (pixray's diffvg drawers do): 4 x 8 x 8 parameters upsampled to the canvas"""

    def __init__(self, settings=None):
        self.z = torch.full((1, 4, 8, 8), 0.5).requires_grad_(True)

    def load_model(self, settings, device):
        pass

    def get_opts(self, decay_divisor):
        return None

    def get_z(self):
        return self.z

    def synth(self, cur_iteration):
        return torch.nn.functional.interpolate(self.z, size=(256, 256), mode="nearest").clamp(0, 1)

    def clip_z(self):
        with torch.no_grad():
            self.z.clamp_(0, 1)


def test_rgba_drawers_are_flattened_over_the_iteration_fill_and_the_transparent_loss():
    """pixray.py:1225-1241, 1383-1386: with --transparent an RGBA image is composited over this iteration's random gray and
    mean(alpha) * transparent_weight joins the loss list; without it the alpha channel is dropped"""
    dr = _RgbaDrawer()
    with torch.no_grad():
        dr.z[:, 3] = 0.25
        dr.z[:, 0] = 1.0
    sess = _cpu_session(drawer=dr)
    sess.args.transparent, sess.args.transparent_weight = True, 0.5
    sess._host_prep(0)
    fill = sess.cur_fill
    assert 0.0 <= fill <= 1.0
    out, alpha = sess.do_synth_and_filter([])
    assert tuple(out.shape) == (1, 3, 256, 256) and tuple(alpha.shape) == (1, 256, 256)
    assert torch.allclose(out[0, 0], torch.full((256, 256), 0.25 * 1.0 + 0.75 * fill), atol=1e-6)
    assert torch.allclose(out[0, 1], torch.full((256, 256), 0.25 * 0.5 + 0.75 * fill), atol=1e-6)
    sess._host_ready = False
    assert sess.train(0)
    assert len(sess.last_losses) == 2 and abs(float(sess.last_losses[-1].detach()) - 0.5 * 0.25) < 1e-6
    assert dr.z.grad[:, 3].abs().sum() > 0                      # the alpha channel is trained through both terms
    assert sess.enable_graph() is False
    plain = _cpu_session(drawer=_RgbaDrawer())
    out, alpha = plain.do_synth_and_filter([])
    assert alpha is None and tuple(out.shape) == (1, 3, 256, 256)
    assert plain.train(0) and len(plain.last_losses) == 1


def test_custom_loss_registration_flow():
    """pixray.py:131-140, 961-995, 2104-2109: name[:weight] chunks, `->` instance arguments, parse_settings / add_globals of
    every instance, KeyError for an unknown name, the TypeError hint for a constructor without device=, and the globals
    reaching get_loss through the Session"""
    from pixray_amd import plugins

    class Tint(LossInterface):
        def instance_settings(self, arglist):
            self.channel = int(arglist[0]) if arglist else 0

        def parse_settings(self, args):
            args.tint_seen = True
            return args

        def add_globals(self, args):
            return {"tint_target": 0.25}

        def get_loss(self, cur_cutouts, out, args, globals=None, lossGlobals=None):
            assert args.tint_seen and globals["cur_iteration"] >= 0
            return (out[:, self.channel].mean() - lossGlobals["tint_target"]) ** 2

    class NoDevice(LossInterface):
        def __init__(self):
            pass

    plugins.add_custom_loss("tint", Tint)
    with pytest.raises(AssertionError):
        plugins.add_custom_loss("bad", object)
    args = _Settings(saturation_weight=1.0)
    losses, loss_globals, args = plugins.setup_custom_losses("tint:0.5->2, tint", args, device="cpu")
    assert [t["weight"] for t in losses] == [0.5, 1] and losses[0]["loss"].channel == 2 and losses[1]["loss"].channel == 0
    assert loss_globals == {"tint_target": 0.25} and args.tint_seen
    assert "style" in plugins.loss_class_table
    with pytest.raises(KeyError):
        plugins.setup_custom_losses("nonexistent", args)
    plugins.add_custom_loss("nodev", NoDevice)
    with pytest.raises(TypeError):
        plugins.setup_custom_losses("nodev", args)
    assert plugins.setup_custom_losses(None, args)[:2] == ([], {})
    sess = _cpu_session(custom_losses=losses, loss_globals=loss_globals)
    sess.args = args
    assert sess.train(0) and len(sess.last_losses) == 3 and all(torch.isfinite(l) for l in sess.last_losses)


def test_drawer_and_filter_tables():
    """pixray.py:72-99 / 612-626 (drawer table, size rounding by num_resolutions) and 54-58 / 650-669 (filter table,
    ValueError for an unknown filter, weights parsed like prompts)"""
    from pixray_amd import plugins
    assert {"vqgan", "fft", "fast_pixel"} <= set(plugins.class_table)

    class ThreeLevels(_CustomDrawer):
        def get_num_resolutions(self):
            return 3

    plugins.add_custom_drawer("three", ThreeLevels)
    with pytest.raises(AssertionError):
        plugins.add_custom_drawer("bad", dict)
    st = _Settings(drawer="three", size=(130, 67))
    drawer, side = plugins.make_drawer(st, "cpu")
    assert isinstance(drawer, ThreeLevels) and side == (128, 64)
    st = _Settings(drawer="fast_pixel", size=(130, 67), pixel_size=(13, 6), pixel_scale=None)
    drawer, side = plugins.make_drawer(st, "cpu")
    assert side == (130, 67)                                   # no resolutions: the size is kept
    with pytest.raises(KeyError):
        plugins.make_drawer(_Settings(drawer="nope", size=(8, 8)), "cpu")
    plugins.add_custom_filter("half", _HalfBrightFilter)
    fl = plugins.setup_filters("half:0.25, half", _Settings(), device="cpu")
    assert [f["weight"] for f in fl] == [0.25, 1] and all(isinstance(f["filter"], _HalfBrightFilter) for f in fl)
    with pytest.raises(ValueError, match="Requested filter not found"):
        plugins.setup_filters("sepia", _Settings())
    assert plugins.setup_filters(None, _Settings()) == []


def test_hypercolumn_index_shuffle_is_the_reference_row_shuffle():
    """style_loss.sample_hypercolumns shuffles a 1-D index array instead of the reference's [H*W, 2] coordinate table
    (StyleLoss.py:52-57): same Fisher-Yates walk, same `random_interval` draws -> same rows kept AND the same np.random
    state afterwards (everything drawn later in the iteration is unchanged)."""
    import numpy as np
    H, W, samps = 37, 52, 100
    xx, xy = np.meshgrid(np.arange(H), np.arange(W))
    xc = np.concatenate([np.expand_dims(xx.flatten(), 1), np.expand_dims(xy.flatten(), 1)], 1)
    np.random.seed(11)
    np.random.shuffle(xc)
    state_ref = np.random.get_state()
    np.random.seed(11)
    idx = np.arange(H * W)
    np.random.shuffle(idx)
    state_new = np.random.get_state()
    idx = idx[:samps]
    assert np.array_equal(idx % H, xc[:samps, 0]) and np.array_equal(idx // H, xc[:samps, 1])
    assert np.array_equal(state_ref[1], state_new[1]) and state_ref[2] == state_new[2]


def _taming_yaml(cfg, target="taming.models.vqgan.VQModel"):
    """the layout of taming's model yamls (what vqgan.py:121 loads with OmegaConf)"""
    dd = dict(double_z=False, z_channels=cfg.z_channels, resolution=cfg.resolution, in_channels=3, out_ch=3, ch=cfg.ch,
              ch_mult=list(cfg.ch_mult), num_res_blocks=cfg.num_res_blocks, attn_resolutions=list(cfg.attn_resolutions), dropout=0.0)
    params = dict(embed_dim=cfg.embed_dim, n_embed=cfg.n_embed, ddconfig=dd,
                  lossconfig=dict(target="taming.modules.losses.vqperceptual.VQLPIPSWithDiscriminator", params=dict(disc_start=0)))
    if target.endswith("Net2NetTransformer"):
        params = dict(first_stage_config=dict(target="taming.models.vqgan.VQModel", params=params),
                      transformer_config=dict(target="taming.modules.transformer.mingpt.GPT", params=dict(vocab_size=cfg.n_embed)))
    return dict(model=dict(base_learning_rate=4.5e-6, target=target, params=params))


@pytest.mark.parametrize("target", ["taming.models.vqgan.VQModel", "taming.models.vqgan.GumbelVQ",
                                    "taming.models.cond_transformer.Net2NetTransformer"])
def test_taming_yaml_and_lightning_checkpoint_load(tmp_path, target):
    """vqgan.py:96-140 without taming and without the download: yaml ddconfig -> VqganConfig, Lightning-format .ckpt
    ({"state_dict": ...} carrying discriminator weights, a GumbelVQ codebook name, or a transformer's first-stage prefix) ->
    exactly the tensors the HIP runner is created from, in the C ABI's order; VqganDrawer.load_model picks the pair up."""
    import argparse
    import yaml
    from pixray_amd import checkpoints, weights
    from pixray_amd.vqgan_drawer import VqganDrawer
    cfg = weights.VQGAN_CONFIGS["tiny_f4"]
    dec = weights.synthetic_vqgan_params(cfg, seed=5)
    enc = weights.synthetic_vqgan_encoder_params(cfg, seed=5, codebook=dec["quantize.embedding.weight"])
    sd = OrderedDict(dec)
    sd.update(enc)
    prefix, gumbel = checkpoints.TAMING_TARGETS[target]
    if gumbel:
        sd["quantize.embed.weight"] = sd.pop("quantize.embedding.weight")
        sd["quantize.proj.weight"] = torch.randn(cfg.n_embed, cfg.embed_dim, 1, 1)     # GumbelQuantize's logits projection: kept for encode()
        sd["quantize.proj.bias"] = torch.randn(cfg.n_embed)
    sd = OrderedDict((prefix + k, v) for k, v in sd.items())
    sd[prefix + "loss.discriminator.main.0.weight"] = torch.zeros(8, 3, 4, 4)          # dropped by `del model.loss`
    if prefix:
        sd["transformer.tok_emb.weight"] = torch.zeros(4, 4)
    ypath, cpath = tmp_path / "vqgan_custom.yaml", tmp_path / "vqgan_custom.ckpt"
    ypath.write_text(yaml.safe_dump(_taming_yaml(cfg, target)))
    torch.save({"state_dict": sd, "global_step": 7, "epoch": 0}, cpath)

    got_cfg, params, got_gumbel = checkpoints.load_taming(str(ypath), str(cpath))
    assert got_cfg == cfg and got_gumbel == gumbel
    want = list(OrderedDict.fromkeys(list(weights.vqgan_param_shapes(cfg)) + list(weights.vqgan_encoder_param_shapes(cfg))))
    assert [k for k in params if not k.startswith("quantize.proj")] == want            # decoder entries in the C ABI's order, then the encoder's (codebook shared)
    assert ("quantize.proj.weight" in params) == gumbel                                 # ... and GumbelVQ's logits projection after them
    for k in weights.vqgan_param_shapes(cfg):
        assert torch.equal(params[k], dec[k]), k
    for k in weights.vqgan_encoder_param_shapes(cfg):
        assert torch.equal(params[k], enc[k]), k

    class _Stop(Exception):
        pass

    # the drawer resolves the same pair from --vqgan_config / --vqgan_checkpoint (no GPU here: stop at handle creation)
    args = VqganDrawer.add_settings(argparse.ArgumentParser()).parse_args(
        ["--vqgan_model", "custom", "--vqgan_config", str(ypath), "--vqgan_checkpoint", str(cpath)])
    args.size = (64, 64)
    drawer = VqganDrawer(args)
    from pixray_amd import ops
    real = ops.VqganHandle
    try:
        def stop(cfg_, params_, *a, **k):
            assert cfg_ == cfg and torch.equal(params_["decoder.conv_in.weight"], dec["decoder.conv_in.weight"])
            raise _Stop()
        ops.VqganHandle = stop
        with pytest.raises(_Stop):
            drawer.load_model(args, "cpu")
    finally:
        ops.VqganHandle = real
    # the flag is consulted: a GumbelVQ checkpoint encodes through its own quantiser (quantize.proj logits + gumbel_softmax,
    # vqgan.py:175-185; tests/test_path_gpu.py gumbel_vq_encode_checks runs it), never through the nearest-code lookup --
    # without the projection weights it refuses loudly
    assert drawer.gumbel == gumbel and VqganDrawer(args).gumbel is False
    if gumbel:
        drawer.state_dict = {k: v for k, v in drawer.state_dict.items() if not k.startswith("quantize.proj")}
        with pytest.raises(KeyError, match="quantize.proj"):
            drawer.init_from_tensor(torch.zeros(1, 3, 64, 64))
    bad = VqganDrawer.add_settings(argparse.ArgumentParser()).parse_args(["--vqgan_config", str(tmp_path / "nope.yaml")])
    bad.size = (64, 64)
    with pytest.raises(FileNotFoundError):
        VqganDrawer(bad).load_model(bad, "cpu")
    with pytest.raises(ValueError, match="unknown model type"):
        checkpoints.vqgan_config_from_taming_yaml(dict(model=dict(target="taming.models.other.Thing", params={})))


def test_clip_torchscript_archive_adapter(tmp_path):
    """slip.py:175 `clip.load(name)` reads OpenAI's TorchScript archives: checkpoints.clip_state_dict_from_archive takes
    `torch.jit.load(path).state_dict()` (or a pickled state dict), clip_config_from_state_dict reads the ViT geometry off the shapes
    the way clip.model.build_model does, and the result passes the visual adapter"""
    import torch.nn as nn
    from pixray_amd import checkpoints, weights
    cfg = weights.CLIP_CONFIGS["tiny-B/32"]
    vis = weights.synthetic_clip_vit_params(cfg, 0)

    def build(prefix_params):
        root = nn.Module()
        for name, t in prefix_params.items():
            m = root
            parts = name.split(".")
            for p_ in parts[:-1]:
                if not hasattr(m, p_):
                    m.add_module(p_, nn.Module())
                m = getattr(m, p_)
            m.register_parameter(parts[-1], nn.Parameter(t.clone().half(), requires_grad=False))      # the archives hold fp16 weights
        return root

    class Wrap(nn.Module):
        def __init__(self, inner):
            super().__init__()
            self.visual = inner

        def forward(self, x):
            return x
    mod = Wrap(build(vis))
    path = str(tmp_path / "ViT-tiny.pt")
    torch.jit.save(torch.jit.script(mod), path)
    sd = checkpoints.clip_state_dict_from_archive(path)
    assert "visual.conv1.weight" in sd and sd["visual.proj"].dtype == torch.float16
    got = checkpoints.clip_config_from_state_dict(sd, "tiny-B/32")
    assert (got.input_resolution, got.patch_size, got.width, got.layers, got.heads, got.output_dim) == \
           (cfg.input_resolution, cfg.patch_size, cfg.width, cfg.layers, cfg.heads, cfg.output_dim)
    params = checkpoints.clip_visual_from_openai(sd, got)
    assert all(torch.allclose(params[k], vis[k].half().float()) for k in vis)
    plain = str(tmp_path / "plain.pt")
    torch.save({"state_dict": {"visual." + k: v for k, v in vis.items()}}, plain)
    assert set(checkpoints.clip_state_dict_from_archive(plain)) == {"visual." + k for k in vis}


def _kernel_scratch(src: str, obj: str = None):
    """[(kernel name, scratch bytes per lane)] of every gfx950 kernel of a .hip source.  Read from the object file the build
    left beside it (the AMDGPU metadata note of the device code object inside its fat binary: a second or two) when that object
    is newer than the source and every header; otherwise the source is compiled with the resource remarks on (minutes)."""
    import glob
    import re
    import shutil
    import subprocess
    import tempfile
    llvm = "/opt/rocm/lib/llvm/bin"
    obj = obj or src[:-4] + ".o"
    deps = [src] + glob.glob(os.path.join(os.path.dirname(src), "*.h")) + glob.glob(os.path.join(os.path.dirname(src), "*.inc"))
    fresh = os.path.exists(obj) and all(os.path.getmtime(obj) >= os.path.getmtime(d) for d in deps)
    if fresh and all(os.path.exists(os.path.join(llvm, t)) for t in ("llvm-objcopy", "clang-offload-bundler", "llvm-readelf")):
        with tempfile.TemporaryDirectory() as tmp:
            fat, co = os.path.join(tmp, "k.fatbin"), os.path.join(tmp, "k.co")
            subprocess.run([os.path.join(llvm, "llvm-objcopy"), f"--dump-section=.hip_fatbin={fat}", obj], check=True)
            targets = subprocess.run([os.path.join(llvm, "clang-offload-bundler"), "--list", "--type=o", f"--input={fat}"],
                                     capture_output=True, text=True, check=True).stdout.split()
            target = next(t for t in targets if "gfx950" in t)
            subprocess.run([os.path.join(llvm, "clang-offload-bundler"), "--unbundle", "--type=o", f"--input={fat}", f"--targets={target}",
                            f"--output={co}"], check=True)
            notes = subprocess.run([os.path.join(llvm, "llvm-readelf"), "--notes", co], capture_output=True, text=True, check=True).stdout
        names = re.findall(r"^\s*\.name:\s+(\S+)", notes, flags=re.M)
        scratch = [int(v) for v in re.findall(r"\.private_segment_fixed_size:\s+(\d+)", notes)]
        assert len(names) == len(scratch) and names, (len(names), len(scratch))
        return list(zip(names, scratch))
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    out = subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-c", src, "-o", os.devnull,
                          "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    names = re.findall(r"Function Name: (\S+)", out.stderr)
    scratch = [int(v) for v in re.findall(r"ScratchSize \[bytes/lane\]: (\d+)", out.stderr)]
    assert len(names) == len(scratch)
    return list(zip(names, scratch))


def test_gemm_engine_kernels_have_no_scratch():
    """The GEMM kernels take their descriptor by value; one epilogue variant written the wrong way made hipcc keep that struct
    on the stack (320 bytes of scratch per lane in EVERY kernel of the file) and the engine ran 3x slower with all parity
    tests green.  The compiler's resource figures are the guard: no kernel of gemm.hip / gemmfit.hip may use scratch."""
    for fname, at_least in (("gemm.hip", 20), ("gemmfit.hip", 6), ("gemmfit_spec_tower.hip", 8), ("gemmfit_spec_dec_a.hip", 8),
                            ("gemmfit_spec_dec_b.hip", 8), ("gemmfit_spec_dec_c.hip", 8), ("gemmfit_f32.hip", 8),
                            ("gemmrow_h6.hip", 8), ("gemmrow_h10.hip", 8), ("gemmrow_b6.hip", 8), ("gemmrow_b10.hip", 8), ("gemmrow_h20.hip", 6), ("gemmrow_b20.hip", 6),
                            ("gemmrowconv_h.hip", 8), ("gemmrowconv_b.hip", 8)):   # (the conv kernels hold a tile's 23 operand fragments: 250 registers)
        kernels = _kernel_scratch(os.path.join(os.path.dirname(HERE), "pixray_amd", "csrc", fname))
        assert len(kernels) >= at_least
        bad = [(n, s) for n, s in kernels if s != 0]
        assert not bad, bad


def test_custom_backward_last_is_the_same_gradient():
    """Session.custom_backward_last differentiates the perceptor terms and the custom-loss terms in two backward() calls (a
    launch-ordering choice for host-heavy plugins): the accumulated gradient is the one-pass gradient"""
    class ImgLoss(LossInterface):
        def get_loss(self, cur_cutouts, out, args, globals=None, lossGlobals=None):
            return [(out ** 2).mean(), 0.3 * sum(c.std() for c in cur_cutouts.values())]

    grads = []
    for flag in (False, True):
        sess = _cpu_session(cutn=2, custom_losses=[{"loss": ImgLoss(device="cpu"), "weight": 0.7}])
        sess.custom_backward_last = flag
        z = sess.drawer.get_z()
        for opt in sess.opts:
            opt.zero_grad(set_to_none=True)
        captured = {}
        real_step = sess.opts[0].step
        sess.opts[0].step = lambda *a, **k: captured.setdefault("g", z.grad.detach().clone())    # look, do not move z
        sess._device_step()
        sess.opts[0].step = real_step
        assert len(sess.last_losses) == 3 and sess._n_path_terms == 1
        grads.append(captured["g"])
    assert torch.allclose(grads[0], grads[1], rtol=1e-5, atol=1e-8)
    assert (grads[0] - grads[1]).abs().max() <= 1e-6 * grads[0].abs().max()


def test_fused_adam_stands_in_only_for_a_plain_device_adam():
    """engine.HipAdam.from_adam (hipGraph replay of a drawer plugin with its own torch Adam): anything the fused kernel does
    not implement -- host tensors, several groups, amsgrad / weight decay / maximize, another optimiser class -- is refused,
    and the session then stays on eager launches"""
    from pixray_amd.engine import HipAdam
    p = torch.zeros(4, requires_grad=True)
    assert HipAdam.from_adam(torch.optim.Adam([p], 0.1)) is None                       # host tensor
    assert HipAdam.from_adam(torch.optim.SGD([p], 0.1)) is None
    assert HipAdam.from_adam(torch.optim.AdamW([p], 0.1)) is None
    assert HipAdam.from_adam(torch.optim.Adam([{"params": [p]}, {"params": [torch.zeros(2, requires_grad=True)]}], 0.1)) is None


def test_gemm_planner_gives_the_8phase_kernel_whole_rounds_of_tiles():
    """The launch planner (gemm.hip plan_8phase, queried through prx_gemm_plan_rows_8phase: host code, no device): the 256 x 256
    8-phase kernel gets problems that fill the 256 CUs with whole rounds of 256 x 256 tiles, the rows of a ragged last round go
    to the 4-wave kernels, and everything small -- the headline's M = 3200 products -- stays on the 4-wave kernels.  The shapes
    are the ViT products of BASELINE.json configs[1..3] as measured in profiles/r03_cfg{2,3}_gemm_shapes.txt."""
    from pixray_amd import _lib
    lib = _lib.load()
    plan = lambda M, N, K: lib.prx_gemm_plan_rows_8phase(None, M, N, K)
    # configs[1], ViT-B/32 at 64 cutouts: 13 row tiles x 3..12 column tiles never reach a round of 256
    for N, K in ((3072, 768), (768, 3072), (2304, 768), (768, 2304), (768, 768)):
        assert plan(3200, N, K) == 0
    # configs[3], ViT-L/14 at 256 cutouts: M = 65 792 = 257 row tiles
    assert plan(65792, 1024, 4096) == 65536          # 256 row tiles x 4 = 4 whole rounds; the 257th row tile is peeled
    assert plan(65792, 1024, 1024) == 65536
    assert plan(65792, 1024, 3072) == 65536
    assert plan(65792, 4096, 1024) == 65792          # 4112 tiles = 16.06 rounds: peeling would leave 94 % of a problem on the 4-wave kernels
    assert plan(65792, 3072, 1024) == 65792
    # configs[2], ViT-B/16 at 128 cutouts: M = 25 216 = 98.5 row tiles
    assert plan(25216, 768, 3072) == 21760           # 85 row tiles x 3 = 255 tiles: one round, 3 456 rows peeled
    assert plan(25216, 768, 768) == 21760
    assert plan(25216, 3072, 768) == 25216
    # not eligible: K not a multiple of 128; too few tiles
    assert plan(65792, 1024, 592) == 0
    assert plan(1024, 1024, 4096) == 0
    for M, N, K in ((65792, 1024, 4096), (25216, 768, 3072), (8192, 8192, 8192)):
        r = plan(M, N, K)
        assert r == M or r % 256 == 0


def test_bench_bare_gpus_form_spawns_its_own_ranks():
    """`python bench.py --gpus N` without a launcher (it raised SystemExit) re-executes itself under
    torch.distributed.run with the driver's own flags; --dry-run checks the mechanics on the host: N ranks rendezvous on
    127.0.0.1, rank 0 prints ONE JSON line"""
    import json
    import subprocess
    import sys
    root = os.path.dirname(HERE)
    sys.path.insert(0, root)
    try:
        import bench
    finally:
        sys.path.remove(root)
    cmd = bench.spawn_command(4, ["--gpus", "4", "--steps", "2"], port=29999)
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and "--nproc-per-node=4" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29999"
    assert cmd[-4:] == ["--gpus", "4", "--steps", "2"] and cmd[-5].endswith("bench.py")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--dry-run"], capture_output=True, text=True,
                         timeout=300, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    rec = json.loads(lines[0])
    assert rec == {"dry_run": True, "world": 2, "ranks": [0, 1], "gpus": 2}


def test_reference_precision_mix_splits_into_f32_decoder_and_fp16_towers():
    from pixray_amd import _lib
    assert _lib.split_precision("ref") == ("f32", "fp16") and _lib.split_precision("REF") == ("f32", "fp16")
    for p in ("fp16", "bf16", "f32", None):
        assert _lib.split_precision(p) == (p, p)
    assert _lib.precision_code(_lib.split_precision("ref")[0]) == _lib.PREC_F32
    assert _lib.precision_code(_lib.split_precision("ref")[1]) == _lib.PREC_F16


def test_default_vector_prompt_table_is_the_references():
    """pixray's default `--vector_prompts textoff` (pixray.py:887-915, 1732): the package ships the reference's table rows for
    the towers it implements, value for value (checked against /root/reference when present), resolved like the reference
    resolves a bare name; a tower without a row is skipped with the reference's warning"""
    import json
    from pixray_amd import api
    t = api.load_vector_table("textoff")
    assert set(t) == {"RN50", "RN101", "RN50x4", "ViT-B/32", "ViT-B/16"}
    assert len(t["ViT-B/32"]) == 1 and len(t["ViT-B/32"][0]) == 512 and len(t["RN50x4"][0]) == 640
    src = "/root/reference/vectors/textoff.json"
    if os.path.exists(src):
        with open(src) as f:
            ref = json.load(f)
        for k in t:
            assert t[k] == ref[k]
    assert api.WORKLOADS["cfg1"]["vector_prompts"] == ("textoff",) and api.WORKLOADS["cfg3"]["vector_prompts"] == ()
    from oracle import step_ref
    from pixray_amd import weights
    pl = step_ref.prompt_list("ViT-B/32", weights.CLIP_CONFIGS["ViT-B/32"], 0)
    assert len(pl) == 2 and pl[1][1] == 0.1 and pl[0][1] == 1.0 and tuple(pl[1][0].shape) == (1, 512)
    assert len(step_ref.prompt_list("tiny-B/32", weights.CLIP_CONFIGS["tiny-B/32"], 0)) == 1
