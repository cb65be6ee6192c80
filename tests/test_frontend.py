"""CPU tests of the settings front end / check-ins / run loop (pixray_amd/frontend.py) against
the reference's own behaviour: pixray.py:1718-2135 (settings), 1145-1201 (check-in, PNG metadata), 1538-1631 (do_run, the
animation ring), cogrun.py:25-52 (the serving generator).  The iteration runs on CPU stand-ins (the oracle's perceptor and
cutouts, the plain-torch pixel-grid drawer): what is tested is the host logic around `engine.Session`."""
import hashlib
import os
import sys

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)

from oracle import prompt_ref, step_ref  # noqa: E402
from pixray_amd import cutouts as pc  # noqa: E402
from pixray_amd import frontend as fe  # noqa: E402
from pixray_amd import weights  # noqa: E402


class _TextPerceptor(step_ref.OraclePerceptor):
    """the oracle tower + a deterministic stand-in text encoder (no CLIP text checkpoint offline): unit vectors seeded by
    the text"""
    texts = []

    def encode_text(self, text):
        items = [text] if isinstance(text, str) else list(text)
        out = []
        for t in items:
            type(self).texts.append(t)
            seed = int.from_bytes(hashlib.sha256(t.encode()).digest()[:4], "big")
            e = torch.randn(self.output_dim, generator=torch.Generator().manual_seed(seed))
            out.append(e / e.norm())
        return torch.stack(out)


def _factories(cutn):
    cfg = weights.CLIP_CONFIGS["tiny-B/32"]

    def perceptor_factory(name, index):
        return _TextPerceptor(cfg, weights.synthetic_clip_vit_params(cfg, 1 + index))

    def cutouts_factory(size, index):
        def sampler(iteration, fill):
            g = torch.Generator().manual_seed(1000 + iteration)
            prm = pc.sample_cutout_params(cutn, size, g, iteration=iteration, fill=fill)
            prm["noise"] = torch.randn(cutn, 3, size, size, generator=g)
            return prm
        return step_ref.OracleMakeCutouts(size, cutn, sampler)

    def prompt_factory(embed, weight=1.0, stop=float("-inf")):
        pm = prompt_ref.Prompt(embed, weight, stop)
        pm.denom = None
        return pm
    return dict(perceptor_factory=perceptor_factory, cutouts_factory=cutouts_factory, prompt_factory=prompt_factory, device="cpu")


def _settings(tmp_path, **kw):
    run = fe.Run()
    base = dict(drawer="fast_pixel", prompts="a red square|a blue circle:0.5", clip_models="tiny-B/32", size=[64, 48], pixel_size=[8, 6],
                num_cuts=2, iterations=6, save_every=2, display_every=3, outdir=str(tmp_path / "out"), seed="42", skip_args=True,
                init_noise="gradient", learning_rate=0.05)
    base.update(kw)
    run.settings = base
    return run, fe.apply_settings(run=run)


# ------------------------------------------------------------------------------------------------ settings
def test_core_option_table_matches_the_reference_parser():
    """every option of the reference's `setup_parser` (pixray.py:1722-1786), read from its source: same flags, destination,
    default and type name"""
    src_path = "/root/reference/pixray.py"
    if not os.path.exists(src_path):
        pytest.skip("needs the reference checkout")
    import ast
    tree = ast.parse(open(src_path).read())
    fn = next(n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "setup_parser")
    ref = {}
    for node in ast.walk(fn):
        if isinstance(node, ast.Call) and getattr(node.func, "attr", "") == "add_argument":
            flags = [a.value for a in node.args]
            kw = {k.arg: k.value for k in node.keywords}
            dest = kw["dest"].value
            default = ast.literal_eval(kw["default"])
            typ = kw["type"].id if "type" in kw else None
            nargs = ast.literal_eval(kw["nargs"]) if "nargs" in kw else None
            ref[dest] = (flags, default, typ, nargs)
    mine = {}
    for short, long_, dest, typ, default, kw in fe.CORE_OPTIONS:
        mine[dest] = (([short] if short else []) + [long_], default, {str: "str", float: "float", int: "int", fe.str2bool: "str2bool"}[typ],
                      kw.get("nargs"))
    assert set(mine) == set(ref)
    for dest in ref:
        assert mine[dest] == ref[dest], dest


def test_quality_size_and_unit_resolution(tmp_path):
    """pixray.py:1824-1929: quality fills clip models / iterations / cutouts / batches / scale; aspect x scale gives the size;
    '%'-or-iteration units resolve against the iteration count; the default vector prompt survives as a list"""
    run, s = _settings(tmp_path, quality="better", clip_models=None, size=None, num_cuts=None, iterations=None, aspect="widescreen",
                       save_every="10%", overlay_until="50%", learning_rate_drops=["75"])
    assert s.clip_models == ["RN50", "ViT-B/32", "ViT-B/16"] and s.iterations == 300 and s.num_cuts == 36 and s.batches == 1
    assert s.size == [576, 324] and abs(s.aspect_width - 576 / 324) < 1e-12
    assert s.save_every == 30 and s.overlay_until == 150 and s.overlay_every == 10 and s.display_every == 3
    assert s.learning_rate_drops == [int(0.75 * 299)]
    assert s.prompts == ["a red square", "a blue circle:0.5"] and s.vector_prompts == ["textoff"]
    run, s = _settings(tmp_path, quality="draft", ezsize="large", aspect="square", size=None, vector_prompts="none")
    assert s.size == [576, 576] and s.vector_prompts == []
    # the non-default settings are what the PNG metadata and settings.yaml carry
    assert run.given_args["quality"] == "draft" and "learning_rate_drops" not in run.given_args
    import yaml
    assert yaml.safe_load(open(os.path.join(s.outdir, "settings.yaml")))["ezsize"] == "large"
    with pytest.raises(ValueError, match="Requested setting not found"):
        _settings(tmp_path, no_such_setting=1)
    with pytest.raises(SystemExit):
        _settings(tmp_path, quality="ultra")
    with pytest.raises(SystemExit):
        _settings(tmp_path, size=None, aspect="cinema")
    # every tower a quality preset names has a configuration (vision + text) and a row in the default vector table (pixray.py:1824-1831)
    from pixray_amd import api, weights
    _, s2 = _settings(tmp_path, quality="supreme", clip_models=None, size=None, num_cuts=None, iterations=None)
    assert s2.clip_models == ["RN50x4", "RN101", "ViT-B/32", "ViT-B/16"] and s2.iterations == 400 and s2.batches == 4
    table = api.load_vector_table("textoff")
    for name in s2.clip_models:
        cfg = weights.CLIP_CONFIGS.get(name) or weights.CLIP_RESNET_CONFIGS[name]
        assert weights.CLIP_TEXT_CONFIGS[name].output_dim == cfg.output_dim == len(table[name][0]), name


def test_outdir_template_and_yaml_config_file(tmp_path):
    """util.py:273-312 (%DATE% / %SEQ%) and pixray.py:2024-2044 (--config_file: scalars override, lists append)"""
    tmpl = str(tmp_path / "o_%DATE%_%SEQ%")
    a = fe.emit_filename(tmpl)
    assert a.endswith("_01") and "%" not in a
    os.makedirs(a)
    assert fe.emit_filename(tmpl).endswith("_02")
    cfg = tmp_path / "settings.yaml"
    cfg.write_text("quality: draft\nnoise_prompt_seeds: [3, 4]\niterations: 7\n")
    run = fe.Run()
    run.settings = dict(drawer="fast_pixel", outdir=str(tmp_path / "y"), noise_prompt_seeds=[1])
    s = fe.apply_settings(argv=["--config_file", str(cfg), "--num_cuts", "5"], run=run)
    assert s.quality == "draft" and s.iterations == 7 and s.num_cuts == 5 and s.noise_prompt_seeds == [1, 3, 4]


def test_seed_resolution_and_brace_glob(tmp_path):
    assert fe.resolve_seed(7) == 7 and fe.resolve_seed("123") == 123
    want = int.from_bytes(hashlib.sha512(b"hello").digest(), "big") % 0x100000000
    assert fe.resolve_seed("hello") == want
    for n in ("a_1.png", "a_2.png", "b_1.png", "c.txt"):
        (tmp_path / n).write_text("x")
    assert [os.path.basename(f) for f in fe.real_glob(str(tmp_path / "{a,b}_*.png"))] == ["a_1.png", "a_2.png", "b_1.png"]
    assert fe.brace_expand("f{01..03}.png") == ["f01.png", "f02.png", "f03.png"] and fe.brace_expand("x{y}z") == ["x{y}z"]


# ------------------------------------------------------------------------------------------------ do_init / do_run / check-ins
def test_do_init_builds_every_prompt_kind_and_do_run_checks_in(tmp_path, capsys):
    from PIL import Image
    target = tmp_path / "target.png"
    Image.fromarray((np.random.RandomState(0).rand(40, 50, 3) * 255).astype(np.uint8)).save(target)
    _TextPerceptor.texts = []
    run, s = _settings(tmp_path, labels="cat", spot_prompts="sky", spot_prompts_off="ground", noise_prompt_seeds=[5], noise_prompt_weights=[0.3],
                       target_images=f"{target}:0.7", image_prompts=str(target), vector_prompts="none")
    sess = fe.do_init(s, run, **_factories(2))
    pms = sess.pmsTable["tiny-B/32"]
    # target image, 2 text prompts, the label's template mean, the noise prompt -- in the reference's order (pixray.py:797-958)
    assert len(pms) == 5 and [round(float(p.weight), 3) for p in pms] == [0.7, 1.0, 0.5, 1.0, 0.3]
    assert _TextPerceptor.texts[:2] == ["a red square", "a blue circle"]
    assert [t for t in _TextPerceptor.texts if "cat" in t] == [t.format("cat") for t in fe.IMAGENET_TEMPLATES]
    assert abs(float(pms[3].embed.norm()) - 1) < 1e-6
    assert len(sess.spotPmsTable["tiny-B/32"]) == 1 and len(sess.spotOffPmsTable["tiny-B/32"]) == 1
    inside, outside = sess.cutoutsTable[224].spot_masks
    assert inside.shape == (3, 224, 224) and bool((inside ^ outside).all())
    assert len(sess.pmsImageTable["tiny-B/32"]) == 1 and sess.pmsImageTable["tiny-B/32"][0].shape[1] == 3
    assert run.seed_used == 42 and sess.max_loss_drops == len(s.learning_rate_drops) and sess.iter_drop_delay == 12
    # run to completion: check-ins at 0, 2, 4 and the final one
    z0 = sess.drawer.get_z_copy()
    assert fe.do_run(s, run=run) is True
    out = capsys.readouterr().out
    lines = [l for l in out.splitlines() if l.startswith("iter:")]
    assert [l.split(",")[0] for l in lines] == ["iter: 0", "iter: 2", "iter: 4", "iter: 6"] and lines[-1].startswith("iter: 6, finished")
    assert "losses:" in lines[0] and lines[0].count(",") >= 2 + 6          # total + one value per loss term
    assert (sess.drawer.get_z() - z0).abs().max() > 1e-4 and sess.cur_iteration == 6
    png = os.path.join(s.outdir, "output.png")
    img = Image.open(png)
    assert img.size == (64, 48)
    assert img.text["Software"].startswith("pixray (") and img.text["pixray_seed_used"] == "42"
    assert img.text["pixray_num_cuts"] == "2" and img.text["pixray_drawer"] == "fast_pixel" and "pixray_quality" not in img.text
    steps = sorted(os.listdir(os.path.join(s.outdir, "steps")))
    assert steps[:4] == ["frame_0000.png", "frame_0002.png", "frame_0004.png", "frame_0006.png"]
    assert os.path.exists(os.path.join(s.outdir, "output.log"))


def test_checkin_image_is_the_state_the_losses_were_computed_on(tmp_path):
    """the reference saves from inside the iteration, before the optimiser step (pixray.py:1477-1479): frame_0000 is the
    start image, not the image after one step"""
    from PIL import Image
    run, s = _settings(tmp_path, iterations=2, save_every=1, vector_prompts="none")
    sess = fe.do_init(s, run, **_factories(2))
    start = np.asarray(run.snapshot())
    fe.do_run(s, run=run)
    f0 = np.asarray(Image.open(os.path.join(s.outdir, "steps", "frame_0000.png")))
    f1 = np.asarray(Image.open(os.path.join(s.outdir, "steps", "frame_0001.png")))
    assert np.array_equal(f0, start) and not np.array_equal(f1, start)


def test_return_display_yields_every_display_every_iterations_like_the_cog_loop(tmp_path, monkeypatch):
    """cogrun.py:47-52: `do_run(settings, return_display=True)` returns False every `display_every` iterations; `predict`
    yields a temporary copy of the current output each time"""
    run, s = _settings(tmp_path, iterations=7, display_every=3, save_every=1, vector_prompts="none")
    fe.do_init(s, run, **_factories(2))
    seen = []
    while True:
        done = fe.do_run(s, return_display=True, run=run)
        seen.append(run.session.cur_iteration)
        if done:
            break
    assert seen == [3, 6, 7]
    # predict(): the same loop behind the module-level API, HIP parts replaced by the stand-ins
    fac = _factories(2)
    orig = fe.do_init
    monkeypatch.setattr(fe, "do_init", lambda settings, run=None: orig(settings, run, **fac))
    paths = list(fe.predict(dict(drawer="fast_pixel", clip_models="tiny-B/32", size=[64, 48], pixel_size=[8, 6], num_cuts=2,
                                 vector_prompts="none", init_noise="snow"), prompts="x", iterations=4, display_every=2, save_every=1,
                            outdir=str(tmp_path / "p")))
    # iterations 2 and 4 return for display, the closing call (train at `iterations`: final check-in) completes: three yields,
    # as the reference's loop gives
    assert len(paths) == 3 and all(os.path.exists(p) and p.endswith(".png") for p in paths)


def test_animation_ring_blends_each_frame_with_its_predecessor(tmp_path):
    """pixray.py:1544-1609: one z per init image, `save_every` iterations per frame and round, outputs named after the inputs
    in the animation directory, frames re-encoded from the blend with the previous frame between rounds"""
    from PIL import Image
    for i, c in enumerate([(255, 0, 0), (0, 255, 0), (0, 0, 255)]):
        Image.new("RGB", (64, 48), c).save(tmp_path / f"in_{i}.png")
    anim = tmp_path / "anim"
    run, s = _settings(tmp_path, init_image=str(tmp_path / "in_{0..2}.png"), init_image_alpha=255, animation_dir=str(anim), iterations=4,
                       save_every=2, vector_prompts="none", prompts="x")
    sess = fe.do_init(s, run, **_factories(2))
    assert len(run.init_images) == 3
    fe.do_run(s, run=run)
    assert sorted(os.listdir(anim)) == ["in_0.png", "in_1.png", "in_2.png"]
    assert len(run.anim_cur_zs) == 3 and run.cur_anim_index == 2
    # each frame started from its own init image and was blended (alpha 128) with its predecessor's image between the two
    # rounds: its own colour and the predecessor's are both strong, the third is not
    for i in range(3):
        arr = np.asarray(Image.open(anim / f"in_{i}.png").convert("RGB")).reshape(-1, 3).mean(0)
        assert arr[i] > 100 and arr[(i - 1) % 3] > 60 and arr[(i + 1) % 3] < 40, (i, arr)


def test_overlay_on_a_checkin_iteration_is_applied_once(tmp_path):
    from PIL import Image
    ov = tmp_path / "ov.png"
    Image.new("RGBA", (64, 48), (255, 255, 255, 255)).save(ov)
    run, s = _settings(tmp_path, overlay_image=str(ov), overlay_every="2 iterations", iterations=3, save_every=2, vector_prompts="none",
                       init_noise="none", prompts="x")
    sess = fe.do_init(s, run, **_factories(2))
    calls = []
    orig = sess.re_average_z
    sess.re_average_z = lambda: (calls.append(sess.cur_iteration), orig())[1]
    fe.do_run(s, run=run)
    assert calls == [0, 2]
    f0 = np.asarray(Image.open(os.path.join(s.outdir, "steps", "frame_0000.png")))
    assert f0.min() == 255            # iteration 0's check-in shows the overlaid (all white) state


def test_command_line_entry_point(tmp_path):
    """`python -m pixray_amd ...` = the reference's `python pixray.py ...` (pixray.py:2126-2135): an unknown drawer is a KeyError
    from the class table, as in the reference (pixray.py:612); without a GPU the HIP parts refuse loudly after the settings
    were resolved and logged"""
    import subprocess
    root = os.path.dirname(HERE)
    env = dict(os.environ, PYTHONPATH=root)
    r = subprocess.run([sys.executable, "-m", "pixray_amd", "--drawer", "nosuchdrawer"], capture_output=True, text=True, cwd=str(tmp_path), env=env,
                       timeout=120)
    assert r.returncode != 0 and "KeyError" in r.stderr
    r = subprocess.run([sys.executable, "-m", "pixray_amd", "--drawer", "fast_pixel", "--quality", "draft", "--outdir", str(tmp_path / "o"),
                        "--prompts", "x", "--iterations", "2"], capture_output=True, text=True, cwd=str(tmp_path), env=env, timeout=180)
    assert "Running with 24x1 = 24 cuts" in r.stdout
    if not torch.cuda.is_available():
        assert r.returncode != 0 and "no ROCm device visible" in r.stderr
    assert os.path.exists(tmp_path / "o" / "settings.yaml")
