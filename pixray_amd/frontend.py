"""Settings front end, session start-up, image check-ins and the outer run loop: SURVEY.md section 8 row f4
(/root/reference/pixray.py:1718-2135 settings and entry points, 557-1020 `do_init`, 1145-1201 `checkin` + PNG metadata,
1538-1631 `do_run` with the animation ring, cogrun.py:25-52 the serving generator).

Everything here is host Python around `engine.Session`: the iteration itself (synth -> cutouts -> CLIP -> loss -> backward
-> Adam) runs on the HIP kernels exactly as `api.build_*_session` assembles it.  The reference keeps its state in module
globals (pixray.py:1022-1063: one run per process); here a `Run` object holds it and the module-level functions
(`reset_settings`, `add_settings`, `get_settings`, `apply_settings`, `do_init`, `do_run`, `run`, `main`, `add_custom_loss`)
keep the reference's names and calling sequence on top of one module-level `Run`, so a notebook / cog script written for
pixray drives this package unchanged:

    import pixray_amd.frontend as pixray
    pixray.reset_settings(); pixray.add_settings(prompts="...", quality="draft", outdir="out"); settings = pixray.apply_settings()
    pixray.do_init(settings); pixray.do_run(settings)

Not carried over (outside SURVEY.md section 8): the notebook display calls, `--palette` / `--transparent_weight` parsing
helpers of util.py that belong to PaletteLoss, the SLIP perceptors, ffmpeg video / gif assembly (the frame files are written;
`make_video` / animation gif need ffmpeg and are skipped with a message when it is absent), the per-frame target-image prompt
table of the animation mode (`pmsTargetTable`, pixray.py:772-795: target images score every frame here), the vdiff drawer (its
source is not in the reference checkout).
"""
from __future__ import annotations

import argparse
import datetime
import glob
import hashlib
import json
import logging
import os
import random
import re
import shutil
import subprocess
import sys
import types
from typing import Callable, Dict, Iterator, List, Optional

import numpy as np
import torch

from . import plugins
from .settings import get_file_path, get_learning_rate_drops, parse_unit, split_pipes
from .prompt import parse_prompt

VERSION = "pixray_amd-0.4"

IMAGENET_TEMPLATES = ("itap of a {}.", "a bad photo of the {}.", "a origami {}.", "a photo of the large {}.",
                      "a {} in a video game.", "art of the {}.", "a photo of the small {}.")      # pixray.py:1514-1522

# ---------------------------------------------------------------------------------------------------------- small helpers


def str2bool(v):
    """util.py:39-47"""
    if isinstance(v, bool):
        return v
    low = str(v).lower()
    if low in ("yes", "true", "t", "y", "1"):
        return True
    if low in ("no", "false", "f", "n", "0"):
        return False
    raise argparse.ArgumentTypeError("Boolean value expected.")


def brace_expand(pattern: str) -> List[str]:
    """bash-style {a,b} / {1..3} expansion (the reference uses the `braceexpand` package, util.py:25-30; not installed here)"""
    m = re.search(r"\{([^{}]*)\}", pattern)
    if not m:
        return [pattern]
    head, body, tail = pattern[:m.start()], m.group(1), pattern[m.end():]
    rng = re.fullmatch(r"(-?\d+)\.\.(-?\d+)", body)
    if rng:
        a, b = int(rng.group(1)), int(rng.group(2))
        width = max(len(rng.group(1)), len(rng.group(2))) if (rng.group(1).startswith("0") or rng.group(2).startswith("0")) else 0
        alts = [str(i).zfill(width) for i in (range(a, b + 1) if a <= b else range(a, b - 1, -1))]
    elif "," in body:
        alts = body.split(",")
    else:
        alts = None
    if alts is None:            # not an expansion: keep the braces literally, go on with the rest
        return [head + "{" + body + "}" + t for t in brace_expand(tail)]
    out: List[str] = []
    for a in alts:
        out += brace_expand(head + a + tail)
    return out


def real_glob(rglob: str) -> List[str]:
    """util.py:25-30: brace expansion, then glob, sorted"""
    files: List[str] = []
    for g in brace_expand(rglob):
        files += glob.glob(g)
    return sorted(files)


def emit_filename(filename: str, template: Optional[dict] = None) -> str:
    """util.py:273-312: %DATE% -> yyyymmdd, %KEY% from `template`, %SEQ% -> the first two-digit number whose path is free"""
    filename = filename.replace("%DATE%", datetime.datetime.now().strftime("%Y%m%d"))
    for k, v in (template or {}).items():
        filename = filename.replace(f"%{k}%", f"{v}")
    if "%SEQ%" in filename:
        seq = 1
        while os.path.exists(filename.replace("%SEQ%", f"{seq:02d}")):
            seq += 1
        filename = filename.replace("%SEQ%", f"{seq:02d}")
    return filename


def resolve_seed(seed) -> int:
    """pixray.py:587-601: None -> a fresh torch seed; an int or digit string -> itself; any other string -> 32 bits of its
    SHA-512"""
    if seed is None:
        return torch.seed()
    if isinstance(seed, int):
        return seed
    if isinstance(seed, str) and seed.isdigit():
        return int(seed)
    return int.from_bytes(hashlib.sha512(str(seed).encode()).digest(), "big") % 0x100000000


def to_pil(t: torch.Tensor):
    """[3|4, H, W] in [0, 1] -> PIL image (torchvision's to_pil_image on a float tensor: x * 255, truncated to uint8)"""
    from PIL import Image
    arr = t.detach().float().cpu().clamp(0, 1).mul(255).byte().permute(1, 2, 0).numpy()
    return Image.fromarray(arr, mode="RGBA" if arr.shape[2] == 4 else "RGB")


def to_tensor(img) -> torch.Tensor:
    """PIL image -> [C, H, W] float in [0, 1] (torchvision's to_tensor)"""
    arr = np.asarray(img, dtype=np.uint8)
    if arr.ndim == 2:
        arr = arr[:, :, None]
    return torch.from_numpy(arr.copy()).permute(2, 0, 1).float().div(255)


def resize_image(image, out_size):
    """pixray.py:514-518: keep the aspect ratio, cap the area at the canvas area"""
    from PIL import Image
    ratio = image.size[0] / image.size[1]
    area = min(image.size[0] * image.size[1], out_size[0] * out_size[1])
    size = round((area * ratio) ** 0.5), round((area / ratio) ** 0.5)
    return image.resize(size, Image.LANCZOS)


def _fractal_noise(side: int, octaves: int, rng: np.random.RandomState) -> np.ndarray:
    """smooth multi-octave noise in [0, 1] (stands in for perlin_numpy.generate_fractal_noise_2d, which is not installed: the
    start image is random either way)"""
    out = np.zeros((side, side))
    amp, res = 1.0, 32
    for _ in range(octaves):
        lat = rng.rand(res + 1, res + 1)
        xs = np.linspace(0, res, side, endpoint=False)
        i0 = xs.astype(int)
        f = xs - i0
        f = f * f * (3 - 2 * f)
        rows = lat[i0][:, i0] * (1 - f)[None, :] + lat[i0][:, i0 + 1] * f[None, :]
        rows1 = lat[i0 + 1][:, i0] * (1 - f)[None, :] + lat[i0 + 1][:, i0 + 1] * f[None, :]
        out += amp * (rows * (1 - f)[:, None] + rows1 * f[:, None])
        amp, res = amp / 2, res * 2
    return (out - out.min()) / (out.max() - out.min())


def starting_image(kind: Optional[str], w: int, h: int):
    """pixray.py:192-246, 683-692: the noise image a run starts from ('pixels' fractal noise with the reference's contrast
    curve, 'gradient', 'snow' uniform noise, anything else a white canvas); numpy's global stream, seeded by do_init"""
    from PIL import Image
    if kind == "pixels":
        side, octaves = (2048, 6) if max(w, h) > 1024 else (1024, 5) if max(w, h) > 512 else (512, 4) if max(w, h) > 256 else (256, 3)
        rng = np.random.RandomState(np.random.randint(0, 2 ** 31 - 1))
        planes = []
        for _ in range(3):
            n = 0.9998 * _fractal_noise(side, octaves, rng) + 0.0001
            planes.append(1 / (1 + np.power(n / (1 - n), -2.0)))                # contrast_noise (pixray.py:200-205)
        return Image.fromarray((255.999 * np.dstack(planes)[:h, :w, :]).astype("uint8"))
    if kind == "gradient":
        stops = (np.random.randint(1, 255), np.random.randint(2, 255), np.random.randint(3, 128))
        starts = (0, 0, np.random.randint(0, 255))
        arr = np.zeros((h, w, 3))
        arr[:, :, 0] = np.tile(np.linspace(starts[0], stops[0], w), (h, 1))
        arr[:, :, 1] = np.tile(np.linspace(starts[1], stops[1], h), (w, 1)).T
        arr[:, :, 2] = np.tile(np.linspace(starts[2], stops[2], h), (w, 1)).T
        return Image.fromarray(np.uint8(arr))
    if kind == "snow":
        return Image.fromarray(np.random.randint(0, 255, (h, w, 3), dtype=np.uint8))
    return Image.new(mode="RGB", size=(w, h), color=(255, 255, 255))


# ---------------------------------------------------------------------------------------------------------- settings
# (short flag, long flag, dest, type, default, extra add_argument keywords): the reference's core option table
# (pixray.py:1722-1786), same flags, destinations and defaults
_S, _F, _I, _B = str, float, int, str2bool
CORE_OPTIONS = [
    ("-p", "--prompts", "prompts", _S, [], {}), ("-sp", "--spot", "spot_prompts", _S, [], {}),
    ("-spo", "--spot_off", "spot_prompts_off", _S, [], {}), ("-spf", "--spot_file", "spot_file", _S, None, {}),
    ("-l", "--labels", "labels", _S, [], {}), ("-vp", "--vector_prompts", "vector_prompts", _S, "textoff", {}),
    ("-ip", "--image_prompts", "image_prompts", _S, [], {}), ("-ipw", "--image_prompt_weight", "image_prompt_weight", _F, None, {}),
    ("-ips", "--image_prompt_shuffle", "image_prompt_shuffle", _B, False, {}), ("-il", "--image_labels", "image_labels", _S, None, {}),
    ("-ilw", "--image_label_weight", "image_label_weight", _F, 1.0, {}), ("-i", "--iterations", "iterations", _I, None, {}),
    ("-se", "--save_every", "save_every", _S, 10, {}), ("-si", "--save_intermediates", "save_intermediates", _B, True, {}),
    ("-de", "--display_every", "display_every", _S, 20, {}), ("-dc", "--display_clear", "display_clear", _B, False, {}),
    ("-ove", "--overlay_every", "overlay_every", _S, "10 iterations", {}),
    ("-ovo", "--overlay_offset", "overlay_offset", _S, "0 iterations", {}), ("-ovu", "--overlay_until", "overlay_until", _S, None, {}),
    ("-ovi", "--overlay_image", "overlay_image", _S, None, {}), (None, "--quality", "quality", _S, "normal", {}),
    ("-asp", "--aspect", "aspect", _S, "widescreen", {}), ("-ezs", "--ezsize", "ezsize", _S, None, {}),
    ("-sca", "--scale", "scale", _F, None, {}), ("-ova", "--overlay_alpha", "overlay_alpha", _I, None, {}),
    ("-s", "--size", "size", _I, None, {"nargs": 2}), ("-ii", "--init_image", "init_image", _S, None, {}),
    ("-iia", "--init_image_alpha", "init_image_alpha", _I, 200, {}), ("-in", "--init_noise", "init_noise", _S, "pixels", {}),
    ("-ti", "--target_images", "target_images", _S, None, {}), ("-anim", "--animation_dir", "animation_dir", _S, None, {}),
    ("-ana", "--animation_alpha", "animation_alpha", _I, 128, {}), ("-iw", "--init_weight", "init_weight", _F, None, {}),
    ("-iwd", "--init_weight_dist", "init_weight_dist", _F, 0.0, {}), ("-iwc", "--init_weight_cos", "init_weight_cos", _F, 0.0, {}),
    ("-iwp", "--init_weight_pix", "init_weight_pix", _F, 0.0, {}), (None, "--perceptors", "perceptors", _S, "clip", {}),
    (None, "--clip_models", "clip_models", _S, None, {}),
    ("-nps", "--noise_prompt_seeds", "noise_prompt_seeds", _I, [], {"nargs": "*"}),
    ("-npw", "--noise_prompt_weights", "noise_prompt_weights", _F, [], {"nargs": "*"}),
    ("-lr", "--learning_rate", "learning_rate", _F, 0.2, {}),
    ("-lrd", "--learning_rate_drops", "learning_rate_drops", _S, [75], {"nargs": "*"}),
    ("-as", "--auto_stop", "auto_stop", _B, False, {}), ("-cuts", "--num_cuts", "num_cuts", _I, None, {}),
    ("-bats", "--batches", "batches", _I, None, {}), ("-cutp", "--cut_power", "cut_pow", _F, 1.0, {}),
    (None, "--seed", "seed", _S, None, {}), ("-opt", "--optimiser", "optimiser", _S, "Adam", {}),
    ("-vid", "--video", "make_video", _B, False, {}), ("-d", "--deterministic", "cudnn_determinism", _B, False, {}),
    ("-cud", "--cuda_device", "cuda_device", _S, "cuda:0", {}), (None, "--palette", "palette", _S, None, {}),
    (None, "--transparent", "transparent", _B, False, {}), (None, "--transparent_weight", "transparent_weight", _F, 0.0, {}),
    (None, "--alpha_use_g", "alpha_use_g", _B, False, {}), (None, "--alpha_gamma", "alpha_gamma", _F, 4.0, {}),
    (None, "--output", "output", _S, "output.png", {}), (None, "--outdir", "outdir", _S, "outputs/%DATE%_%SEQ%", {}),
]
# options of this package only: operand precision of the HIP runners (include/prx.h PRX_PREC_*; "ref" = f32 decoder + fp16 towers)
EXTRA_OPTIONS = [(None, "--precision", "precision", _S, None, {})]

QUALITY_CLIP_MODELS = {            # pixray.py:1824-1832 (the 'clip' family; the SLIP families need the SLIP checkpoints' code)
    "draft": "ViT-B/16", "normal": "ViT-B/32,ViT-B/16", "better": "RN50,ViT-B/32,ViT-B/16",
    "best": "RN50x4,ViT-B/32,ViT-B/16", "supreme": "RN50x4,RN101,ViT-B/32,ViT-B/16"}
QUALITY_TABLES = {                 # pixray.py:1849-1879: iterations, size scale, cutouts per batch, batches
    "draft": (200, 1, 24, 1), "normal": (250, 2, 30, 1), "better": (300, 3, 36, 1), "best": (350, 4, 12, 2), "supreme": (400, 5, 8, 4)}
EZSIZE_SCALE = {"small": 1, "medium": 2, "large": 4}                                       # pixray.py:1896-1900
ASPECT_BASE = {"square": (144, 144), "portrait": (128, 160), "widescreen": (192, 108)}     # pixray.py:1901-1905


def setup_parser(parser: argparse.ArgumentParser) -> argparse.ArgumentParser:
    for short, long_, dest, typ, default, kw in CORE_OPTIONS + EXTRA_OPTIONS:
        flags = ([short] if short else []) + [long_]
        parser.add_argument(*flags, type=typ, default=default, dest=dest, **kw)
    return parser


def _parse_known_with_yaml(parser, settings: dict, argv: List[str]):
    """pixray.py:2024-2044: `--config_file settings.yaml`; scalar values override, list values append to what the settings
    already hold.  Returns (first-pass namespace, the merged settings dictionary, argv without the --config_file option): the
    reference applies the file's values to the first pass only when it runs from a bare command line (its second pass starts
    from a fresh namespace, and the open file object ends up among the recorded settings); here the merged values reach
    the second pass too and the file object is dropped."""
    import yaml
    pre = argparse.ArgumentParser(add_help=False)
    pre.add_argument("--config_file", dest="config_file", type=argparse.FileType(mode="r"))
    found, rest = pre.parse_known_args(argv)
    merged = dict(settings)
    if found.config_file:
        for key, value in (yaml.load(found.config_file, Loader=yaml.SafeLoader) or {}).items():
            if isinstance(value, list):
                merged[key] = list(merged.get(key) or []) + list(value)
            else:
                merged[key] = value
        found.config_file.close()
    core, _unknown = parser.parse_known_args(args=rest, namespace=types.SimpleNamespace(**merged))
    return core, merged, rest


def process_args(parser: argparse.ArgumentParser, namespace=None, argv=None, run: Optional["Run"] = None):
    """pixray.py:1788-1997: parse, record the non-default settings, resolve the output directory, fill quality-dependent
    defaults, work out the canvas size, split the pipe lists, turn '30%' / '20 iterations' into iteration counts"""
    if namespace is None:
        args = parser.parse_args(args=argv)
    elif hasattr(namespace, "skip_args"):
        args = parser.parse_args(args=[], namespace=namespace)
    else:
        args = parser.parse_args(args=argv, namespace=namespace)
    given = {a.dest: getattr(args, a.dest) for a in parser._option_string_actions.values()
             if hasattr(args, a.dest) and a.default != getattr(args, a.dest)}
    args.outdir = emit_filename(args.outdir)
    if args.outdir != "" and not os.path.exists(args.outdir):
        os.makedirs(args.outdir)
    _initialize_logging(args, given)
    if args.perceptors != "clip":
        raise ValueError(f"perceptors={args.perceptors!r}: only the OpenAI CLIP family is provided (SURVEY.md section 8 f2)")
    if args.quality not in QUALITY_TABLES:
        print("Qualitfy setting not understood, aborting -> ", args.quality)
        sys.exit(1)
    q_iter, q_scale, q_cuts, q_batches = QUALITY_TABLES[args.quality]
    if args.clip_models is None:
        args.clip_models = QUALITY_CLIP_MODELS[args.quality]
    if args.iterations is None:
        args.iterations = q_iter
    if args.num_cuts is None:
        args.num_cuts = q_cuts
    if args.batches is None:
        args.batches = q_batches
    if args.ezsize is None and args.scale is None:
        args.scale = q_scale
    if args.size is None:
        scale = args.scale
        if scale is None:
            if args.ezsize not in EZSIZE_SCALE:
                print("EZ Size not understood, aborting -> ", args.ezsize)
                sys.exit(1)
            scale = EZSIZE_SCALE[args.ezsize]
        if args.aspect in ASPECT_BASE:
            bw, bh = ASPECT_BASE[args.aspect]
            args.size = [int(scale * bw), int(scale * bh)]
        elif args.aspect == "retain" and args.init_image is not None:
            from PIL import Image
            w, h = Image.open(real_glob(args.init_image)[0]).size
            args.size = [int(144 * scale), int(144 * (h / w) * scale)]
        else:
            print("aspect not understood, aborting -> ", args.aspect)
            sys.exit(1)
    args.aspect_width = args.size[0] / args.size[1]                 # `global_aspect_width` (pixray.py:1931)
    if args.init_noise is not None and str(args.init_noise).lower() == "none":
        args.init_noise = None
    for name in ("prompts", "target_images", "spot_prompts", "spot_prompts_off", "labels"):
        setattr(args, name, split_pipes(getattr(args, name)))
    for name in ("overlay_offset", "overlay_until", "overlay_every", "display_every", "save_every"):
        setattr(args, name, parse_unit(getattr(args, name), args.iterations, name, "i"))
    if args.image_prompts:
        args.image_prompts = real_glob(args.image_prompts)
    vp = args.vector_prompts
    if vp and not (str(vp).lower() == "none" or vp == "0"):
        args.vector_prompts = [p.strip() for p in vp.split("|")] if isinstance(vp, str) else list(vp)
    else:
        args.vector_prompts = []
    if args.overlay_image is not None and args.overlay_every <= 0:
        args.overlay_image = None
    args.clip_models = [m.strip() for m in args.clip_models.split(",")] if isinstance(args.clip_models, str) else list(args.clip_models)
    if args.make_video:
        os.makedirs(os.path.join(args.outdir, "video"), exist_ok=True)
    args.learning_rate_drops = get_learning_rate_drops(args.learning_rate_drops, args.iterations)
    if run is not None:
        run.reset(given)
    return args


def _initialize_logging(args, given: dict) -> None:
    """pixray.py:2046-2053: <outdir>/<output>.log and <outdir>/settings.yaml with the non-default settings"""
    if args.outdir is not None and args.outdir.strip() != "":
        import yaml
        logging.basicConfig(level=logging.DEBUG, filename=get_file_path(args.outdir, args.output, ".log"), filemode="w+", force=True)
        with open(os.path.join(args.outdir, "settings.yaml"), "w+") as f:
            yaml.dump(given, f, allow_unicode=True, default_flow_style=False)


# ---------------------------------------------------------------------------------------------------------- the run
class Run:
    """What the reference keeps in module globals between `apply_settings`, `do_init` and `do_run`: the pending settings,
    the non-default settings (for the PNG metadata), the session, the animation ring."""

    def __init__(self):
        self.settings: Dict[str, object] = {}
        self.given_args: Dict[str, object] = {}
        self.session = None
        self.seed_used = None
        self.device = None
        self.init_images: List = []
        self.overlay_images: List = []
        self._png_info = None
        self.reset({})

    def reset(self, given: dict) -> None:
        self.given_args = dict(given)
        self.cur_anim_index: Optional[int] = None
        self.best_loss, self.best_iter = 1e20, 0
        self.anim_output_files: List[str] = []
        self.anim_cur_zs: List = []
        self._png_info = None

    # -- PNG metadata (pixray.py:1145-1156) -----------------------------------------------------------------------------
    def png_info(self):
        from PIL import PngImagePlugin
        if self._png_info is None:
            info = PngImagePlugin.PngInfo()
            info.add_text("Software", f"pixray ({VERSION})")
            for k, v in self.given_args.items():
                info.add_text(f"pixray_{k}", str(v))
            info.add_text("pixray_seed_used", str(self.seed_used))
            self._png_info = info
        return self._png_info

    # -- check-in (pixray.py:1158-1201) ---------------------------------------------------------------------------------
    @torch.no_grad()
    def snapshot(self):
        """the current image as the loop sees it: drawer.synth + the filter chain (+ alpha), as PIL"""
        sess = self.session
        timg, alpha = sess.do_synth_and_filter([])
        timg = timg[0]
        if alpha is not None:
            timg = torch.cat([timg, alpha[0][None]], dim=0)
        return to_pil(timg)

    def checkin(self, args, it: int, losses, img=None) -> str:
        sess = self.session
        if losses is not None:
            vals = [float(l) for l in losses]
            line = f"iter: {it}, loss: {sum(vals):1.3g}, losses: {', '.join(f'{v:2.3g}' for v in vals)}"
        else:
            line = f"iter: {it}, finished"
        if args.animation_dir is not None:
            line = f"anim: {self.cur_anim_index}/{len(self.anim_output_files)} {line}"
        else:
            # the reference tracks the best loss every iteration (checkdrop, pixray.py:1091-1109: one device->host sync per
            # iteration); without --auto_stop this loop only reads losses at check-ins, so "best" is over the check-ins
            if sess.auto_stop and sess.best_loss is not None:
                self.best_loss, self.best_iter = sess.best_loss, sess.best_iter
            elif losses is not None and sum(vals) < self.best_loss:
                self.best_loss, self.best_iter = sum(vals), it
            line = f"{line} (-{it - self.best_iter}=>{self.best_loss:2.4g})"
        img = img if img is not None else self.snapshot()
        outfile = get_file_path(args.outdir, args.output, ".png") if self.cur_anim_index is None \
            else self.anim_output_files[self.cur_anim_index]
        img.save(outfile, pnginfo=self.png_info())
        if args.save_intermediates:
            steps = os.path.join(args.outdir, "steps")
            os.makedirs(steps, exist_ok=True)
            img.save(get_file_path(steps, f"frame_{it:04d}", ".png"))
        if args.make_video:
            img.save(os.path.join(args.outdir, "video", f"frame_{it:04d}.png"))
        if self.cur_anim_index is not None and self.cur_anim_index == len(self.anim_output_files) - 1:
            make_gif(args)
        print(line)
        logging.info(line)
        return outfile

    # -- one iteration with its check-in (pixray.py:1436-1512) -----------------------------------------------------------
    def train(self, args, cur_it: int) -> bool:
        sess = self.session
        if cur_it == 0 and self.init_images and self.cur_anim_index is not None:             # pixray.py:1452-1455
            frame = self.init_images[self.cur_anim_index % len(self.init_images)]
            sess.drawer.reapply_from_tensor(to_tensor(frame).to(self.device).unsqueeze(0) * 2 - 1)
        if self.overlay_images and self.cur_anim_index is not None:                          # pixray.py:1457-1460
            sess.overlay_image_rgba = self.overlay_images[self.cur_anim_index % len(self.overlay_images)]
        save = cur_it < args.iterations and cur_it % args.save_every == 0
        # the reference saves from inside the iteration, after the forward pass and before the optimiser step: the image of
        # the state the losses were computed on.  The overlay re-encode (if due) happens first there too, so apply it here.
        if save and sess.apply_overlay(cur_it):
            sess.re_average_z()
            held, sess.overlay_image_rgba = sess.overlay_image_rgba, None
            img = self.snapshot()
            keep_going = sess.train(cur_it)
            sess.overlay_image_rgba = held
        else:
            img = self.snapshot() if save else None
            keep_going = sess.train(cur_it)
        if save:
            self.checkin(args, cur_it, sess.last_losses, img)
        if cur_it == args.iterations:
            self.checkin(args, cur_it, None)
        return keep_going


def make_gif(args) -> Optional[str]:
    """pixray.py:1071-1083 (needs ffmpeg)"""
    out = os.path.join(args.animation_dir, "anim.gif")
    if shutil.which("ffmpeg") is None:
        print("ffmpeg not found: animation frames are in", args.animation_dir)
        return None
    if os.path.exists(out):
        os.remove(out)
    try:
        subprocess.check_output(["ffmpeg", "-framerate", "10", "-pattern_type", "glob", "-i", f"{args.animation_dir}/*.png",
                                 "-loop", "0", out])
    except subprocess.CalledProcessError as e:
        print("Ignoring non-zero exit: ", e.output)
    return out


def frames_to_video(frames: List[str], output_file: str, comment: Optional[str] = None, length: int = 14) -> Optional[str]:
    """pixray.py:1634-1711 `step_to_video` / `do_video`: the frame files piped to ffmpeg at clip(len / 14 s, 10, 60) fps"""
    if shutil.which("ffmpeg") is None or not frames:
        print("ffmpeg not found (or no frames): no video written")
        return None
    fps = int(np.clip(len(frames) / length, 10, 60))
    cmd = ["ffmpeg", "-y", "-f", "image2pipe", "-vcodec", "png", "-r", str(fps), "-i", "-", "-vcodec", "libx264", "-r", str(fps),
           "-pix_fmt", "yuv420p", "-crf", "17", "-preset", "veryslow"]
    if comment is not None:
        cmd += ["-metadata", f"comment={comment}"]
    p = subprocess.Popen(cmd + [output_file], stdin=subprocess.PIPE)
    for path in frames + [frames[-1]] * fps:
        with open(path, "rb") as f:
            p.stdin.write(f.read())
    p.stdin.close()
    p.wait()
    return output_file


# ---------------------------------------------------------------------------------------------------------- do_init
def _hip_parts(args, device):
    """the product's parts: HIP perceptors, HIP cutouts, HIP prompt loss (no CPU fallback: fails loudly without a GPU)"""
    from . import _lib
    from .cutouts import MakeCutouts
    from .perceptor import get_clip_perceptor
    from .prompt import Prompt
    _lib.load()
    _lib.require_device()
    prec = _lib.split_precision(getattr(args, "precision", None))[1]

    def perceptor_factory(name, index):
        return get_clip_perceptor(name, device, max_batch=args.num_cuts, seed=getattr(args, "weight_seed", 0) + 1 + 10 * index,
                                  precision=prec)

    def cutouts_factory(cut_size, index):
        return MakeCutouts(cut_size, args.num_cuts, cut_pow=args.cut_pow, aspect_width=args.aspect_width,
                           generator=torch.Generator().manual_seed(1000 + int(args.seed_used % (2 ** 31)) + index))
    return perceptor_factory, cutouts_factory, Prompt


def do_init(args, run: Optional[Run] = None, *, perceptor_factory: Optional[Callable] = None,
            cutouts_factory: Optional[Callable] = None, prompt_factory: Optional[Callable] = None, device=None):
    """pixray.py:579-1020: seed, drawer, perceptors and cutout tables, filters, start image, overlay, every kind of prompt,
    custom losses -> `engine.Session` (returned; also kept in `run`).  The three factories default to the HIP parts; tests
    hand in CPU stand-ins."""
    from PIL import Image
    from .engine import Session
    run = run if run is not None else _RUN
    seed = resolve_seed(args.seed)
    print("Using seed:", seed)
    run.seed_used = args.seed_used = seed
    torch.manual_seed(seed)
    np.random.seed(int(seed) % (2 ** 30))
    random.seed(int(seed) % (2 ** 30))
    if device is None:
        device = torch.device(args.cuda_device if torch.cuda.is_available() else "cpu")
    run.device = device = torch.device(device)
    if not hasattr(args, "precision"):
        args.precision = None
    # "ref" = the reference's own GPU mix (fp32 decoder + fp16 towers): the drawer sees its half of it
    from . import _lib
    session_precision = args.precision
    args.precision = _lib.split_precision(session_precision)[0]
    try:
        drawer, (sideX, sideY) = plugins.make_drawer(args, device)                            # pixray.py:612-626
    finally:
        args.precision = session_precision
    if perceptor_factory is None:
        perceptor_factory, cutouts_factory, prompt_factory = _hip_parts(args, device)
    perceptors, cutouts = {}, {}
    for i, name in enumerate(args.clip_models):                                               # pixray.py:633-649
        perceptors[name] = perceptor_factory(name, i)
        size = perceptors[name].input_resolution
        if size not in cutouts:
            cutouts[size] = cutouts_factory(size, i)
    filters = plugins.setup_filters(getattr(args, "filters", None), args, device)             # pixray.py:650-669

    # ---- start image (pixray.py:674-729)
    init_image_tensor, z_orig = None, None
    run.init_images, run.overlay_images = [], []
    if args.init_image or args.init_noise:
        start = starting_image(args.init_noise, args.size[0], args.size[1]).convert("RGB").resize((sideX, sideY), Image.LANCZOS)
        if args.init_image:
            for f in real_glob(args.init_image):
                src = Image.open(f)
                init_image_tensor = to_tensor(src.convert("RGB").resize((sideX, sideY), Image.LANCZOS)).to(device).unsqueeze(0)
                top = src.convert("RGBA").resize((sideX, sideY), Image.LANCZOS)
                if args.init_image_alpha and args.init_image_alpha >= 0:
                    top.putalpha(args.init_image_alpha)
                frame = start.copy()
                frame.paste(top, (0, 0), top)
                run.init_images.append(frame)
            if init_image_tensor is None:
                raise FileNotFoundError(f"init_image matched no file: {args.init_image}")
            drawer.init_from_tensor(init_image_tensor * 2 - 1)
            z_orig = drawer.get_z_copy()
        else:
            drawer.init_from_tensor(to_tensor(start).to(device).unsqueeze(0) * 2 - 1)
    else:
        drawer.init_from_tensor(None)
    overlay = None
    if args.overlay_image is not None:                                                        # pixray.py:731-747
        for f in real_glob(args.overlay_image):
            o = Image.open(f).convert("RGBA").resize((sideX, sideY), Image.LANCZOS)
            if args.overlay_alpha:
                o.putalpha(args.overlay_alpha)
            run.overlay_images.append(o)
        overlay = run.overlay_images[0] if run.overlay_images else None

    # ---- prompts
    pms = {m: [] for m in args.clip_models}
    spot, spot_off, image_prompts = {m: [] for m in args.clip_models}, {m: [] for m in args.clip_models}, {}
    mk_prompt = lambda e, w=1.0, s=float("-inf"): prompt_factory(e.to(device).float(), w, s).to(device)   # noqa: E731
    for target in (args.target_images or []):                                                 # pixray.py:797-831
        f1, weight, stop = parse_prompt(target)
        files = real_glob(f1)
        for m in args.clip_models:
            res = perceptors[m].input_resolution
            batch = torch.stack([_clip_preprocess(Image.open(f).convert("RGB"), res) for f in files]).to(device)
            # the reference normalises the target images with CLIP's mean / std itself (`do_image_features`, pixray.py:567-575) and
            # THEN hands them to the perceptor wrapper, whose encode_image renormalises (batch min / max) and normalises again
            # (slip.py:58-66): kept, so that a target image lands where pixray puts it
            mean = torch.tensor([0.48145466, 0.4578275, 0.40821073], device=device)[:, None, None]
            std = torch.tensor([0.26862954, 0.26130258, 0.27577711], device=device)[:, None, None]
            batch = (batch - mean) / std
            with torch.no_grad():
                feats = perceptors[m].encode_image(batch).float()
            pms[m].append(mk_prompt(feats, weight, stop))
    z_labels = []
    if args.image_labels is not None:                                                         # pixray.py:833-849
        cur = []
        for f in real_glob(args.image_labels):
            t = to_tensor(Image.open(f).convert("RGB").resize((sideX, sideY), Image.LANCZOS)).to(device).unsqueeze(0) * 2 - 1
            cur.append(drawer.get_z_from_tensor(t))
        emb = torch.stack(cur)
        emb = emb / emb.norm(dim=-1, keepdim=True)
        emb = emb.mean(dim=0)
        z_labels.append((emb / emb.norm()).unsqueeze(0))
    if z_orig is not None:
        z_orig = drawer.get_z_copy()
    for prompt in (args.prompts or []):                                                       # pixray.py:859-877
        txt, weight, stop = parse_prompt(prompt)
        for m in args.clip_models:
            pms[m].append(mk_prompt(perceptors[m].encode_text(txt).float(), weight, stop))
    from .api import load_vector_table
    for vp in args.vector_prompts:                                                            # pixray.py:887-915
        f1, weight, stop = parse_prompt(vp)
        table = load_vector_table(f1)
        for m in args.clip_models:
            if m not in table:
                print(f"WARNING: no vector for {m} in {f1}!")
                print("Continuing without this vector... (BUT THIS RESULT MIGHT NOT BE WHAT YOU WANT)")
                continue
            pms[m].append(mk_prompt(torch.tensor(np.array(table[m]), dtype=torch.float32), 0.1 * weight, stop))
    for src, dst in ((args.spot_prompts, spot), (args.spot_prompts_off, spot_off)):            # pixray.py:917-931
        for prompt in (src or []):
            txt, weight, stop = parse_prompt(prompt)
            for m in args.clip_models:
                dst[m].append(mk_prompt(perceptors[m].encode_text(txt).float(), weight, stop))
    for label in (args.labels or []):                                                         # pixray.py:933-945
        txt, weight, stop = parse_prompt(label)
        for m in args.clip_models:
            ce = perceptors[m].encode_text([t.format(txt) for t in IMAGENET_TEMPLATES]).float()
            ce = ce / ce.norm(dim=-1, keepdim=True)
            ce = ce.mean(dim=0)
            pms[m].append(mk_prompt((ce / ce.norm()).unsqueeze(0), weight, stop))
    if args.image_prompts:                                                                    # pixray.py:947-953
        imgs = [to_tensor(resize_image(Image.open(p).convert("RGB"), (sideX, sideY))).unsqueeze(0).to(device) for p in args.image_prompts]
        image_prompts = {m: imgs for m in args.clip_models}
    last = args.clip_models[-1]            # pixray.py:955-958 appends noise prompts to the LAST perceptor's list only
    for s, w in zip(args.noise_prompt_seeds, args.noise_prompt_weights):
        e = torch.empty([1, perceptors[last].output_dim]).normal_(generator=torch.Generator().manual_seed(s))
        pms[last].append(mk_prompt(e, w))
    if args.spot_prompts or args.spot_prompts_off:
        for size, mk in cutouts.items():
            mk.spot_masks = load_spot_masks(args.spot_file, size, args.aspect_width)
    custom, loss_globals, args = plugins.setup_custom_losses(getattr(args, "custom_loss", None), args, device)   # pixray.py:961-995

    sess = Session(drawer, perceptors, cutouts, pms, learning_rate=args.learning_rate, iterations=args.iterations,
                   batches=args.batches, learning_rate_drops=args.learning_rate_drops, custom_losses=custom, filters=filters,
                   args=args, init_weight=args.init_weight or 0.0, init_weight_dist=args.init_weight_dist, z_orig=z_orig,
                   seed=int(seed) % (2 ** 31), auto_stop=args.auto_stop, image_prompts=image_prompts or None,
                   image_prompt_weight=args.image_prompt_weight, image_prompt_shuffle=args.image_prompt_shuffle,
                   z_labels=z_labels, image_label_weight=args.image_label_weight, init_weight_pix=args.init_weight_pix,
                   init_weight_cos=args.init_weight_cos, init_image_tensor=init_image_tensor, spot_prompts=spot,
                   spot_prompts_off=spot_off, overlay_image=overlay, overlay_every=args.overlay_every,
                   overlay_offset=args.overlay_offset, overlay_until=args.overlay_until, overlay_alpha=None,
                   prompt_factory=prompt_factory, loss_globals=loss_globals)
    sess.max_loss_drops = len(args.learning_rate_drops)                                       # pixray.py:1979
    sess.iter_drop_delay = 12                                                                 # pixray.py:1980
    run.session = sess
    print("Using device:", device)
    print("Optimising using:", args.optimiser)
    for label, value in (("text prompts", args.prompts), ("spot prompts", args.spot_prompts), ("spot off prompts", args.spot_prompts_off)):
        if value:
            print(f"Using {label}:", value)
    return sess


def _clip_preprocess(img, res: int) -> torch.Tensor:
    """torchvision Compose([Resize(res, BICUBIC), CenterCrop(res), ToTensor()]) of pixray.py:777-781 (the CLIP mean / std
    normalisation is inside `encode_image` on this path)"""
    from PIL import Image
    w, h = img.size
    s = res / min(w, h)
    img = img.resize((max(res, round(w * s)), max(res, round(h * s))), Image.BICUBIC)
    w, h = img.size
    left, top = int(round((w - res) / 2.0)), int(round((h - res) / 2.0))
    return to_tensor(img.crop((left, top, left + res, top + res)))


def load_spot_masks(spot_file: Optional[str], S: int, aspect_width: float = 1.0):
    """pixray.py:368-394 `fetch_spot_indexes`: the spot image (--spot_file, else inputs/spot_wide.png on a non-square canvas,
    else inputs/spot_square.png, relative to the working directory as in the reference) resized to the cutout size, and
    the pair (value >= 0.5, value < 0.5) of bool [3,S,S] masks `MakeCutouts.forward(spot=...)` blanks the pooled image with.
    When no file is there, a centred disc of radius S/4 stands in (the reference's images are not part of this package)."""
    from PIL import Image
    path = spot_file if spot_file is not None else ("inputs/spot_wide.png" if aspect_width != 1 else "inputs/spot_square.png")
    if os.path.exists(path):
        m = to_tensor(Image.open(path).convert("RGB").resize((S, S), Image.LANCZOS))
    elif spot_file is not None:
        raise FileNotFoundError(spot_file)
    else:
        ys, xs = torch.meshgrid(torch.arange(S).float(), torch.arange(S).float(), indexing="ij")
        m = (((xs - (S - 1) / 2) ** 2 + (ys - (S - 1) / 2) ** 2) < (S / 4) ** 2).float()[None].expand(3, S, S)
    return m.ge(0.5), m.lt(0.5)


# ---------------------------------------------------------------------------------------------------------- do_run
def _pick_filelist(old_src, old, cur_src, cur):
    """pixray.py:1524-1536: the animation follows the LONGEST file list among overlay / target / init images"""
    if old_src is None or len(old) < len(cur):
        return cur_src, cur
    return old_src, old


def do_run(args, return_display: bool = False, run: Optional[Run] = None) -> bool:
    """pixray.py:1538-1631.  Returns True when the run is complete; with `return_display` it returns False every
    `display_every` iterations so that a caller can publish the current output file and call again (cogrun.py:47-52)."""
    run = run if run is not None else _RUN
    sess = run.session
    if args.animation_dir is not None:
        _run_animation(args, run)
    else:
        try:
            keep_going = True
            while keep_going:
                it = sess.cur_iteration
                try:
                    keep_going = run.train(args, it)
                except RuntimeError as e:
                    print("Oops: runtime error: ", e)
                    print("Try reducing --num-cuts to save memory")
                    raise
                if it == args.iterations:
                    break
                sess.cur_iteration = it + 1
                if keep_going and return_display and sess.cur_iteration % args.display_every == 0:
                    return False
        except KeyboardInterrupt:
            pass
    if args.make_video:
        folder = os.path.join(args.outdir, "video")
        frames = [os.path.join(folder, f"frame_{i:04d}.png") for i in range(1, sess.cur_iteration)]
        frames_to_video([f for f in frames if os.path.exists(f)], get_file_path(args.outdir, args.output, ".mp4"), str(args.prompts))
    if args.save_intermediates and shutil.which("ffmpeg") is not None:
        steps = os.path.join(args.outdir, "steps")
        frames_to_video(sorted(glob.glob(os.path.join(steps, "frame_*.png"))), os.path.join(steps, "output.mp4"))
    return True


def _run_animation(args, run: Run) -> None:
    """pixray.py:1544-1609: one z per frame of the longest input file list; every round each frame trains `save_every`
    iterations from its own z, then every frame is blended with its predecessor's image (alpha `animation_alpha`) and
    re-encoded into its z for the next round"""
    sess = run.session
    os.makedirs(args.animation_dir, exist_ok=True)
    src, files = None, []
    if args.overlay_image is not None:
        src, files = _pick_filelist(src, files, "overlay_images", real_glob(args.overlay_image))
    if args.target_images:
        cur = []
        for t in args.target_images:
            cur += real_glob(parse_prompt(t)[0])
        src, files = _pick_filelist(src, files, "target_images", cur)
    if args.init_image is not None:
        src, files = _pick_filelist(src, files, "init_images", real_glob(args.init_image))
    if args.image_prompts:
        src, files = _pick_filelist(src, files, "image_prompts", list(args.image_prompts))
    print(f"==> animation filelist {src} ({len(files)} files)")
    n = len(files)
    run.anim_output_files = [os.path.join(args.animation_dir, os.path.basename(f)) for f in files]
    run.anim_cur_zs = [sess.drawer.get_z_copy() for _ in range(n)]
    step = 0
    while n > 0:
        images = []
        for i in range(n):
            run.cur_anim_index = i
            sess.drawer.set_z(run.anim_cur_zs[i])
            it = step
            for _ in range(args.save_every):
                run.train(args, it)
                it += 1
            run.anim_cur_zs[i] = sess.drawer.get_z_copy()
            images.append(sess.drawer.to_image())
        step += args.save_every
        sess.cur_iteration = step
        if step >= args.iterations:
            break
        for i in range(n):
            base = images[i].copy()
            prev = images[(i + n - 1) % n].copy().convert("RGBA")
            prev.putalpha(args.animation_alpha)
            base.paste(prev, (0, 0), prev)
            sess.drawer.reapply_from_tensor(to_tensor(base.convert("RGB")).to(run.device).unsqueeze(0) * 2 - 1)
            run.anim_cur_zs[i] = sess.drawer.get_z_copy()


# ---------------------------------------------------------------------------------------------------------- module-level API
_RUN = Run()


def reset_settings() -> None:
    _RUN.settings = {}


def add_settings(**kwargs) -> None:
    _RUN.settings.update(kwargs)


def get_settings() -> dict:
    return dict(_RUN.settings)


def add_custom_loss(name: str, customloss: type) -> None:
    plugins.add_custom_loss(name, customloss)


def apply_settings(argv=None, run: Optional[Run] = None):
    """pixray.py:2055-2102: a first pass finds the drawer / filters / custom losses (each contributes its own options), then
    the full parser runs over the settings dictionary (+ the command line, unless `skip_args` is set).  A setting no parser
    knows is a ValueError."""
    run = run if run is not None else _RUN
    parser = argparse.ArgumentParser(description="Image generation on the MI355X-native pixray hot path")
    parser.add_argument("--drawer", type=str, default="vqgan", dest="drawer")
    parser.add_argument("--filters", type=str, default=None, dest="filters")
    parser.add_argument("--losses", "--custom_loss", type=str, default=None, dest="custom_loss")
    argv = [] if "skip_args" in run.settings else (sys.argv[1:] if argv is None else list(argv))
    core, merged, argv = _parse_known_with_yaml(parser, run.settings, argv)
    setup_parser(parser)
    plugins.class_table[core.drawer].add_settings(parser)
    for spec, table in ((core.filters, plugins.filters_class_table), (core.custom_loss, plugins.loss_class_table)):
        if spec is not None:
            for chunk in [c.strip() for c in spec.split(",")]:
                table[chunk.split("->")[0].split(":")[0]].add_settings(parser)
    namespace = None
    if merged:
        dests = {a.dest for a in parser._actions}
        for k, v in merged.items():
            if k not in dests and k != "skip_args":
                raise ValueError(f"Requested setting not found, aborting: {k}={v}")
        namespace = types.SimpleNamespace(**merged)
    settings = process_args(parser, namespace, argv=argv, run=run)
    logging.debug(json.dumps(settings, default=lambda o: getattr(o, "__dict__", str(o)), sort_keys=True, indent=4))
    return settings


def run(prompts=None, drawer="vqgan", **kwargs):
    """pixray.py:2119-2124: one-stop entry point for notebooks and scripts"""
    reset_settings()
    add_settings(prompts=prompts, drawer=drawer, skip_args=True, **kwargs)
    settings = apply_settings()
    do_init(settings)
    do_run(settings)
    return settings


def main(argv=None):
    settings = apply_settings(argv)
    print(f"Running with {settings.num_cuts}x{settings.batches} = {settings.num_cuts * settings.batches} cuts")
    do_init(settings)
    do_run(settings)


def predict(base_settings: Optional[dict] = None, **kwargs) -> Iterator[str]:
    """The serving contract of cogrun.py:25-52 without the cog dependency: settings from a dictionary (the reference reads
    cogs/<name>.yaml) + keyword overrides, then one path per `display_every` iterations -- a temporary COPY of the current
    output image, as the reference yields, so that the consumer never reads a file that is being rewritten."""
    import tempfile
    reset_settings()
    add_settings(**(base_settings or {}))
    add_settings(**kwargs)
    add_settings(skip_args=True)
    settings = apply_settings()
    do_init(settings)
    done = False
    while not done:
        done = do_run(settings, return_display=True)
        out = os.path.join(settings.outdir, settings.output)
        tmp = os.path.join(tempfile.gettempdir(), "tempfile" + os.path.splitext(out)[1])
        shutil.copy2(out, tmp)
        yield os.path.realpath(tmp)


if __name__ == "__main__":
    main()
