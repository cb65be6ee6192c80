"""Which part of a configs[3]-shaped session refuses hipGraph capture?  Builds small sessions that differ in one piece
(canvas shape, plugin stack) and prints Session.enable_graph's verdict for each (run on the GPU box)."""
import argparse
import os
import sys
import types
import warnings

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from pixray_amd import api, weights
from pixray_amd import style_loss as sl
from pixray_amd.cutouts import MakeCutouts
from pixray_amd.engine import Session
from pixray_amd.fft_drawer import FftDrawer
from pixray_amd.interfaces import LossInterface
from pixray_amd.perceptor import get_clip_perceptor
from pixray_amd.prompt import Prompt

DEV = torch.device("cuda", 0)


class Saturation(LossInterface):
    supports_graph_replay = True

    def get_loss(self, cur_cutouts, out, args, globals=None, lossGlobals=None):
        cut = next(iter(cur_cutouts.values()))
        return -cut.std(dim=1).mean() * 0.2


def build(size, losses, skip="1", max_hw=None):
    st = types.SimpleNamespace(size=size, fft_use="fft", fft_decay=1.5, fft_lrate=0.3)
    dr = FftDrawer(st)
    dr.load_model(st, DEV)
    dr.init_from_tensor(None)
    perc = get_clip_perceptor("tiny-B/32", DEV, max_batch=8)
    mk = MakeCutouts(224, 8, generator=torch.Generator().manual_seed(3), aspect_width=size[0] / size[1])
    pm = Prompt(api.seeded_unit_vectors(1, 128, 9).to(DEV), 1.0, float("-inf")).to(DEV)
    args = sl.StyleLoss.add_settings(argparse.ArgumentParser()).parse_args(["--styleloss_skip", skip, "--styleloss_content_weight", "8"])
    custom = []
    if "style" in losses:
        ext = None
        if max_hw:
            ext = sl.Vgg16Extractor(params=weights.synthetic_vgg16_params(0), device=DEV, max_hw=max_hw)
        style = sl.StyleLoss(extractor=ext, vgg_params=weights.synthetic_vgg16_params(0),
                             style_image=torch.rand(1, 3, 50, 60, generator=torch.Generator().manual_seed(4)), device=DEV)
        args = style.parse_settings(args)
        custom.append({"loss": style, "weight": 1.0})
    if "sat" in losses:
        custom.append({"loss": Saturation(device=DEV), "weight": 1.0})
    return Session(dr, {"tiny-B/32": perc}, {224: mk}, {"tiny-B/32": [pm]}, args=args, seed=1, custom_losses=custom)


cases = [("96x80 no plugins", (96, 80), ()), ("128x128 no plugins", (128, 128), ()), ("96x80 saturation", (96, 80), ("sat",)),
         ("128x128 style skip0 reserved", (128, 128), ("style",), "0", (128, 128)),
         ("128x128 style skip1", (128, 128), ("style",), "1"),
         ("96x80 style skip0 reserved", (96, 80), ("style",), "0", (96, 96)),
         ("96x80 style skip1", (96, 80), ("style",), "1"),
         ("96x80 style+sat skip1", (96, 80), ("style", "sat"), "1")]
only = sys.argv[1:]
for c in cases:
    name, size, losses = c[0], c[1], c[2]
    if only and not any(o in name for o in only):
        continue
    np.random.seed(0)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        sess = build(size, losses, *(c[3:]))
        ok = sess.enable_graph(warmup=2)
        if ok:
            for _ in range(3):
                sess.train()
            torch.cuda.synchronize()
    print(f"{name:32s} -> {'captured + replayed' if ok else 'REFUSED: ' + str(sess.graph_error).splitlines()[0]}", flush=True)
