"""Stage-by-stage run-to-run comparison of the tiny session (diagnostic): same state, two passes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pixray_amd import api

kw = dict(size=(64, 64), vqgan_model="tiny_f4", clip_model="tiny-B/32", num_cuts=8, seed=3)
if len(sys.argv) > 1 and sys.argv[1] == "big":
    kw = dict(size=(256, 256), vqgan_model="imagenet_f16_16384", clip_model="ViT-B/32", num_cuts=64, seed=3)
s = api.build_vqgan_clip_session(**kw)
for mk in s.cutoutsTable.values():
    mk.noise_fac = 0.0
mk = list(s.cutoutsTable.values())[0]
name, perc = list(s.perceptors.items())[0]
prompt = s.pmsTable[name][0]
z = s.drawer.get_z()

def one_pass(it):
    res = {}
    if z.grad is not None:
        z.grad = None
    mk.prepare(iteration=it, fill=0.5)
    prm = mk.last_params
    img = s.drawer.synth(it)
    img.retain_grad()
    res["img"] = img.detach().clone()
    mk.fixed_params = prm
    cut = mk(img)
    mk.fixed_params = None
    cut.retain_grad()
    res["cut"] = cut.detach().clone()
    emb = perc.encode_image(cut).float()
    emb.retain_grad()
    res["emb"] = emb.detach().clone()
    loss = prompt(emb)
    res["loss"] = loss.detach().clone()
    loss.backward()
    res["d_emb"] = emb.grad.clone(); res["d_cut"] = cut.grad.clone(); res["d_img"] = img.grad.clone(); res["d_z"] = z.grad.clone()
    return res, prm

r1, prm = one_pass(5)
mk.fixed_params = prm
for rep in range(3):
    r2, _ = one_pass(5)
    print("rep", rep, " ".join(f"{k}:{((r1[k] - r2[k]).norm() / (r1[k].norm() + 1e-30)).item():.2e}" for k in r1))
