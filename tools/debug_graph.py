"""eager-vs-eager and graph-vs-eager teacher-forced drift of the tiny session (diagnostic)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pixray_amd import api

def run(graph):
    kw = dict(size=(64, 64), vqgan_model="tiny_f4", clip_model="tiny-B/32", num_cuts=8, seed=3)
    a = api.build_vqgan_clip_session(**kw)
    b = api.build_vqgan_clip_session(**kw)
    for mk in list(a.cutoutsTable.values()) + list(b.cutoutsTable.values()):
        mk.noise_fac = 0.0
    if graph:
        assert b.enable_graph(warmup=2)
    else:
        for it in range(3):
            b.train(it)
    for it in range(3):
        a.train(it)
    za, zb = a.drawer.get_z(), b.drawer.get_z()
    oa, ob = a.opts[0], b.opts[0]
    for it in range(3, 7):
        with torch.no_grad():
            zb.copy_(za)
            for k in ("exp_avg", "exp_avg_sq"):
                ob.state[zb][k].copy_(oa.state[za][k])
        a.train(it); b.train(it)
        d = (za.detach() - zb.detach()).abs()
        ga, gb = za.grad, zb.grad
        gr = None if ga is None or gb is None else ((ga - gb).norm() / ga.norm()).item()
        print("graph" if graph else "eager", it, "frac>1e-3", (d > 1e-3).float().mean().item(), "max", d.max().item(), "grad rel", gr,
              "loss", float(sum(a.last_losses)), float(sum(b.last_losses)))

run(False)
run(True)
