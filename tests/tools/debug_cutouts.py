"""per-cutout forward / backward error of MakeCutouts vs the oracle (diagnostic; run on the GPU box)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from pixray_amd import cutouts as pc
from oracle import cutouts_ref
cutn, S, HW, it = 10, 224, 256, 0
g = torch.Generator().manual_seed(100 + cutn + it)
img = torch.rand(1, 3, HW, HW, generator=g)
prm = pc.sample_cutout_params(cutn, S, g, iteration=it)
prm["noise"] = torch.randn(cutn, 3, S, S, generator=g)
imr = img.clone().requires_grad_(True)
ref = cutouts_ref.make_cutouts(imr, prm, S)
mk = pc.MakeCutouts(S, cutn); mk.fixed_params = prm
imd = img.cuda().requires_grad_(True)
out = mk(imd)
d = (out.detach().cpu() - ref.detach()).abs()
desc = mk.transforms
for i in range(cutn):
    go = torch.zeros(cutn, 3, S, S); go[i] = torch.randn(3, S, S, generator=torch.Generator().manual_seed(i))
    (gr,) = torch.autograd.grad(ref, imr, go, retain_graph=True)
    (gd,) = torch.autograd.grad(out, imd, go.cuda(), retain_graph=True)
    rel = ((gd.cpu() - gr).norm() / gr.norm()).item()
    print(i, "fwd max", f"{d[i].max().item():.3e}", "bwd rel", f"{rel:.3e}", "modes", desc[i, 18].item(), desc[i, 19].item(), "jit", desc[i, 21].item())
