import sys, os, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from oracle import cutouts_ref as cr
from pixray_amd import ops
S = 224
g = torch.Generator().manual_seed(0)
img = torch.rand(1, 3, S, S, generator=g)
for (sat, hue, sf) in [(1.05, 0.03, True), (0.93, -0.07, False), (1.0999, 0.0999, True)]:
    desc = torch.zeros(1, 32, dtype=torch.float64)
    desc[0, 0] = desc[0, 4] = desc[0, 8] = 1; desc[0, 9] = desc[0, 13] = desc[0, 17] = 1
    desc[0, 21] = 1; desc[0, 22] = sat; desc[0, 23] = float(torch.tensor(hue) * cr.TWO_PI); desc[0, 24] = float(sf)
    xr = img.clone().requires_grad_(True)
    ref = cr.color_jitter(xr, torch.tensor([True]), torch.tensor([sat]), torch.tensor([hue]), sf)
    gout = torch.randn(1, 3, S, S, generator=g)
    (gref,) = torch.autograd.grad(ref, xr, gout)
    xd = img.cuda().requires_grad_(True)
    out = ops.make_cutouts(xd, desc.cuda(), None, S)
    (gd,) = torch.autograd.grad(out, xd, gout.cuda())
    d = (gd.cpu() - gref).abs().amax(1)[0]
    fd = (out.detach().cpu() - ref.detach()).abs().amax(1)[0]
    print(f"sat={sat} hue={hue} sf={sf}: fwd max {fd.max():.2e}  bwd max {d.max():.3e}  n>1e-3 {(d > 1e-3).sum().item()} of {d.numel()}")
    idx = torch.argsort(d.flatten(), descending=True)[:4]
    for i in idx.tolist():
        y, x = i // S, i % S
        print("   px", y, x, "rgb", img[0, :, y, x].tolist(), "gpu", gd[0, :, y, x].tolist(), "ref", gref[0, :, y, x].tolist(), "fwd diff", fd[y, x].item())
