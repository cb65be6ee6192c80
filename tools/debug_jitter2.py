import sys, os, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import cutouts_ref as cr
from pixray_amd import ops, cutouts as pc
S, HW, cutn = 224, 256, 10
g = torch.Generator().manual_seed(3)
img = torch.rand(1, 3, HW, HW, generator=g)
img = torch.nn.functional.interpolate(torch.rand(1, 3, HW // 8, HW // 8, generator=g), size=(HW, HW), mode="bicubic").clamp(0, 1)
img[:, :, : HW // 4] = img[:, :, : HW // 4].round()
prm = pc.sample_cutout_params(cutn, S, g, iteration=0)
import copy
p0 = copy.deepcopy(prm); p0["z_jit_apply"][:] = False; p0["w_jit_apply"][:] = False; p0["noise"] = None
xb = cr.make_cutouts(img, p0, S).detach()     # pre-jitter batch from the oracle
nz = int(0.6 * cutn)
for n in range(cutn):
    z = n < nz
    i = n if z else n - nz
    app = bool((prm["z_jit_apply"] if z else prm["w_jit_apply"])[i])
    if not app: continue
    sat = (prm["z_sat"] if z else prm["w_sat"])[i:i+1]; hue = (prm["z_hue"] if z else prm["w_hue"])[i:i+1]
    sf = bool(prm["z_sat_first"] if z else prm["w_sat_first"])
    xr = xb[n:n+1].clone().requires_grad_(True)
    ref = cr.color_jitter(xr, torch.tensor([True]), sat, hue, sf)
    gout = torch.randn(1, 3, S, S, generator=g)
    (gref,) = torch.autograd.grad(ref, xr, gout)
    desc = torch.zeros(1, 32, dtype=torch.float64)
    desc[0, 0] = desc[0, 4] = desc[0, 8] = 1; desc[0, 9] = desc[0, 13] = desc[0, 17] = 1
    desc[0, 21] = 1; desc[0, 22] = float(sat); desc[0, 23] = float(hue.double() * 2 * math.pi); desc[0, 24] = float(sf)
    xd = xb[n:n+1].cuda().requires_grad_(True)
    out = ops.make_cutouts(xd, desc.cuda(), None, S)
    (gd,) = torch.autograd.grad(out, xd, gout.cuda())
    d = (gd.cpu() - gref).abs().amax(1)[0]
    fd = (out.detach().cpu() - ref.detach()).abs().amax(1)[0]
    print(f"cut {n} sat={float(sat):.4f} hue={float(hue):.4f} sf={sf}: fwd max {fd.max():.2e} bwd max {d.max():.3e} n>1e-3 {(d > 1e-3).sum().item()}")
    idx = torch.argsort(d.flatten(), descending=True)[:3]
    for k in idx.tolist():
        y, x = k // S, k % S
        if d[y, x] < 1e-3: break
        print("   px", y, x, "rgb", [f"{v:.9g}" for v in xb[n, :, y, x].tolist()], "gpu", [f"{v:.4f}" for v in gd[0, :, y, x].tolist()], "ref", [f"{v:.4f}" for v in gref[0, :, y, x].tolist()], "gout", [f"{v:.3f}" for v in gout[0,:,y,x].tolist()])
