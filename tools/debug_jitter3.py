import sys, os, math, copy
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
p0 = copy.deepcopy(prm); p0["z_jit_apply"][:] = False; p0["w_jit_apply"][:] = False; p0["noise"] = None; p0["noise_fac"][:] = 0
xb = cr.make_cutouts(img, p0, S).detach()
mk = pc.MakeCutouts(S, cutn, noise_fac=0.0); mk.fixed_params = p0
out = mk(img.cuda()).cpu()
d = (out - xb).abs()
print("pre-jitter max diff", d.max().item(), "n nonzero diff", (d > 0).sum().item(), "of", d.numel())
def cls(t):
    r, gg, b = t[:, 0], t[:, 1], t[:, 2]
    return (r == gg).int() + 2 * (gg == b).int() + 4 * (r == b).int() + 8 * (r > gg).int() + 16 * (gg > b).int() + 32 * (r > b).int()
c1, c2 = cls(out), cls(xb)
mis = (c1 != c2)
print("pixels with different channel ordering/tie class:", mis.sum().item())
idx = mis.nonzero()[:12]
for (n, y, x) in idx.tolist():
    print(n, y, x, "gpu", [f"{v:.9g}" for v in out[n, :, y, x].tolist()], "ref", [f"{v:.9g}" for v in xb[n, :, y, x].tolist()])
