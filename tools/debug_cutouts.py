import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import cutouts_ref
from pixray_amd import cutouts as pc

def rel(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return ((a - b).norm() / (b.norm() + 1e-30)).item()

def run(tag, cutn=10, S=224, HW=256, it=0, mod=None, sat_img=True, seed=3, smooth=False, clampy=False):
    g = torch.Generator().manual_seed(seed)
    img = torch.rand(1, 3, HW, HW, generator=g)
    if smooth:
        img = torch.nn.functional.interpolate(torch.rand(1, 3, HW // 8, HW // 8, generator=g), size=(HW, HW), mode="bicubic").clamp(0, 1)
    if clampy:
        img = (torch.nn.functional.interpolate(torch.rand(1, 3, HW // 8, HW // 8, generator=g), size=(HW, HW), mode="bicubic") * 1.6 - 0.3).clamp(0, 1)
    if sat_img:
        img[:, :, : HW // 4] = img[:, :, : HW // 4].round()
    prm = pc.sample_cutout_params(cutn, S, g, iteration=it)
    prm["noise"] = torch.randn(cutn, 3, S, S, generator=g)
    if mod: mod(prm)
    ir = img.clone().requires_grad_(True)
    ref = cutouts_ref.make_cutouts(ir, prm, S)
    gout = torch.randn(cutn, 3, S, S, generator=g)
    (gref,) = torch.autograd.grad(ref, ir, gout)
    mk = pc.MakeCutouts(S, cutn); mk.fixed_params = prm
    idv = img.cuda().requires_grad_(True)
    out = mk(idv)
    (gd,) = torch.autograd.grad(out, idv, gout.cuda())
    d = (gd.cpu() - gref).abs()
    print(f"{tag:40s} fwd {rel(out, ref):.2e}  bwd {rel(gd, gref):.2e}  max|d| {d.max():.3e}  n(|d|>1e-3) {(d>1e-3).sum().item()}  of {d.numel()}", flush=True)
    return gd.cpu(), gref

def nojit(p): p["z_jit_apply"][:] = False; p["w_jit_apply"][:] = False
def nopersp(p): p["z_persp_apply"][:] = False; p["w_persp_apply"][:] = False
def both(p): nojit(p); nopersp(p)

run("clampy", clampy=True, sat_img=False)
run("clampy 64", clampy=True, sat_img=False, cutn=64, it=1)
run("baseline")
run("smooth", smooth=True)
run("smooth unsat", smooth=True, sat_img=False)
run("smooth 64", smooth=True, cutn=64)
run("no saturated region", sat_img=False)
run("no jitter", mod=nojit)
run("no jitter, unsat", mod=nojit, sat_img=False)
run("no persp", mod=nopersp)
run("no jitter no persp", mod=both)
run("no jitter no persp unsat", mod=both, sat_img=False)
run("border it=1 nojit", mod=nojit, it=1)
