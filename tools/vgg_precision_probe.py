import math, sys, warnings
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from pixray_amd import ops, weights, style_loss as sl
from oracle import vgg_ref
DEV = "cuda"
def rel(a, b): return ((a.float() - b.float()).norm() / (b.float().norm() + 1e-30)).item()
def cos(a, b):
    a, b = a.double().flatten(), b.double().flatten(); return float(a @ b / (a.norm() * b.norm()))
params = weights.synthetic_vgg16_params(0)
for prec in ("f32", "fp16", "bf16"):
    for (H, W) in [(64, 48), (128, 128)]:
        handle = ops.Vgg16Handle(params, (128, 128), torch.device(DEV), precision=prec)
        g = torch.Generator().manual_seed(5)
        x = (torch.rand(1, 3, H, W, generator=g) * 2 - 1)
        xn = vgg_ref.normalise(x)
        xo = xn.clone().requires_grad_(True)
        ref = vgg_ref.forward_base(params, xo)[1:]
        xd = xn.to(DEV).requires_grad_(True)
        got = ops.vgg16_features(xd, handle)
        fr = [rel(f.permute(0, 3, 1, 2).cpu(), r.detach()) for f, r in zip(got, ref)]
        rs = [torch.randn(r.shape, generator=g) / math.sqrt(r.numel()) for r in ref]
        sum((r_ * f_).sum() for r_, f_ in zip(rs, ref)).backward()
        sum((r_.permute(0, 2, 3, 1).to(DEV) * f_).sum() for r_, f_ in zip(rs, got)).backward()
        print(prec, H, W, "feat max rel %.2e" % max(fr), "grad cos %.6f rel %.3e" % (cos(xd.grad.cpu(), xo.grad), rel(xd.grad.cpu(), xo.grad)), flush=True)
class OracleExtractor:
    def __call__(self, x): return [f.permute(0, 2, 3, 1).contiguous() for f in vgg_ref.forward(params, x, "uniform")]
    def forward_samples_hypercolumn(self, X, samps=100): return sl.sample_hypercolumns(self(X), samps)
g = torch.Generator().manual_seed(17)
img = torch.rand(1, 3, 96, 80, generator=g); style = torch.rand(1, 3, 96, 80, generator=g)
a = img.clone().requires_grad_(True)
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    np.random.seed(3); la = sl.strotss_loss(a, style, 16.0, extractor=OracleExtractor())
    (ga,) = torch.autograd.grad(la, a)
    for prec in ("f32", "fp16", "bf16"):
        b = img.clone().to(DEV).requires_grad_(True)
        np.random.seed(3); lb = sl.strotss_loss(b, style.to(DEV), 16.0, extractor=sl.Vgg16Extractor(params=params, device=DEV, max_hw=(96, 80), precision=prec))
        (gb,) = torch.autograd.grad(lb, b)
        print("strotss", prec, "value rel %.3e" % (abs(float(la) - float(lb)) / abs(float(la))), "grad cos %.6f rel %.3e" % (cos(gb.cpu(), ga), rel(gb.cpu(), ga)), flush=True)
