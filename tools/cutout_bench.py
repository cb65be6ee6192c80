"""MakeCutouts forward / backward time at the headline size (run on the GPU box): 64 cutouts of 224^2 from a 256^2 image,
even (reflection) and odd (border) iterations, torch events around the autograd calls."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pixray_amd import cutouts as pc

dev = "cuda"
cutn, S, HW = 64, 224, 256
g = torch.Generator().manual_seed(0)
img = torch.rand(1, 3, HW, HW, generator=g).to(dev)
gout = torch.randn(cutn, 3, S, S, generator=g).to(dev)
for it in (0, 1):
    prm = pc.sample_cutout_params(cutn, S, g, iteration=it)
    prm["noise"] = torch.randn(cutn, 3, S, S, generator=g)
    mk = pc.MakeCutouts(S, cutn)
    tf, tb = [], []
    for rep in range(25):
        mk.fixed_params = prm
        mk.transforms = None
        x = img.clone().requires_grad_(True)
        e0, e1, e2 = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        e0.record()
        out = mk(x)
        e1.record()
        (gx,) = torch.autograd.grad(out, x, gout)
        e2.record()
        torch.cuda.synchronize()
        if rep >= 5:
            tf.append(e0.elapsed_time(e1)); tb.append(e1.elapsed_time(e2))
    tf.sort(); tb.sort()
    print(f"iteration parity {it} ({'reflection' if it % 2 == 0 else 'border'} padding): forward {1e3 * tf[len(tf) // 2]:.1f} us, "
          f"backward {1e3 * tb[len(tb) // 2]:.1f} us (median of {len(tf)})")
