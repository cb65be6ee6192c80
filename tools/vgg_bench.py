"""Timing of the HIP VGG16 extractor (StyleLoss plugin): forward and forward+backward at the canvas sizes of
BASELINE.json configs[3]/[1], and one full STROTSS loss evaluation.  Run on the GPU box."""
import os
import sys
import time
import warnings

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from pixray_amd import ops, weights
from pixray_amd import style_loss as sl

dev = torch.device("cuda", 0)
params = weights.synthetic_vgg16_params(0)
cfg = [(3, 64), (64, 64), (64, 128), (128, 128), (128, 256), (256, 256), (256, 256), (256, 512), (512, 512), (512, 512), (512, 512), (512, 512), (512, 512)]
stage = [0, 0, 1, 1, 2, 2, 2, 3, 3, 3, 4, 4, 4]
for S in (256, 512):
    h = ops.Vgg16Handle(params, (S, S), dev)
    x = torch.randn(1, 3, S, S, device=dev, requires_grad=True)
    gflop = sum(2.0 * 9 * ci * co * (S >> st) * (S >> st) for (ci, co), st in zip(cfg, stage)) / 1e9

    def fwd():
        with torch.no_grad():
            return ops.vgg16_features(x, h)

    def fwdbwd():
        f = ops.vgg16_features(x, h)
        torch.autograd.grad(sum(t.sum() for t in f), x)
    for name, fn, work in (("forward", fwd, gflop), ("forward+backward", fwdbwd, 2 * gflop)):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 10
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        ms = 1e3 * (time.perf_counter() - t0) / n
        print(f"vgg16 {S}x{S} {name}: {ms:.3f} ms  ({work:.1f} GFLOP -> {work / ms:.1f} TFLOP/s)")
    del h
img = torch.rand(1, 3, 256, 256, device=dev, requires_grad=True)
style = torch.rand(1, 3, 256, 256, device=dev)
ex = sl.Vgg16Extractor(params=params, device=dev, max_hw=(256, 256))
np.random.seed(0)
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    for i in range(3):
        if i == 1:
            torch.cuda.synchronize(); t0 = time.perf_counter()
        loss = sl.strotss_loss(img, style, 32.0, extractor=ex)
        torch.autograd.grad(loss, img)
    torch.cuda.synchronize()
print(f"StyleLoss (STROTSS, 3 scales, 27 extractor passes) value+gradient at 256x256: {1e3 * (time.perf_counter() - t0) / 2:.1f} ms")
