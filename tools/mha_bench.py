"""General-T attention kernels at the BASELINE configurations' shapes (GPU): forward and backward time per layer.
    python tools/mha_bench.py            # workgroup-per-head kernels (64 < T <= 512)
    PRX_MHA_TILES=1 python tools/mha_bench.py   # the one-wave-per-tile kernels they replaced"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from pixray_amd._lib import call, current_stream as stream

DEV = "cuda"
SHAPES = [("ViT-B/16 @128 cutouts", 128, 197, 768, 12), ("ViT-L/14 @256 cutouts", 256, 257, 1024, 16),
          ("ViT-L/14 @32 cutouts", 32, 257, 1024, 16), ("RN50x4 attention pool @128", 128, 82, 2560, 40)]
for name, N, T, C, heads in SHAPES:
    qkv = torch.randn(N * T, 3 * C, device=DEV).bfloat16()
    out = torch.empty(N * T, C, dtype=torch.bfloat16, device=DEV)
    lse = torch.empty(N * heads * T, device=DEV)
    do = torch.randn(N * T, C, device=DEV).bfloat16()
    dqkv = torch.empty(N * T, 3 * C, dtype=torch.bfloat16, device=DEV)
    fwd = lambda: call("prx_k_mha_fwd_gen", qkv, out, lse, N, T, C, heads, stream())
    bwd = lambda: call("prx_k_mha_bwd_gen", qkv, out, do, lse, dqkv, N, T, C, heads, stream())
    res = []
    for fn in (fwd, bwd):
        for _ in range(3):
            fn()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(20):
            fn()
        b.record()
        torch.cuda.synchronize()
        res.append(a.elapsed_time(b) / 20 * 1e3)
    flop_f = 4.0 * N * heads * T * T * 64
    print(f"{name:30s} T={T:4d}: forward {res[0]:8.1f} us ({flop_f / res[0] / 1e6:6.1f} TFLOP/s)   backward {res[1]:8.1f} us "
          f"({2.5 * flop_f / res[1] / 1e6:6.1f} TFLOP/s)")
