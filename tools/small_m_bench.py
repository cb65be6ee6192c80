"""Diagnostic (GPU box): weight-heavy products with few output rows (the decoder's 16^2 / 32^2 levels as row-major GEMMs of the same
shape) on the 4-wave kernels (split-K + reduce) and the ring fit tiles with 4 / 8 K groups, hot (one buffer set) and cold
(rotating over > 256 MB of buffer sets).   python tools/small_m_bench.py"""
import math
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pixray_amd import _lib
from pixray_amd._lib import GemmArgs, call

dev = "cuda"
ws = torch.empty(64 << 20, dtype=torch.uint8, device=dev)
lib = _lib.load()
ctx = _lib.tool_ctx()
lib.prx_gemm_tile_override(ctx, -12, 0, 1)      # forced tiles of the shapes both kernel families have mean the fit kernel
s = _lib.current_stream()


def timeit(fns, iters=60):
    n = len(fns)
    for i in range(8):
        fns[i % n]()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(iters):
        fns[i % n]()
    e1.record(); torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / iters


def empty_kernel_us():
    x = torch.zeros(64, device=dev)
    return timeit([lambda: x.add_(1.0)], 200)


print(f"launch floor (a 64-element torch add, back to back): {empty_kernel_us():.2f} us")
for (M, N, K, tiles) in [(256, 512, 4608, [(16, 32), (16, 64)]), (1024, 256, 2304, [(16, 64), (32, 64)]),
                         (1024, 512, 4608, [(32, 64), (64, 64)]), (64, 512, 4608, [(16, 32)]), (16, 512, 4608, [(16, 32)])]:
    for nset in (1, 40):
        sets = []
        for i in range(nset):
            A = torch.randn(M, K, device=dev).to(torch.float16)
            Bt = (torch.randn(N, K, device=dev) / math.sqrt(K)).to(torch.float16)
            out = torch.empty(M, N, device=dev)
            g = GemmArgs()
            g.A = A.data_ptr(); g.lda = K; g.B = Bt.data_ptr(); g.ldb = K; g.M, g.N, g.K = M, N, K
            g.alpha = 1.0; g.f32 = 2; g.out_f32 = out.data_ptr(); g.ldc_f32 = N
            sets.append((A, Bt, out, g))
        row = [f"M={M:5d} N={N:4d} K={K:5d} sets={nset:2d}:"]
        lib.prx_gemm_tile_override(ctx, 0, 0, 0)
        lib.prx_gemm_tile_override(ctx, -7, 0, 0)
        row.append(f"4-wave+splitK {timeit([(lambda g=t[3]: call('prx_k_gemm', g, ws, ws.numel(), s)) for t in sets]):6.1f}")
        ref = sets[0][2].clone()
        lib.prx_gemm_tile_override(ctx, -7, 0, 1)
        for tile in tiles:
            lib.prx_gemm_tile_override(ctx, tile[0], tile[1], 1)
            us = timeit([(lambda g=t[3]: call('prx_k_gemm', g, ws, ws.numel(), s)) for t in sets])
            err = ((sets[0][2] - ref).norm() / ref.norm()).item()
            row.append(f"{tile[0]}x{tile[1]} {us:6.1f} (d {err:.0e})")
        lib.prx_gemm_tile_override(ctx, 0, 0, 0)
        row.append(f"vendor {timeit([(lambda t=t: torch.matmul(t[0], t[1].t())) for t in sets]):6.1f}")
        print("  ".join(row), flush=True)
