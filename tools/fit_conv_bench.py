"""Diagnostic (run on the GPU box): where the time of the decoder's implicit 3x3 convolutions on the fit kernel goes --
whole launch / K loop only / epilogue only (gemmfit.hip fit_flags bits 3 and 2), with and without the GroupNorm sums the
decoder accumulates in the epilogue (forward sums; backward sums of a dgrad).

    python tools/fit_conv_bench.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pixray_amd import _lib
from pixray_amd._lib import GemmArgs, call

dev = "cuda"
lib = _lib.load()
ctx = _lib.tool_ctx()
h16 = torch.float16
s = _lib.current_stream()
# (H, W, Cin, Cout, up) of the headline decoder (VQGAN f16, 256^2), most frequent first
shapes = [(64, 64, 256, 256, 0), (128, 128, 128, 128, 0), (16, 16, 512, 512, 0), (32, 32, 256, 256, 0), (32, 32, 512, 512, 0),
          (256, 256, 128, 128, 0), (128, 128, 256, 256, 0), (64, 64, 512, 256, 0)]


def timeit(fn, iters=60):
    for _ in range(8):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / iters


for H, W, Cin, Cout, up in shapes:
    M, K = H * W, 9 * Cin
    x = torch.randn(M, Cin, device=dev).to(h16)
    w = (torch.randn(Cout, K, device=dev) / K ** 0.5).to(h16)
    bias = torch.randn(Cout, device=dev)
    resid = torch.randn(M, Cout, device=dev)
    out = torch.empty(M, Cout, device=dev)
    o16 = torch.empty(M, Cout, device=dev, dtype=h16)
    stats = torch.zeros(64, device=dev, dtype=torch.float64)
    xg = torch.randn(M, Cout, device=dev)
    fst = torch.stack([xg.double().view(M, 32, -1).sum(dim=(0, 2)), (xg.double() ** 2).view(M, 32, -1).sum(dim=(0, 2))], 1).reshape(-1).contiguous()
    gamma, beta = torch.randn(Cout, device=dev), torch.randn(Cout, device=dev)
    gs = Cout // 32
    xr = torch.randn(M, K, device=dev).to(h16)       # the same product with a row-major A (no gather arithmetic)

    def args(kind):
        g = GemmArgs()
        g.A = x.data_ptr(); g.a_mode = 1; g.lda = Cin; g.B = w.data_ptr(); g.ldb = K
        g.M, g.N, g.K = M, Cout, K
        g.H, g.W, g.Cin, g.up = H, W, Cin, up
        g.alpha = 1.0; g.f32 = 2
        g.out_f32 = out.data_ptr(); g.ldc_f32 = Cout
        if kind == "rowmajor":
            g.A = xr.data_ptr(); g.a_mode = 0; g.lda = K; g.H = g.W = g.Cin = g.up = 0
        if kind in ("fwd", "fwd+gn", "rowmajor"):
            g.bias_n = bias.data_ptr(); g.resid = resid.data_ptr(); g.ldr = Cout
            g.out_bf16 = o16.data_ptr(); g.ldc_bf16 = Cout
        return g
    res = {}
    for kind in ("rowmajor", "fwd", "fwd+gn", "dgrad", "dgrad+gnb"):
        g = args(kind)
        for name, flags in (("all", 1), ("loop", 1 + 4), ("epi", 1 + 8)):
            lib.prx_gemm_tile_override(ctx, -8, 0, flags)
            if kind == "fwd+gn":
                fn = lambda: call("prx_k_gemm_gn", g, stats, gs, None, None, None, None, 0, 1e-6, None, 0, s)
            elif kind == "dgrad+gnb":
                fn = lambda: call("prx_k_gemm_gn", g, stats, gs, xg, fst, gamma, beta, 1, 1e-6, None, 0, s)
            else:
                fn = lambda: call("prx_k_gemm", g, None, 0, s)
            res[(kind, name)] = timeit(fn)
    lib.prx_gemm_tile_override(ctx, -8, 0, 1)
    fl = 2.0 * M * Cout * K
    print(f"{H}x{W} {Cin}->{Cout} (M={M} N={Cout} K={K}): " + " | ".join(
        f"{kind}: {res[(kind, 'all')]:5.1f} us ({fl / res[(kind, 'all')] / 1e6:4.0f} TF; loop {res[(kind, 'loop')]:5.1f}, epilogue {res[(kind, 'epi')]:5.1f})"
        for kind in ("rowmajor", "fwd", "fwd+gn", "dgrad", "dgrad+gnb")), flush=True)
