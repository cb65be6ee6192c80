"""Diagnostic (run on the GPU box): the fit-tile GEMM kernel (gemmfit.hip) against the 4-wave engine kernels and the vendor
library on the ViT-B/32 products of the headline iteration (M = 3200), each with the epilogue it carries in the tower.

    python tools/fit_bench.py [cold]

Prints refcheck (rel-L2 vs an fp32 torch product of the same 16-bit operands, transposition-sensitive: asymmetric random
data with a per-column scale) and microseconds per launch, hot (one buffer set, back to back) and -- with `cold` -- rotating
over buffer sets larger than the 256 MB Infinity Cache, which is closer to what a launch sees inside the iteration."""
import math
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pixray_amd import _lib
from pixray_amd._lib import GemmArgs, call

dev = "cuda"
cold = len(sys.argv) > 1 and sys.argv[1] == "cold"
ws = torch.empty(64 << 20, dtype=torch.uint8, device=dev)
lib = _lib.load()
ctx = _lib.tool_ctx()
lib.prx_gemm_tile_override(ctx, -12, 0, 1)      # forced tiles of the shapes both kernel families have mean the fit kernel
h16 = torch.float16

# (M, N, K, tag, epilogue, fit tile)
shapes = [
    (3200, 3072, 768, "FC1 (bias, QuickGELU, 2 x 16-bit out)", "fc1", (160, 256)),
    (3200, 3072, 768, "W2T dgrad (dQuickGELU(aux), 16-bit out)", "dgelu", (160, 256)),
    (3200, 768, 3072, "FC2 (bias, resid, f32 out)", "resid", (80, 128)),
    (3200, 768, 3072, "W1T dgrad (f32 out)", "f32", (80, 128)),
    (3200, 2304, 768, "QKV (bias, 16-bit out)", "bias16", (160, 192)),
    (3200, 768, 2304, "WqkvT dgrad (f32 out)", "f32", (80, 128)),
    (3200, 768, 768, "proj (bias, resid, f32 out)", "resid", (80, 128)),
    (3200, 768, 768, "WoT dgrad (16-bit out)", "plain16", (80, 128)),
    (3136, 768, 3072, "patch embed (f32 out)", "f32", (80, 128)),
    (1600, 3072, 768, "FC1 @32 cutouts", "fc1", (80, 128)),
    (1000, 200, 1152, "ragged M, N (bias, resid)", "resid", (80, 128)),
    (333, 520, 256, "ragged M, N (bias16)", "bias16", (160, 128)),
    (333, 520, 256, "ragged M, N (bias16) 160x192", "bias16", (160, 192)),
]


def make(M, N, K, epi, seed):
    g = torch.Generator(device=dev).manual_seed(seed)
    A = torch.randn(M, K, device=dev, generator=g).to(h16)
    Bt = (torch.randn(N, K, device=dev, generator=g) * torch.linspace(0.5, 1.5, N, device=dev)[:, None] / math.sqrt(K)).to(h16)
    t = dict(A=A, Bt=Bt)
    if epi in ("fc1", "resid", "bias16"):
        t["bias"] = torch.randn(N, device=dev, generator=g)
    if epi == "resid":
        t["resid"] = torch.randn(M, N, device=dev, generator=g)
    if epi == "dgelu":
        t["aux"] = torch.randn(M, N, device=dev, generator=g).to(h16)
    if epi in ("resid", "f32"):
        t["out_f32"] = torch.empty(M, N, device=dev)
    else:
        t["out16"] = torch.empty(M, N, device=dev, dtype=h16)
    if epi == "fc1":
        t["pre16"] = torch.empty(M, N, device=dev, dtype=h16)
    return t


def args_of(t, M, N, K, epi):
    g = GemmArgs()
    g.A = t["A"].data_ptr(); g.a_is_f32 = 0; g.a_mode = 0; g.lda = K
    g.B = t["Bt"].data_ptr(); g.ldb = K; g.M, g.N, g.K = M, N, K
    g.alpha = 1.0; g.f32 = 2      # PRX_PREC_F16
    if "bias" in t: g.bias_n = t["bias"].data_ptr()
    if "resid" in t: g.resid = t["resid"].data_ptr(); g.ldr = N
    if "aux" in t: g.aux = t["aux"].data_ptr(); g.ldaux = N; g.act = 2
    if epi == "fc1": g.act = 1; g.out_bf16_pre = t["pre16"].data_ptr()
    if "out_f32" in t: g.out_f32 = t["out_f32"].data_ptr(); g.ldc_f32 = N
    if "out16" in t: g.out_bf16 = t["out16"].data_ptr(); g.ldc_bf16 = N
    return g


def reference(t, epi):
    v = t["A"].float() @ t["Bt"].float().T
    if "bias" in t: v = v + t["bias"]
    if epi == "dgelu":
        a = t["aux"].float(); s = torch.sigmoid(1.702 * a)
        v = v * (s * (1 + 1.702 * a * (1 - s)))
    if "resid" in t: v = v + t["resid"]
    pre = None
    if epi == "fc1":
        pre = v.to(h16).float()
        v = pre * torch.sigmoid(1.702 * pre)
    return v, pre


def rel(a, b):
    return ((a.float() - b).norm() / (b.norm() + 1e-30)).item()


def timeit(fns, iters=40):
    n = len(fns)
    for i in range(6):
        fns[i % n]()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(iters):
        fns[i % n]()
    e1.record(); torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / iters


def set_tile(bm, bn, fit):
    lib.prx_gemm_tile_override(ctx, -7, 0, fit)
    lib.prx_gemm_tile_override(ctx, bm, bn, 1 if bm else 0)


s = _lib.current_stream()
for M, N, K, tag, epi, tile in shapes:
    nset = 1
    if cold:
        per = (M * K + N * K) * 2 + M * N * 10
        nset = max(2, min(24, (400 << 20) // per))
    sets = [make(M, N, K, epi, 100 + i) for i in range(nset)]
    gs = [args_of(t, M, N, K, epi) for t in sets]
    ref, pre = reference(sets[0], epi)
    res = {}
    variants = dict(engine=(0, 0, 0, 0), fit0=(tile[0], tile[1], 1, 0), fit=(tile[0], tile[1], 1, 1),
                    loop=(tile[0], tile[1], 1, 1 + 4), epi=(tile[0], tile[1], 1, 1 + 8))
    for name, (bm, bn, fit, flags) in variants.items():
        lib.prx_gemm_tile_override(ctx, -8, 0, flags)
        set_tile(bm, bn, fit)
        t0 = sets[0]
        for k in ("out_f32", "out16", "pre16"):
            if k in t0: t0[k].fill_(float("nan"))
        call("prx_k_gemm", gs[0], ws, ws.numel(), s)
        torch.cuda.synchronize()
        out = t0["out_f32"] if "out_f32" in t0 else t0["out16"]
        err = rel(out, ref)
        epre = rel(t0["pre16"], pre) if pre is not None else 0.0
        # rerun: bitwise-stable?
        keep = out.clone()
        call("prx_k_gemm", gs[0], ws, ws.numel(), s)
        torch.cuda.synchronize()
        same = bool((keep == out).all().item()) or bool(torch.equal(keep.view(torch.uint8), out.view(torch.uint8)))
        us = timeit([(lambda g=g: call("prx_k_gemm", g, ws, ws.numel(), s)) for g in gs])
        res[name] = (us, err, epre, same)
    set_tile(0, 0, 1)
    Btt = [t["Bt"].t() for t in sets]
    us_lib = timeit([(lambda t=t, b=b: torch.matmul(t["A"], b)) for t, b in zip(sets, Btt)])
    fl = 2.0 * M * N * K
    e, f = res["engine"], res["fit"]
    print(f"{tag:42s} M={M:5d} N={N:5d} K={K:5d} sets={nset:2d}: engine {e[0]:6.1f} us {fl / e[0] / 1e6:5.0f} TF (err {e[1]:.1e}) | "
          f"fit {tile[0]}x{tile[1]} flags 0/1: {res['fit0'][0]:6.1f} {f[0]:6.1f} us {fl / f[0] / 1e6:5.0f} TF "
          f"(err {max(res['fit0'][1], f[1]):.1e} pre {f[2]:.1e} rerun-same {f[3] and res['fit0'][3]}) | "
          f"loop only {res['loop'][0]:5.1f} epilogue only {res['epi'][0]:5.1f} | vendor plain {us_lib:6.1f} us {fl / us_lib / 1e6:5.0f} TF", flush=True)
