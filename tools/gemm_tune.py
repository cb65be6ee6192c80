"""Sweep tile / split-K choices of the v2 GEMM on the hot-path shapes (run on the GPU box)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pixray_amd import _lib
from pixray_amd._lib import GemmArgs, call
dev = "cuda"
ws = torch.empty(256 << 20, dtype=torch.uint8, device=dev)
lib = _lib.load()

def run(M, N, K, conv, cfgs, iters=20):
    A = torch.randn(M if conv is None else M // (4 if conv[3] else 1), K if conv is None else conv[2], device=dev).to(torch.bfloat16)
    Bt = torch.randn(N, K, device=dev).to(torch.bfloat16)
    out = torch.empty(M, N, device=dev)
    g = GemmArgs()
    g.A = A.data_ptr(); g.a_mode = 0 if conv is None else 1; g.lda = K if conv is None else conv[2]
    g.B = Bt.data_ptr(); g.ldb = K; g.M, g.N, g.K = M, N, K
    if conv is not None: g.H, g.W, g.Cin, g.up = conv
    g.alpha = 1.0; g.out_f32 = out.data_ptr(); g.ldc_f32 = N
    s = _lib.current_stream()
    res = []
    for (bm, bn, sp) in cfgs:
        lib.prx_gemm_tile_override(_lib.tool_ctx(), bm, bn, sp)
        for _ in range(3): call("prx_k_gemm", g, ws, ws.numel(), s)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters): call("prx_k_gemm", g, ws, ws.numel(), s)
        e1.record(); torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) / iters * 1e3)
    lib.prx_gemm_tile_override(_lib.tool_ctx(), 0, 0, 0)
    best = min(range(len(res)), key=lambda i: res[i])
    print(f"M={M:6d} N={N:5d} K={K:5d} conv={str(conv):20s} " + "  ".join(f"{c}:{t:6.1f}" for c, t in zip(cfgs, res)) + f"   best {cfgs[best]} {2.0*M*N*K/res[best]/1e6:.0f} TF", flush=True)

import sys as _s
if len(_s.argv) > 1 and _s.argv[1] == "xcd":
    for sw in (0, 1):
        lib.prx_gemm_tile_override(_lib.tool_ctx(), -1, 0, sw)
        print("== xcd swizzle", sw)
        for (M, N, K) in [(3200, 2304, 768), (3200, 768, 2304), (3200, 768, 768), (3200, 3072, 768), (3200, 768, 3072), (4096, 4096, 4096), (8192, 8192, 8192)]:
            run(M, N, K, None, [(0, 0, 0)])
        for (H, C, Co, up) in [(64, 256, 256, 0), (128, 256, 256, 1), (128, 128, 128, 0), (256, 128, 128, 0), (256, 128, 128, 1)]:
            run(H * H, Co, 9 * C, (H, H, C, up), [(0, 0, 0)])
    lib.prx_gemm_tile_override(_lib.tool_ctx(), -1, 0, 1)
    _s.exit(0)
if len(_s.argv) > 1 and _s.argv[1] == "deep":
    # LDS ring depth x tile on the M = 3200 products of the headline's ViT: with <= 256 tiles a launch is one workgroup per CU
    # whatever the LDS footprint, so the "3 stages halve the occupancy" argument against deep rings on 128-wide tiles is void
    for st in (0, 2, 3, 4):
        lib.prx_gemm_tile_override(_lib.tool_ctx(), -2, 0, st)
        print("== stages", st, "(0 = heuristic)")
        cf = [(0, 0, 0), (128, 128, 1), (128, 64, 1), (64, 64, 1), (256, 128, 1)]
        for (M, N, K) in [(3200, 3072, 768), (3200, 768, 3072), (3200, 2304, 768), (3200, 768, 2304), (3200, 768, 768)]:
            run(M, N, K, None, cf)
    lib.prx_gemm_tile_override(_lib.tool_ctx(), -2, 0, 0)
    _s.exit(0)
if len(_s.argv) > 1 and _s.argv[1] == "dbg":
    lib.prx_gemm_tile_override(_lib.tool_ctx(), -3, 0, 0)
    for dbg in (0, 1, 2, 3):
        lib.prx_gemm_tile_override(_lib.tool_ctx(), -4, 0, dbg)
        print("== dbg", dbg, "(bit0: no DMA issue, bit1: no LDS-read/MFMA)")
        cf = [(128, 128, 1), (128, 64, 1), (64, 64, 1)]
        for (M, N, K) in [(3200, 768, 2304), (3200, 768, 3072), (3200, 3072, 768), (8192, 8192, 8192)]:
            run(M, N, K, None, cf)
        for (H, C, Co, up) in [(64, 256, 256, 0), (128, 256, 256, 1), (256, 128, 128, 0)]:
            run(H * H, Co, 9 * C, (H, H, C, up), cf)
    lib.prx_gemm_tile_override(_lib.tool_ctx(), -4, 0, 0); lib.prx_gemm_tile_override(_lib.tool_ctx(), -3, 0, 1)
    _s.exit(0)
if len(_s.argv) > 1 and _s.argv[1] == "wsplit":
    for wsp in (0, 1):
        for st in (2, 3):
            lib.prx_gemm_tile_override(_lib.tool_ctx(), -3, 0, wsp); lib.prx_gemm_tile_override(_lib.tool_ctx(), -2, 0, st)
            print("== wsplit", wsp, "stages", st)
            cf = [(0, 0, 0), (64, 64, 1)]
            for (M, N, K) in [(3200, 768, 2304), (3200, 768, 768), (3200, 768, 3072), (3200, 2304, 768), (3200, 3072, 768)]:
                run(M, N, K, None, cf)
            for (H, C, Co, up) in [(16, 512, 512, 0), (32, 512, 512, 1), (32, 256, 256, 0), (64, 256, 256, 0), (128, 256, 256, 1), (128, 128, 128, 0), (256, 128, 128, 0)]:
                run(H * H, Co, 9 * C, (H, H, C, up), cf + [(64, 64, 4)])
            for (M, N, K) in [(256, 1536, 512), (1024, 256, 512), (16384, 128, 256)]:
                run(M, N, K, None, cf)
    lib.prx_gemm_tile_override(_lib.tool_ctx(), -2, 0, 0); lib.prx_gemm_tile_override(_lib.tool_ctx(), -3, 0, 1)
    _s.exit(0)
if len(_s.argv) > 1 and _s.argv[1] == "c64":
    for c64 in (0, 1):
        for st in (2, 3):
            lib.prx_gemm_tile_override(_lib.tool_ctx(), -5, 0, c64); lib.prx_gemm_tile_override(_lib.tool_ctx(), -2, 0, st)
            print("== conv scalar-tap gather", c64, "stages", st)
            cf = [(0, 0, 0), (128, 128, 1), (128, 64, 1), (64, 64, 1), (64, 64, 4), (128, 64, 4), (128, 128, 4)]
            for (H, C, Co, up) in [(16, 512, 512, 0), (32, 512, 512, 1), (32, 256, 256, 0), (64, 256, 256, 0), (128, 256, 256, 1), (128, 128, 128, 0), (256, 128, 128, 0), (256, 128, 128, 1)]:
                run(H * H, Co, 9 * C, (H, H, C, up), cf)
    lib.prx_gemm_tile_override(_lib.tool_ctx(), -2, 0, 0); lib.prx_gemm_tile_override(_lib.tool_ctx(), -5, 0, 1)
    _s.exit(0)
if len(_s.argv) > 1 and _s.argv[1] == "big":
    for st in (2, 3):
        lib.prx_gemm_tile_override(_lib.tool_ctx(), -2, 0, st)
        print("== 256x128 tiles, stages", st)
        cf = [(0, 0, 0), (256, 128, 1), (256, 128, 2), (256, 128, 3), (256, 128, 4), (128, 128, 2), (128, 128, 1), (64, 64, 1)]
        for (M, N, K) in [(3200, 768, 3072), (3200, 768, 2304), (3200, 768, 768), (3200, 3072, 768), (3200, 2304, 768), (4096, 4096, 4096), (8192, 8192, 8192)]:
            run(M, N, K, None, cf)
        for (H, C, Co, up) in [(64, 256, 256, 0), (128, 256, 256, 1), (128, 128, 128, 0), (256, 128, 128, 0), (256, 128, 128, 1)]:
            run(H * H, Co, 9 * C, (H, H, C, up), cf)
    lib.prx_gemm_tile_override(_lib.tool_ctx(), -2, 0, 0)
    _s.exit(0)
if len(_s.argv) > 1 and _s.argv[1] == "stages":
    for st in (2, 3, 4):
        lib.prx_gemm_tile_override(_lib.tool_ctx(), -2, 0, st)
        print("== stages", st)
        cf = [(0, 0, 0), (128, 128, 1), (128, 64, 1), (64, 64, 1)]
        for (M, N, K) in [(3200, 2304, 768), (3200, 768, 2304), (3200, 768, 768), (3200, 3072, 768), (3200, 768, 3072), (4096, 4096, 4096), (8192, 8192, 8192)]:
            run(M, N, K, None, cf)
        for (H, C, Co, up) in [(16, 512, 512, 0), (32, 512, 512, 1), (32, 256, 256, 0), (64, 256, 256, 0), (128, 256, 256, 1), (128, 128, 128, 0), (256, 128, 128, 0), (256, 128, 128, 1)]:
            run(H * H, Co, 9 * C, (H, H, C, up), cf + [(64, 64, 4), (128, 64, 4)])
        for (M, N, K) in [(256, 1536, 512), (1024, 256, 512), (16384, 128, 256)]:
            run(M, N, K, None, cf)
    lib.prx_gemm_tile_override(_lib.tool_ctx(), -2, 0, 0)
    _s.exit(0)
cfgs = [(0, 0, 0), (128, 128, 1), (128, 128, 2), (128, 64, 1), (128, 64, 2), (64, 64, 1)]
for (M, N, K) in [(3200, 2304, 768), (3200, 768, 2304), (3200, 768, 768), (3200, 3072, 768), (3200, 768, 3072), (3200, 3072, 768)]:
    run(M, N, K, None, cfgs)
ccfgs = [(0, 0, 0), (128, 128, 1), (128, 64, 1), (64, 64, 1), (64, 64, 4), (64, 64, 8), (64, 64, 16), (128, 64, 4), (128, 128, 4), (128,128,8)]
for (H, C, Co, up) in [(16, 512, 512, 0), (16, 256, 512, 0), (32, 512, 512, 1), (32, 256, 256, 0), (32, 512, 256, 0), (64, 256, 256, 0), (128, 256, 256, 1), (128, 128, 128, 0), (128, 256, 128, 0)]:
    run(H * H, Co, 9 * C, (H, H, C, up), ccfgs)
for (M, N, K) in [(256, 1536, 512), (256, 256, 512), (256, 512, 256), (256, 512, 512), (1024, 256, 512), (16384, 128, 256)]:
    run(M, N, K, None, ccfgs)
