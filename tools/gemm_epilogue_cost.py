"""Diagnostic (run on the GPU box): what the fused epilogue costs per kernel family, stand-alone and hot.

For the ViT products of BASELINE.json configs[2] / [3] the same C[M,N] = A[M,K] Bt[N,K]^T is timed with the epilogues the towers use
(plain 16-bit store; + bias; + bias, QuickGELU and the saved pre-activation (FC1 forward); x dQuickGELU(aux) (FC2's dgrad);
+ fp32 residual, fp32 and 16-bit outputs (proj / FC2 forward)) on the planner's choice, on the 8-phase kernel forced, and on the
128 x 128 4-wave kernel forced.  profiles/r03_cfg3_gemm_shapes.txt has the K = 1024 products at 528-724 TFLOP/s in the pipeline
against 971-1035 at K >= 3072; this separates "epilogue" from "cold operands" (DESIGN.md section 6).

    python tools/gemm_epilogue_cost.py [fp16|bf16]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pixray_amd import _lib
from pixray_amd._lib import GemmArgs, call

dev = "cuda"
prec = sys.argv[1] if len(sys.argv) > 1 else "fp16"
t16 = torch.float16 if prec == "fp16" else torch.bfloat16
ws = torch.empty(256 << 20, dtype=torch.uint8, device=dev)
lib = _lib.load()
ctx = _lib.tool_ctx()
shapes = [(65792, 4096, 1024, "L/14 FC1"), (65792, 1024, 4096, "L/14 FC2"), (65792, 1024, 1024, "L/14 proj"), (65792, 3072, 1024, "L/14 QKV"),
          (25216, 3072, 768, "B/16 FC1"), (25216, 768, 3072, "B/16 FC2"), (25216, 768, 768, "B/16 proj")]


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


for M, N, K, tag in shapes:
    A = torch.randn(M, K, device=dev).to(t16)
    Bt = (torch.randn(N, K, device=dev) / K ** 0.5).to(t16)
    bias = torch.randn(N, device=dev)
    resid = torch.randn(M, N, device=dev)
    aux = torch.randn(M, N, device=dev).to(t16)
    out_f = torch.empty(M, N, device=dev)
    out_b = torch.empty(M, N, device=dev, dtype=t16)
    out_p = torch.empty(M, N, device=dev, dtype=t16)

    def args(kind):
        g = GemmArgs()
        g.A = A.data_ptr(); g.a_is_f32 = 0; g.a_mode = 0; g.lda = K
        g.B = Bt.data_ptr(); g.ldb = K; g.M, g.N, g.K = M, N, K
        g.alpha = 1.0; g.ldc_bf16 = N; g.ldc_f32 = N; g.ldaux = N; g.ldr = N
        if hasattr(g, "f32"):
            g.f32 = 2 if prec == "fp16" else 0               # include/prx.h PRX_PREC_*
        g.out_bf16 = out_b.data_ptr()
        if kind in ("bias", "gelu"):
            g.bias_n = bias.data_ptr()
        if kind == "gelu":
            g.act = 1; g.out_bf16_pre = out_p.data_ptr()      # PRX_ACT_QUICKGELU + saved pre-activation
        if kind == "dgelu":
            g.act = 2; g.aux = aux.data_ptr()                 # PRX_ACT_MUL_DQUICKGELU
        if kind == "resid":
            g.bias_n = bias.data_ptr(); g.resid = resid.data_ptr(); g.out_f32 = out_f.data_ptr()
        return g

    fl = 2.0 * M * N * K
    row = f"{tag:10s} M={M:6d} N={N:5d} K={K:5d} |"
    for kind in ("plain", "bias", "gelu", "dgelu", "resid"):
        g = args(kind)
        s = _lib.current_stream()
        cell = []
        for name, tile in (("plan", (0, 0)), ("8p", (256, 256)), ("4w", (128, 128))):
            lib.prx_gemm_tile_override(ctx, tile[0], tile[1], 1 if tile[0] else 0)
            t = timeit(lambda: call("prx_k_gemm", g, ws, ws.numel(), s))
            cell.append(f"{name} {1e3 * t:6.0f}us {fl / t / 1e9:5.0f}TF")
        lib.prx_gemm_tile_override(ctx, 0, 0, 0)
        row += f" {kind}: " + ", ".join(cell) + " |"
    print(row, flush=True)
