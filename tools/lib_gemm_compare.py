"""Diagnostic (run on the GPU box): the engine's row-major bf16 GEMM against the vendor library (torch.matmul -> hipBLASLt /
rocBLAS) on the hot shapes of the headline iteration, hot cache, plain epilogue.  The library is NOT used by the product;
this answers "is the shape or the kernel the limit"."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pixray_amd import _lib
from pixray_amd._lib import GemmArgs, call

dev = "cuda"
ws = torch.empty(256 << 20, dtype=torch.uint8, device=dev)
shapes = [(3200, 3072, 768, "FC1 / W2T dgrad"), (3200, 768, 3072, "FC2 / W1T dgrad"), (3200, 2304, 768, "QKV"),
          (3200, 768, 2304, "WqkvT dgrad"), (3200, 768, 768, "proj"), (25216, 768, 768, "B/16 proj @128"),
          (65792, 1024, 4096, "L/14 FC2 @256"), (8192, 8192, 8192, "large")]


def timeit(fn, iters=30):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


for M, N, K, tag in shapes:
    A = torch.randn(M, K, device=dev).to(torch.bfloat16)
    Bt = (torch.randn(N, K, device=dev) / K ** 0.5).to(torch.bfloat16)
    out = torch.empty(M, N, device=dev)
    out_b = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    g = GemmArgs()
    g.A = A.data_ptr(); g.a_is_f32 = 0; g.a_mode = 0; g.lda = K
    g.B = Bt.data_ptr(); g.ldb = K; g.M, g.N, g.K = M, N, K
    g.alpha = 1.0; g.out_bf16 = out_b.data_ptr(); g.ldc_bf16 = N
    s = _lib.current_stream()
    lib = _lib.load()
    lib.prx_gemm_tile_override(_lib.tool_ctx(), 0, 0, 0)
    t_prx = timeit(lambda: call("prx_k_gemm", g, ws, ws.numel(), s))
    ref_b = out_b.clone()
    # the 256 x 256 8-phase kernel (gemm8p.hip) forced on, with the XCD-aware tile order on and off; result checked against
    # the 4-wave kernels' (same operands, same fp32 accumulation order per K tile up to the MFMA's own)
    t8 = {}
    for xcd in (1, 0):
        lib.prx_gemm_tile_override(_lib.tool_ctx(), 256, 256, 1)
        lib.prx_gemm_tile_override(_lib.tool_ctx(), -1, 0, xcd)
        out_b.zero_()
        t8[xcd] = timeit(lambda: call("prx_k_gemm", g, ws, ws.numel(), s))
        err = (out_b.float() - ref_b.float()).abs().max().item() / max(ref_b.float().abs().max().item(), 1e-30)
    lib.prx_gemm_tile_override(_lib.tool_ctx(), 0, 0, 0)
    lib.prx_gemm_tile_override(_lib.tool_ctx(), -1, 0, 2)
    Btt = Bt.t()
    t_lib = timeit(lambda: torch.matmul(A, Btt))
    fl = 2.0 * M * N * K
    print(f"{tag:18s} M={M:6d} N={N:5d} K={K:5d}: engine {1e3 * t_prx:7.1f} us {fl / t_prx / 1e9:7.0f} TF | 8-phase 256^2 "
          f"{1e3 * t8[1]:7.1f} us {fl / t8[1] / 1e9:7.0f} TF (no XCD order {1e3 * t8[0]:7.1f} us; max rel diff vs engine {err:.1e}) | "
          f"torch.matmul (vendor library) {1e3 * t_lib:7.1f} us {fl / t_lib / 1e9:7.0f} TF", flush=True)
