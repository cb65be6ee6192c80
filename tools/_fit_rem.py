import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pixray_amd import _lib
from pixray_amd._lib import GemmArgs, call
dev = "cuda"; lib = _lib.load(); ctx = _lib.tool_ctx()
h16 = torch.float16
def bench(M, N, K, tile, fit, resid=False):
    A = torch.randn(M, K, device=dev).to(h16); B = (torch.randn(N, K, device=dev) / math.sqrt(K)).to(h16)
    bias = torch.randn(N, device=dev); out = torch.empty(M, N, device=dev, dtype=h16)
    r16 = torch.randn(M, N, device=dev).to(h16)
    g = GemmArgs(); g.A = A.data_ptr(); g.lda = K; g.B = B.data_ptr(); g.ldb = K; g.M, g.N, g.K = M, N, K
    g.alpha = 1.0; g.f32 = 2; g.bias_n = bias.data_ptr(); g.out_bf16 = out.data_ptr(); g.ldc_bf16 = N
    if resid: g.resid = r16.data_ptr(); g.ldr = N; g.row16 = 1
    lib.prx_gemm_tile_override(ctx, -12, 0, 1 if fit else 0)
    if tile: lib.prx_gemm_tile_override(ctx, tile[0], tile[1], 1)
    else: lib.prx_gemm_tile_override(ctx, 0, 0, 0)
    s = _lib.current_stream()
    for _ in range(5): call("prx_k_gemm", g, None, 0, s)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(50): call("prx_k_gemm", g, None, 0, s)
    b.record(); torch.cuda.synchronize()
    lib.prx_gemm_tile_override(ctx, 0, 0, 0); lib.prx_gemm_tile_override(ctx, -12, 0, 0)
    return a.elapsed_time(b) / 50 * 1e3
for (M, N, K) in [(3456, 768, 3072), (3456, 768, 768), (3456, 768, 2304), (3456, 3072, 768), (3456, 2304, 768)]:
    row = [f"{M}x{N}x{K}: default {bench(M, N, K, None, False):6.1f} us"]
    for tile, fit in [((64, 64), False), ((128, 128), False), ((128, 128), True), ((128, 64), True), ((80, 128), True), ((160, 128), True), ((160, 256), True), ((256, 128), True)]:
        try: row.append(f"{'fit' if fit else '4w'}{tile[0]}x{tile[1]} {bench(M, N, K, tile, fit):6.1f}")
        except Exception as e: row.append(f"{'fit' if fit else '4w'}{tile[0]}x{tile[1]} n/a")
    print(" | ".join(row))
