"""Time both GEMM kernels on the hot-path shapes (run on the GPU box)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pixray_amd import _lib
from pixray_amd._lib import GemmArgs, call

dev = "cuda"
ws = torch.empty(256 << 20, dtype=torch.uint8, device=dev)


def bench(M, N, K, conv=None, iters=20):
    if conv is None:
        A = torch.randn(M, K, device=dev).to(torch.bfloat16)
    else:
        A = torch.randn(M // (4 if conv[3] else 1), conv[2], device=dev).to(torch.bfloat16)
    Bt = torch.randn(N, K, device=dev).to(torch.bfloat16)
    out = torch.empty(M, N, device=dev)
    g = GemmArgs()
    g.A = A.data_ptr(); g.a_is_f32 = 0; g.a_mode = 0 if conv is None else 1
    g.lda = K if conv is None else conv[2]
    g.B = Bt.data_ptr(); g.ldb = K; g.M, g.N, g.K = M, N, K
    if conv is not None:
        g.H, g.W, g.Cin, g.up = conv[0], conv[1], conv[2], conv[3]
    g.alpha = 1.0; g.out_f32 = out.data_ptr(); g.ldc_f32 = N
    s = _lib.current_stream()
    res = []
    for variant in (0, 1):
        _lib.load().prx_gemm_tile_override(_lib.tool_ctx(), -3, 0, variant)
        for _ in range(3):
            call("prx_k_gemm", g, ws, ws.numel(), s)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            call("prx_k_gemm", g, ws, ws.numel(), s)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / iters
        res.append((ms * 1e3, 2.0 * M * N * K / ms / 1e9))
    print(f"M={M:6d} N={N:5d} K={K:5d} conv={str(conv):22s} v1 {res[0][0]:8.1f} us {res[0][1]:7.1f} TF | "
          f"v2 {res[1][0]:8.1f} us {res[1][1]:7.1f} TF", flush=True)


if __name__ == "__main__":
    print(torch.cuda.get_device_name(0))
    for (M, N, K) in [(3200, 2304, 768), (3200, 768, 2304), (3200, 768, 768), (3200, 3072, 768), (3200, 768, 3072),
                      (4096, 4096, 4096), (8192, 8192, 8192)]:
        bench(M, N, K)
    for (H, C, Co, up) in [(16, 512, 512, 0), (32, 512, 512, 1), (64, 256, 256, 0), (128, 256, 256, 1),
                           (128, 128, 128, 0), (256, 128, 128, 0), (256, 128, 128, 1)]:
        bench(H * H, Co, 9 * C, conv=(H, H, C, up))
