"""Hot-cache vs cold-cache timing of the hot GEMM shapes (run on the GPU box): between timed launches either nothing
(hot: operands stay in L2 / Infinity Cache) or a 1 GB streaming write (cold: operands come from HBM, as in the iteration,
where every weight matrix is touched once per step)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pixray_amd import _lib
from pixray_amd._lib import GemmArgs, call
dev = "cuda"
ws = torch.empty(64 << 20, dtype=torch.uint8, device=dev)
flush = torch.empty(1 << 30, dtype=torch.uint8, device=dev)

def run(M, N, K, conv=None, mode="hot", what="both", iters=10):
    A = torch.randn(M if conv is None else M // (4 if conv[3] == 1 else 1), K if conv is None else conv[2], device=dev).to(torch.bfloat16)
    Bt = torch.randn(N, K, device=dev).to(torch.bfloat16)
    out = torch.empty(M, N, device=dev)
    g = GemmArgs()
    g.A = A.data_ptr(); g.a_mode = 0 if conv is None else 1; g.lda = K if conv is None else conv[2]
    g.B = Bt.data_ptr(); g.ldb = K; g.M, g.N, g.K = M, N, K
    if conv is not None: g.H, g.W, g.Cin, g.up = conv
    g.alpha = 1.0; g.out_f32 = out.data_ptr(); g.ldc_f32 = N
    s = _lib.current_stream()
    for _ in range(3): call("prx_k_gemm", g, ws, ws.numel(), s)
    tot = 0.0
    for _ in range(iters):
        if mode == "cold":
            flush.fill_(1)
            if what == "B":          # re-warm A only (activations were just produced in the iteration)
                A.add_(0)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); call("prx_k_gemm", g, ws, ws.numel(), s); e1.record(); torch.cuda.synchronize()
        tot += e0.elapsed_time(e1)
    return tot / iters * 1e3

for (M, N, K, conv) in [(3200, 3072, 768, None), (3200, 768, 3072, None), (3200, 768, 768, None), (3200, 2304, 768, None),
                        (65536, 128, 1152, (256, 256, 128, 0)), (4096, 256, 2304, (64, 64, 256, 0)), (256, 512, 4608, (16, 16, 512, 0))]:
    h = run(M, N, K, conv, "hot"); c = run(M, N, K, conv, "cold"); cb = run(M, N, K, conv, "cold", "B")
    print(f"M={M:6d} N={N:5d} K={K:5d} conv={conv}: hot {h:6.1f} us   cold(A+B) {c:6.1f} us   cold(B only) {cb:6.1f} us", flush=True)
