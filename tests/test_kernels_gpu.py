"""Kernel-level parity tests (GPU): every HIP kernel alone, through the C ABI, against a plain
PyTorch fp32 reference of the same op evaluated on the same (bf16-rounded where the kernel
consumes bf16) inputs.  Tolerances are written next to each check."""
import math
import os

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from pixray_amd import _lib
from pixray_amd._lib import GemmArgs, call

DEV = "cuda"


def stream():
    return _lib.current_stream()


@pytest.fixture(params=[1, 0], ids=["glds_v2", "regstage_v1"])
def gemm_variant(request):
    """both GEMM kernels: direct-to-LDS (default) and register-staged"""
    _lib.load().prx_gemm_tile_override(_lib.tool_ctx(), -3, 0, request.param)
    yield request.param
    _lib.load().prx_gemm_tile_override(_lib.tool_ctx(), -3, 0, 1)


def bf(x):
    return x.to(torch.bfloat16)


def rel_l2(a, b):
    return ((a.float() - b.float()).norm() / (b.float().norm() + 1e-30)).item()


def run_gemm(A, Bt, M, N, K, *, a_mode=0, lda=None, H=0, W=0, Cin=0, up=0, alpha=1.0, bias_n=None, bias_m=None,
             aux=None, resid=None, act=0, want_f32=True, want_bf16=False, want_pre=False, ws=None):
    g = GemmArgs()
    g.A = A.data_ptr(); g.a_is_f32 = int(A.dtype == torch.float32); g.a_mode = a_mode
    g.lda = lda if lda is not None else A.shape[-1]
    g.B = Bt.data_ptr(); g.ldb = Bt.shape[-1]
    g.M, g.N, g.K = M, N, K
    g.H, g.W, g.Cin, g.up = H, W, Cin, up
    g.alpha = alpha
    g.bias_n = bias_n.data_ptr() if bias_n is not None else None
    g.bias_m = bias_m.data_ptr() if bias_m is not None else None
    g.aux = aux.data_ptr() if aux is not None else None
    g.ldaux = N
    g.resid = resid.data_ptr() if resid is not None else None
    g.ldr = N
    g.act = act
    out_f32 = torch.full((M, N), float("nan"), device=DEV) if want_f32 else None
    out_bf16 = torch.full((M, N), float("nan"), device=DEV, dtype=torch.bfloat16) if want_bf16 else None
    out_pre = torch.full((M, N), float("nan"), device=DEV, dtype=torch.bfloat16) if want_pre else None
    g.out_f32 = out_f32.data_ptr() if want_f32 else None
    g.ldc_f32 = N
    g.out_bf16 = out_bf16.data_ptr() if want_bf16 else None
    g.out_bf16_pre = out_pre.data_ptr() if want_pre else None
    g.ldc_bf16 = N
    if ws is None:
        ws = torch.empty(64 << 20, dtype=torch.uint8, device=DEV)
    call("prx_k_gemm", g, ws, ws.numel(), stream())
    torch.cuda.synchronize()
    return out_f32, out_bf16, out_pre


@pytest.mark.parametrize("M,N,K", [
    (128, 128, 64), (64, 64, 64), (256, 512, 4608),   # exercises split-K
    (3200, 768, 768), (3200, 2304, 768), (3136, 768, 3072), (3200, 3072, 768), (64, 512, 768),
    (65536, 128, 1152), (100, 72, 136),               # ragged M/N, K not a multiple of 64
])
def test_gemm_rowmajor_bf16(M, N, K, gemm_variant):
    torch.manual_seed(M + N + K)
    A = bf(torch.randn(M, K, device=DEV))
    # asymmetric B (rule: transpose-detecting)
    Bt = bf(torch.randn(N, K, device=DEV) * torch.linspace(0.5, 1.5, N, device=DEV)[:, None])
    out, _, _ = run_gemm(A, Bt, M, N, K)
    ref = A.float() @ Bt.float().T
    # fp32 accumulate of exact bf16 products; only summation order differs -> 1e-5 relative
    assert rel_l2(out, ref) < 2e-5, rel_l2(out, ref)
    assert torch.isfinite(out).all()


def test_gemm_f32_A_and_epilogues(gemm_variant):
    torch.manual_seed(0)
    M, N, K = 3200, 768, 3072
    A32 = torch.randn(M, K, device=DEV)
    Bt = bf(torch.randn(N, K, device=DEV) / math.sqrt(K))
    bias = torch.randn(N, device=DEV)
    resid = torch.randn(M, N, device=DEV)
    out, ob, _ = run_gemm(A32, Bt, M, N, K, bias_n=bias, resid=resid, want_bf16=True)
    ref = bf(A32).float() @ Bt.float().T + bias + resid
    assert rel_l2(out, ref) < 2e-5
    assert rel_l2(ob, ref) < 4e-3  # bf16 output rounding (2^-9 relative per element)
    # QuickGELU forward epilogue (+ pre-activation copy) and its backward multiplier
    out, ob, pre = run_gemm(bf(A32), Bt, M, N, K, bias_n=bias, act=1, want_bf16=True, want_pre=True)
    t = bf(A32).float() @ Bt.float().T + bias
    assert rel_l2(pre, t) < 4e-3
    tq = pre.float()
    assert rel_l2(out, tq * torch.sigmoid(1.702 * tq)) < 1e-5
    aux = pre
    out2, _, _ = run_gemm(bf(A32), Bt, M, N, K, act=2, aux=aux)
    s = torch.sigmoid(1.702 * tq)
    ref2 = (bf(A32).float() @ Bt.float().T) * (s * (1 + 1.702 * tq * (1 - s)))
    assert rel_l2(out2, ref2) < 2e-5
    # bias_m + alpha
    bm = torch.randn(M, device=DEV)
    out3, _, _ = run_gemm(bf(A32), Bt, M, N, K, bias_m=bm, alpha=0.5)
    assert rel_l2(out3, 0.5 * (bf(A32).float() @ Bt.float().T) + bm[:, None]) < 2e-5


@pytest.mark.parametrize("prec", ["bf16", "fp16"])
def test_gemm_8phase_kernel_ragged_edges_and_epilogues(prec):
    """gemm8p_kernel (256 x 256 tiles, 8 waves, 8-phase main loop) forced on shapes whose M and N are NOT multiples of 256 (the
    source-side clamp of rows >= M / columns >= N), K = 128 (a single loop body, the drain path only) up to K = 1536, both 16-bit
    formats, with the fused epilogue (bias + residual, fp32 + 16-bit outputs; bias + QuickGELU with the pre-activation copy);
    against an fp32 product of the same rounded operands"""
    lib = _lib.load()
    dt = torch.bfloat16 if prec == "bf16" else torch.float16
    tol16 = 4e-3 if prec == "bf16" else 5e-4
    torch.manual_seed(23)
    try:
        lib.prx_gemm_tile_override(_lib.tool_ctx(), 256, 256, 1)
        for (M, N, K) in [(300, 520, 256), (257, 255 + 1, 128), (1000, 264, 1536), (513, 768, 384)]:
            A = torch.randn(M, K, device=DEV).to(dt)
            Bt = (torch.randn(N, K, device=DEV) * torch.linspace(0.5, 1.5, N, device=DEV)[:, None] / math.sqrt(K)).to(dt)
            bias = torch.randn(N, device=DEV)
            resid = torch.randn(M, N, device=DEV)
            prod = A.float() @ Bt.float().T
            g = GemmArgs()
            g.A = A.data_ptr(); g.lda = K; g.B = Bt.data_ptr(); g.ldb = K; g.M, g.N, g.K = M, N, K
            g.alpha = 1.0; g.f32 = 0 if prec == "bf16" else 2
            g.bias_n = bias.data_ptr(); g.resid = resid.data_ptr(); g.ldr = N
            out = torch.full((M, N), float("nan"), device=DEV); g.out_f32 = out.data_ptr(); g.ldc_f32 = N
            o16 = torch.full((M, N), float("nan"), device=DEV, dtype=dt); g.out_bf16 = o16.data_ptr(); g.ldc_bf16 = N
            call("prx_k_gemm", g, None, 0, stream())
            torch.cuda.synchronize()
            ref = prod + bias + resid
            assert rel_l2(out, ref) < 2e-5, (M, N, K, rel_l2(out, ref))
            assert rel_l2(o16, ref) < tol16 and not torch.isnan(o16.float()).any()
            g2 = GemmArgs()
            g2.A = A.data_ptr(); g2.lda = K; g2.B = Bt.data_ptr(); g2.ldb = K; g2.M, g2.N, g2.K = M, N, K
            g2.alpha = 1.0; g2.f32 = g.f32; g2.bias_n = bias.data_ptr(); g2.act = 1
            u = torch.full((M, N), float("nan"), device=DEV, dtype=dt); g2.out_bf16 = u.data_ptr()
            t = torch.full((M, N), float("nan"), device=DEV, dtype=dt); g2.out_bf16_pre = t.data_ptr(); g2.ldc_bf16 = N
            call("prx_k_gemm", g2, None, 0, stream())
            torch.cuda.synchronize()
            pre = (prod + bias).to(dt).float()
            assert rel_l2(t, pre) < tol16 and rel_l2(u, pre * torch.sigmoid(1.702 * pre)) < tol16
    finally:
        lib.prx_gemm_tile_override(_lib.tool_ctx(), 0, 0, 0)


def row_kernel_checks(cases=None):
    """gemmrow_kernel (gemmrow_kernel.h: weights of a column slab resident in LDS, 16-row wave tiles streamed through registers, the
    MFMA transposed so that a lane owns 8 consecutive output columns) on the shapes of the ModifiedResNet runner's 1x1
    convolutions, scaled down in M: ragged M (clamped loads, masked stores), K = 64 .. 320 including K % 32 != 0 (zero-filled K
    tail) and both K ranges (next-tile prefetch / none), 160-, 128- and 80-column slabs (the last with a lone 16-column tile), every
    epilogue of the runner -- conv3 forward (bias + 16-bit identity + ReLU), downsample forward (bias only), conv1 dgrad (16-bit
    identity gradient + ReLU mask of the block below), block 0's fp32 output; on N = 80: conv1 forward (bias + ReLU), conv3 dgrad
    (ReLU mask of the saved activation), its unmasked fp32 form; in bf16 the fp32 identity with both outputs.  Against the TILED
    kernels on the same descriptor (forced 128 x 128: the engine's previous path; same epilogue arithmetic, another summation
    order) and an fp32 product; the launch counter proves which ran."""
    lib = _lib.load()
    torch.manual_seed(31)
    lib.prx_gemm_tile_override(_lib.tool_ctx(), -14, 0, 1 << 20)       # the engine's own rule starts at 24 Mi output elements
    try:
        _row_kernel_cases(lib, cases)
    finally:
        lib.prx_gemm_tile_override(_lib.tool_ctx(), -14, 0, 0)


def _row_kernel_cases(lib, cases):
    cases = cases or [("fp16", 16405, 320, 80), ("fp16", 8200, 640, 160), ("fp16", 13700, 384, 96), ("fp16", 8192 + 7, 640, 192),
                      ("fp16", 4200, 1280, 320), ("fp16", 30003, 80, 320), ("fp16", 26000, 80, 80), ("fp16", 12000, 240, 200),
                      ("bf16", 20483, 256, 64), ("bf16", 16400, 320, 80), ("bf16", 8200, 640, 320), ("bf16", 26003, 80, 320),
                      ("fp16", 8200, 160, 640), ("fp16", 1100, 2560, 640), ("bf16", 4100, 320, 520)]      # K in (320, 640]: 80-column slabs
    RELU, MRM, RMP = 3, 4, 5          # include/prx.h PRX_ACT_RELU, PRX_ACT_MUL_RELUMASK, PRX_ACT_RELUMASK_POST
    for prec, M, N, K in cases:
        dt = torch.float16 if prec == "fp16" else torch.bfloat16
        tol16 = 5e-4 if prec == "fp16" else 4e-3
        A = torch.randn(M, K, device=DEV).to(dt)
        Bt = (torch.randn(N, K, device=DEV) * torch.linspace(0.5, 1.5, N, device=DEV)[:, None] / math.sqrt(K)).to(dt)
        bias = torch.randn(N, device=DEV)
        res32 = torch.randn(M, N, device=DEV)
        res = res32.to(dt) if prec == "fp16" else res32
        mask = torch.randn(M, N, device=DEV).to(dt)
        prod = A.float() @ Bt.float().T

        def run(act, use_bias, use_res, want32, want16, forced):
            g = GemmArgs()
            g.A = A.data_ptr(); g.lda = K; g.B = Bt.data_ptr(); g.ldb = K; g.M, g.N, g.K = M, N, K
            g.alpha = 1.0; g.f32 = 2 if prec == "fp16" else 0; g.act = act
            if use_bias: g.bias_n = bias.data_ptr()
            if use_res: g.resid = res.data_ptr(); g.ldr = N; g.row16 = 1 if prec == "fp16" else 0
            if act in (MRM, RMP): g.aux = mask.data_ptr(); g.ldaux = N
            o32 = torch.full((M, N), float("nan"), device=DEV) if want32 else None
            o16 = torch.full((M + 1, N), float("nan"), device=DEV, dtype=dt) if want16 else None      # + a guard row
            if want32: g.out_f32 = o32.data_ptr(); g.ldc_f32 = N
            if want16: g.out_bf16 = o16.data_ptr(); g.ldc_bf16 = N
            try:
                if forced: lib.prx_gemm_tile_override(_lib.tool_ctx(), 128, 128, 1)
                n0 = lib.prx_gemm_row_launches()
                call("prx_k_gemm", g, None, 0, stream())
                torch.cuda.synchronize()
                assert lib.prx_gemm_row_launches() - n0 == (0 if forced else 1), (prec, M, N, K, act, forced)
            finally:
                if forced: lib.prx_gemm_tile_override(_lib.tool_ctx(), 0, 0, 0)
            if want16:
                assert torch.isnan(o16[M].float()).all()        # nothing written past row M - 1
                o16 = o16[:M]
            return o32, o16

        rr = res.float()
        if N % 160 != 0 and N % 128 != 0:       # 80-column slabs: conv1 forward, conv3 dgrad and its unmasked fp32 form
            patterns = [(RELU, True, False, False, True, torch.relu(prod + bias)),
                        (MRM, False, False, prec == "bf16", True, prod * (mask.float() > 0)),
                        (0, False, False, True, False, prod),
                        (0, False, True, True, False, prod + rr)]                                             # conv1 dgrad of the first block
        else:
            patterns = [(RELU, True, True, prec == "bf16", True, torch.relu(prod + bias + rr)),               # conv3 forward
                        (0, True, False, False, True, prod + bias),                                           # downsample forward
                        (RMP, False, True, prec == "bf16", True, (prod + rr) * (mask.float() > 0)),           # conv1 dgrad
                        (0, False, True, True, False, prod + rr)]                                             # ... of block 0
        for act, use_bias, use_res, want32, want16, ref in patterns:
            o32, o16 = run(act, use_bias, use_res, want32, want16, False)
            t32, t16 = run(act, use_bias, use_res, want32, want16, True)
            if want32:
                assert rel_l2(o32, ref) < 2e-5, (prec, M, N, K, act, rel_l2(o32, ref))
                assert rel_l2(o32, t32) < 2e-6 and (o32 - t32).abs().max() <= 2e-5 * ref.abs().max()
            if want16:
                assert rel_l2(o16, ref) < tol16 and not torch.isnan(o16.float()).any(), (prec, M, N, K, act, rel_l2(o16, ref))
                # same epilogue arithmetic on sums that differ in their last fp32 bits: a 16-bit rounding boundary is crossed rarely
                assert (o16 != t16).float().mean().item() < 2e-3, (prec, M, N, K, act)
                assert ((o16.float() > 0) == (t16.float() > 0)).float().mean().item() > 0.9999


def test_gemm_row_streaming_kernel_vs_tiled_kernels():
    row_kernel_checks()


def row_conv_checks(cases=None):
    """gemmrowconv_kernel (gemmrowconv_kernel.h: all weights resident in LDS, 16-pixel wave tiles whose taps are loaded straight into
    MFMA operand registers) on the ModifiedResNet runner's small-channel 3x3 convolutions, scaled down: Cin, N in {40, 80} and Cin = N =
    160 (40-column slabs end in half a tile), image borders / several images / a ragged last tile, bias + ReLU (forward) and the ReLU mask of
    a saved activation (dgrad), both formats; against F.conv2d on the same rounded operands and the tiled kernels (forced 128 x
    128) on the same descriptor"""
    lib = _lib.load()
    torch.manual_seed(37)
    cases = cases or [("fp16", 2, 70, 90, 80, 80), ("fp16", 3, 50, 62, 40, 40), ("fp16", 1, 96, 100, 40, 80), ("fp16", 2, 64, 66, 80, 40),
                      ("bf16", 2, 50, 50, 80, 80), ("bf16", 1, 72, 130, 40, 80),
                      ("fp16", 2, 72, 75, 160, 160), ("bf16", 1, 60, 61, 160, 160)]       # four 40-column slabs, the ring K loop
    lib.prx_gemm_tile_override(_lib.tool_ctx(), -14, 0, 1 << 16)
    try:
        for prec, NB, H, W, Cin, Cout in cases:
            dt = torch.float16 if prec == "fp16" else torch.bfloat16
            tol16 = 5e-4 if prec == "fp16" else 4e-3
            M, K = NB * H * W, 9 * Cin
            x = torch.randn(NB, H, W, Cin, device=DEV).to(dt)
            w = (torch.randn(Cout, Cin, 3, 3, device=DEV) / math.sqrt(K)).to(dt)
            Bt = w.permute(0, 2, 3, 1).reshape(Cout, K).contiguous()              # [co][tap * Cin + ci]
            bias = torch.randn(Cout, device=DEV)
            mask = torch.randn(M, Cout, device=DEV).to(dt)
            conv = F.conv2d(x.float().permute(0, 3, 1, 2), w.float(), padding=1).permute(0, 2, 3, 1).reshape(M, Cout)
            for act, ref in [(3, torch.relu(conv + bias)), (4, conv * (mask.float() > 0))]:
                outs = []
                for forced in (False, True):
                    g = GemmArgs()
                    g.A = x.data_ptr(); g.a_mode = 1; g.lda = Cin; g.B = Bt.data_ptr(); g.ldb = K; g.M, g.N, g.K = M, Cout, K
                    g.H, g.W, g.Cin = H, W, Cin
                    g.alpha = 1.0; g.f32 = 2 if prec == "fp16" else 0; g.act = act
                    if act == 3: g.bias_n = bias.data_ptr()
                    else: g.aux = mask.data_ptr(); g.ldaux = Cout
                    o16 = torch.full((M + 1, Cout), float("nan"), device=DEV, dtype=dt)
                    g.out_bf16 = o16.data_ptr(); g.ldc_bf16 = Cout
                    try:
                        if forced: lib.prx_gemm_tile_override(_lib.tool_ctx(), 128, 128, 1)
                        n0 = lib.prx_gemm_row_launches()
                        call("prx_k_gemm", g, None, 0, stream())
                        torch.cuda.synchronize()
                        assert lib.prx_gemm_row_launches() - n0 == (0 if forced else 1), (prec, NB, H, W, Cin, Cout, act, forced)
                    finally:
                        if forced: lib.prx_gemm_tile_override(_lib.tool_ctx(), 0, 0, 0)
                    assert torch.isnan(o16[M].float()).all()
                    outs.append(o16[:M])
                o16, t16 = outs
                assert rel_l2(o16, ref) < tol16 and not torch.isnan(o16.float()).any(), (prec, NB, H, W, Cin, Cout, act, rel_l2(o16, ref))
                assert (o16 != t16).float().mean().item() < 2e-3
    finally:
        lib.prx_gemm_tile_override(_lib.tool_ctx(), -14, 0, 0)


def test_gemm_row_streaming_conv3x3_vs_conv2d_and_tiled_kernels():
    row_conv_checks()


@pytest.mark.parametrize("tile", [(256, 128), (128, 128), (128, 64), (64, 64)], ids=lambda t: f"{t[0]}x{t[1]}")
@pytest.mark.parametrize("stages", [2, 3, 4])
@pytest.mark.parametrize("splits", [1, 3])
def test_gemm_forced_tiles_stages_splitk(tile, stages, splits):
    """every tile shape x LDS pipeline depth x split-K of the direct-to-LDS kernel (normally picked by the heuristic),
    on ragged row-major shapes with the fused epilogue, and on an implicit conv that uses the scalar-tap gather"""
    lib = _lib.load()
    torch.manual_seed(7)
    try:
        lib.prx_gemm_tile_override(_lib.tool_ctx(), tile[0], tile[1], splits)
        lib.prx_gemm_tile_override(_lib.tool_ctx(), -2, 0, stages)
        for (M, N, K) in [(3200, 768, 768), (1000, 200, 1096), (257, 136, 64)]:
            A = bf(torch.randn(M, K, device=DEV))
            Bt = bf(torch.randn(N, K, device=DEV) * torch.linspace(0.5, 1.5, N, device=DEV)[:, None] / math.sqrt(K))
            bias = torch.randn(N, device=DEV)
            resid = torch.randn(M, N, device=DEV)
            out, ob, _ = run_gemm(A, Bt, M, N, K, bias_n=bias, resid=resid, want_bf16=True)
            ref = A.float() @ Bt.float().T + bias + resid
            assert rel_l2(out, ref) < 2e-5, (M, N, K, rel_l2(out, ref))
            assert rel_l2(ob, ref) < 4e-3
        for (H, W, Cin, Cout, up, NB) in [(32, 32, 128, 128, 1, 1), (24, 40, 64, 72, 0, 2), (16, 16, 40, 64, 0, 1)]:
            hin, win = (H // 2, W // 2) if up else (H, W)
            x = torch.randn(NB, Cin, hin, win, device=DEV)
            w = torch.randn(Cout, Cin, 3, 3, device=DEV) / math.sqrt(9 * Cin)
            bias = torch.randn(Cout, device=DEV)
            x_nhwc = bf(x.permute(0, 2, 3, 1).contiguous())
            w_pack = bf(w.permute(0, 2, 3, 1).reshape(Cout, 9 * Cin).contiguous())
            out, _, _ = run_gemm(x_nhwc, w_pack, NB * H * W, Cout, 9 * Cin, a_mode=1, lda=Cin, H=H, W=W, Cin=Cin, up=up,
                                 bias_n=bias)
            xr = bf(x).float()
            if up:
                xr = F.interpolate(xr, scale_factor=2.0, mode="nearest")
            ref = F.conv2d(xr, bf(w).float(), bias, padding=1).permute(0, 2, 3, 1).reshape(NB * H * W, Cout)
            assert rel_l2(out, ref) < 2e-5, (H, W, Cin, Cout, up, rel_l2(out, ref))
            # fp32 A converted on load: a register-staged kernel (the 256 x 128 tile has none and is re-planned to 128 x 128)
            out32, _, _ = run_gemm(x.permute(0, 2, 3, 1).contiguous(), w_pack, NB * H * W, Cout, 9 * Cin, a_mode=1, lda=Cin, H=H,
                                   W=W, Cin=Cin, up=up, bias_n=bias)
            assert rel_l2(out32, ref) < 2e-5, (H, W, Cin, Cout, up, rel_l2(out32, ref))
    finally:
        lib.prx_gemm_tile_override(_lib.tool_ctx(), 0, 0, 0)
        lib.prx_gemm_tile_override(_lib.tool_ctx(), -2, 0, 0)


@pytest.mark.parametrize("tile", [(160, 256), (160, 128), (160, 192), (256, 128), (128, 128), (80, 128), (128, 64), (64, 64), (32, 64), (16, 64), (16, 32), (256, 16)])
@pytest.mark.parametrize("prec", ["bf16", "fp16"])
@pytest.mark.parametrize("stagger", [1, 0])
def test_gemm_fit_tiles(tile, prec, stagger):
    """the fit-tile kernel (gemmfit.hip: every tile shape, 1 / 2 / 4 / 8 K groups) forced on ragged and exact shapes, every
    epilogue form the ViT tower uses (bias + QuickGELU with both 16-bit outputs, dQuickGELU(aux), bias + residual -> fp32,
    plain 16-bit), both 16-bit formats, staggered wave groups on and off; vs an fp32 product of the same rounded operands"""
    lib = _lib.load()
    dt = torch.bfloat16 if prec == "bf16" else torch.float16
    tol16 = 4e-3 if prec == "bf16" else 5e-4
    torch.manual_seed(11)
    try:
        lib.prx_gemm_tile_override(_lib.tool_ctx(), -12, 0, 1)        # the shapes both families have: the fit kernel
        lib.prx_gemm_tile_override(_lib.tool_ctx(), tile[0], tile[1], 1)
        lib.prx_gemm_tile_override(_lib.tool_ctx(), -8, 0, stagger)
        for (M, N, K) in [(3200, 768, 768), (1000, 200, 1152), (333, 520, 512), (81, 136, 1024), (160, 256, 1536)]:
            A = torch.randn(M, K, device=DEV).to(dt)
            Bt = (torch.randn(N, K, device=DEV) * torch.linspace(0.5, 1.5, N, device=DEV)[:, None] / math.sqrt(K)).to(dt)
            bias = torch.randn(N, device=DEV)
            resid = torch.randn(M, N, device=DEV)
            aux = torch.randn(M, N, device=DEV).to(dt)
            prod = A.float() @ Bt.float().T

            def run(**kw):
                g = GemmArgs()
                g.A = A.data_ptr(); g.a_is_f32 = 0; g.a_mode = 0; g.lda = K
                g.B = Bt.data_ptr(); g.ldb = K; g.M, g.N, g.K = M, N, K
                g.alpha = kw.get("alpha", 1.0); g.f32 = 0 if prec == "bf16" else 2
                outs = {}
                if kw.get("bias"): g.bias_n = bias.data_ptr()
                if kw.get("resid"): g.resid = resid.data_ptr(); g.ldr = N
                if kw.get("aux"): g.aux = aux.data_ptr(); g.ldaux = N
                g.act = kw.get("act", 0)
                if kw.get("f32"):
                    outs["f32"] = torch.full((M, N), float("nan"), device=DEV); g.out_f32 = outs["f32"].data_ptr(); g.ldc_f32 = N
                if kw.get("o16"):
                    outs["o16"] = torch.full((M, N), float("nan"), device=DEV, dtype=dt); g.out_bf16 = outs["o16"].data_ptr()
                if kw.get("pre"):
                    outs["pre"] = torch.full((M, N), float("nan"), device=DEV, dtype=dt); g.out_bf16_pre = outs["pre"].data_ptr()
                g.ldc_bf16 = N
                call("prx_k_gemm", g, None, 0, stream())
                torch.cuda.synchronize()
                return outs

            o = run(bias=True, resid=True, f32=True, o16=True, alpha=0.5)
            ref = 0.5 * prod + bias + resid
            assert rel_l2(o["f32"], ref) < 2e-5, (M, N, K, rel_l2(o["f32"], ref))
            assert rel_l2(o["o16"], ref) < tol16
            o = run(bias=True, act=1, o16=True, pre=True)
            pre = (prod + bias).to(dt).float()
            assert rel_l2(o["pre"], pre) < tol16 and rel_l2(o["o16"], pre * torch.sigmoid(1.702 * pre)) < tol16
            o = run(aux=True, act=2, o16=True)
            sg = torch.sigmoid(1.702 * aux.float())
            assert rel_l2(o["o16"], prod * (sg * (1 + 1.702 * aux.float() * (1 - sg)))) < tol16
            o = run(o16=True)
            assert rel_l2(o["o16"], prod) < tol16
            assert not torch.isnan(o["o16"].float()).any()
            o = run(bias=True, act=3, f32=True)                                   # PRX_ACT_RELU
            assert rel_l2(o["f32"], torch.relu(prod + bias)) < 2e-5
            o = run(aux=True, act=4, f32=True)                                    # PRX_ACT_MUL_RELUMASK
            assert rel_l2(o["f32"], prod * (aux.float() > 0)) < 2e-5
            o = run(aux=True, resid=True, act=5, f32=True)                        # PRX_ACT_RELUMASK_POST: residual AND mask -> 4-wave kernels
            assert rel_l2(o["f32"], (prod + resid) * (aux.float() > 0)) < 2e-5
    finally:
        lib.prx_gemm_tile_override(_lib.tool_ctx(), 0, 0, 0)
        lib.prx_gemm_tile_override(_lib.tool_ctx(), -8, 0, 1)
        lib.prx_gemm_tile_override(_lib.tool_ctx(), -12, 0, 0)


@pytest.mark.parametrize("kind", ["gemm", "conv", "gn"])
def test_gemm_engine_random_shapes_on_the_device(kind):
    """tests/_emu_fuzz.py on the GPU: 600 random products / 600 random implicit convolutions / 300 GroupNorm-epilogue launches over
    every kernel family, spare rows and columns of the outputs checked for stray writes (first run on the device in round 5:
    profiles/r05_first_call/r05_sweeps.log)"""
    import _emu_fuzz
    lib = _lib.load()
    if kind == "gemm":
        bad = _emu_fuzz.gemm_cases(lib, 11, 600, device=DEV)
    elif kind == "conv":
        bad, rejected = _emu_fuzz.conv_cases(lib, 11, 600, device=DEV)
        assert rejected > 0
    else:
        bad = _emu_fuzz.gn_cases(lib, 11, 300, device=DEV)
    assert bad == [], "\n".join(bad[:20])


FIT_TILES = [(160, 256), (160, 128), (160, 192), (256, 128), (128, 128), (80, 128), (128, 64), (64, 64), (32, 64), (16, 64), (16, 32), (256, 16)]
FIT_KS = {(160, 256): 1, (160, 128): 1, (160, 192): 1, (256, 128): 1, (128, 128): 1, (80, 128): 2, (128, 64): 2, (64, 64): 2, (32, 64): 4, (16, 64): 4, (16, 32): 8, (256, 16): 1}


@pytest.mark.parametrize("tile", FIT_TILES)
@pytest.mark.parametrize("prec", ["bf16", "fp16"])
def test_gemm_fit_tiles_implicit_conv_and_groupnorm_sums(tile, prec):
    """every fit tile on implicit 3x3 convolutions (plain and through the fused nearest-2x upsample, ragged M against the tile,
    batch 2) with the decoder's epilogues: bias + residual + fp32 / 16-bit outputs + the NEXT GroupNorm's sums, and a dgrad-shaped
    launch accumulating a GroupNorm-BACKWARD's sums; against torch conv2d on the same rounded operands and fp64 sums"""
    lib = _lib.load()
    dt = torch.bfloat16 if prec == "bf16" else torch.float16
    torch.manual_seed(5)
    ks = FIT_KS[tile]
    ran = 0
    try:
        lib.prx_gemm_tile_override(_lib.tool_ctx(), -12, 0, 1)
        lib.prx_gemm_tile_override(_lib.tool_ctx(), tile[0], tile[1], 1)
        for (H, W, Cin, Cout, up, NB) in [(32, 32, 128, 128, 1, 1), (24, 40, 64, 256, 0, 2), (16, 16, 512, 128, 0, 1), (16, 24, 256, 128, 1, 2)]:
            if (9 * Cin) % (64 * ks) or tile[0] % 80 == 0:           # K must split over the K groups; the 80-row-granular tiles are row-major only
                continue
            ran += 1
            hin, win = (H // 2, W // 2) if up else (H, W)
            x = torch.randn(NB, Cin, hin, win, device=DEV)
            w = torch.randn(Cout, Cin, 3, 3, device=DEV) / math.sqrt(9 * Cin)
            bias = torch.randn(Cout, device=DEV)
            M = NB * H * W
            resid = torch.randn(M, Cout, device=DEV)
            x_nhwc = x.permute(0, 2, 3, 1).contiguous().to(dt)
            w_pack = w.permute(0, 2, 3, 1).reshape(Cout, 9 * Cin).contiguous().to(dt)
            xr = x.to(dt).float()
            if up:
                xr = F.interpolate(xr, scale_factor=2.0, mode="nearest")
            ref = F.conv2d(xr, w.to(dt).float(), bias, padding=1).permute(0, 2, 3, 1).reshape(M, Cout) + resid
            gs = Cout // 32
            stats = torch.zeros(64, device=DEV, dtype=torch.float64)

            def args():
                g = GemmArgs()
                g.A = x_nhwc.data_ptr(); g.a_mode = 1; g.lda = Cin; g.B = w_pack.data_ptr(); g.ldb = 9 * Cin
                g.M, g.N, g.K = M, Cout, 9 * Cin
                g.H, g.W, g.Cin, g.up = H, W, Cin, up
                g.alpha = 1.0; g.f32 = 0 if prec == "bf16" else 2
                return g
            g = args()
            g.bias_n = bias.data_ptr(); g.resid = resid.data_ptr(); g.ldr = Cout
            out = torch.full((M, Cout), float("nan"), device=DEV); g.out_f32 = out.data_ptr(); g.ldc_f32 = Cout
            o16 = torch.full((M, Cout), float("nan"), device=DEV, dtype=dt); g.out_bf16 = o16.data_ptr(); g.ldc_bf16 = Cout
            call("prx_k_gemm_gn", g, stats, gs, None, None, None, None, 0, 1e-6, None, 0, stream())
            torch.cuda.synchronize()
            assert rel_l2(out, ref) < 2e-5, (tile, H, W, Cin, Cout, up, rel_l2(out, ref))
            assert rel_l2(o16, ref) < (4e-3 if prec == "bf16" else 5e-4)
            o64 = out.double().view(M, 32, gs)
            want = torch.stack([o64.sum(dim=(0, 2)), (o64 ** 2).sum(dim=(0, 2))], dim=1).reshape(-1)
            assert torch.allclose(stats, want, rtol=1e-5, atol=1e-3), (tile, (stats - want).abs().max())
            # GroupNorm-backward sums of a dgrad-shaped launch: out = d(GN output); xg = that GroupNorm's forward input
            xg = torch.randn(M, Cout, device=DEV)
            x64 = xg.double().view(M, 32, gs)
            fstats = torch.stack([x64.sum(dim=(0, 2)), (x64 ** 2).sum(dim=(0, 2))], dim=1).reshape(-1).contiguous()
            gamma, beta = torch.randn(Cout, device=DEV), torch.randn(Cout, device=DEV)
            bst = torch.zeros(64, device=DEV, dtype=torch.float64)
            g2 = args()
            out2 = torch.full((M, Cout), float("nan"), device=DEV); g2.out_f32 = out2.data_ptr(); g2.ldc_f32 = Cout
            call("prx_k_gemm_gn", g2, bst, gs, xg, fstats, gamma, beta, 1, 1e-6, None, 0, stream())
            torch.cuda.synchronize()
            assert rel_l2(out2, ref - resid - bias) < 2e-5
            n = M * gs
            mean = fstats.view(32, 2)[:, 0] / n
            var = (fstats.view(32, 2)[:, 1] / n - mean ** 2).clamp(min=0)
            rstd = 1.0 / torch.sqrt(var + 1e-6)
            xh = (x64 - mean.view(1, 32, 1)) * rstd.view(1, 32, 1)
            y = xh.float().double() * gamma.double().view(1, 32, gs) + beta.double().view(1, 32, gs)
            sg = torch.sigmoid(y)
            dxh = out2.double().view(M, 32, gs) * (sg * (1 + y * (1 - sg))) * gamma.double().view(1, 32, gs)
            wantb = torch.stack([dxh.sum(dim=(0, 2)), (dxh * xh).sum(dim=(0, 2))], dim=1).reshape(-1)
            assert torch.allclose(bst, wantb, rtol=5e-4, atol=5e-2), (tile, (bst - wantb).abs().max())
        assert ran or tile[0] % 80 == 0
    finally:
        lib.prx_gemm_tile_override(_lib.tool_ctx(), 0, 0, 0)
        lib.prx_gemm_tile_override(_lib.tool_ctx(), -12, 0, 0)


SPEC_TILES = [(160, 256), (160, 192), (160, 128), (80, 128), (256, 128), (128, 128), (128, 64), (64, 64), (32, 64), (16, 64), (16, 32)]


def fit_spec_vs_generic(tile, shapes=None, conv_shapes=None):
    """the fit kernels with a compile-time epilogue (gemmfit_kernel.h FIT_EPI_*: OUT16, RES16, GELU, DGELU, GN, RES16_GN, GNB; IEEE-half
    operands) against the generic kernel of the same tile on the same launch (switch-word bit 6): the arithmetic is the generic
    epilogue's term by term, so 16-bit outputs are BIT-identical wherever no sigmoid is evaluated; QuickGELU / its derivative use
    v_exp + v_rcp instead of a correctly rounded division (the saved pre-activation stays bit-identical, the activation moves by
    at most one half-precision ulp); GroupNorm sums agree to the rounding of their fp32 per-lane partials"""
    lib = _lib.load()
    dt = torch.float16
    ks = FIT_KS.get(tile, 2)                      # (256, 256): the 8-phase kernel, K % 128 == 0
    tower = tile[0] % 80 == 0 or tile == (256, 256)
    torch.manual_seed(21)
    ctx = _lib.tool_ctx()
    ran = set()

    def both(fn, spec=True):
        res = []
        for flags in (1, 65):
            lib.prx_gemm_tile_override(ctx, -8, 0, flags)
            n0 = lib.prx_gemm_fit_spec_launches()
            res.append(fn())
            assert lib.prx_gemm_fit_spec_launches() - n0 == (1 if flags == 1 and spec else 0), (tile, flags)      # the specialised kernel ran / did not run
        lib.prx_gemm_tile_override(ctx, -8, 0, 1)
        return res

    def bits(t):
        return t.view(torch.int16)
    try:
        lib.prx_gemm_tile_override(ctx, -12, 0, 1)
        lib.prx_gemm_tile_override(ctx, tile[0], tile[1], 1)
        for (M, N, K) in (shapes or [(3200, 768, 768), (1000, 200, 1152), (333, 520, 512), (81, 136, 1024)]):
            if K % (64 * ks):
                continue
            A = torch.randn(M, K, device=DEV).to(dt)
            Bt = (torch.randn(N, K, device=DEV) * torch.linspace(0.5, 1.5, N, device=DEV)[:, None] / math.sqrt(K)).to(dt)
            bias = torch.randn(N, device=DEV)
            ldr = N + 8
            resid = torch.randn(M, ldr, device=DEV).to(dt)
            aux = torch.randn(M, N, device=DEV).to(dt)
            prod = A.float() @ Bt.float().T

            def run(**kw):
                g = GemmArgs()
                g.A = A.data_ptr(); g.a_mode = 0; g.lda = K; g.B = Bt.data_ptr(); g.ldb = K; g.M, g.N, g.K = M, N, K
                g.alpha = kw.get("alpha", 1.0); g.f32 = 2; g.act = kw.get("act", 0)
                if kw.get("bias"): g.bias_n = bias.data_ptr()
                if kw.get("resid"): g.resid = resid.data_ptr(); g.ldr = ldr; g.row16 = 1
                if kw.get("aux"): g.aux = aux.data_ptr(); g.ldaux = N
                ldc = N + 16
                o16 = torch.full((M + 2, ldc), float("nan"), device=DEV, dtype=dt); g.out_bf16 = o16.data_ptr(); g.ldc_bf16 = ldc
                pre = None
                if kw.get("pre"):
                    pre = torch.full((M + 2, ldc), float("nan"), device=DEV, dtype=dt); g.out_bf16_pre = pre.data_ptr()
                call("prx_k_gemm", g, None, 0, stream())
                torch.cuda.synchronize()
                assert torch.isnan(o16[M:].float()).all() and torch.isnan(o16[:, N:].float()).all()       # spare rows / columns untouched
                return o16[:M, :N].clone(), (pre[:M, :N].clone() if pre is not None else None)

            (s16, _), (g16, _) = both(lambda: run(bias=True, alpha=0.5))                                   # OUT16
            assert torch.equal(bits(s16), bits(g16)) and rel_l2(s16, 0.5 * prod + bias) < 5e-4
            (s16, _), (g16, _) = both(lambda: run(bias=True, resid=True))                                  # RES16 (16-bit residual stream)
            assert torch.equal(bits(s16), bits(g16)) and rel_l2(s16, prod + bias + resid[:, :N].float()) < 5e-4
            ran.update(["out16", "res16"])
            (s16, spre), (g16, gpre) = both(lambda: run(bias=True, act=1, pre=True), spec=tower)           # GELU (the tower tiles have it)
            assert torch.equal(bits(spre), bits(gpre))
            pre = (prod + bias).to(dt).float()
            assert rel_l2(s16, pre * torch.sigmoid(1.702 * pre)) < 5e-4
            d = (s16.float() - g16.float()).abs()
            assert bool((d <= 2.0 ** -10 * g16.float().abs() + 1e-7).all()), d.max()                       # one half ulp of movement at most
            (s16, _), (g16, _) = both(lambda: run(aux=True, act=2), spec=tower)                            # DGELU
            sg = torch.sigmoid(1.702 * aux.float())
            assert rel_l2(s16, prod * (sg * (1 + 1.702 * aux.float() * (1 - sg)))) < 5e-4
            d = (s16.float() - g16.float()).abs()
            assert bool((d <= 2.0 ** -10 * g16.float().abs() + 1e-7).all()), d.max()
            ran.update(["gelu", "dgelu"])
        for (H, W, Cin, Cout, up, NB) in (conv_shapes or [(32, 32, 128, 128, 1, 1), (24, 40, 64, 256, 0, 2), (16, 16, 512, 128, 0, 1)]):
            if tower or (9 * Cin) % (64 * ks):
                continue
            hin, win = (H // 2, W // 2) if up else (H, W)
            x_nhwc = torch.randn(NB, hin, win, Cin, device=DEV).to(dt)
            w_pack = (torch.randn(Cout, 9 * Cin, device=DEV) / math.sqrt(9 * Cin)).to(dt)
            bias = torch.randn(Cout, device=DEV)
            M = NB * H * W
            resid = torch.randn(M, Cout, device=DEV).to(dt)
            gs = Cout // 32
            xg = torch.randn(M, Cout, device=DEV).to(dt)
            x64 = xg.double().view(M, 32, gs)
            fstats = torch.stack([x64.sum(dim=(0, 2)), (x64 ** 2).sum(dim=(0, 2))], dim=1).reshape(-1).contiguous()
            gamma, beta = torch.randn(Cout, device=DEV), torch.randn(Cout, device=DEV)

            def runc(kind):
                g = GemmArgs()
                g.A = x_nhwc.data_ptr(); g.a_mode = 1; g.lda = Cin; g.B = w_pack.data_ptr(); g.ldb = 9 * Cin
                g.M, g.N, g.K = M, Cout, 9 * Cin
                g.H, g.W, g.Cin, g.up = H, W, Cin, up
                g.alpha = 1.0; g.f32 = 2; g.bias_n = bias.data_ptr()
                o16 = torch.full((M + 2, Cout), float("nan"), device=DEV, dtype=dt); g.out_bf16 = o16.data_ptr(); g.ldc_bf16 = Cout
                st = torch.zeros(64, device=DEV, dtype=torch.float64)
                if kind == "res16_gn":
                    g.resid = resid.data_ptr(); g.ldr = Cout; g.row16 = 1
                if kind == "gnb":
                    g.bias_n = None; g.row16 = 2
                    call("prx_k_gemm_gn", g, st, gs, xg, fstats, gamma, beta, 1, 1e-6, None, 0, stream())
                else:
                    call("prx_k_gemm_gn", g, st, gs, None, None, None, None, 0, 1e-6, None, 0, stream())
                torch.cuda.synchronize()
                assert torch.isnan(o16[M:].float()).all()
                return o16[:M].clone(), st
            for kind in ("gn", "res16_gn", "gnb"):
                (s16, sst), (g16, gst) = both(lambda: runc(kind))
                assert torch.equal(bits(s16), bits(g16)), (tile, kind)
                assert torch.allclose(sst, gst, rtol=2e-5 if kind == "gnb" else 2e-6, atol=1e-2 if kind == "gnb" else 1e-3), (tile, kind, (sst - gst).abs().max())      # fp32 per-lane partials: the two code shapes contract their multiply-adds differently
                assert float(sst.abs().sum()) > 0
                ran.add(kind)
    finally:
        lib.prx_gemm_tile_override(ctx, 0, 0, 0)
        lib.prx_gemm_tile_override(ctx, -8, 0, 1)
        lib.prx_gemm_tile_override(ctx, -12, 0, 0)
    return ran


@pytest.mark.parametrize("tile", SPEC_TILES + [(256, 256)])
def test_gemm_fit_specialised_epilogues_match_the_generic_kernel(tile):
    """... and the 8-phase 256 x 256 kernel's tower epilogues (gemm8p.hip) by the same rule"""
    ran = fit_spec_vs_generic(tile)
    assert {"out16", "res16", "gelu", "dgelu"} <= ran and (tile[0] % 80 == 0 or tile == (256, 256) or {"gn", "res16_gn", "gnb"} <= ran)


def fit_f32_conv_and_stats_checks(tiles):
    """fp32-operand fit kernels (FIT_EPI_F32) forced on small implicit convolutions: output vs float64, fused GroupNorm forward sums and
    GroupNorm-backward sums vs float64; then a convolution they are not eligible for (Cin = 48), where the engine must deliver the
    same statistics through its fallback (4-wave fp32 kernel + the norm kernels' statistics pass)"""
    lib = _lib.load()
    ctx = _lib.tool_ctx()
    torch.manual_seed(9)
    try:
        for tile in list(tiles) + [None]:
            lib.prx_gemm_tile_override(ctx, -12, 0, 1)
            if tile is not None:
                lib.prx_gemm_tile_override(ctx, tile[0], tile[1], 1)
                H, W, Cin, Cout = 8, 8, 64 * FIT_KS[tile] // (2 if FIT_KS[tile] > 1 else 1), 128      # K = 9 Cin a multiple of 32 KS
            else:
                lib.prx_gemm_tile_override(ctx, 0, 0, 0)
                H, W, Cin, Cout = 8, 12, 48, 128
            M, K = H * W, 9 * Cin
            x = torch.randn(1, H, W, Cin, device=DEV)
            w = torch.randn(Cout, K, device=DEV) / math.sqrt(K)
            bias = torch.randn(Cout, device=DEV); resid = torch.randn(M, Cout, device=DEV)
            gs = Cout // 32
            xr = x.permute(0, 3, 1, 2).double().cpu()
            wr = w.view(Cout, 3, 3, Cin).permute(0, 3, 1, 2).double().cpu()
            ref = F.conv2d(xr, wr, bias.double().cpu(), padding=1).permute(0, 2, 3, 1).reshape(M, Cout) + resid.double().cpu()

            def args():
                g = GemmArgs()
                g.A = x.data_ptr(); g.a_mode = 1; g.lda = Cin; g.B = w.data_ptr(); g.ldb = K; g.M, g.N, g.K = M, Cout, K
                g.H, g.W, g.Cin, g.up = H, W, Cin, 0
                g.alpha = 1.0; g.f32 = 1
                return g
            g = args(); g.bias_n = bias.data_ptr(); g.resid = resid.data_ptr(); g.ldr = Cout
            out = torch.full((M + 2, Cout), float("nan"), device=DEV); g.out_f32 = out.data_ptr(); g.ldc_f32 = Cout
            st = torch.zeros(64, device=DEV, dtype=torch.float64)
            n0 = lib.prx_gemm_fit_spec_launches()
            call("prx_k_gemm_gn", g, st, gs, None, None, None, None, 0, 1e-6, None, 0, stream())
            torch.cuda.synchronize()
            assert lib.prx_gemm_fit_spec_launches() - n0 == (1 if tile is not None else 0), tile        # the fp32 fit kernel ran / the fallback did
            assert torch.isnan(out[M:]).all()
            o = out[:M].double().cpu()
            assert float((o - ref).norm() / ref.norm()) < 2e-6, (tile, float((o - ref).norm() / ref.norm()))
            o64 = o.view(M, 32, gs)
            want = torch.stack([o64.sum(dim=(0, 2)), (o64 ** 2).sum(dim=(0, 2))], dim=1).reshape(-1)
            assert torch.allclose(st.cpu(), want, rtol=1e-5, atol=1e-3), (tile, float((st.cpu() - want).abs().max()))
            # GroupNorm-backward sums of a dgrad-shaped launch (no bias / residual): out2 = d(GN output), xg = that GroupNorm's input
            xg = torch.randn(M, Cout, device=DEV)
            x64 = xg.double().cpu().view(M, 32, gs)
            fstats = torch.stack([x64.sum(dim=(0, 2)), (x64 ** 2).sum(dim=(0, 2))], dim=1).reshape(-1).contiguous().to(DEV)
            gamma, beta = torch.randn(Cout, device=DEV), torch.randn(Cout, device=DEV)
            bst = torch.zeros(64, device=DEV, dtype=torch.float64)
            g2 = args()
            out2 = torch.full((M, Cout), float("nan"), device=DEV); g2.out_f32 = out2.data_ptr(); g2.ldc_f32 = Cout
            call("prx_k_gemm_gn", g2, bst, gs, xg, fstats, gamma, beta, 1, 1e-6, None, 0, stream())
            torch.cuda.synchronize()
            o2 = out2.double().cpu()
            assert float((o2 - (ref - resid.double().cpu() - bias.double().cpu())).norm() / o2.norm()) < 2e-6
            n = M * gs
            mean = fstats.cpu().view(32, 2)[:, 0] / n
            var = (fstats.cpu().view(32, 2)[:, 1] / n - mean ** 2).clamp(min=0)
            rstd = 1.0 / torch.sqrt(var + 1e-6)
            xh = (x64 - mean.view(1, 32, 1)) * rstd.view(1, 32, 1)
            y = xh.float().double() * gamma.double().cpu().view(1, 32, gs) + beta.double().cpu().view(1, 32, gs)
            sg = torch.sigmoid(y)
            dxh = o2.view(M, 32, gs) * (sg * (1 + y * (1 - sg))) * gamma.double().cpu().view(1, 32, gs)
            wantb = torch.stack([dxh.sum(dim=(0, 2)), (dxh * xh).sum(dim=(0, 2))], dim=1).reshape(-1)
            assert torch.allclose(bst.cpu(), wantb, rtol=5e-4, atol=5e-3), (tile, float((bst.cpu() - wantb).abs().max()))
    finally:
        lib.prx_gemm_tile_override(ctx, 0, 0, 0)
        lib.prx_gemm_tile_override(ctx, -12, 0, 0)


def test_gemm_fit_fp32_operand_kernels_conv_and_groupnorm_sums():
    fit_f32_conv_and_stats_checks([(256, 128), (128, 128), (128, 64), (64, 64), (32, 64), (16, 64), (16, 32)])


def test_gemm_fit_tiles_are_what_the_headline_tower_runs_on():
    """the planner gives the ViT-B/32 products of 64 cutouts (M = 3200) one workgroup per CU: 240 tiles of 160 x 256, 160 x 192 or
    80 x 128, and the result is the 4-wave kernels' to fp32 round-off (another K summation order on the two-K-group tile)"""
    lib = _lib.load()
    torch.manual_seed(3)
    # ... and the sharded batches of 2 / 4 / 8 GPUs (32 / 16 / 8 cutouts per rank: M = 1600 / 800 / 400 token rows, multiples of 80)
    for (M, N, K) in [(3200, 3072, 768), (3200, 2304, 768), (3200, 768, 3072), (1600, 3072, 768), (1600, 768, 3072), (800, 2304, 768),
                      (800, 768, 3072), (400, 3072, 768), (400, 768, 768), (400, 768, 3072)]:
        A = torch.randn(M, K, device=DEV).to(torch.float16)
        Bt = (torch.randn(N, K, device=DEV) / math.sqrt(K)).to(torch.float16)
        outs = []
        for fit in (1, 0):
            lib.prx_gemm_tile_override(_lib.tool_ctx(), -7, 0, fit)
            g = GemmArgs()
            g.A = A.data_ptr(); g.lda = K; g.B = Bt.data_ptr(); g.ldb = K; g.M, g.N, g.K = M, N, K
            g.alpha = 1.0; g.f32 = 2
            out = torch.full((M, N), float("nan"), device=DEV); g.out_f32 = out.data_ptr(); g.ldc_f32 = N
            call("prx_k_gemm", g, None, 0, stream())
            torch.cuda.synchronize()
            outs.append(out)
        lib.prx_gemm_tile_override(_lib.tool_ctx(), -7, 0, 1)
        assert rel_l2(outs[0], outs[1]) < 2e-6
        assert rel_l2(outs[0], A.float() @ Bt.float().T) < 2e-5
    # the decoder's 256^2 level (two row tiles per workgroup) and a 16^2 convolution (8 K groups), planner's choice vs 4-wave kernels
    for (H, Cin, Cout) in [(256, 128, 128), (16, 512, 512)]:
        x = torch.randn(1, Cin, H, H, device=DEV)
        w = torch.randn(Cout, Cin, 3, 3, device=DEV) / math.sqrt(9 * Cin)
        x_nhwc = x.permute(0, 2, 3, 1).contiguous().to(torch.float16)
        w_pack = w.permute(0, 2, 3, 1).reshape(Cout, 9 * Cin).contiguous().to(torch.float16)
        ws = torch.empty(64 << 20, dtype=torch.uint8, device=DEV)
        outs = []
        for fit in (1, 0):
            lib.prx_gemm_tile_override(_lib.tool_ctx(), -7, 0, fit)
            g = GemmArgs()
            g.A = x_nhwc.data_ptr(); g.a_mode = 1; g.lda = Cin; g.B = w_pack.data_ptr(); g.ldb = 9 * Cin
            g.M, g.N, g.K = H * H, Cout, 9 * Cin
            g.H, g.W, g.Cin, g.up = H, H, Cin, 0
            g.alpha = 1.0; g.f32 = 2
            out = torch.full((H * H, Cout), float("nan"), device=DEV); g.out_f32 = out.data_ptr(); g.ldc_f32 = Cout
            call("prx_k_gemm", g, ws, ws.numel(), stream())
            torch.cuda.synchronize()
            outs.append(out)
        lib.prx_gemm_tile_override(_lib.tool_ctx(), -7, 0, 1)
        ref = F.conv2d(x.to(torch.float16).float(), w.to(torch.float16).float(), None, padding=1).permute(0, 2, 3, 1).reshape(H * H, Cout)
        assert rel_l2(outs[0], outs[1]) < 2e-6 and rel_l2(outs[0], ref) < 2e-5


@pytest.mark.parametrize("H,W,Cin,Cout,up,NB", [
    (16, 16, 256, 512, 0, 1), (32, 32, 512, 256, 1, 1), (64, 64, 128, 128, 0, 1), (8, 12, 32, 40, 1, 2),
    (256, 256, 128, 128, 0, 1), (64, 64, 8, 128, 0, 1),
])
def test_gemm_conv3x3(H, W, Cin, Cout, up, NB, gemm_variant):
    """implicit-GEMM 3x3/pad-1 conv on NHWC (optionally through a fused nearest-2x upsample)"""
    torch.manual_seed(H * W + Cin)
    hin, win = (H // 2, W // 2) if up else (H, W)
    x = torch.randn(NB, Cin, hin, win, device=DEV)
    w = torch.randn(Cout, Cin, 3, 3, device=DEV) / math.sqrt(9 * Cin)
    bias = torch.randn(Cout, device=DEV)
    x_nhwc = bf(x.permute(0, 2, 3, 1).contiguous())
    w_pack = bf(w.permute(0, 2, 3, 1).reshape(Cout, 9 * Cin).contiguous())
    out, _, _ = run_gemm(x_nhwc, w_pack, NB * H * W, Cout, 9 * Cin, a_mode=1, lda=Cin, H=H, W=W, Cin=Cin, up=up,
                         bias_n=bias)
    xr = bf(x).float()
    if up:
        xr = F.interpolate(xr, scale_factor=2.0, mode="nearest")
    ref = F.conv2d(xr, bf(w).float(), bias, padding=1).permute(0, 2, 3, 1).reshape(NB * H * W, Cout)
    assert rel_l2(out, ref) < 2e-5, rel_l2(out, ref)
    # f32 activation input converted on load
    x32 = x.permute(0, 2, 3, 1).contiguous()
    out2, _, _ = run_gemm(x32, w_pack, NB * H * W, Cout, 9 * Cin, a_mode=1, lda=Cin, H=H, W=W, Cin=Cin, up=up,
                          bias_n=bias)
    assert rel_l2(out2, ref) < 2e-5


@pytest.mark.parametrize("H,W,C,Cout,NB", [(16, 16, 128, 128, 1), (12, 20, 64, 72, 2), (128, 128, 128, 128, 1)])
def test_gemm_conv3x3_stride2_down(H, W, C, Cout, NB):
    """taming Downsample (encoder): zero pad (0,1,0,1) then 3x3 stride-2 conv, as the engine's up == 2 gather"""
    torch.manual_seed(H + W + C)
    x = torch.randn(NB, C, 2 * H, 2 * W, device=DEV)
    w = torch.randn(Cout, C, 3, 3, device=DEV) / math.sqrt(9 * C)
    bias = torch.randn(Cout, device=DEV)
    x_nhwc = bf(x.permute(0, 2, 3, 1).contiguous())
    w_pack = bf(w.permute(0, 2, 3, 1).reshape(Cout, 9 * C).contiguous())
    out, _, _ = run_gemm(x_nhwc, w_pack, NB * H * W, Cout, 9 * C, a_mode=1, lda=C, H=H, W=W, Cin=C, up=2, bias_n=bias)
    ref = F.conv2d(F.pad(bf(x).float(), (0, 1, 0, 1)), bf(w).float(), bias, stride=2)
    ref = ref.permute(0, 2, 3, 1).reshape(NB * H * W, Cout)
    assert rel_l2(out, ref) < 2e-5, rel_l2(out, ref)


@pytest.mark.parametrize("P,C,swish", [(256, 512, 1), (1024, 256, 1), (4096, 128, 0), (65536, 128, 1)])
def test_groupnorm_fwd_bwd(P, C, swish):
    torch.manual_seed(P + C)
    NB = 1
    x = (torch.randn(NB, P, C, device=DEV) * 1.5 + 0.3).requires_grad_(True)
    gamma = torch.randn(C, device=DEV) * 0.2 + 1.0
    beta = torch.randn(C, device=DEV) * 0.1
    stats = torch.zeros(NB * 64, dtype=torch.float64, device=DEV)
    out_bf16 = torch.empty(NB, P, C, dtype=torch.bfloat16, device=DEV)
    out_f32 = torch.empty(NB, P, C, device=DEV)
    call("prx_k_groupnorm_fwd", x, gamma, beta, stats, out_bf16, out_f32, NB, P, C, swish, 1e-6, stream())
    y = F.group_norm(x.permute(0, 2, 1), 32, gamma, beta, eps=1e-6).permute(0, 2, 1)
    ref = y * torch.sigmoid(y) if swish else y
    torch.cuda.synchronize()
    assert rel_l2(out_f32, ref) < 1e-5
    assert rel_l2(out_bf16, ref) < 4e-3
    g = torch.randn(NB, P, C, device=DEV)
    add = torch.randn(NB, P, C, device=DEV)
    (gx,) = torch.autograd.grad(ref, x, g)
    bstats = torch.zeros(NB * 64, dtype=torch.float64, device=DEV)
    dx = torch.empty(NB, P, C, device=DEV)
    call("prx_k_groupnorm_bwd", g, x.detach(), gamma, beta, stats, bstats, add, dx, NB, P, C, swish, 1e-6, stream())
    torch.cuda.synchronize()
    assert rel_l2(dx, gx + add) < 2e-5, rel_l2(dx, gx + add)


@pytest.mark.parametrize("rows,C", [(3200, 768), (64, 768), (257, 1024)])
def test_layernorm_fwd_bwd(rows, C):
    torch.manual_seed(rows)
    x = (torch.randn(rows, C, device=DEV) * 2 + 0.5).requires_grad_(True)
    gamma = torch.randn(C, device=DEV) * 0.2 + 1.0
    beta = torch.randn(C, device=DEV) * 0.1
    ob = torch.empty(rows, C, dtype=torch.bfloat16, device=DEV)
    of = torch.empty(rows, C, device=DEV)
    mean = torch.empty(rows, device=DEV)
    rstd = torch.empty(rows, device=DEV)
    call("prx_k_layernorm_fwd", x, C, gamma, beta, ob, of, mean, rstd, rows, C, 1e-5, stream())
    ref = F.layer_norm(x, (C,), gamma, beta, eps=1e-5)
    torch.cuda.synchronize()
    assert rel_l2(of, ref) < 1e-5
    assert rel_l2(ob, ref) < 4e-3
    g = torch.randn(rows, C, device=DEV)
    add = torch.randn(rows, C, device=DEV)
    (gx,) = torch.autograd.grad(ref, x, g)
    dx = torch.empty(rows, C, device=DEV)
    call("prx_k_layernorm_bwd", g, C, x.detach(), C, gamma, mean, rstd, add, C, dx, C, rows, C, stream())
    torch.cuda.synchronize()
    assert rel_l2(dx, gx + add) < 1e-5


def test_transpose_softmax_upsample_layout():
    torch.manual_seed(1)
    a = bf(torch.randn(256, 1536, device=DEV))
    out = torch.empty(512, 256, dtype=torch.bfloat16, device=DEV)
    call("prx_k_transpose_bf16", a[:, 512:], 1536, out, 256, 256, 512, stream())
    torch.cuda.synchronize()
    assert torch.equal(out, a[:, 512:1024].T.contiguous())  # bit-exact data movement
    S = torch.randn(256, 256, device=DEV) * 20
    P = torch.empty(256, 256, dtype=torch.bfloat16, device=DEV)
    PT = torch.empty(256, 256, dtype=torch.bfloat16, device=DEV)
    scale = 512 ** -0.5
    call("prx_k_softmax_rows", S, 256, scale, P, 256, PT, 256, 256, 256, stream())
    ref = torch.softmax(S * scale, dim=-1)
    torch.cuda.synchronize()
    assert rel_l2(P, ref) < 4e-3 and torch.equal(P.T.contiguous(), PT)
    dP = torch.randn(256, 256, device=DEV)
    dS = torch.empty_like(P)
    dST = torch.empty_like(P)
    call("prx_k_softmax_rows_bwd", P, 256, dP, 256, scale, dS, 256, dST, 256, 256, 256, stream())
    pf = P.float()
    refd = scale * pf * (dP - (pf * dP).sum(-1, keepdim=True))
    torch.cuda.synchronize()
    assert rel_l2(dS, refd) < 4e-3 and torch.equal(dS.T.contiguous(), dST)
    hi = torch.randn(2, 16, 24, 128, device=DEV)
    low = torch.empty(2, 8, 12, 128, device=DEV)
    call("prx_k_upsample2x_bwd", hi, low, 2, 8, 12, 128, stream())
    ref = hi.view(2, 8, 2, 12, 2, 128).sum(dim=(2, 4))
    torch.cuda.synchronize()
    assert rel_l2(low, ref) < 1e-6
    x = torch.randn(2, 3, 50, device=DEV)
    o32 = torch.empty(2, 50, 8, device=DEV)
    ob = torch.empty(2, 50, 8, dtype=torch.bfloat16, device=DEV)
    call("prx_k_nchw_to_nhwc", x, o32, ob, 2, 3, 50, 8, stream())
    back = torch.empty(2, 3, 50, device=DEV)
    call("prx_k_nhwc_to_nchw", o32, 8, back, 2, 3, 50, stream())
    torch.cuda.synchronize()
    assert torch.equal(back, x) and torch.equal(o32[..., 3:], torch.zeros_like(o32[..., 3:]))
    assert torch.equal(ob, bf(o32))


def test_image_head():
    torch.manual_seed(2)
    HW, C, ld = 4096, 3, 4
    x = torch.randn(1, HW, ld, device=DEV) * 1.5
    img = torch.empty(1, C, HW, device=DEV)
    call("prx_k_image_head_fwd", x, ld, img, 1, C, HW, stream())
    u = (x[..., :C].permute(0, 2, 1) + 1) / 2
    torch.cuda.synchronize()
    assert torch.equal(img, u.clamp(0, 1))
    g = torch.randn(1, C, HW, device=DEV)
    dx = torch.empty(1, HW, 8, device=DEV)
    dxb = torch.empty(1, HW, 8, dtype=torch.bfloat16, device=DEV)
    call("prx_k_image_head_bwd", x, ld, g, dx, dxb, 8, 1, C, HW, stream())
    # ClampWithGrad.backward (vqgan.py:76-79) then d/dx of (x+1)/2
    ref = (g * (g * (u - u.clamp(0, 1)) >= 0)) * 0.5
    torch.cuda.synchronize()
    assert torch.equal(dx[..., :C], ref.permute(0, 2, 1)) and (dx[..., C:] == 0).all()
    assert torch.equal(dxb, bf(dx))


@pytest.mark.parametrize("N,T", [(64, 50), (3, 64), (2, 17)])
def test_mha_fwd_bwd(N, T):
    torch.manual_seed(N * T)
    C, heads = 768, 12
    qkv = bf(torch.randn(N * T, 3 * C, device=DEV))
    out = torch.full((N * T, C), float("nan"), dtype=torch.bfloat16, device=DEV)
    call("prx_k_mha_fwd", qkv, out, N, T, C, heads, stream())
    q, k, v = [t.reshape(N, T, heads, 64).permute(0, 2, 1, 3).float().requires_grad_(True)
               for t in qkv.float().split(C, dim=1)]
    att = torch.softmax(q @ k.transpose(-1, -2) * 0.125, dim=-1)
    ref = (att @ v).permute(0, 2, 1, 3).reshape(N * T, C)
    torch.cuda.synchronize()
    # P is rounded to bf16 before PV and the output is bf16: 2^-8-class errors
    assert rel_l2(out, ref) < 8e-3, rel_l2(out, ref)
    do = bf(torch.randn(N * T, C, device=DEV))
    dqkv = torch.full((N * T, 3 * C), float("nan"), dtype=torch.bfloat16, device=DEV)
    call("prx_k_mha_bwd", qkv, do, dqkv, N, T, C, heads, stream())
    gq, gk, gv = torch.autograd.grad(ref, (q, k, v), do.float())
    refd = torch.cat([t.permute(0, 2, 1, 3).reshape(N * T, C) for t in (gq, gk, gv)], dim=1)
    torch.cuda.synchronize()
    for i, nm in enumerate("qkv"):
        e = rel_l2(dqkv[:, i * C:(i + 1) * C], refd[:, i * C:(i + 1) * C])
        assert e < 1.2e-2, (nm, e)


@pytest.mark.parametrize("N,T,qs", [(2, 65, 1.0), (3, 197, 1.0), (2, 257, 1.0), (4, 50, 1.0), (2, 82, 1.0), (1, 130, 1.0), (1, 300, 1.0),
                                    (1, 512, 1.0), (1, 577, 1.0), (2, 257, 6.0), (1, 197, 6.0)])
def test_mha_general_fwd_bwd(N, T, qs):
    """flash-style attention for any sequence length (ViT-B/16: 197 tokens, ViT-L/14: 257, RN50x4 attention pool: 82):
    workgroup-per-head kernels up to 512 tokens (one, two or four workgroups per head), tile kernels beyond (577 = ViT-L/14
    at 336 px); `qs` scales the queries so that the running maximum of the online softmax moves between key blocks"""
    torch.manual_seed(N * T + 1)
    C, heads = 256, 4
    qkv = torch.randn(N * T, 3 * C, device=DEV)
    qkv[:, :C] *= qs
    qkv = bf(qkv)
    out = torch.full((N * T, C), float("nan"), dtype=torch.bfloat16, device=DEV)
    lse = torch.full((N * heads * T,), float("nan"), device=DEV)
    call("prx_k_mha_fwd_gen", qkv, out, lse, N, T, C, heads, stream())
    q, k, v = [t.reshape(N, T, heads, 64).permute(0, 2, 1, 3).float().requires_grad_(True)
               for t in qkv.float().split(C, dim=1)]
    sc = q @ k.transpose(-1, -2) * 0.125
    att = torch.softmax(sc, dim=-1)
    ref = (att @ v).permute(0, 2, 1, 3).reshape(N * T, C)
    torch.cuda.synchronize()
    assert rel_l2(out, ref) < 8e-3, rel_l2(out, ref)
    assert rel_l2(lse.reshape(N, heads, T), torch.logsumexp(sc, dim=-1)) < 1e-5
    do = bf(torch.randn(N * T, C, device=DEV))
    dqkv = torch.full((N * T, 3 * C), float("nan"), dtype=torch.bfloat16, device=DEV)
    call("prx_k_mha_bwd_gen", qkv, out, do, lse, dqkv, N, T, C, heads, stream())
    gq, gk, gv = torch.autograd.grad(ref, (q, k, v), do.float())
    refd = torch.cat([t.permute(0, 2, 1, 3).reshape(N * T, C) for t in (gq, gk, gv)], dim=1)
    torch.cuda.synchronize()
    assert torch.isfinite(dqkv.float()).all()
    for i, nm in enumerate("qkv"):
        e = rel_l2(dqkv[:, i * C:(i + 1) * C], refd[:, i * C:(i + 1) * C])
        assert e < 1.5e-2, (nm, e)


@pytest.mark.parametrize("n,m", [(300, 777), (640, 200), (1024, 5000)])
@pytest.mark.parametrize("l2", [False, True])
def test_strotss_relaxed_emd_kernels_match_the_composed_torch_expression(n, m, l2):
    """csrc/strotss.hip against the plugin's own chain (Losses/StyleLoss.py:272-293) on the device: value, and the gradient
    over the n + m selected pairs against autograd through the dense distance matrix.  n < m: the column branch of the max
    carries the gradient; n > m: the row branch.  l2: the 3-channel palette form (cosine + L2)."""
    from pixray_amd import ops, style_loss as sl
    g = torch.Generator(device=DEV).manual_seed(100 * n + m + l2)
    d = 3 if l2 else 515
    X = (torch.randn(n, d, device=DEV, generator=g).abs() + 0.05 * torch.randn(n, d, device=DEV, generator=g))
    Y = (torch.randn(m, d, device=DEV, generator=g).abs() + 0.05 * torch.randn(m, d, device=DEV, generator=g))
    # hubs, as real feature columns have them: half of the style columns sit next to one of five rows of X, so those rows collect
    # ~m / 10 column minima each -- several 32-pair chunks of the backward, reduced in order
    hub = torch.arange(0, m, 2, device=DEV)
    Y[hub] = X[hub % 5] * (1.0 + 0.1 * torch.rand(hub.numel(), 1, device=DEV, generator=g)) + 0.01 * torch.randn(hub.numel(), d, device=DEV, generator=g)
    Xa = X.clone().requires_grad_(True)
    va = sl._remd_composed(Xa, Y, l2)
    (va * 1.7).backward()
    Xb = X.clone().requires_grad_(True)
    vb = ops.strotss_remd(Xb, Y, l2=l2)
    (vb * 1.7).backward()
    assert abs(float(va.detach()) - float(vb.detach())) <= 2e-6 * abs(float(va.detach())), (float(va.detach()), float(vb.detach()))
    rel = float((Xb.grad - Xa.grad).norm() / Xa.grad.norm())
    assert rel < 2e-5, rel
    assert int((Xa.grad.abs().sum(1) > 0).sum()) == int((Xb.grad.abs().sum(1) > 0).sum())      # the same rows are selected
    # reproducible: packed {value, position} minima, a fixed pair order, no floating-point atomics
    Xc = X.clone().requires_grad_(True)
    (ops.strotss_remd(Xc, Y, l2=l2) * 1.7).backward()
    assert torch.equal(Xc.grad, Xb.grad)


def test_strotss_self_similarity_kernels_match_the_composed_torch_expression():
    """`content_loss` (Losses/StyleLoss.py:246-265): both operands are differentiated"""
    from pixray_amd import ops, style_loss as sl
    g = torch.Generator(device=DEV).manual_seed(11)
    for n, d in ((257, 130), (1024, 2179)):
        X = torch.randn(n, d, device=DEV, generator=g).abs()
        Y = (X + 0.3 * torch.randn(n, d, device=DEV, generator=g)).abs()
        Xa, Ya = X.clone().requires_grad_(True), Y.clone().requires_grad_(True)
        va = sl._selfsim_composed(Xa, Ya)
        (va * 0.6).backward()
        Xb, Yb = X.clone().requires_grad_(True), Y.clone().requires_grad_(True)
        vb = ops.strotss_selfsim(Xb, Yb)
        (vb * 0.6).backward()
        assert abs(float(va.detach()) - float(vb.detach())) <= 2e-6 * abs(float(va.detach())), (float(va.detach()), float(vb.detach()))
        for a, b in ((Xa.grad, Xb.grad), (Ya.grad, Yb.grad)):
            rel = float((b - a).norm() / a.norm())
            assert rel < 5e-5, (n, d, rel)


def test_hypercolumns_match_the_composed_torch_expression():
    """StyleLoss `spatial_feature_extract` (Losses/StyleLoss.py:169-223) in one gather launch: the CUDA branch of
    style_loss._bilinear_columns against its own composed torch branch (what runs on CPU tensors and is pinned to the
    reference's goldens) -- forward bit-identical, gradients to fp32 atomics' reordering"""
    import numpy as np
    from pixray_amd import ops, style_loss as sl
    g = torch.Generator().manual_seed(5)
    shapes = [(40, 56, 3), (40, 56, 64), (40, 56, 64), (20, 28, 128), (20, 28, 128), (10, 14, 256), (10, 14, 256), (10, 14, 256),
              (5, 7, 512), (2, 3, 512)]
    fa = [torch.randn(1, h, w, c, generator=g).to(DEV).requires_grad_(True) for h, w, c in shapes]
    fb = [torch.randn(1, h, w, c, generator=g).to(DEV).requires_grad_(True) for h, w, c in shapes]
    np.random.seed(2)
    xx, xy = sl._sample_grid(40, 56)
    xx, xy = xx[:300].astype(np.float64) + 0.37, xy[:300].astype(np.float64) + 0.61      # off-grid: all four taps weigh in
    a, b = sl._bilinear_columns(fa, fb, xx, xy)
    go = torch.randn(a.shape, generator=g).to(DEV)
    ga = torch.autograd.grad((a * go).sum() + (b * go).sum() * 0.5, fa + fb)
    # the composed form: the same function on CPU copies
    fa_c = [f.detach().cpu().requires_grad_(True) for f in fa]
    fb_c = [f.detach().cpu().requires_grad_(True) for f in fb]
    a_c, b_c = sl._bilinear_columns(fa_c, fb_c, xx, xy)
    gc = torch.autograd.grad((a_c * go.cpu()).sum() + (b_c * go.cpu()).sum() * 0.5, fa_c + fb_c)
    assert tuple(a.shape) == tuple(a_c.shape) == (1, sum(c for _, _, c in shapes) + 2, 300, 1)
    assert torch.equal(a.cpu(), a_c) and torch.equal(b.cpu(), b_c)
    for x, y in zip(ga, gc):
        assert rel_l2(x.cpu(), y) < 1e-6
