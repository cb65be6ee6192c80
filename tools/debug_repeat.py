"""Run-to-run bit-equality of single kernels and of the two model runners (diagnostic)."""
import os, sys, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pixray_amd import _lib, ops, weights
from pixray_amd._lib import GemmArgs, call
DEV = "cuda"
s = _lib.current_stream

def report(name, outs_a, outs_b):
    bad = []
    for i, (a, b) in enumerate(zip(outs_a, outs_b)):
        if not torch.equal(a, b):
            bad.append((i, ((a.double() - b.double()).norm() / (a.double().norm() + 1e-300)).item()))
    print(f"{name:50s} {'BIT-EXACT' if not bad else 'DIFFERS ' + str(bad)}", flush=True)

# --- GEMM variants with GN stats -------------------------------------------------------------------------------
def gemm_once(M, N, K, conv=None, stats=False, tile=(0, 0, 0), stages=0):
    torch.manual_seed(1)
    lib = _lib.load()
    if conv is None:
        A = torch.randn(M, K, device=DEV).bfloat16()
    else:
        A = torch.randn(M // (4 if conv[3] else 1), conv[2], device=DEV).bfloat16()
    Bt = (torch.randn(N, K, device=DEV) / math.sqrt(K)).bfloat16()
    bias = torch.randn(N, device=DEV)
    ws = torch.empty(64 << 20, dtype=torch.uint8, device=DEV)
    outs = []
    lib.prx_gemm_tile_override(_lib.tool_ctx(), *tile); lib.prx_gemm_tile_override(_lib.tool_ctx(), -2, 0, stages)
    for rep in range(2):
        out = torch.full((M, N), float("nan"), device=DEV)
        st = torch.zeros(64, dtype=torch.float64, device=DEV)
        g = GemmArgs()
        g.A = A.data_ptr(); g.a_mode = 0 if conv is None else 1; g.lda = K if conv is None else conv[2]
        g.B = Bt.data_ptr(); g.ldb = K; g.M, g.N, g.K = M, N, K
        if conv is not None: g.H, g.W, g.Cin, g.up = conv
        g.alpha = 1.0; g.out_f32 = out.data_ptr(); g.ldc_f32 = N; g.bias_n = bias.data_ptr()
        if stats:
            g.gn_stats = st.data_ptr(); g.gn_gs = N // 32
        call("prx_k_gemm", g, ws, ws.numel(), s())
        torch.cuda.synchronize()
        outs.append((out, st))
    lib.prx_gemm_tile_override(_lib.tool_ctx(), 0, 0, 0); lib.prx_gemm_tile_override(_lib.tool_ctx(), -2, 0, 0)
    return outs

for (H, C, Co, up) in [(16, 512, 512, 0), (32, 512, 256, 1), (64, 256, 256, 0), (128, 128, 128, 0), (256, 128, 128, 1)]:
    for stats in (False, True):
        o = gemm_once(H * H, Co, 9 * C, (H, H, C, up), stats)
        report(f"conv {H}x{H} {C}->{Co} up{up} stats{int(stats)}", o[0], o[1])
for (M, N, K) in [(3200, 768, 3072), (3200, 3072, 768), (256, 512, 256), (256, 1536, 512)]:
    o = gemm_once(M, N, K)
    report(f"gemm {M}x{N}x{K}", o[0], o[1])

# --- norms ----------------------------------------------------------------------------------------------------
for (P, C) in [(256, 512), (4096, 256), (65536, 128)]:
    torch.manual_seed(2)
    x = torch.randn(1, P, C, device=DEV); gamma = torch.randn(C, device=DEV); beta = torch.randn(C, device=DEV)
    g = torch.randn(1, P, C, device=DEV)
    res = []
    for rep in range(2):
        stats = torch.zeros(64, dtype=torch.float64, device=DEV); ob = torch.empty(1, P, C, dtype=torch.bfloat16, device=DEV); of = torch.empty(1, P, C, device=DEV)
        call("prx_k_groupnorm_fwd", x, gamma, beta, stats, ob, of, 1, P, C, 1, 1e-6, s())
        bst = torch.zeros(64, dtype=torch.float64, device=DEV); dx = torch.empty(1, P, C, device=DEV)
        call("prx_k_groupnorm_bwd", g, x, gamma, beta, stats, bst, None, dx, 1, P, C, 1, 1e-6, s())
        torch.cuda.synchronize()
        res.append((stats, ob, of, bst, dx))
    report(f"groupnorm fwd+bwd P={P} C={C}", res[0], res[1])

# --- runners ---------------------------------------------------------------------------------------------------
for vq_name, clip_name, size, cutn in [("tiny_f4", "tiny-B/32", 64, 8), ("imagenet_f16_16384", "ViT-B/32", 256, 64)]:
    vcfg = weights.VQGAN_CONFIGS[vq_name]
    vp = weights.synthetic_vqgan_params(vcfg, seed=0)
    f = 2 ** (len(vcfg.ch_mult) - 1)
    vh = ops.VqganHandle(vcfg, vp, (size // f, size // f), device=DEV)
    torch.manual_seed(3)
    z0 = torch.randn(1, vcfg.z_channels, size // f, size // f, device=DEV)
    res = []
    for rep in range(2):
        z = z0.clone().requires_grad_(True)
        img = ops.vqgan_synth(z, vh)
        gi = torch.ones_like(img) * torch.linspace(-1, 1, img.numel(), device=DEV).reshape(img.shape)
        (dz,) = torch.autograd.grad(img, z, gi)
        torch.cuda.synchronize()
        res.append((img.detach(), dz))
    report(f"vqgan {vq_name} synth + backward", res[0], res[1])
    ccfg = weights.CLIP_CONFIGS[clip_name]
    cp = weights.synthetic_clip_vit_params(ccfg, seed=0)
    ch = ops.ClipVitHandle(ccfg, cp, max_batch=cutn, device=DEV)
    cut0 = torch.rand(cutn, 3, ccfg.input_resolution, ccfg.input_resolution, device=DEV)
    res = []
    for rep in range(2):
        cut = cut0.clone().requires_grad_(True)
        emb = ops.clip_encode_image(cut, ch)
        ge = torch.linspace(-1, 1, emb.numel(), device=DEV).reshape(emb.shape)
        (dc,) = torch.autograd.grad(emb, cut, ge)
        torch.cuda.synchronize()
        res.append((emb.detach(), dc))
    report(f"clip {clip_name} encode + backward", res[0], res[1])

# --- stage-by-stage bisect of the decoder forward ----------------------------------------------------------------
print("decoder forward, stage by stage (two passes, same z):")
vcfg = weights.VQGAN_CONFIGS["imagenet_f16_16384"]
vp = weights.synthetic_vqgan_params(vcfg, seed=0)
vh = ops.VqganHandle(vcfg, vp, (16, 16), device=DEV)
torch.manual_seed(3)
z = torch.randn(1, 256, 16, 16, device=DEV)
dump = []
for rep in range(2):
    img = ops.vqgan_synth(z, vh)
    bufs = []
    stage = -2
    while True:
        dst = torch.zeros(65536 * 128, device=DEV)
        n = call("prx_vqgan_debug_stage", vh.h, stage, dst, dst.numel(), s())
        if n is None or n < 0:
            break
        bufs.append(dst[:n].clone())
        stage += 1
    torch.cuda.synchronize()
    dump.append(bufs)
for i, (a, b) in enumerate(zip(dump[0], dump[1])):
    eq = torch.equal(a, b)
    rel = 0.0 if eq else ((a.double() - b.double()).norm() / (a.double().norm() + 1e-300)).item()
    print(f"  stage {i - 2:3d} n={a.numel():9d} {'bit-exact' if eq else f'DIFFERS rel {rel:.3e}'}")
