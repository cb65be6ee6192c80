"""The product's HIP kernels, run on the CPU.

`tools/hipemu` compiles the SAME sources as libprx_hip.so (pixray_amd/csrc/*.hip, all of them) for the host against a stand-in
<hip/hip_runtime.h>: every work-item is a fiber, __syncthreads / wave shuffles / ballots suspend until the workgroup / wave
has arrived, `v_mfma_*` (32x32x16, 16x16x32, 32x32x2 f32) and `global_load_lds` are emulated with the hardware's register
layouts, atomics are plain read-modify-writes.  `tests/_emu.py` points the ctypes loader at that library, so the same Python
wrappers and the same C ABI drive it on CPU tensors.  These tests are the GPU suite's own test functions (`-m gpu`, imported and
called with their device switched to "cpu") on a subset sized for a CPU: they check the kernels' LOGIC -- indexing, tiling,
swizzles, epilogues, reductions, the runners' data flow -- against the oracle in a container that has no GPU.  They say nothing
about speed, and the product never loads this library (the HIP path still fails loudly without a device).

What the emulation cannot see: anything that depends on real concurrency between waves (missing barriers that happen to work
sequentially), register / LDS capacity, and instruction-level hazards."""
import os
import shutil
import sys

import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)

import _emu  # noqa: E402

pytestmark = pytest.mark.skipif(not os.path.exists("/opt/rocm/lib/llvm/bin/clang++") or shutil.which("make") is None,
                                reason="needs the ROCm host clang++ and make to build tools/hipemu")


@pytest.fixture(scope="module")
def emu():
    with _emu.enable() as lib:
        import test_kernels_gpu as tk
        import test_path_gpu as tp
        for m in (tk, tp):
            m.DEV = "cpu"
        yield types_ns(lib=lib, tk=tk, tp=tp)
        for m in (tk, tp):
            m.DEV = "cuda"


def types_ns(**kw):
    import types
    return types.SimpleNamespace(**kw)


def test_emulated_library_exports_the_whole_c_abi(emu):
    from pixray_amd import _lib
    assert emu.lib.prx_abi_version() == 3
    assert len(_lib._protos) >= 79 and all(hasattr(emu.lib, name) for name in _lib._protos)


# ------------------------------------------------------------------------------------------------ GEMM engine
def test_gemm_kernels_4wave_fit_and_8phase(emu):
    tk = emu.tk
    tk.test_gemm_rowmajor_bf16(100, 72, 136, 1)                                   # ragged M / N, K not a multiple of 64 (4-wave, 32x32x16)
    tk.test_gemm_rowmajor_bf16(64, 64, 64, 0)                                     # the register-staged variant
    tk.test_gemm_forced_tiles_stages_splitk((64, 64), 3, 3)                       # 3-deep DMA ring + split-K workspace + reduce
    tk.test_gemm_8phase_kernel_ragged_edges_and_epilogues("fp16")                 # 256 x 256 8-phase kernel, staggered wave rows


def test_fit_kernel_implicit_conv_and_groupnorm_epilogue_sums(emu):
    emu.tk.test_gemm_fit_tiles_implicit_conv_and_groupnorm_sums((64, 64), "fp16")  # 16x16x32 MFMA, K groups through LDS, GN / GN-backward sums
    emu.tk.test_gemm_conv3x3(16, 16, 128, 128, 0, 1, 1)
    emu.tk.test_gemm_conv3x3_stride2_down(12, 20, 64, 72, 2)


def test_fit_kernels_with_compile_time_epilogues_match_the_generic_kernel(emu):
    """every specialised epilogue (OUT16, RES16, GELU, DGELU, GN, RES16_GN, GNB) of a tower tile, a K-group tower tile and decoder
    tiles with 1 / 2 / 8 K groups: bit-identical 16-bit outputs to the generic kernel of the same launch (sigmoid forms: one half
    ulp), spare rows untouched, GroupNorm sums equal; the launch counter proves which kernel ran (tests/test_kernels_gpu.py)"""
    small = [(200, 136, 512), (81, 264, 1024)]
    convs = [(8, 12, 64, 128, 0, 1), (8, 8, 512, 128, 1, 1)]
    for tile in [(160, 192), (80, 128), (128, 128), (64, 64), (16, 32)]:
        ran = emu.tk.fit_spec_vs_generic(tile, shapes=small, conv_shapes=convs)
        assert {"out16", "res16", "gelu", "dgelu"} <= ran and (tile[0] % 80 == 0 or {"gn", "res16_gn", "gnb"} <= ran), (tile, ran)
    ran = emu.tk.fit_spec_vs_generic((256, 256), shapes=[(300, 264, 256), (257, 136, 384)])       # the 8-phase kernel's tower epilogues
    assert {"out16", "res16", "gelu", "dgelu"} <= ran


def test_row_streaming_kernel_vs_tiled_kernels(emu):
    """gemmrow.hip on the emulator: the slab permutation, the transposed MFMA, the K tail, ragged M, every runner epilogue, both formats"""
    emu.tk.row_kernel_checks([("fp16", 16405, 320, 80), ("fp16", 8200, 640, 160), ("bf16", 20483, 256, 64), ("fp16", 4200, 1280, 320),
                              ("fp16", 16403, 80, 200), ("bf16", 13140, 80, 80), ("fp16", 6600, 160, 640)])


def test_row_streaming_conv3x3_vs_conv2d_and_tiled_kernels(emu):
    """gemmrowconv_kernel.h on the emulator: tap decode per 16-byte chunk, border zeros, the half tile of N = 40, both epilogues"""
    emu.tk.row_conv_checks([("fp16", 2, 30, 34, 80, 80), ("fp16", 1, 40, 45, 40, 40), ("bf16", 1, 33, 40, 40, 80), ("fp16", 1, 48, 40, 80, 40),
                            ("fp16", 1, 24, 27, 160, 160)])


def test_gemm_engine_random_shapes_every_kernel_family(emu):
    """300 random row-major products and 300 random implicit convolutions (tests/_emu_fuzz.py): ragged shapes against every tile,
    strided operands and outputs, fp32 A converted on load, every fused epilogue, every kernel family (planner's choice, forced
    fit tiles, register-staged kernels, split-K, the 8-phase tile).  Unsupported combinations have
    to be refused, not computed wrong.  (Found: a forced / big-tile 256 x 128 launch with an fp32 A fell through to the 64 x 64
    register-staged kernel on the 256 x 128 grid and left most of the output unwritten.)"""
    import _emu_fuzz
    assert _emu_fuzz.gemm_cases(emu.lib, 1, 300) == []
    bad, rejected = _emu_fuzz.conv_cases(emu.lib, 1, 300)
    assert bad == [] and rejected > 0


def test_gemm_groupnorm_epilogues_random_shapes(emu):
    """the decoder's fused GroupNorm sums (forward, and a GroupNorm-backward's of a dgrad-shaped launch) over random conv shapes
    and kernel families incl. split-K; spare output rows and a second block of sums behind the first must stay untouched"""
    import _emu_fuzz
    assert _emu_fuzz.gn_cases(emu.lib, 0, 150) == []


def test_device_allocations_are_red_zoned_and_the_decoder_fits_a_wide_latent(emu):
    """tools/hipemu gives every hipMalloc a red zone checked after each launch.  Found with it: a decoder whose z_channels exceed
    its widest block (256 latent channels into a 128-channel network) wrote conv_in's input gradient past the gradient stream
    buffers (sized from the blocks alone).  Runs that configuration; an overrun aborts the process."""
    import test_path_gpu as tp
    from pixray_amd import weights
    weights.VQGAN_CONFIGS["wide-latent"] = weights.VqganConfig(ch=128, ch_mult=(1,), num_res_blocks=2, attn_resolutions=(16,),
                                                               resolution=16, z_channels=256, embed_dim=256, n_embed=64)
    try:
        ref, out, gref, gd, idx_ref, idx = tp._vqgan_case("wide-latent", (2, 3), 3)
        assert tp.rel_l2(out, ref) < 2e-2 and tp.rel_l2(gd, gref) < 5e-2 and torch.equal(idx, idx_ref)
    finally:
        del weights.VQGAN_CONFIGS["wide-latent"]
    # two attention resolutions that are both visited cannot be expressed in the C ABI: refused by name, not built wrong
    from pixray_amd import ops
    cfg = weights.VqganConfig(ch=128, ch_mult=(1, 2), num_res_blocks=1, attn_resolutions=(16, 32), resolution=32, z_channels=128,
                              embed_dim=128, n_embed=64)
    with pytest.raises(ValueError, match="attention at 2 resolutions"):
        ops.VqganHandle(cfg, weights.synthetic_vqgan_params(cfg, 0), (2, 2), "cpu")
    assert ops._single_attn_resolution(weights.VqganConfig()) == 16
    assert ops._single_attn_resolution(weights.VqganConfig(attn_resolutions=(16, 32), resolution=32, ch_mult=(1,))) == 32


def test_runners_on_random_geometries(emu):
    """a few random geometries per runner (tests/_emu_fuzz.py; `python tests/_emu_fuzz.py vit|vqgan|runners SEED N` for long sweeps):
    ViT (patch / grid / width / depth / batch), VQGAN decoder (depth, multipliers, latent channels, attention placement, latents
    down to 1 x 1), ModifiedResNet (exact-f32 and fp16) and the VQGAN encoder, each against the fp32 oracle under the red zones"""
    import _emu_fuzz
    assert _emu_fuzz.vit_cases(emu.lib, 2, 2) == []
    assert _emu_fuzz.vqgan_cases(emu.lib, 2, 4) == []
    assert _emu_fuzz.resnet_encoder_cases(emu.lib, 2, 6) == []


def test_handles_refuse_weights_whose_shapes_do_not_fit_the_configuration(emu):
    """the C ABI takes bare pointers: a tensor smaller than the configuration implies would be read out of bounds on the device
    (AddressSanitizer on the emulated create shows exactly that), so the Python handles check names and shapes first"""
    from pixray_amd import ops, weights
    cfg = weights.VQGAN_CONFIGS["tiny_f4"]
    p = dict(weights.synthetic_vqgan_params(cfg, 0))
    name = next(k for k in p if k.endswith("conv_in.weight"))
    bad = dict(p); bad[name] = bad[name][:, :-1].contiguous()
    with pytest.raises(ValueError, match="the configuration implies"):
        ops.VqganHandle(cfg, bad, (4, 4), "cpu")
    del bad[name]
    with pytest.raises(KeyError, match="no tensor named"):
        ops.VqganHandle(cfg, bad, (4, 4), "cpu")
    vcfg = weights.CLIP_CONFIGS["tiny-B/32"]
    vp = dict(weights.synthetic_clip_vit_params(vcfg, 0))
    vp["conv1.weight"] = vp["conv1.weight"][:, :, :16].contiguous()
    with pytest.raises(ValueError, match="conv1.weight"):
        ops.ClipVitHandle(vcfg, vp, max_batch=1, device="cpu")
    g = dict(weights.synthetic_vgg16_params(0)); g["features.0.bias"] = g["features.0.bias"][:32]
    with pytest.raises(ValueError, match="features.0.bias"):
        ops.Vgg16Handle(g, (32, 32), torch.device("cpu"), precision="f32")
    rcfg = weights.CLIP_RESNET_CONFIGS["tiny-RN"]
    rp = dict(weights.synthetic_clip_resnet_params(rcfg, 0)); rp.pop("layer1.0.bn1.running_var")
    with pytest.raises(KeyError, match="running_var"):
        ops.ClipResNetHandle(rcfg, rp, max_batch=1, device="cpu")


def test_cutout_noise_drawn_in_the_kernel(emu):
    """the in-kernel Philox noise of the stage-B cutout kernel on the emulated kernels (the checks live in tests/test_path_gpu.py,
    where the device runs them too and compares its draws with these)"""
    emu.tp.cutout_noise_checks()


def test_perceptor_preprocess_surface_on_the_emulated_tower(emu):
    """CLIP_Base.preprocess / apply_preprocess=False (slip.py:58-66) against the fused default path"""
    emu.tp.perceptor_preprocess_checks()


def test_gumbel_vq_encode_on_the_emulated_encoder(emu):
    """GumbelVQ.encode (vqgan.py:174-185) behind VqganDrawer.init_from_tensor"""
    emu.tp.gumbel_vq_encode_checks()


def test_dma_rings_under_the_eager_completion_model_too(emu):
    """the default model of the emulation retires a `global_load_lds` only at the counted wait that covers it (the latest legal
    completion); the rings of the 4-wave, fit and 8-phase kernels once more with the data landing AT ISSUE (the earliest), which is
    what exposes a stage that is overwritten while another wave still reads it"""
    try:
        emu.lib.hipemu_set_dma_eager(1)
        emu.tk.test_gemm_forced_tiles_stages_splitk((64, 64), 3, 1)
        emu.tk.test_gemm_8phase_kernel_ragged_edges_and_epilogues("bf16")
        emu.tk.test_gemm_fit_tiles_implicit_conv_and_groupnorm_sums((64, 64), "bf16")
    finally:
        emu.lib.hipemu_set_dma_eager(0)


def test_norms_attention_layout_and_image_head_kernels(emu):
    tk = emu.tk
    tk.test_groupnorm_fwd_bwd(256, 512, 1)
    tk.test_layernorm_fwd_bwd(257, 1024)
    tk.test_transpose_softmax_upsample_layout()
    tk.test_image_head()
    tk.test_mha_fwd_bwd(3, 64)
    tk.test_mha_general_fwd_bwd(2, 65, 1.0)


def test_strotss_and_hypercolumn_kernels(emu):
    emu.tk.test_strotss_relaxed_emd_kernels_match_the_composed_torch_expression(300, 777, False)
    emu.tk.test_strotss_relaxed_emd_kernels_match_the_composed_torch_expression(640, 200, True)
    emu.tk.test_hypercolumns_match_the_composed_torch_expression()


# ------------------------------------------------------------------------------------------------ the path's own ops
def test_cutouts_forward_and_backward_kernels_vs_oracle(emu):
    tp = emu.tp
    tp.test_make_cutouts_vs_oracle(5, 64, 40, 3, "noise")
    tp.test_make_cutouts_non_square_canvas_vs_oracle(64, 112, 64, 1)
    tp.test_make_cutouts_spot_masks_vs_oracle()
    tp.test_make_cutouts_cached_transform_path_vs_oracle(0)


def test_prompt_vq_and_adam_kernels_vs_oracle(emu):
    tp = emu.tp
    tp.test_prompt_loss_vs_oracle(64, 3, 512, -0.5, float("-inf"))
    tp.test_adam_clamp_vs_torch()
    tp.test_vq_nearest_vs_oracle()


def test_clip_vit_runner_vs_oracle(emu):
    emu.tp.test_clip_vit_vs_oracle("tiny-B/32", 4)


def test_the_other_runners_vs_oracle(emu):
    """VQGAN encoder (stride-2 gather), CLIP text tower (causal general-T attention), CLIP ModifiedResNet (ReLU-mask epilogues,
    attention pool), the cached-transform cutout path and the sharded cutout table: the runners of SURVEY.md section 8(f1, f2)"""
    tp = emu.tp
    tp.test_vqgan_encode_vs_oracle("tiny_f4", (40, 56))
    tp.test_clip_text_tower_vs_oracle("tiny-B/32", 5)
    tp.test_clip_resnet_vs_oracle("tiny-RN", 3, "fp16")
    tp.test_resnet_lean_streams_ab()
    tp.test_make_cutouts_shard_matches_full()
    tp.test_cutout_align_corners_convention_is_a_descriptor_field("crop_align_corners")


def test_vgg16_extractor_exact_mode_vs_oracle(emu):
    """the StyleLoss plugin's VGG16 runner (13 implicit convolutions with the ReLU in the epilogue, argmax max-pools, the dgrad
    chain) in the exact-f32 mode: features and input gradient at fp32 round-off"""
    import test_f32_mode_gpu as tf
    tf.DEV = "cpu"
    try:
        tf.test_vgg16_f32_vs_oracle(64, 48)
    finally:
        tf.DEV = "cuda"


def test_fp32_operand_fit_kernels_and_their_groupnorm_sums(emu):
    """the exact mode's decoder products on the fp32-operand fit kernels (gemmfit_f32.hip, v_mfma_f32_16x16x4_f32): implicit
    convolutions on a 1-, a 2- and an 8-K-group tile with bias + fp32 residual, the next GroupNorm's sums and a GroupNorm-backward's
    sums in the epilogue, against float64; and a shape those kernels do not take (Cin % 32 != 0): the engine runs the 4-wave fp32
    kernel and the norm kernels' own statistics pass instead -- same sums"""
    emu.tk.fit_f32_conv_and_stats_checks([(128, 128), (64, 64), (16, 32)])


def test_one_iteration_of_the_reduced_configuration_vs_oracle(emu):
    """synth (VQ + VQGAN decode + clamp) -> cutouts -> CLIP ViT -> prompt loss -> backward to z: the smoke test's toy graph,
    IEEE-half operands, every kernel emulated.  The numbers reproduce the GPU's (profiles/: dz rel-L2 2.4e-2 on this graph)."""
    from oracle import step_ref
    r = step_ref.compare_one_iteration(vqgan_model="tiny_f4", clip_model="tiny-B/32", size=(64, 64), cutn=8, seed=0, precision="fp16",
                                       device="cpu")
    assert r["indices_equal"] and r["loss_abs_err"] < 2e-3
    assert r["image_rel_l2"] < 2e-3 and r["embeds_rel_l2"] < 3e-3
    assert r["dz_rel_l2"] < 3e-2 and r["dz_cosine"] > 0.999, r


@pytest.mark.parametrize("size", [(96, 64), (45, 32), (64, 33)])
def test_fft_drawer_hip_path_vs_the_explicit_dft_oracle(emu, size):
    """csrc/fft_drawer.hip (pack, two exact-f32 GEMM stages against twiddle matrices, std / colour / sigmoid tail;
    the transposed chain backwards) against oracle/fft_ref.py: image and d/d(spectrum), even and odd canvas sizes"""
    from oracle import fft_ref
    from pixray_amd import ops
    h = ops.FftDrawerHandle(size[0], size[1])
    p = fft_ref.rand_init(size, 3)
    assert h.freq_columns == p.shape[3]
    q = p.detach().clone().requires_grad_(True)
    a, b = ops.fft_synth(q, h, 0.9), fft_ref.synth(p, size)
    assert float((a - b).detach().abs().max()) < 2e-6
    proj = torch.randn(a.shape, generator=torch.Generator().manual_seed(1))
    (ga,) = torch.autograd.grad((a * proj).sum(), q)
    (gb,) = torch.autograd.grad((b * proj).sum(), p)
    assert float((ga - gb).norm() / gb.norm()) < 5e-6
    if size[0] % 2:                                      # the surplus frequency column of an odd width: no gradient
        assert float(ga[..., -1, :].abs().max()) == 0.0


def test_fft_drawer_plugin_switches_to_the_hip_path(emu, monkeypatch):
    import types
    from oracle import fft_ref
    from pixray_amd.fft_drawer import FftDrawer
    st = types.SimpleNamespace(size=(40, 24), fft_use="fft", fft_decay=1.5, fft_lrate=0.3, weight_seed=2, fft_hip_force=True)
    dr = FftDrawer(st)
    dr.load_model(st, "cpu")
    dr.init_from_tensor(None)
    assert dr.hip
    img = dr.synth(0)
    ref = fft_ref.synth(dr.params[0].detach(), (40, 24))
    assert float((img - ref).detach().abs().max()) < 2e-6
    img.sum().backward()
    assert dr.params[0].grad is not None and torch.isfinite(dr.params[0].grad).all()
    opt = dr.get_opts()[0]
    opt.step()                                            # the plugin's own Adam over the spectrum (fftdrawer.py:65-69)


def test_results_do_not_depend_on_the_order_waves_and_workgroups_run_in(emu):
    """the product claims bit-reproducible gradients (no float atomics between waves; per-wave accumulator planes and fixed
    summation orders in the cutout scatter, K groups and split-K partials added in group order): the emulation runs the waves of
    every workgroup, and the workgroups of every grid, once first-to-last and once last-to-first (lanes of a wave keep their
    order, as the hardware's LDS atomics do) -- cutouts forward / backward and a tower forward / backward must not change by a bit"""
    from pixray_amd import cutouts as pc, ops, weights
    g = torch.Generator().manual_seed(5)
    img = torch.rand(1, 3, 40, 40, generator=g)
    prm = pc.sample_cutout_params(6, 64, g, iteration=0)
    prm["noise"] = torch.randn(6, 3, 64, 64, generator=g)
    gout = torch.randn(6, 3, 64, 64, generator=g)
    cfg = weights.CLIP_CONFIGS["tiny-B/32"]
    params = weights.synthetic_clip_vit_params(cfg, 2)
    cuts = torch.rand(4, 3, 224, 224, generator=g)
    res = {}
    try:
        for order in (0, 1):
            emu.lib.hipemu_set_reverse_order(order)
            mk = pc.MakeCutouts(64, 6)
            mk.fixed_params = prm
            x = img.clone().requires_grad_(True)
            out = mk(x)
            (gx,) = torch.autograd.grad(out, x, gout)
            h = ops.ClipVitHandle(cfg, params, 4, "cpu", precision="fp16")
            c = cuts.clone().requires_grad_(True)
            e = ops.clip_encode_image(c, h)
            (gc,) = torch.autograd.grad(e, c, torch.ones_like(e))
            res[order] = (out.detach().clone(), gx.clone(), e.detach().clone(), gc.clone())
    finally:
        emu.lib.hipemu_set_reverse_order(0)
    for a, b in zip(res[0], res[1]):
        assert torch.equal(a, b)


# ------------------------------------------------------------------------------------------------ N > 1 on the emulated kernels
def _dist_worker(rank, world, port, cutn, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(2)
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    with _emu.enable():
        from pixray_amd import api
        sess = api.build_vqgan_clip_session(size=(64, 64), vqgan_model="tiny_f4", clip_model="tiny-B/32", num_cuts=cutn, seed=3, device="cpu",
                                            group=dist.group.WORLD if world > 1 else None, rank=rank, world_size=world, precision="fp16",
                                            learning_rate=0.1, iterations=10)
        # the same explicit draws (incl. the noise tensor, which a session otherwise draws per rank on the device) everywhere
        from pixray_amd import cutouts as pc
        for size, mk in sess.cutoutsTable.items():
            g = torch.Generator().manual_seed(77)
            prm = pc.sample_cutout_params(cutn, size, g, iteration=0, fill=0.5)
            prm["noise"] = torch.randn(cutn, 3, size, size, generator=g)
            mk.fixed_params = prm
        sess.train(0)
        z = sess.drawer.get_z()
        q.put((rank, z.detach().numpy().copy(), z.grad.detach().numpy().copy(), float(sum(l.detach() for l in sess.last_losses))))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def test_two_ranks_on_the_emulated_kernels_match_one_process():
    """SURVEY.md section 8(e) with the PRODUCT's parts end to end: two gloo ranks, each running the emulated HIP kernels on its
    half of the cutouts -- the batch-global min / max all-reduce in front of the tower, the fp64 renormalisation sums behind its
    backward, the global-mean prompt denominator, the dL/d(image) all-reduce in front of the replicated decoder backward, the same
    Adam step on every rank -- against one process on the whole batch."""
    import torch.multiprocessing as mp
    import test_dist_cpu as td
    _emu.build()
    ctx = mp.get_context("spawn")
    cutn = 4

    def run(world):
        q, port = ctx.Queue(), td._free_port()
        procs = [ctx.Process(target=_dist_worker, args=(r, world, port, cutn, q)) for r in range(world)]
        for p in procs:
            p.start()
        try:
            res = sorted(td._collect(q, procs, world, timeout=600), key=lambda t: t[0])
            for p in procs:
                p.join(timeout=120)
                assert p.exitcode == 0
            return res
        finally:
            for p in procs:
                if p.is_alive():
                    p.terminate()
    (_, z0, g0, l0), (_, z1, g1, l1) = run(2)
    ((_, zr, gr, lr),) = run(1)
    z0, g0, z1, g1, zr, gr = [torch.from_numpy(t) for t in (z0, g0, z1, g1, zr, gr)]
    assert torch.equal(z0, z1) and torch.equal(g0, g1), "ranks diverged"
    # the same cutouts through the same kernels; the tower's GEMMs see 100 instead of 200 token rows (other tiles, another
    # summation order in fp32) and the image gradient is summed across the ranks: half-precision round-off, not more
    rel = float((g0 - gr).norm() / gr.norm())
    assert rel < 5e-3, rel
    assert abs((l0 + l1) - lr) < 2e-3
    # the first Adam step moves every entry by +-lr = 0.1 (sign of its gradient): entries whose gradient is round-off around zero may go
    # the other way (a difference of up to 0.2), all the others land on the same value
    assert float((z0 - zr).abs().max()) < 0.21 and float(((z0 - zr).abs() > 1e-3).float().mean()) < 0.02


# ------------------------------------------------------------------------------------------------ the real architectures (minutes)
_SLOW = pytest.mark.skipif(os.environ.get("PRX_EMU_SLOW", "0") != "1",
                           reason="full-depth runners on the CPU emulation take minutes: PRX_EMU_SLOW=1 (ViT-B/32 1 min, VQGAN 256^2 4 min, "
                                  "the headline iteration at 16 cutouts 10 min, seven session-level GPU tests 1 min each)")


@_SLOW
def test_full_depth_vit_b32_runner_vs_oracle(emu):
    emu.tp.test_clip_vit_vs_oracle("ViT-B/32", 8)


@_SLOW
def test_full_size_vqgan_decoder_256_vs_oracle(emu):
    """taming `imagenet_f16_16384` at its own size: z [1,256,16,16] -> image [1,3,256,256] and back (253 + 253 GFLOP through the
    emulated MFMAs), every decoder convolution on the fit kernels"""
    emu.tp.test_vqgan_synth_vs_oracle("imagenet_f16_16384", 16)


@_SLOW
def test_headline_configuration_one_iteration_16_cutouts(emu):
    """BASELINE.json configs[1] (VQGAN 256^2 + ViT-B/32, 2 prompts) at 16 cutouts, every kernel emulated"""
    from oracle import step_ref
    r = step_ref.compare_one_iteration(precision="fp16", cutn=16, device="cpu")
    assert r["indices_equal"] and r["loss_abs_err"] < 1e-3 and r["dz_rel_l2"] < 2e-2 and r["dz_cosine"] > 0.999, r


@_SLOW
@pytest.mark.parametrize("name", ["test_widescreen_one_iteration_vs_oracle", "test_custom_loss_and_filter_compose_with_native_ops",
                                  "test_image_prompts_and_init_regularisers_vs_oracle", "test_two_perceptors_share_one_decoder_pass",
                                  "test_text_prompt_vector_prompt_and_init_image_session", "test_overlay_image_goes_through_the_hip_encoder",
                                  "test_cutout_shards_reproduce_the_unsharded_gradient"])
def test_session_level_gpu_tests_on_the_emulated_kernels(emu, monkeypatch, tmp_path, name):
    """the GPU suite's session-level tests (tests/test_e2e_gpu.py: image prompts and init regularisers, two perceptors on one decoder
    pass, text / vector prompts + init image, the overlay through the HIP encoder, custom losses and filters, the widescreen canvas,
    cutout shards) with the session built on the emulated device; about a minute each"""
    import inspect
    import test_e2e_gpu as te
    from pixray_amd import api
    build = api.build_vqgan_clip_session
    monkeypatch.setattr(api, "build_vqgan_clip_session", lambda *a, **k: build(*a, **{**k, "device": "cpu"}))
    monkeypatch.setattr(te, "DEV", "cpu")
    fn = getattr(te, name)
    kw = {}
    if "precision" in inspect.signature(fn).parameters:
        kw["precision"] = "fp16"
    if "tmp_path" in inspect.signature(fn).parameters:
        kw["tmp_path"] = tmp_path
    fn(**kw)


@_SLOW
def test_front_end_on_the_emulated_hip_parts(emu, monkeypatch, tmp_path):
    """tests/test_zz_frontend_gpu.py's settings -> PNG run (HIP VQGAN drawer, cutouts, CLIP tower, prompt loss; 7 iterations of the
    reduced configuration) with the session on the emulated device; the optimiser is torch's Adam here (the fused kernel is chosen
    for device tensors), everything else is the product path"""
    import test_zz_frontend_gpu as tf
    monkeypatch.setattr(tf, "DEV", "cpu")
    tf.test_settings_to_png_on_the_hip_path(tmp_path)
