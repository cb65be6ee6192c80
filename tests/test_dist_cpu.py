"""CPU tests (no GPU): the N>1 path on the `gloo` backend, world_size 2.

The Session shards the cutout batch over the ranks, all-reduces dL/d(image) before the drawer's backward and takes
the same optimiser step everywhere.  Here the per-rank arithmetic is supplied by the CPU oracle parts (test
infrastructure); what is under test is the product's host logic: shard assignment, global-mean loss scaling, the
image-gradient all-reduce hook, and the batch-global min/max protocol.  The sharded result must equal the
single-process result on the full batch."""
import os
import socket
import sys
import types

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))


def _collect(q, procs, world, timeout=240):
    """results of all ranks, failing fast when a worker died instead of waiting out the queue timeout"""
    import queue as _queue
    import time
    out, t0 = [], time.time()
    while len(out) < world:
        try:
            out.append(q.get(timeout=2))
        except _queue.Empty:
            dead = [p for p in procs if p.exitcode not in (None, 0)]
            assert not dead, f"worker exited with {dead[0].exitcode}"
            assert time.time() - t0 < timeout, "workers timed out"
    return out


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class SaturationLoss:
    """same arithmetic as the reference plugin of that name (Losses/SaturationLoss.py:15-30): a colourfulness score from
    std / mean over ALL cutout pixels -- batch-coupled, so sharded runs must gather the batch for it.  It says so with the
    documented plugin attribute (engine.needs_full_batch); class names are not consulted."""
    needs_full_batch = True

    def get_loss(self, cur_cutouts, out, args, globals=None, lossGlobals=None):
        res = []
        for _, cutouts in cur_cutouts.items():
            px = cutouts.permute(0, 2, 3, 1).reshape(-1, 3)
            rg = px[:, 0] - px[:, 1]
            yb = 0.5 * (px[:, 0] + px[:, 1]) - px[:, 2]
            rg_std, rg_mean = torch.std_mean(rg)
            yb_std, yb_mean = torch.std_mean(yb)
            res.append(-(torch.sqrt(rg_std ** 2 + yb_std ** 2) + 0.3 * torch.sqrt(rg_mean ** 2 + yb_mean ** 2)) / 10.0)
        return res


class GlobalMeanPrompt(torch.nn.Module):
    """the oracle Prompt with the global-mean denominator the Session sets on sharded runs (what the HIP Prompt does natively)"""

    def __init__(self, embed, weight=1.0, stop=float("-inf")):
        super().__init__()
        self.embed, self.weight, self.stop, self.denom = embed, weight, stop, None

    def forward(self, x):
        from oracle import prompt_ref
        full = prompt_ref.Prompt(self.embed, self.weight, self.stop)(x)                     # mean over the local pairs
        if self.denom is None:
            return full
        return full * (x.shape[0] * self.embed.shape[0]) / self.denom                        # rescale to the global mean


def _rgba_drawer_class():
    from pixray_amd.pixel_grid_drawer import PixelGridDrawer

    class RgbaGridDrawer(PixelGridDrawer):
        """an RGBA drawer (pixray.py:1225-1241): the colours of the pixel grid plus an alpha channel derived from z"""

        def synth(self, cur_iteration):
            rgb = super().synth(cur_iteration)
            alpha = torch.sigmoid(4.0 * (rgb.mean(dim=1, keepdim=True) - 0.5))
            return torch.cat([rgb, alpha], dim=1)
    return RgbaGridDrawer


def _build(cutn, world, rank, group, coupled_loss=False, extras=False):
    from oracle import prompt_ref, step_ref
    from pixray_amd import cutouts as pc, weights
    from pixray_amd.engine import Session
    from pixray_amd.pixel_grid_drawer import PixelGridDrawer
    cfg = weights.CLIP_CONFIGS["tiny-B/32"]
    params = weights.synthetic_clip_vit_params(cfg, 1)
    st = types.SimpleNamespace(size=(96, 96), pixel_size=(12, 12), pixel_scale=None)
    drawer = (_rgba_drawer_class() if extras == "spot_rgba" else PixelGridDrawer)(st)
    drawer.load_model(st, "cpu")
    g = torch.Generator().manual_seed(7)
    drawer.init_from_tensor(torch.rand(1, 3, 96, 96, generator=g) * 2.6 - 1.3)     # some pixels start out of range
    perceptor = step_ref.OraclePerceptor(cfg, params, group=group)

    def sampler(iteration, fill):
        gg = torch.Generator().manual_seed(1000 + iteration)
        prm = pc.sample_cutout_params(cutn, 224, gg, iteration=iteration, fill=fill)
        prm["noise"] = torch.randn(cutn, 3, 224, 224, generator=gg)
        return prm
    mk = step_ref.OracleMakeCutouts(224, cutn, sampler)
    e = torch.randn(2, cfg.output_dim, generator=g)
    pm = prompt_ref.Prompt(e, 1.0, float("-inf"))
    pm.denom = None

    custom = [{"loss": SaturationLoss(), "weight": 3.0}] if coupled_loss else []
    if coupled_loss == "unregistered":
        # the reference's own class arrives WITHOUT the attribute and without going through plugins.add_custom_loss: only
        # its name says what it is (Session.__init__ -> engine.resolve_full_batch)
        bare = type("SaturationLoss", (), {"get_loss": SaturationLoss.get_loss})
        assert not hasattr(bare, "needs_full_batch")
        custom = [{"loss": bare(), "weight": 3.0}]
    kw = {}
    if extras == "spot_rgba":
        # spot / spot-off prompts (pixray.py:1270-1292) on an RGBA drawer with the transparency term (1383-1386): the spot
        # tables take the GLOBAL mean denominator under sharding, and the alpha term is a replicated (undivided) one
        inside = torch.zeros(3, 224, 224, dtype=torch.bool)
        inside[:, 60:170, 40:150] = True
        mk.spot_masks = (inside, ~inside)
        e2 = torch.randn(1, cfg.output_dim, generator=g)
        e3 = torch.randn(2, cfg.output_dim, generator=g)
        kw = dict(spot_prompts={"tiny-B/32": [GlobalMeanPrompt(e2, 0.8)]}, spot_prompts_off={"tiny-B/32": [GlobalMeanPrompt(e3, 0.6)]},
                  args=types.SimpleNamespace(transparent=True, transparent_weight=0.35))
    elif extras:    # image prompt (cached transforms, embeddings all-gathered) + the z / pixel regularisers of pixray.py:1344-1375
        target = torch.rand(1, 3, 96, 96, generator=g)
        init = torch.rand(1, 3, 96, 96, generator=g)
        kw = dict(image_prompts={"tiny-B/32": [target]}, image_prompt_weight=0.7, z_orig=drawer.get_z_copy().detach() * 0.9,
                  init_weight=0.3, init_weight_dist=0.2, init_weight_pix=0.4, init_weight_cos=0.1, init_image_tensor=init)
    return Session(drawer, {"tiny-B/32": perceptor}, {224: mk}, {"tiny-B/32": [GlobalMeanPrompt(e)]}, learning_rate=0.05,
                   iterations=10, seed=3, world_size=world, rank=rank, group=group, custom_losses=custom,
                   prompt_factory=GlobalMeanPrompt, **kw)


def _worker(rank, world, port, cutn, q, coupled_loss=False, extras=False):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    sess = _build(cutn, world, rank, dist.group.WORLD, coupled_loss, extras)
    for it in range(2):
        sess.train(it)
    q.put((rank, sess.drawer.get_z().detach().numpy().copy(), sess.drawer.get_z().grad.detach().numpy().copy(),
           float(sum(l.detach() for l in sess.last_losses))))      # numpy: pickled by value
    dist.barrier()
    dist.destroy_process_group()


def _run_world(world, cutn, coupled_loss=False, extras=False):
    """spawn `world` gloo ranks, return their (rank, z, grad, loss) sorted by rank; never leaves a worker behind"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, cutn, q, coupled_loss, extras)) for r in range(world)]
    for p in procs:
        p.start()
    try:
        res = sorted(_collect(q, procs, world), key=lambda t: t[0])
        for p in procs:
            p.join(timeout=60)
            assert p.exitcode == 0
        return res
    finally:
        for p in procs:
            if p.is_alive():
                p.terminate()


def test_world2_gloo_matches_single_process():
    cutn, world = 4, 2
    res = _run_world(world, cutn)
    # single process, full batch
    torch.set_num_threads(4)
    ref = _build(cutn, 1, 0, None)
    for it in range(2):
        ref.train(it)
    z_ref, g_ref = ref.drawer.get_z().detach(), ref.drawer.get_z().grad.detach()
    (_, z0, g0, l0), (_, z1, g1, l1) = res
    z0, g0, z1, g1 = [torch.from_numpy(t) for t in (z0, g0, z1, g1)]
    assert torch.equal(z0, z1) and torch.equal(g0, g1), "ranks diverged"
    # identical math, different summation order across the shard boundary: fp32 round-off only
    assert (g0 - g_ref).abs().max().item() < 1e-5 * max(1.0, g_ref.abs().max().item())
    assert (z0 - z_ref).abs().max().item() < 1e-5
    # each rank's loss is its share of the global mean; the shares add up to the full-batch loss
    assert abs((l0 + l1) - float(sum(l.detach() for l in ref.last_losses))) < 1e-5


def test_world4_gloo_matches_single_process():
    """the same with four ranks and one cutout each (the zoom / wide split and the per-rank descriptor slices are by global
    index, so every rank count that divides the cutout count must give the single-process iteration)"""
    cutn, world = 4, 4
    res = _run_world(world, cutn)
    torch.set_num_threads(4)
    ref = _build(cutn, 1, 0, None)
    for it in range(2):
        ref.train(it)
    z_ref, g_ref = ref.drawer.get_z().detach(), ref.drawer.get_z().grad.detach()
    zs = [torch.from_numpy(r[1]) for r in res]
    gs = [torch.from_numpy(r[2]) for r in res]
    for z, g in zip(zs[1:], gs[1:]):
        assert torch.equal(z, zs[0]) and torch.equal(g, gs[0]), "ranks diverged"
    assert (gs[0] - g_ref).abs().max().item() < 1e-5 * max(1.0, g_ref.abs().max().item())
    assert (zs[0] - z_ref).abs().max().item() < 1e-5
    assert abs(sum(r[3] for r in res) - float(sum(l.detach() for l in ref.last_losses))) < 1e-5


def test_world2_batch_coupled_custom_loss_is_scored_on_the_gathered_batch():
    """a SaturationLoss-style plugin (std over all cutout pixels) on 2 ranks: the loop gathers the cutout shards for it,
    so z, its gradient and the loss value equal the single-process run (the per-shard std would not)"""
    cutn, world = 4, 2
    res = _run_world(world, cutn, True)
    torch.set_num_threads(4)
    ref = _build(cutn, 1, 0, None, True)
    for it in range(2):
        ref.train(it)
    z_ref, g_ref = ref.drawer.get_z().detach(), ref.drawer.get_z().grad.detach()
    (_, z0, g0, l0), (_, z1, g1, l1) = res
    z0, g0, z1, g1 = [torch.from_numpy(t) for t in (z0, g0, z1, g1)]
    assert torch.equal(z0, z1) and torch.equal(g0, g1), "ranks diverged"
    assert (g0 - g_ref).abs().max().item() < 1e-5 * max(1.0, g_ref.abs().max().item())
    assert (z0 - z_ref).abs().max().item() < 1e-5
    # the prompt shares add up; the coupled loss is the same full-batch value on both ranks (counted once)
    sat_ref = float(ref.last_losses[-1].detach())
    assert abs((l0 + l1) - (float(sum(l.detach() for l in ref.last_losses)) + sat_ref)) < 1e-5


def test_world2_unregistered_reference_named_loss_is_still_gathered():
    """a reference `SaturationLoss` instance handed straight to Session(custom_losses=...) -- never registered,
    no `needs_full_batch` attribute -- must not be scored per shard: the Session resolves the flag once from the class name"""
    cutn, world = 4, 2
    res = _run_world(world, cutn, "unregistered")
    torch.set_num_threads(4)
    ref = _build(cutn, 1, 0, None, "unregistered")
    for it in range(2):
        ref.train(it)
    z_ref, g_ref = ref.drawer.get_z().detach(), ref.drawer.get_z().grad.detach()
    (_, z0, g0, l0), (_, z1, g1, l1) = res
    z0, g0, z1, g1 = [torch.from_numpy(t) for t in (z0, g0, z1, g1)]
    assert torch.equal(z0, z1) and torch.equal(g0, g1), "ranks diverged"
    assert (g0 - g_ref).abs().max().item() < 1e-5 * max(1.0, g_ref.abs().max().item())
    assert (z0 - z_ref).abs().max().item() < 1e-5


def test_explicit_needs_full_batch_false_is_respected():
    from pixray_amd.engine import needs_full_batch, resolve_full_batch
    per_shard = type("SaturationLoss", (), {"needs_full_batch": False})()
    assert not needs_full_batch(resolve_full_batch(per_shard))
    assert needs_full_batch(resolve_full_batch(type("AestheticLoss", (), {})()))
    assert not needs_full_batch(resolve_full_batch(type("MyLoss", (), {})()))


def test_world2_image_prompts_and_regularisers_match_single_process():
    """image prompts (every rank compares its cutouts with ALL target embeddings: all-gather) and the z / pixel regularisers
    (replicated on z, split over the ranks on the image) on 2 ranks equal the single-process run"""
    cutn, world = 4, 2
    res = _run_world(world, cutn, False, True)
    torch.set_num_threads(4)
    ref = _build(cutn, 1, 0, None, False, True)
    for it in range(2):
        ref.train(it)
    z_ref, g_ref = ref.drawer.get_z().detach(), ref.drawer.get_z().grad.detach()
    (_, z0, g0, _), (_, z1, g1, _) = res
    z0, g0, z1, g1 = [torch.from_numpy(t) for t in (z0, g0, z1, g1)]
    assert torch.equal(z0, z1) and torch.equal(g0, g1), "ranks diverged"
    assert (g0 - g_ref).abs().max().item() < 1e-5 * max(1.0, g_ref.abs().max().item())
    assert (z0 - z_ref).abs().max().item() < 1e-5


def test_world2_spot_prompts_and_rgba_drawer_match_single_process():
    """spot / spot-off prompts (three differentiable encode_image calls per iteration) and an RGBA drawer with the
    transparency term on 2 ranks equal the single-process run: z, its gradient, and the summed loss"""
    cutn, world = 4, 2
    res = _run_world(world, cutn, False, "spot_rgba")
    torch.set_num_threads(4)
    ref = _build(cutn, 1, 0, None, False, "spot_rgba")
    for it in range(2):
        ref.train(it)
    z_ref, g_ref = ref.drawer.get_z().detach(), ref.drawer.get_z().grad.detach()
    (_, z0, g0, l0), (_, z1, g1, l1) = res
    z0, g0, z1, g1 = [torch.from_numpy(t) for t in (z0, g0, z1, g1)]
    assert torch.equal(z0, z1) and torch.equal(g0, g1), "ranks diverged"
    assert (g0 - g_ref).abs().max().item() < 1e-5 * max(1.0, g_ref.abs().max().item())
    assert (z0 - z_ref).abs().max().item() < 1e-5
    # prompt shares add up; the transparency term is replicated (both ranks hold the whole term): counted once
    alpha_ref = float(ref.last_losses[-1].detach())
    assert abs((l0 + l1) - (float(sum(l.detach() for l in ref.last_losses)) + alpha_ref)) < 1e-5


def test_cutn_must_divide_world_size():
    from pixray_amd.engine import Session
    with pytest.raises(ValueError):
        mk = types.SimpleNamespace(cutn=5, shard=None)
        Session(types.SimpleNamespace(get_opts=lambda d: [], get_z=lambda: None), {}, {224: mk}, {}, world_size=2, rank=0)
