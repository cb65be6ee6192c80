"""The per-iteration loop of the reference (`train`, pixray.py:1436-1512; `ascend_txt`, 1243-1406;
`do_synth_and_filter`, 1203-1241; `rebuild_optimisers`, 520-555; `checkdrop`, 1091-1109) as an object
instead of module globals (pixray.py:1022-1063).

A `Session` is generic over its parts: it only relies on the duck-typed plugin surface of SURVEY.md §8b
(drawer.synth / get_z / clip_z / get_opts, perceptor.encode_image, MakeCutouts(out), Prompt(embeds),
LossInterface.get_loss, FilterInterface.forward).  The product factory (`pixray_amd.api`) fills it with the
HIP-backed parts only; tests may assemble one from other parts (e.g. the CPU oracle) to exercise the host
logic without a GPU.

Multi-GPU: the cutout batch is sharded over the ranks of `group`; every rank holds the same
z and replicates the drawer; the gradient of the loss w.r.t. the synthesised image is all-reduced (SUM)
before it enters the drawer's backward, so every rank then computes the identical, exact dL/dz and takes
the identical optimiser step.  (The reduction is done on dL/d(image) rather than on dL/dz because
ClampWithGrad's backward, vqgan.py:76-79, is not linear in the incoming gradient.)
"""
from __future__ import annotations

import functools
import operator
import types
from typing import Dict, List, Optional, Sequence

import torch
import torch.nn.functional as F

from .prompt import Prompt


def spherical_dist_loss(x, y):
    """pixray.py:262-265 (used by the z regularisers, 1344-1356)"""
    x = F.normalize(x, dim=-1)
    y = F.normalize(y, dim=-1)
    return (x - y).norm(dim=-1).div(2).arcsin().pow(2).mul(2)


class _GatherShards(torch.autograd.Function):
    """all-gather of the ranks' cutout (or embedding) shards along dim 0 for batch-coupled custom losses (SURVEY.md §8e:
    `SaturationLoss` takes a std over all cutout pixels, `AestheticLoss` sizes its target by num_cuts).  Every rank then
    evaluates the same full-batch loss; its backward keeps only the gradient of the rank's own shard, so the
    all-reduce of dL/d(image) that follows adds the shards' contributions up exactly once."""

    @staticmethod
    def forward(ctx, x, group, rank, world):
        import torch.distributed as dist
        parts = [torch.empty_like(x) for _ in range(world)]
        dist.all_gather(parts, x.contiguous(), group=group)
        ctx.rank, ctx.n = rank, x.shape[0]
        return torch.cat(parts, 0)

    @staticmethod
    def backward(ctx, g):
        return g[ctx.rank * ctx.n:(ctx.rank + 1) * ctx.n].contiguous(), None, None, None


def needs_full_batch(loss) -> bool:
    """A custom loss that couples the whole cutout batch (a statistic over all cutouts, a target sized by `num_cuts`, ...)
    declares it with a class or instance attribute `needs_full_batch = True`.  Under cutout sharding
    such a loss is scored on the all-gathered batch; every other loss is scored on the local shard with weight / world.
    The reference's own batch-coupled plugins (`SaturationLoss`: std over all cutout pixels, Losses/SaturationLoss.py:19-28;
    `AestheticLoss`: target sized by num_cuts, Losses/AestheticLoss.py:26; `ResmemLoss`) predate the attribute, so
    `plugins.add_custom_loss` / `mark_full_batch` set it on those classes when they are registered, and `Session.__init__`
    resolves it ONCE for instances handed over directly (`resolve_full_batch`) -- there is no matching by class name at
    scoring time."""
    return bool(getattr(loss, "needs_full_batch", False))


def resolve_full_batch(loss):
    """Decide the batch-coupling flag of a custom-loss instance once, when a Session takes it: an explicit
    `needs_full_batch` attribute (class or instance, True or False) wins; otherwise an instance of one of the reference's
    own batch-coupled classes (by class name anywhere in its MRO) is marked, so that an unregistered
    `SaturationLoss()` passed straight to `Session(custom_losses=...)` is not silently scored per shard."""
    if not hasattr(loss, "needs_full_batch") and any(c.__name__ in REFERENCE_FULL_BATCH_LOSSES for c in type(loss).__mro__):
        loss.needs_full_batch = True
    return loss


# reference plugin classes that are batch-coupled but cannot declare it themselves (see needs_full_batch)
REFERENCE_FULL_BATCH_LOSSES = ("SaturationLoss", "AestheticLoss", "ResmemLoss")


def mark_full_batch(loss_or_class):
    """Declare a (reference) loss class or instance batch-coupled; returns it.  Called when a plugin is REGISTERED."""
    loss_or_class.needs_full_batch = True
    return loss_or_class


class HipAdam(torch.optim.Optimizer):
    """`optim.Adam([z], lr)` (pixray.py:539) on the fused HIP Adam(+clip_z) kernel. One fp32 tensor.

    The step-dependent scalars (lr / bias_correction1, sqrt(bias_correction2)) live in a small device tensor that
    `prepare_step()` refreshes from a pinned host buffer, so `step()` launches the same kernel with the same
    arguments every iteration and can be replayed from a captured hipGraph."""

    def __init__(self, params, lr=0.2, betas=(0.9, 0.999), eps=1e-8, bounds=None):
        super().__init__(list(params), dict(lr=lr, betas=betas, eps=eps))
        self.bounds = bounds   # (zmin[C], zmax[C]) or None
        self._t = 0
        self._pending = False
        p = self.param_groups[0]["params"][0]
        from .cutouts import PinnedRing
        self._ring = PinnedRing((4,), torch.float32, p.device) if p.is_cuda else None
        self._hyper = self._ring.dev if self._ring is not None else torch.zeros(4, dtype=torch.float32, device=p.device)
        self.clamped_last_step = False   # did the last step() apply the fused clip_z bounds?

    @classmethod
    def from_adam(cls, opt: "torch.optim.Adam"):
        """A plain `torch.optim.Adam` over ONE group of device tensors (what a drawer plugin's `get_opts` returns: FftDrawer,
        fftdrawer.py:65-69) as the replayable fused kernel: same update rule, moments and step count carried over.  None
        when the optimiser uses anything the kernel does not implement."""
        if type(opt) is not torch.optim.Adam or len(opt.param_groups) != 1:
            return None
        g = opt.param_groups[0]
        if g.get("amsgrad") or g.get("maximize") or g.get("weight_decay", 0) != 0 or g.get("differentiable") or \
                isinstance(g["lr"], torch.Tensor) or not all(p.is_cuda and p.dtype == torch.float32 for p in g["params"]):
            return None
        new = cls(g["params"], lr=g["lr"], betas=tuple(g["betas"]), eps=g["eps"])
        steps = set()
        for p in g["params"]:
            st = opt.state.get(p)
            if st:
                steps.add(int(st["step"]))
                new.state[p] = {"step": int(st["step"]), "exp_avg": st["exp_avg"], "exp_avg_sq": st["exp_avg_sq"]}
        if len(steps) > 1 or (steps and len(new.state) != len(g["params"])):
            return None                      # parameters at different step counts: one shared bias correction cannot serve them
        new._t = steps.pop() if steps else 0
        return new

    def prepare_step(self):
        """host side of the next step(): advance t and stage {lr/bc1, sqrt(bc2)} (stream-ordered H2D)"""
        g = self.param_groups[0]
        self._t += 1
        b1, b2 = g["betas"]
        bc1 = 1.0 - b1 ** self._t
        bc2 = 1.0 - b2 ** self._t
        # through a ring of pinned buffers: the host may be iterations ahead of the queued H2D copies
        self._ring.stage(torch.tensor([g["lr"] / bc1, bc2 ** 0.5, 0.0, 0.0], dtype=torch.float32))
        self._pending = True

    @torch.no_grad()
    def step(self, closure=None):
        from . import ops
        if not self._pending:
            self.prepare_step()
        self._pending = False
        self.clamped_last_step = False
        for group in self.param_groups:
            for p in group["params"]:
                if p.grad is None:
                    continue
                st = self.state[p]
                if not st:
                    st["step"] = 0
                    st["exp_avg"] = torch.zeros_like(p)
                    st["exp_avg_sq"] = torch.zeros_like(p)
                elif int(st["step"]) != self._t - 1:
                    # state was set from outside (tests teacher-force the moments): follow its step count
                    self._t = int(st["step"])
                    self.prepare_step()
                    self._pending = False
                st["step"] = self._t
                zmin, zmax = self.bounds if self.bounds is not None else (None, None)
                ops.adam_clamp_step_dev(p, st["exp_avg"], st["exp_avg_sq"], p.grad.contiguous(), zmin, zmax, self._hyper,
                                        group["betas"], group["eps"])
                self.clamped_last_step = zmin is not None


class Session:
    def __init__(self, drawer, perceptors: Dict[str, object], cutouts: Dict[int, object],
                 prompts: Dict[str, Sequence[object]], *, learning_rate: float = 0.2, iterations: int = 250,
                 batches: int = 1, learning_rate_drops: Sequence[int] = (), custom_losses: Sequence[dict] = (),
                 filters: Sequence[dict] = (), args=None, init_weight: float = 0.0, init_weight_dist: float = 0.0,
                 z_orig=None, optimiser_factory=None, seed: int = 0, group=None, rank: int = 0, world_size: int = 1,
                 auto_stop: bool = False, image_prompts: Optional[Dict[str, Sequence[torch.Tensor]]] = None,
                 image_prompt_weight: Optional[float] = None, image_prompt_shuffle: bool = False,
                 z_labels: Sequence[torch.Tensor] = (), image_label_weight: float = 1.0, init_weight_pix: float = 0.0,
                 init_weight_cos: float = 0.0, init_image_tensor: Optional[torch.Tensor] = None,
                 spot_prompts: Optional[Dict[str, Sequence[object]]] = None,
                 spot_prompts_off: Optional[Dict[str, Sequence[object]]] = None,
                 overlay_image=None, overlay_every: int = 10, overlay_offset: int = 0, overlay_until: Optional[int] = None,
                 overlay_alpha: Optional[int] = None, prompt_factory=None, loss_globals: Optional[dict] = None, comm=None):
        self.drawer = drawer
        # optional C-ABI exchange (pixray_amd.comm.OneShotComm over `prx_allreduce_grad`): carries the per-step all-reduce of
        # dL/d(image) instead of torch.distributed; None = RCCL through torch.distributed
        self.comm = comm
        self.comm_check_every = 50        # iterations between prx_comm_status checks (each one synchronises the device)
        if comm is not None:              # the perceptors' two scalar-sized collectives ride the same exchange (ops._ClipEncodeFn)
            for p_ in perceptors.values():
                p_.comm = comm
        self.perceptors = perceptors
        self.cutoutsTable = cutouts
        self.cutoutSizeTable = {name: p.input_resolution for name, p in perceptors.items()}
        self.pmsTable = prompts
        self.learning_rate = learning_rate
        self.iterations = iterations
        self.batches = batches
        self.learning_rate_drops = list(learning_rate_drops)
        for _t in custom_losses:
            resolve_full_batch(_t["loss"])
        self.custom_losses = list(custom_losses)     # [{"loss": LossInterface, "weight": float}] (pixray.py:961-995)
        # Scheduling, not arithmetic: differentiate the perceptor terms first and the custom-loss terms in a second
        # backward() (gradients accumulate in the leaves exactly as in one pass; the shared drawer graph is walked twice).
        # autograd runs the most recently created nodes first, i.e. a host-heavy plugin (StyleLoss: thousands of small
        # launches) BEFORE the perceptor's backward, so the GPU idles while the host works through the plugin and the host
        # idles while the GPU runs the tower.  Measured neutral on configs[3] (the two already overlap across iterations); off.
        self.custom_backward_last = False
        self._n_path_terms = None                    # how many entries of ascend_txt()'s list precede the custom-loss terms
        self.filters = list(filters)                 # [{"filter": FilterInterface, "weight": float}] (650-669)
        self.args = args if args is not None else types.SimpleNamespace()
        self.init_weight = init_weight
        self.init_weight_dist = init_weight_dist
        self.z_orig = z_orig
        # image prompts (pixray.py:823-835, 1307-1336): target images [1,3,H,W] in [0,1], cut out every iteration with the
        # iteration's cached transforms so their cutouts line up with the current ones
        self.pmsImageTable = {k: list(v) for k, v in (image_prompts or {}).items()}
        self.image_prompt_weight = image_prompt_weight
        self.image_prompt_shuffle = image_prompt_shuffle
        self.z_labels = list(z_labels)                   # pixray.py:837-849, 1344-1349
        self.image_label_weight = image_label_weight
        self.init_weight_pix = init_weight_pix           # pixray.py:1363-1368
        self.init_weight_cos = init_weight_cos           # pixray.py:1370-1375
        self.init_image_tensor = init_image_tensor
        # spot prompts (pixray.py:1270-1292): Prompts scored on cutouts whose spot region (spotPmsTable) or its complement
        # (spotOffPmsTable) was blanked; the cutout tables need `.spot_masks`
        self.spotPmsTable = {k: list(v) for k, v in (spot_prompts or {}).items()}
        self.spotOffPmsTable = {k: list(v) for k, v in (spot_prompts_off or {}).items()}
        # overlay (pixray.py:731-747, 1408-1420, 1431-1434, 1457-1461): every `overlay_every` iterations the current image gets a
        # PIL RGBA image pasted over it (its alpha as mask) and is re-encoded into the drawer (`reapply_from_tensor`)
        self.overlay_image_rgba = None
        self.overlay_every, self.overlay_offset, self.overlay_until = overlay_every, overlay_offset, overlay_until
        if overlay_image is not None:
            from PIL import Image
            size = drawer.to_image().size                       # (sideX, sideY)
            img = overlay_image if isinstance(overlay_image, Image.Image) else Image.open(overlay_image)
            img = img.convert("RGBA").resize(size, Image.LANCZOS)
            if overlay_alpha:
                img.putalpha(overlay_alpha)
            self.overlay_image_rgba = img
        self.optimiser_factory = optimiser_factory
        # class of the throwaway Prompts built per iteration for image prompts (pixray.py:1331-1333); the HIP Prompt unless a
        # caller assembles the loop from other parts (CPU tests)
        self.prompt_factory = prompt_factory if prompt_factory is not None else Prompt
        self.group, self.rank, self.world_size = group, rank, world_size
        self.auto_stop = auto_stop
        self.cur_iteration = 0
        self.num_loss_drop = 0
        self.max_loss_drops = 2
        self.iter_drop_delay = 20
        self.best_loss = None
        self.best_iter = 0
        self.lossGlobals = dict(loss_globals or {})      # pixray.py:993-995 (plugins.setup_custom_losses builds it)
        self.rng = torch.Generator().manual_seed(seed)       # fill colour stream (python random in the reference)
        self.last_losses: Optional[List[torch.Tensor]] = None
        self.last_embeds = None
        self._graph = None
        self.graph_error = None          # why the last enable_graph() refused (None: not asked, or captured)
        self._host_ready = False
        self.cur_fill = 0.0
        self._shard_cutouts()
        self.opts = self.rebuild_optimisers()

    # ------------------------------------------------------------------ setup
    def _shard_cutouts(self):
        for mk in self.cutoutsTable.values():
            cutn = mk.cutn
            if self.world_size > 1:
                if cutn % self.world_size:
                    raise ValueError(f"num_cuts {cutn} must be divisible by the number of GPUs {self.world_size}")
                per = cutn // self.world_size
                mk.shard = (self.rank * per, (self.rank + 1) * per)
        # every Prompt scored on a shard of the cutout batch takes its mean over the GLOBAL (cutn x n_embed) pairs
        # (pixray.py:280): the main prompts and the spot / spot-off prompts alike
        for table in (self.pmsTable, self.spotPmsTable, self.spotOffPmsTable):
            for name, pms in table.items():
                cutn = self.cutoutsTable[self.cutoutSizeTable[name]].cutn
                for pm in pms:
                    if self.world_size > 1 and hasattr(pm, "denom"):
                        pm.denom = float(cutn * pm.embed.shape[0])

    def rebuild_optimisers(self):
        """pixray.py:520-555"""
        drop_divisor = 10 ** self.num_loss_drop
        new_opts = self.drawer.get_opts(drop_divisor)
        if new_opts is None:
            lr = self.learning_rate / drop_divisor
            z = self.drawer.get_z()
            if hasattr(self.drawer, "_fused_clamp"):
                self.drawer._fused_clamp = False         # set again below only if THIS optimiser clamps in its kernel
            if self.optimiser_factory is not None:
                new_opts = [self.optimiser_factory([z], lr)]
            elif z.is_cuda:
                bounds = None
                if hasattr(self.drawer, "_zmin_flat"):
                    bounds = (self.drawer._zmin_flat, self.drawer._zmax_flat)
                    self.drawer._fused_clamp = True
                new_opts = [HipAdam([z], lr=lr, bounds=bounds)]
            else:
                new_opts = [torch.optim.Adam([z], lr=lr)]
        return new_opts

    # ------------------------------------------------------------------ forward of one iteration
    def do_synth_and_filter(self, loss_list):
        """pixray.py:1203-1241; returns (image [1,3,H,W], alpha or None).  A drawer may return RGBA: with `args.transparent` the
        colours are composited over this iteration's gray fill (the random squash of pixray.py:1231-1236), otherwise the
        alpha channel is dropped."""
        out = self.drawer.synth(self.cur_iteration)
        for f in self.filters:
            out, new_losses = f["filter"](out)
            if not isinstance(new_losses, (list, tuple)):
                loss_list.append(f["weight"] * new_losses)
            else:
                loss_list += [f["weight"] * l for l in new_losses]
        alpha = None
        if out.shape[1] == 4:
            colors = out[:, 0:3, :, :]
            if getattr(self.args, "transparent", False):
                alpha = out[:, 3, :, :]
                out = alpha[:, None] * colors + (1 - alpha[:, None]) * float(self.cur_fill)
            else:
                out = colors
        return out, alpha

    def ascend_txt(self):
        """pixray.py:1243-1406"""
        it = self.cur_iteration
        if not self._host_ready:
            self._host_prep(it)
        self._host_ready = False
        result: List[torch.Tensor] = []
        out, img_alpha = self.do_synth_and_filter(result)
        if (self.world_size > 1 or getattr(self, "_force_hook_group", None) is not None) and out.requires_grad:
            import torch.distributed as dist

            def _allreduce(g):
                g = g.contiguous()
                if self.comm is not None and g.is_cuda:
                    if g.data_ptr() % 16 or g.numel() % 4:
                        g = g.clone()
                    self.comm.all_reduce_sum_(g)          # one-shot direct-write over the peers' IPC windows (csrc/comm.hip)
                else:
                    dist.all_reduce(g, op=dist.ReduceOp.SUM, group=self.group if self.group is not None else self._force_hook_group)
                return g
            out.register_hook(_allreduce)
        cur_cutouts = {}
        for size, mk in self.cutoutsTable.items():
            cur_cutouts[size] = mk(out)
        cur_spot, cur_spot_off = {}, {}
        if any(self.spotPmsTable.values()):                                                    # pixray.py:1270-1276
            for size, mk in self.cutoutsTable.items():
                cur_spot[size] = mk(out, spot=1)
        if any(self.spotOffPmsTable.values()):
            for size, mk in self.cutoutsTable.items():
                cur_spot_off[size] = mk(out, spot=0)
        iii = None
        for name, perceptor in self.perceptors.items():
            for table, cuts in ((self.spotPmsTable, cur_spot), (self.spotOffPmsTable, cur_spot_off)):   # pixray.py:1282-1292
                if table.get(name):
                    iii_s = perceptor.encode_image(cuts[self.cutoutSizeTable[name]]).float()
                    for prompt in table[name]:
                        result.append(prompt(iii_s))
            # image-prompt targets (pixray.py:1307-1336) are encoded BEFORE the differentiable pass on the current cutouts: a
            # perceptor handle keeps one forward's activations, so the last forward before backward() should be the one that
            # is differentiated (ops._ClipEncodeFn re-runs a forward whose activations were replaced, so any order is correct;
            # this one costs nothing).  The embeddings do not depend on the order.
            mk = self.cutoutsTable[self.cutoutSizeTable[name]]
            target_embeds = []
            for timg in self.pmsImageTable.get(name, ()):
                if self.image_prompt_shuffle:
                    mk.transforms = None
                with torch.no_grad():
                    embed = perceptor.encode_image(mk(timg)).float()
                    if self.world_size > 1:      # every rank needs ALL target embeddings: each cutout is compared with all
                        import torch.distributed as dist
                        parts = [torch.empty_like(embed) for _ in range(self.world_size)]
                        dist.all_gather(parts, embed.contiguous(), group=self.group)
                        embed = torch.cat(parts, 0)
                target_embeds.append(embed)
            iii = perceptor.encode_image(cur_cutouts[self.cutoutSizeTable[name]]).float()     # pixray.py:1295
            for prompt in self.pmsTable[name]:
                result.append(prompt(iii))                                                     # pixray.py:1297-1299
            # image prompts: throwaway Prompts from this iteration's cutouts of each target image (pixray.py:1331-1336)
            for embed in target_embeds:
                w = self.image_prompt_weight if self.image_prompt_weight is not None else 1.0
                pm = self.prompt_factory(embed, w).to(embed.device)
                if self.world_size > 1 and hasattr(pm, "denom"):
                    pm.denom = float(mk.cutn * embed.shape[0])
                result.append(pm(iii))
        for mk in self.cutoutsTable.values():
            mk.transforms = None                                                               # pixray.py:1339-1342
        # regularisers on z are replicated on every rank (they do not pass through `out`)
        for z_label in self.z_labels:                                                          # pixray.py:1344-1349
            f = self.drawer.get_z().reshape(1, -1)
            result.append(spherical_dist_loss(f, z_label.reshape(1, -1)) * self.image_label_weight)
        if self.init_weight and self.z_orig is not None:
            f = self.drawer.get_z().reshape(1, -1)
            result.append((spherical_dist_loss(f, self.z_orig.reshape(1, -1)) * self.init_weight)[0])
        if self.init_weight_dist and self.z_orig is not None:
            result.append(F.mse_loss(self.drawer.get_z(), self.z_orig) * self.init_weight_dist / 2)
        if self.init_weight_pix:                                                               # pixray.py:1363-1368
            if self.init_image_tensor is None:
                print("OOPS IIT is 0")
            else:
                w = self.init_weight_pix / self.world_size if self.world_size > 1 else self.init_weight_pix
                result.append(F.l1_loss(out, self.init_image_tensor.to(out.device)) * w / 2)
        if self.init_weight_cos and self.z_orig is not None:                                   # pixray.py:1370-1375
            f = self.drawer.get_z().reshape(1, -1)
            f2 = self.z_orig.reshape(1, -1)
            result.append(F.cosine_embedding_loss(f, f2, torch.ones_like(f[0])) * self.init_weight_cos)
        needed_globals = {"cur_iteration": it, "embeds": iii}                                  # pixray.py:1377-1381
        t_w = getattr(self.args, "transparent_weight", 0.0)
        if img_alpha is not None and t_w != 0:                                                 # pixray.py:1383-1386
            # replicated term: `img_alpha` is read off the drawer output BEFORE the composite, so its gradient does not pass
            # through the all-reduce hook on `out` -- every rank holds the whole term, like the z regularisers above
            result.append(t_w * torch.mean(img_alpha))
        full_cutouts = full_globals = None
        self._n_path_terms = len(result)
        for t in self.custom_losses:
            w = t["weight"] / self.world_size if self.world_size > 1 else t["weight"]
            cuts, glb = cur_cutouts, needed_globals
            if self.world_size > 1 and needs_full_batch(t["loss"]):
                # batch-coupled loss: every rank scores the FULL cutout batch (gathered, differentiable through the rank's
                # own shard only), so its weight is not divided by the world size.  Such a loss must not also read `out`.
                if full_cutouts is None:
                    gather = lambda x: _GatherShards.apply(x, self.group, self.rank, self.world_size)
                    full_cutouts = {size: gather(c) for size, c in cur_cutouts.items()}
                    full_globals = dict(needed_globals, embeds=gather(iii) if iii is not None else None)
                cuts, glb, w = full_cutouts, full_globals, t["weight"]
            new_losses = t["loss"].get_loss(cuts, out, self.args, globals=glb,
                                            lossGlobals=self.lossGlobals)
            if not isinstance(new_losses, (list, tuple)):
                result.append(w * new_losses)
            else:
                result += [w * l for l in new_losses]
        self.last_embeds = iii
        return result

    def _host_prep(self, it):
        """Host side of an iteration: the random draws the reference makes in Python / kornia (fill colour
        pixray.py:1255-1258, padding-mode parity 1250-1253, augmentation parameters) and their staging to the device."""
        fill = float(torch.rand((), generator=self.rng, dtype=torch.float64))
        self.cur_fill = fill
        for mk in self.cutoutsTable.values():
            if hasattr(mk, "prepare"):
                mk.prepare(iteration=it, fill=fill)
            else:
                mk.iteration, mk.fill = it, fill
        # custom losses with a host half (StyleLoss: numpy sampling tables).  Only once static buffers are on: in plain eager
        # use a plugin draws inside get_loss, where the reference draws -- the order of numpy's global stream across plugins
        # is then the reference's by construction.
        for t in self.custom_losses:
            if getattr(t["loss"], "graph_capturable", False) and hasattr(t["loss"], "host_prep"):
                t["loss"].host_prep(self.args, it)
        self._host_ready = True

    def _device_step(self):
        """Device side of train(): zero_grad -> ascend_txt -> backward -> step -> clip_z (no host decisions inside,
        so it can be captured once and replayed)."""
        for opt in self.opts:
            opt.zero_grad(set_to_none=True)
        for i in range(self.batches):
            lossAll = self.ascend_txt()
            n_path = self._n_path_terms
            if self.custom_backward_last and self.custom_losses and n_path and 0 < n_path < len(lossAll):
                custom = [l for l in lossAll[n_path:] if isinstance(l, torch.Tensor) and l.requires_grad]
                sum(lossAll[:n_path]).backward(retain_graph=bool(custom))
                if custom:
                    sum(custom).backward()
            else:
                loss = functools.reduce(operator.add, lossAll)      # one add per extra term (sum() starts from 0: one launch more)
                loss.backward(gradient=self._unit_grad(loss))       # a resident 1.0 instead of a ones_like fill per iteration
            # values only: holding the loss tensors themselves would keep this iteration's autograd graph -- every Function's
            # ctx with its workspace and its runner handle -- alive until the NEXT iteration replaces them, i.e. a handle
            # could be destroyed (hipFree) in the middle of a later iteration, which a hipGraph capture does not survive
            self.last_losses = [l.detach() if isinstance(l, torch.Tensor) else l for l in lossAll]
            del lossAll
        for opt in self.opts:
            opt.step()
        self._clip_z()

    def _unit_grad(self, loss):
        u = getattr(self, "_unit", None)
        if u is None or u.device != loss.device or u.dtype != loss.dtype or u.shape != loss.shape:
            u = self._unit = torch.ones_like(loss)
        return u

    def _clip_z(self):
        """drawer.clip_z() (pixray.py:1487); a drawer with a fused Adam+clamp kernel skips its own clamp only when every
        optimiser reports that it did clamp in this step"""
        if hasattr(self.drawer, "_fused_clamp"):
            self.drawer._fused_clamp = bool(self.opts) and all(getattr(o, "clamped_last_step", False) for o in self.opts)
        self.drawer.clip_z()

    def _drop_graph(self):
        """back to eager launches: the next ascend_txt() must make its own host draws"""
        self._graph = None
        self._staged_for_replay = False
        self._host_ready = False
        for mk in self.cutoutsTable.values():
            if hasattr(mk, "_prepared"):
                mk._prepared = False

    # ------------------------------------------------------------------ hipGraph capture
    def _custom_graph_state(self, it):
        """what the custom losses bake into a captured iteration besides tensor values (e.g. StyleLoss's skip / every
        schedule): a replay is valid only while this is unchanged"""
        return tuple(t["loss"].graph_state(self.args, it) if hasattr(t["loss"], "graph_state") else None for t in self.custom_losses)

    def enable_graph(self, warmup: int = 3):
        """Capture the device side of one iteration (≈590 kernel launches for the headline; ≈7 000 for configs[3], whose
        StyleLoss plugin is otherwise bound by the host's launch rate) in a hipGraph and replay it from then on.
        Host-drawn inputs reach the graph through fixed device buffers (cutout descriptors, Adam scalars, a plugin's sampling
        tables: `host_prep`).  Falls back to eager launches when something in the session cannot be captured (foreign
        optimisers, plugins that do not declare `supports_graph_replay`, batches > 1, ...)."""
        import os
        from . import graph_replay_refusal
        why = graph_replay_refusal()
        if why:
            # ROCm 7.2: with packet capture on, a replayed graph's kernel arguments live in memory that a later hipMalloc may
            # be handed -- any fresh allocation between replays can corrupt them (tools/debug_capture.py).
            # The flag must have been in place BEFORE the runtime started: a value set after the first HIP call is never read
            return self._no_graph(why)
        if warmup < 1:
            # a plugin with host-drawn inputs (StyleLoss) stages them in host_prep only once it has seen one evaluation; capturing
            # a cold first evaluation would bake its draws and pinned uploads into the graph
            return self._no_graph("warmup >= 1 is required: plugins size their staging buffers during one eager iteration")
        if self.batches != 1 or self.auto_stop or not self.opts:
            return self._no_graph("batches > 1, auto_stop, or no optimiser")
        if (self.world_size > 1 or getattr(self, "_force_hook_group", None) is not None) and self.comm is None and \
                os.environ.get("PRX_GRAPH_WITH_COLLECTIVES") != "1":
            # the three collectives of the iteration would be captured with it.  On the C-ABI one-shot exchange (self.comm) each is
            # a plain kernel whose sequence number lives on the device (csrc/comm.hip): captured and replayed like any other
            # launch.  RCCL inside a capture has not been run on a node: eager there.
            return self._no_graph("world_size > 1 on torch.distributed collectives: not validated inside a capture "
                                  "(PRX_ONESHOT_ALLREDUCE=1 puts them on the capturable C-ABI exchange; PRX_GRAPH_WITH_COLLECTIVES=1 to try RCCL)")
        params = [p for o in self.opts for g in o.param_groups for p in g["params"]]
        if not params or not all(p.is_cuda for p in params):
            return self._no_graph("optimised tensors are not on a GPU")
        dev = params[0].device
        # the fused Adam kernel reads its step scalars from a fixed device buffer; a drawer plugin's plain torch Adam (FftDrawer)
        # keeps them on the host, so it is swapped for the kernel (same rule, state carried over) -- for a replayed session only
        opts = [o if isinstance(o, HipAdam) else HipAdam.from_adam(o) for o in self.opts]
        if any(o is None for o in opts):
            return self._no_graph("an optimiser the fused Adam kernel cannot stand in for")
        if getattr(self.args, "transparent", False):      # the RGBA squash uses this iteration's host-drawn gray as a constant
            return self._no_graph("--transparent draws a host-side gray per iteration")
        # image / spot prompts go through the cached-transform path, which stages a fresh descriptor table per call
        if any(self.pmsImageTable.values()) or any(self.spotPmsTable.values()) or any(self.spotOffPmsTable.values()):
            return self._no_graph("image / spot prompts stage a fresh descriptor table per call")
        if not all(getattr(t["loss"], "supports_graph_replay", False) for t in self.custom_losses):
            return self._no_graph("a custom loss does not declare supports_graph_replay")   # a plugin's get_loss may draw, upload or branch on the host: it has to say that it does not
        for mk in self.cutoutsTable.values():
            if not hasattr(mk, "enable_static_buffers") or getattr(mk, "fixed_params", None) is not None:
                return self._no_graph("a cutout module without static descriptor buffers (or with fixed parameters)")
        for mk in self.cutoutsTable.values():
            mk.enable_static_buffers(dev)
        for t in self.custom_losses:
            if hasattr(t["loss"], "enable_static_buffers"):
                t["loss"].enable_static_buffers(dev)
        self.opts = opts
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            for _ in range(warmup):
                # ordinary train() calls (eager: no graph yet), so learning-rate drops, the overlay schedule and the
                # iteration limit are honoured during the warm-up exactly as without a graph
                if not self.train():
                    break
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        if self.cur_iteration >= self.iterations or self.cur_iteration in self.learning_rate_drops or \
                self.apply_overlay(self.cur_iteration) or not all(isinstance(o, HipAdam) for o in self.opts):
            return self._no_graph("the next iteration is not a plain one")     # the next iteration is not a plain one: stay eager (call enable_graph again later)
        self._host_prep(self.cur_iteration)
        for o in self.opts:
            o.prepare_step()
        graph = torch.cuda.CUDAGraph()
        try:
            with torch.cuda.graph(graph):
                self._device_step()
        except Exception as e:           # an op that cannot be captured (a sync, a pageable copy, a hipFree, ...)
            reason = f"capture failed: {type(e).__name__}: {str(e).splitlines()[0] if str(e) else ''}"
            try:                         # is the device usable again (an invalidated capture may stay open on the capture stream)?
                torch.cuda.synchronize(dev)
                torch.zeros(1).to(dev).cpu()
            except Exception as e2:
                raise RuntimeError(f"hipGraph {reason}; the aborted capture left the device unusable in this process "
                                   f"({type(e2).__name__}): run without enable_graph()") from e
            import warnings
            warnings.warn(f"hipGraph {reason}; staying on eager launches")
            for o in self.opts:          # the aborted capture consumed the staged step: rewind it, the eager step stages again
                o._t -= 1
                o._pending = False
                for st in o.state.values():
                    st["step"] = o._t
            self._drop_graph()
            return self._no_graph(reason)
        # the capture pass does not execute: its staged inputs (cutout descriptors, Adam scalars, sampling tables) are
        # consumed by the first replay, i.e. by the next train() call, which therefore must not draw again
        self._graph = graph
        self._graph_state = self._custom_graph_state(self.cur_iteration)
        self._staged_for_replay = True
        self.graph_error = None
        return True

    def _no_graph(self, reason: str) -> bool:
        """enable_graph's refusals, kept for the caller (`graph_error`)"""
        self.graph_error = reason
        return False

    def apply_overlay(self, cur_it: int) -> bool:
        """pixray.py:1431-1434"""
        return self.overlay_image_rgba is not None and (cur_it % self.overlay_every) == self.overlay_offset and \
            (self.overlay_until is None or cur_it < self.overlay_until)

    def re_average_z(self):
        """pixray.py:1408-1420: current image -> paste the overlay through its alpha -> back into the drawer (for the VQGAN
        drawer that is the HIP encoder + nearest-code lookup); the optimiser state is kept, as in the reference"""
        import numpy as np
        from PIL import Image
        cur = self.drawer.to_image().convert("RGB")
        size = cur.size
        cur.paste(self.overlay_image_rgba, (0, 0), mask=self.overlay_image_rgba)
        cur = cur.resize(size, Image.LANCZOS)
        t = torch.from_numpy(np.asarray(cur, dtype=np.float32) / 255.0).permute(2, 0, 1).unsqueeze(0)
        z = self.drawer.get_z()
        self.drawer.reapply_from_tensor((t.to(z.device) if z is not None else t) * 2 - 1)

    # ------------------------------------------------------------------ one optimiser step
    def train(self, cur_it: Optional[int] = None) -> bool:
        """pixray.py:1436-1512 (image saving and the animation ring are outside the hot path)"""
        if cur_it is None:
            cur_it = self.cur_iteration
        self.cur_iteration = cur_it
        rebuild = False
        if self.comm is not None and cur_it % self.comm_check_every == self.comm_check_every - 1:
            self.comm.check()            # a timed-out wait of the one-shot exchange poisoned a gradient with NaN: fail loudly (synchronises)
        if cur_it < self.iterations:
            if self.apply_overlay(cur_it):
                self.re_average_z()
            if cur_it in self.learning_rate_drops:
                rebuild = True
            if self._graph is not None and not getattr(self, "_staged_for_replay", False) and \
                    self._graph_state != self._custom_graph_state(cur_it):
                self._drop_graph()       # a plugin's schedule changed what the captured iteration baked in: eager from here
            if self._graph is not None:
                if getattr(self, "_staged_for_replay", False):
                    self._staged_for_replay = False     # inputs staged by enable_graph for the captured iteration
                else:
                    self._host_prep(cur_it)
                    for opt in self.opts:
                        opt.prepare_step()
                self._graph.replay()
                self._host_ready = False            # consumed by the replay (ascend_txt does not run during a replay)
                for mk in self.cutoutsTable.values():
                    if hasattr(mk, "_prepared"):
                        mk._prepared = False
                for opt in self.opts:               # keep the Python-side step count in sync with the replayed kernels
                    for st in opt.state.values():
                        st["step"] = opt._t
            elif self.auto_stop or self.batches != 1:
                for opt in self.opts:
                    opt.zero_grad()
                for i in range(self.batches):
                    lossAll = self.ascend_txt()
                    if i == 0 and not rebuild and self.auto_stop:
                        rebuild = self.checkdrop(cur_it, lossAll)
                    loss = sum(lossAll)
                    loss.backward()
                    self.last_losses = [l.detach() if isinstance(l, torch.Tensor) else l for l in lossAll]
                for opt in self.opts:
                    opt.step()
                self._clip_z()
            else:
                self._device_step()
        if cur_it == self.iterations:
            return False
        if rebuild:
            self.num_loss_drop += 1
            if self.num_loss_drop > self.max_loss_drops:
                return False
            self.best_iter = cur_it
            self.best_loss = None
            self.opts = self.rebuild_optimisers()
            self._drop_graph()               # captured kernels reference the old optimiser state: back to eager
        self.cur_iteration = cur_it + 1
        return True

    def do_run(self, return_display: bool = False, display_every: int = 20) -> bool:
        """The outer loop of `do_run` without the animation ring (pixray.py:1614-1631): train until `iterations` (or until the
        learning-rate drops run out); with `return_display` it returns False every `display_every` iterations so that a
        serving front end can show progress and call again (cogrun.py:47-52); True when the run is complete."""
        keep_going = True
        while keep_going:
            it = self.cur_iteration
            try:
                keep_going = self.train(it)
            except RuntimeError as e:
                print("Oops: runtime error: ", e)
                print("Try reducing --num-cuts to save memory")
                raise
            if it == self.iterations:
                break
            self.cur_iteration = it + 1
            if keep_going and return_display and self.cur_iteration % display_every == 0:
                return False
        return True

    def checkdrop(self, it, losses) -> bool:
        """pixray.py:1091-1109 -- the comparison forces a device->host sync, exactly as in the reference; it only
        runs when auto_stop is enabled."""
        loss_sum = float(sum(losses))
        if self.best_loss is None or loss_sum < self.best_loss:
            self.best_loss, self.best_iter = loss_sum, it
            return False
        return (it - self.best_iter) >= self.iter_drop_delay
