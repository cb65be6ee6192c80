"""`VqganDrawer` with the reference's drawer surface (/root/reference/vqgan.py:83-214) on the HIP VQGAN runner.

Call sites the loop relies on: add_settings (pixray.py:2070), __init__(settings) (612),
load_model(settings, device) (613), get_num_resolutions (614), init_from_tensor (718-727),
reapply_from_tensor (1420), get_z_from_tensor (843), get_opts (525), synth(cur_iteration) (1206),
to_image (1413), clip_z (1487), get_z / set_z / get_z_copy (537, 1104, 1346).

The VQGAN *encoder* (init / overlay / label images: init_from_tensor, reapply_from_tensor, get_z_from_tensor,
vqgan.py:174-185) is the first of the "next" rows: it runs on the same HIP engine, forward only, and
is built lazily on first use.  `init_from_tensor(None)` draws a random z instead (the reference would fail there).
Checkpoint download is out of scope; `settings.vqgan_state_dict` takes a taming state dict (pixray_amd/checkpoints.py).
"""
import torch

from . import ops
from .interfaces import DrawingInterface
from .weights import VQGAN_CONFIGS, synthetic_vqgan_params, synthetic_vqgan_encoder_params, vqgan_encoder_param_shapes


class VqganDrawer(DrawingInterface):
    @staticmethod
    def add_settings(parser):
        parser.add_argument("--vqgan_model", type=str, help="VQGAN model", default='imagenet_f16_16384', dest='vqgan_model')
        parser.add_argument("--vqgan_config", type=str, help="VQGAN config", default=None, dest='vqgan_config')
        parser.add_argument("--vqgan_checkpoint", type=str, help="VQGAN checkpoint", default=None, dest='vqgan_checkpoint')
        return parser

    def __init__(self, settings):
        super(DrawingInterface, self).__init__()
        self.vqgan_model = getattr(settings, "vqgan_model", "imagenet_f16_16384")
        self.size = tuple(getattr(settings, "size", (256, 256)))          # (width, height) as in the reference
        self.state_dict = getattr(settings, "vqgan_state_dict", None)     # taming state dict, if the caller has one
        self.weight_seed = getattr(settings, "weight_seed", 0)
        self.precision = getattr(settings, "precision", None)             # None = "fp16" | "bf16" | "f32" (exact-f32 MFMA parity mode)
        self.z = None
        # GumbelVQ (vqgan.py:149-153): set by load_taming from the yaml's target, or by the caller of a ready state dict
        self.gumbel = bool(getattr(settings, "vqgan_gumbel", False))
        self._fused_clamp = False

    def load_model(self, settings, device):
        # vqgan.py:100-108: explicit --vqgan_config / --vqgan_checkpoint, else models/vqgan_<name>.{yaml,ckpt}; files that
        # are not there are NOT downloaded (vqgan.py:110-113 would wget them) -- without them the seeded synthetic weights
        # of the named architecture are used, or the caller's `settings.vqgan_state_dict`
        import os
        cfg_path = getattr(settings, "vqgan_config", None) or f"models/vqgan_{self.vqgan_model}.yaml"
        ckpt_path = getattr(settings, "vqgan_checkpoint", None) or f"models/vqgan_{self.vqgan_model}.ckpt"
        explicit = bool(getattr(settings, "vqgan_config", None) or getattr(settings, "vqgan_checkpoint", None))
        if self.state_dict is None and os.path.exists(cfg_path) and os.path.exists(ckpt_path):
            from .checkpoints import load_taming
            self.cfg, self.state_dict, self.gumbel = load_taming(cfg_path, ckpt_path)
        elif explicit and self.state_dict is None:
            raise FileNotFoundError(f"VQGAN config / checkpoint not found: {cfg_path}, {ckpt_path} (nothing is downloaded)")
        elif self.vqgan_model not in VQGAN_CONFIGS:
            raise ValueError(f"unknown model type: {self.vqgan_model}")
        else:
            self.cfg = VQGAN_CONFIGS[self.vqgan_model]
        self.device = torch.device(device)
        params = self.state_dict if self.state_dict is not None else synthetic_vqgan_params(self.cfg, self.weight_seed)
        f = 2 ** (self.cfg.num_resolutions - 1)
        w, h = self.size
        if w % f or h % f:
            # the reference's drawer takes its latent size from the init tensor, which do_init has already rounded down to
            # a multiple of 2^(num_resolutions-1) (pixray.py:621-626); this drawer fixes its size here, so it rounds the same way
            w, h = (w // f) * f, (h // f) * f
            if w == 0 or h == 0:
                raise ValueError(f"size {self.size} is smaller than one latent cell ({f} x {f} pixels)")
            self.size = (w, h)
        self.latent_hw = (h // f, w // f)
        self._params = params
        self.handle = ops.VqganHandle(self.cfg, params, self.latent_hw, self.device, precision=self.precision)
        self.e_dim = self.cfg.embed_dim
        self.n_toks = self.cfg.n_embed
        zmin, zmax = self.handle.z_bounds()                                # vqgan.py:155-158
        self.z_min = zmin[None, :, None, None]
        self.z_max = zmax[None, :, None, None]
        self._zmin_flat, self._zmax_flat = zmin, zmax

    def get_opts(self, decay_divisor):
        return None

    def rand_init(self, seed=1):
        g = torch.Generator().manual_seed(seed)
        z = torch.randn(1, self.cfg.z_channels, *self.latent_hw, generator=g).to(self.device)
        self.z = z.maximum(self.z_min).minimum(self.z_max).requires_grad_(True)

    # -- encoder side (vqgan.py:174-185): `z, *_ = self.model.encode(t)`, t in [-1,1] [1,3,H,W] ---------------------------
    def _encoder(self):
        enc = getattr(self, "_enc_handle", None)
        if enc is None:
            names = vqgan_encoder_param_shapes(self.cfg).keys()
            if self.state_dict is not None and all(k in self.state_dict for k in names):
                params = self.state_dict
            elif self.state_dict is not None:
                raise KeyError("the VQGAN state dict has no `encoder.*` / `quant_conv.*` entries: cannot encode an image")
            else:       # no checkpoints offline: seeded random encoder sharing the decoder's codebook
                params = synthetic_vqgan_encoder_params(self.cfg, self.weight_seed, codebook=self._params["quantize.embedding.weight"])
            w, h = self.size
            enc = self._enc_handle = ops.VqganEncHandle(self.cfg, params, (h, w), self.device)
        return enc

    def _encode(self, t):
        if t.dim() != 4 or t.shape[0] != 1 or t.shape[1] != 3:
            raise ValueError(f"expected an image tensor [1,3,H,W] in [-1,1], got {tuple(t.shape)}")
        if self.gumbel:
            # taming's GumbelVQ.encode (vqgan.py:175-185): Encoder -> quant_conv -> GumbelQuantize, which in eval mode is
            # one_hot(argmax(proj(h) + Gumbel noise)) @ embed -- `F.gumbel_softmax(logits, tau, dim=1, hard=True)` draws its noise
            # in eval mode too.  The encoder and quant_conv run on the HIP runner (the latent BEFORE its nearest-code step); the
            # 1x1 logits projection, the noise (torch's generator, as the reference's) and the code lookup are three tensor ops
            # on a [n_embed, h*w] matrix, once per init / overlay -- not on the iteration's path
            sd = self.state_dict or {}
            if "quantize.proj.weight" not in sd:
                raise KeyError("the GumbelVQ state dict has no `quantize.proj.*` entries: cannot encode an image "
                               "(checkpoints.load_taming keeps them)")
            _, _, pre = ops.vqgan_encode(t.to(self.device), self._encoder(), return_pre=True)
            wp = sd["quantize.proj.weight"].to(self.device).float().reshape(self.cfg.n_embed, -1)
            bp = sd["quantize.proj.bias"].to(self.device).float()
            logits = torch.einsum("nc,bchw->bnhw", wp, pre) + bp.view(1, -1, 1, 1)
            gumbels = -torch.empty_like(logits).exponential_().log()                 # F.gumbel_softmax's own draw
            idx = (logits + gumbels).argmax(dim=1)                                     # hard=True: the forward value is the one-hot
            cb = self._params["quantize.embedding.weight"].to(self.device)
            self.last_encode_indices = idx.reshape(-1).int()
            return cb[idx].permute(0, 3, 1, 2).contiguous()
        z, idx = ops.vqgan_encode(t.to(self.device), self._encoder())
        self.last_encode_indices = idx
        return z

    def init_from_tensor(self, init_tensor):
        if init_tensor is None:
            self.rand_init()
            return
        self.z = self._encode(init_tensor)
        self.z.requires_grad_(True)

    def reapply_from_tensor(self, new_tensor):
        new_z = self._encode(new_tensor)
        with torch.no_grad():
            self.z.copy_(new_z)

    def get_z_from_tensor(self, ref_tensor):
        return self._encode(ref_tensor)

    def get_num_resolutions(self):
        return self.cfg.num_resolutions

    def synth(self, cur_iteration):
        return ops.vqgan_synth(self.z, self.handle, True)

    @torch.no_grad()
    def to_image(self):
        from PIL import Image
        out = self.synth(None)
        arr = out[0].mul(255).clamp(0, 255).byte().permute(1, 2, 0).cpu().numpy()
        return Image.fromarray(arr)

    def clip_z(self):
        if self._fused_clamp:
            return          # the fused Adam+clamp kernel already applied the bounds this step
        with torch.no_grad():
            self.z.copy_(self.z.maximum(self.z_min).minimum(self.z_max))

    def get_z(self):
        return self.z

    def set_z(self, new_z):
        with torch.no_grad():
            return self.z.copy_(new_z)

    def get_z_copy(self):
        return self.z.clone()
