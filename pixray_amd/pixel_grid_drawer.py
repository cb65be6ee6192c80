"""`PixelGridDrawer`: a grid of RGB cells rendered by nearest-neighbour replication.

This is the CPU-runnable plumbing drawer of BASELINE.json configs[0] ("pixeldrawer 16x16 grid"): for
integer cell sizes the reference's rect-grid PixelDrawer (diffvg, pixeldrawer.py:330-367) and its
FastPixelDrawer (fast_pixeldrawer.py:83-91) both reduce to `clamp_with_grad(nearest_upsample(z), 0, 1)`.
It is a drawer *plugin* written against the duck-typed drawer API, not part of the HIP
hot path; it runs on whatever device it is given.
"""
import torch
import torch.nn.functional as F

from .interfaces import DrawingInterface


class _SoftClamp01(torch.autograd.Function):
    """forward clamp(x,0,1); backward blocks only gradients that push further out of range
    (same rule as the reference's ClampWithGrad, vqgan.py:66-79)."""

    @staticmethod
    def forward(ctx, x):
        ctx.save_for_backward(x)
        return x.clamp(0, 1)

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        keep = (g * (x - x.clamp(0, 1))) >= 0
        return g * keep


class PixelGridDrawer(DrawingInterface):
    @staticmethod
    def add_settings(parser):
        parser.add_argument("--pixel_size", nargs=2, type=int, default=None, dest="pixel_size",
                            help="grid size (columns rows)")
        parser.add_argument("--pixel_scale", type=float, default=None, dest="pixel_scale", help="grid scale")
        return parser

    def __init__(self, settings):
        width, height = settings.size
        grid = getattr(settings, "pixel_size", None)
        if grid is None:
            grid = (40, 40) if width == height else ((40, 50) if width < height else (80, 45))
        cols, rows = grid
        scale = getattr(settings, "pixel_scale", None)
        if scale:
            cols, rows = int(cols / scale), int(rows / scale)
        self.grid_hw = (min(rows, height), min(cols, width))
        self.canvas_hw = (height, width)
        self.device = torch.device("cpu")
        self.z = None

    def load_model(self, settings, device):
        self.device = torch.device(device)

    def get_num_resolutions(self):
        return None

    def get_opts(self, decay_divisor):
        return None

    def get_z_from_tensor(self, ref_tensor):            # ref in [-1, 1]
        return F.interpolate(ref_tensor * 0.5 + 0.5, size=self.grid_hw, mode="bilinear", align_corners=False)

    def init_from_tensor(self, init_tensor):
        if init_tensor is None:
            init_tensor = torch.zeros(1, 3, *self.canvas_hw, device=self.device)
        self.z = self.get_z_from_tensor(init_tensor.to(self.device)).detach().requires_grad_(True)

    def reapply_from_tensor(self, new_tensor):
        with torch.no_grad():
            self.z.copy_(self.get_z_from_tensor(new_tensor.to(self.device)))

    def synth(self, cur_iteration):
        return _SoftClamp01.apply(F.interpolate(self.z, size=self.canvas_hw, mode="nearest"))

    @torch.no_grad()
    def to_image(self):
        from PIL import Image
        rgb = self.synth(None)[0].mul(255).round().clamp(0, 255).byte().permute(1, 2, 0).cpu().numpy()
        return Image.fromarray(rgb)

    def clip_z(self):
        with torch.no_grad():
            self.z.clamp_(0, 1)

    def get_z(self):
        return self.z

    def set_z(self, new_z):
        with torch.no_grad():
            return self.z.copy_(new_z)

    def get_z_copy(self):
        return self.z.clone()
