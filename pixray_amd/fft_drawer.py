"""`FftDrawer` with the reference's drawer surface (/root/reference/fftdrawer.py:13-110): the image is a learnable
Fourier spectrum (BASELINE.json configs[3]: "fftdrawer 512x512").

A drawer *plugin* (the reference's drawers are plugins too): `synth()` hands the loop a [1,3,H,W] tensor in [0,1] and
everything downstream (cutouts, CLIP tower, loss, backward to the image) runs on the HIP kernels.  The spectrum -> image map
itself runs on `csrc/fft_drawer.hip` when the drawer lives on a GPU: the inverse real transform as exact-f32 GEMMs on the engine
against float64-built twiddle matrices, the std / colour / sigmoid tail and the whole backward in small kernels (no FFT library,
17 launches per forward + backward instead of ~40 torch ops; bit-reproducible: the std's sums are taken in a fixed order).  A
drawer created on the CPU (the CPU plumbing tests, configs[0]-style runs) evaluates the same map with plain torch
(`torch.fft.irfftn`).  Both are checked against the explicit-DFT oracle (oracle/fft_ref.py).

The reference delegates the parameterisation to `aphantasia.image.fft_image / to_valid_rgb` (eps696/aphantasia@7e6b3bb,
requirements.txt, absent offline) [UPSTREAM]; restated here from the published algorithm (the transform is pinned to its
definition by the oracle; aphantasia's constants -- colour matrix, 1/max(W,H) floor -- stay from the published source):
  params  = N(0, 0.01) real/imag spectrum [1, 3, H, W/2+1, 2]
  scale   = sqrt(W*H) / max(|f|, 1/max(W,H)) ** decay         (|f| = radial rfft2 frequency)
  image   = irfft2(scale * params, norm="ortho");  image *= contrast / image.std()
  rgb     = sigmoid(image projected through the normalised colour-correlation matrix, first axis / colors)
Only `--fft_use fft` is provided (`dwt` needs pytorch_wavelets; `pixel` is PixelGridDrawer's job)."""
import math

import numpy as np
import torch

from .interfaces import DrawingInterface

_COLOR_CORRELATION_SVD_SQRT = np.asarray([[0.26, 0.09, 0.02], [0.27, 0.00, -0.05], [0.27, -0.09, 0.03]], dtype=np.float32)


def rfft2d_freqs(h: int, w: int) -> np.ndarray:
    fy = np.fft.fftfreq(h)[:, None]
    fx = np.fft.fftfreq(w)[: w // 2 + (2 if w % 2 == 1 else 1)]
    return np.sqrt(fx * fx + fy * fy)


def color_matrix(colors: float) -> torch.Tensor:
    m = _COLOR_CORRELATION_SVD_SQRT / np.asarray([colors, 1.0, 1.0], dtype=np.float32)
    m = m / np.max(np.linalg.norm(m, axis=0))
    return torch.tensor(m.T)


class FftDrawer(DrawingInterface):
    @staticmethod
    def add_settings(parser):
        parser.add_argument("--fft_use", type=str, help="use fft or dwt or pixel", default="fft", dest='fft_use')
        parser.add_argument('--fft_decay', default=1.5, type=float, dest='fft_decay')
        parser.add_argument('--fft_wave', default='coif2', help='wavelets: db[1..], coif[1..], haar, dmey', dest='fft_wave')
        parser.add_argument('--fft_sharp', default=0.3, type=float, dest='fft_sharp')
        parser.add_argument('--fft_colors', default=1.5, type=float, dest='fft_colors')
        parser.add_argument('--fft_lrate', default=0.3, type=float, help='Learning rate', dest='fft_lrate')
        return parser

    def __init__(self, settings):
        super(DrawingInterface, self).__init__()
        self.canvas_width, self.canvas_height = settings.size[0], settings.size[1]
        self.fft_use = getattr(settings, "fft_use", "fft")
        self.decay = getattr(settings, "fft_decay", 1.5)
        self.lrate = getattr(settings, "fft_lrate", 0.3)
        self.seed = getattr(settings, "weight_seed", 0)
        self.img = None
        self.params = None
        self.opts = None

    def load_model(self, settings, device):
        self.device = torch.device(device)
        # on a GPU the spectrum -> image map runs on the exact-f32 GEMM engine (csrc/fft_drawer.hip), not on torch.fft / rocFFT
        # (oracle/fft_ref.py agreement on the device: tests/test_zz_frontend_gpu.py::test_fft_drawer_hip_path; `fft_hip_force`
        # lets the CPU emulation of the kernels stand in for the device in tests/test_emu_cpu.py)
        self.hip = self.device.type == "cuda" or bool(getattr(settings, "fft_hip_force", False))
        self._handle = None

    def rand_init(self, toksX=None, toksY=None):
        self.init_from_tensor(None)

    def init_from_tensor(self, init_tensor):
        if self.fft_use != "fft":
            raise ValueError(f"fft drawer does not know how to apply fft_use={self.fft_use}")
        h, w = self.canvas_height, self.canvas_width
        freqs = rfft2d_freqs(h, w)
        scale = 1.0 / np.maximum(freqs, 1.0 / max(w, h)) ** self.decay * math.sqrt(w * h)
        self._scale = torch.tensor(scale, dtype=torch.float32, device=self.device)[None, None, ..., None]
        self._colors = color_matrix(1.5).to(self.device)          # fftdrawer.py:62 hard-codes colors=1.5
        g = torch.Generator().manual_seed(self.seed)
        spectrum = torch.randn(1, 3, *freqs.shape, 2, generator=g) * 0.01
        if init_tensor is not None:
            # start from an image: invert sigmoid + colour projection + scaling (the reference round-trips through a PNG
            # and aphantasia's `img2fft`; this is the same idea without the file)
            x = init_tensor.detach().float().cpu()
            if x.min() < 0:                          # [-1, 1] input (what pixray passes: pixray.py:718)
                x = x.add(1).div(2)
            x = x.clamp(1e-3, 1 - 1e-3)
            logit = torch.log(x) - torch.log1p(-x)
            dec = torch.einsum("ndhw,dc->nchw", logit, torch.linalg.inv(color_matrix(1.5)))
            spec = torch.view_as_real(torch.fft.rfftn(dec, s=(h, w), dim=(-2, -1), norm="ortho"))
            spectrum = spec / self._scale.cpu()
        self.params = [spectrum.to(self.device).requires_grad_(True)]

    def reapply_from_tensor(self, new_tensor):
        self.init_from_tensor(new_tensor)

    def get_opts(self, decay_divisor=1):
        self.opts = [torch.optim.Adam(self.params, self.lrate / decay_divisor)]       # fftdrawer.py:65-69
        return self.opts

    def get_z_from_tensor(self, ref_tensor):
        return None

    def get_num_resolutions(self):
        return None

    def synth(self, cur_iteration):
        if cur_iteration is not None and cur_iteration < 0:
            return self.img
        h, w = self.canvas_height, self.canvas_width
        if getattr(self, "hip", False):
            from . import ops
            if self._handle is None:
                self._handle = ops.FftDrawerHandle(w, h, decay=self.decay, colors=1.5)
            self.img = ops.fft_synth(self.params[0], self._handle, 0.9)
            return self.img
        spec = torch.view_as_complex((self._scale * self.params[0]).contiguous())
        image = torch.fft.irfftn(spec, s=(h, w), dim=(-2, -1), norm="ortho")
        image = image * 0.9 / image.std()                                             # contrast=0.9 (fftdrawer.py:84)
        image = torch.einsum("nchw,cd->ndhw", image, self._colors)
        self.img = torch.sigmoid(image)
        return self.img

    @torch.no_grad()
    def to_image(self):
        from PIL import Image
        img = self.synth(None) if self.img is None else self.img
        arr = (img.detach().cpu().numpy()[0].transpose(1, 2, 0).clip(0, 1) * 255).astype(np.uint8)
        return Image.fromarray(arr, mode="RGB")

    def clip_z(self):
        pass

    def get_z(self):
        return None

    def get_z_copy(self):
        return None

    def set_z(self, new_z):
        return None
