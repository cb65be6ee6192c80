"""StyleLoss custom-loss plugin (BASELINE.json configs[3]; reference Losses/StyleLoss.py) on the HIP VGG16 extractor.

The reference plugin is STROTSS ("Style Transfer by Relaxed Optimal Transport and Self-Similarity") evaluated as a loss on
the drawer's output: over a pyramid of image scales it compares hyper-column VGG16 features of the (Laplacian-refolded)
image with features sampled from a style image -- relaxed earth mover's distance on cosine distances, first/second moment
matching, a palette term -- and with the image's own features (self-similarity "content" term).  This module restates that
arithmetic (same order of numpy random draws, same formulas; citations per function) with two structural differences:

* the nine VGG16 feature maps come from `ops.vgg16_features` (hand-written HIP, MFMA implicit-GEMM convs, see
  csrc/vgg.hip) and stay **channels-last** ([1,h,w,C]); the samplers gather rows of the [h*w, C] matrix instead of
  fancy-indexing an NCHW tensor;
* there is no download: VGG16 weights come from a torchvision-format state dict (`params=`, or the file named by
  $PIXRAY_VGG16_CKPT).  Without weights or without a GPU the plugin raises -- there is no CPU fallback.  Tests inject an
  extractor (`StyleLoss(extractor=...)`).
"""
from __future__ import annotations

import math
import os
from typing import List, Sequence

import numpy as np
import torch
import torch.nn.functional as F

from .interfaces import LossInterface

VGG_MEAN = (0.485, 0.456, 0.406)
VGG_STD = (0.229, 0.224, 0.225)
N_FEATURE_CHANNELS = 3 + 2 * 64 + 128 * 2 + 256 * 3 + 512 * 2      # StyleLoss.py:298 (every captured channel, no coordinates)


# ---------------------------------------------------------------------------------------------- host <-> device plumbing
_CONST = {}


def _const(key, values, device):
    """small constant tables (VGG mean/std, the YUV matrix) uploaded once per device: a `torch.tensor(list, device=...)`
    in the loss body is a blocking pageable H2D copy per call"""
    k = (key, str(device))
    t = _CONST.get(k)
    if t is None:
        t = _CONST[k] = torch.tensor(values, dtype=torch.float32, device=device)
    return t


def _upload(a: np.ndarray, device):
    """one numpy table -> device through the pinned caching allocator, stream-ordered (no host/GPU sync: the sampling tables
    are drawn by the host while the GPU is still busy with the previous kernels)"""
    t = torch.from_numpy(np.ascontiguousarray(a))
    if torch.device(device).type == "cuda":
        return t.pin_memory().to(device, non_blocking=True)
    return t.to(device)


# ---------------------------------------------------------------------------------------------- extractor
class Vgg16Extractor:
    """`Vgg16_Extractor` (StyleLoss.py:24-81): returns [input, relu1_1, relu1_2, relu2_1, relu2_2, relu3_1, relu3_2, relu3_3,
    relu4_3, relu5_3] as channels-last maps [1,h,w,C]."""

    def __init__(self, space: str = "uniform", params=None, device=None, max_hw=(64, 64), precision=None):
        self.space = space
        self.precision = precision      # None / "fp16" / "bf16" -> bf16 extractor | "f32" (exact-f32 MFMA parity mode)
        self.device = torch.device(device) if device is not None else torch.device("cuda")
        if params is None:
            path = os.environ.get("PIXRAY_VGG16_CKPT")
            if not path:
                raise RuntimeError("StyleLoss needs VGG16 weights: pass params= (torchvision state dict) or set "
                                   "PIXRAY_VGG16_CKPT to a torchvision vgg16 checkpoint (no download is attempted)")
            from .checkpoints import vgg16_from_torchvision
            params = vgg16_from_torchvision(torch.load(path, map_location="cpu"))
        self.params = params
        self.handle = None
        self.max_hw = (0, 0)
        self.reserve(*max_hw)

    def reserve(self, H: int, W: int):
        """make sure inputs up to H x W fit (the runner's temporaries are sized at creation)"""
        if self.handle is None or H > self.max_hw[0] or W > self.max_hw[1]:
            from . import ops
            self.max_hw = (max(H, self.max_hw[0]), max(W, self.max_hw[1]))
            self.handle = ops.Vgg16Handle(self.params, self.max_hw, self.device, precision=self.precision)

    def normalise(self, x):
        """StyleLoss.py:41-45"""
        if self.space != "vgg":
            x = (x + 1.0) / 2.0
            x = x - _const("vgg_mean", VGG_MEAN, x.device).to(x.dtype).view(1, -1, 1, 1)
            x = x / _const("vgg_std", VGG_STD, x.device).to(x.dtype).view(1, -1, 1, 1)
        return x

    def __call__(self, x) -> List[torch.Tensor]:
        from . import ops
        self.reserve(int(x.shape[2]), int(x.shape[3]))
        xn = self.normalise(x)
        return [xn.permute(0, 2, 3, 1)] + list(ops.vgg16_features(xn, self.handle))

    forward = __call__

    def forward_samples_hypercolumn(self, X, samps: int = 100):
        return sample_hypercolumns(self(X), samps)


def sample_hypercolumns(feat: Sequence[torch.Tensor], samps: int) -> torch.Tensor:
    """`forward_samples_hypercolumn` (StyleLoss.py:49-81) on channels-last maps: `samps` random pixel positions of the
    input, followed down the pyramid by halving (the reference halves whenever a map is smaller than the one before it),
    one column of all 2179 channels per position -> [1, 2179, samps], detached."""
    H, W = feat[0].shape[1], feat[0].shape[2]
    # The reference shuffles the [H*W, 2] coordinate table in place and keeps the first `samps` rows.  numpy's row shuffle
    # of a 2-D array is an interpreted per-row swap (0.3 s at 512x512); shuffling a 1-D index array walks the same
    # Fisher-Yates loop with the same `random_interval` draws in C (tests/test_host_logic.py pins both the permutation and
    # the generator state afterwards), so the draws -- and np.random's stream for everything after -- are unchanged.
    idx = np.arange(H * W)
    np.random.shuffle(idx)
    samples = min(samps, idx.shape[0])
    idx = idx[:samples]
    # meshgrid(arange(H), arange(W)) is [W, H]-shaped ('xy' indexing): flat position p holds (p % H, p // H)
    xx = (idx % H).astype(np.int64)
    yy = (idx // H).astype(np.int64)
    rows = np.empty((len(feat), samples), np.int64)
    for i, layer in enumerate(feat):
        if i > 0 and layer.shape[1] < feat[i - 1].shape[1]:
            xx = xx / 2.0
            yy = yy / 2.0
        xx = np.clip(xx, 0, layer.shape[1] - 1).astype(np.int32)
        yy = np.clip(yy, 0, layer.shape[2] - 1).astype(np.int32)
        rows[i] = xx.astype(np.int64) * layer.shape[2] + yy.astype(np.int64)
    rows_d = _upload(rows, feat[0].device)
    cols = [layer.reshape(-1, layer.shape[3]).index_select(0, rows_d[i]).detach() for i, layer in enumerate(feat)]   # [samples, C]
    return torch.cat(cols, 1).t().unsqueeze(0).contiguous()


# ---------------------------------------------------------------------------------------------- image pyramid
def _resample(t, size):
    """`tensor_resample` (StyleLoss.py:91-92)"""
    return F.interpolate(t, size, mode="bilinear", align_corners=False)


def _laplacian(x):
    """StyleLoss.py:133-135: x - up(down(x))"""
    h, w = x.shape[2], x.shape[3]
    return x - _resample(_resample(x, [h // 2, w // 2]), [h, w])


def _laplace_pyramid(x, levels):
    """StyleLoss.py:137-144"""
    pyr, cur = [], x
    for _ in range(levels):
        pyr.append(_laplacian(cur))
        cur = _resample(cur, (max(cur.shape[2] // 2, 1), max(cur.shape[3] // 2, 1)))
    pyr.append(cur)
    return pyr


def _fold_pyramid(pyr):
    """StyleLoss.py:146-151"""
    cur = pyr[-1]
    for lvl in reversed(pyr[:-1]):
        cur = lvl + _resample(cur, (lvl.shape[2], lvl.shape[3]))
    return cur


# ---------------------------------------------------------------------------------------------- sampling
def _sample_grid(h: int, w: int):
    """`sample_indices` (StyleLoss.py:153-167): a strided grid with random offsets, at most ~128^2 positions"""
    const = 128 ** 2
    big = h * w
    stride_x = int(max(math.floor(math.sqrt(big // const)), 1))
    offset_x = np.random.randint(stride_x)
    stride_y = int(max(math.ceil(math.sqrt(big // const)), 1))
    offset_y = np.random.randint(stride_y)
    xx, xy = np.meshgrid(np.arange(h)[offset_x::stride_x], np.arange(w)[offset_y::stride_y])
    return xx.flatten(), xy.flatten()


def _bilinear_columns(feat_a: Sequence[torch.Tensor], feat_b: Sequence[torch.Tensor], xx, xy):
    """`spatial_feature_extract` (StyleLoss.py:169-223): bilinear samples of every map of both feature lists at the same
    positions (halved whenever the resolution drops), concatenated over channels, plus the two (finally halved) coordinate
    channels -> two [1, 2181, n, 1] tensors."""
    dev = feat_a[0].device
    L, n = len(feat_a), len(xx)
    rows = np.empty((L, 4, n), np.int64)
    wts = np.empty((L * 4 + 2, n), np.float32)        # per layer w00 w01 w10 w11, then the two coordinate channels
    for i in range(L):
        fa = feat_a[i]
        if i > 0 and feat_a[i - 1].shape[1] > fa.shape[1]:
            xx = xx / 2.0
            xy = xy / 2.0
        xxm = np.floor(xx).astype(np.float32)
        xxr = xx - xxm
        xym = np.floor(xy).astype(np.float32)
        xyr = xy - xym
        wts[4 * i + 0] = (1. - xxr) * (1. - xyr)
        wts[4 * i + 1] = (1. - xxr) * xyr
        wts[4 * i + 2] = xxr * (1. - xyr)
        wts[4 * i + 3] = xxr * xyr
        hh, ww = fa.shape[1], fa.shape[2]
        xi = np.clip(xxm.astype(np.int32), 0, hh - 1).astype(np.int64)
        yi = np.clip(xym.astype(np.int32), 0, ww - 1).astype(np.int64)
        xi1, yi1 = np.clip(xi + 1, 0, hh - 1), np.clip(yi + 1, 0, ww - 1)
        rows[i, 0] = xi * ww + yi
        rows[i, 1] = xi * ww + yi1
        rows[i, 2] = xi1 * ww + yi
        rows[i, 3] = xi1 * ww + yi1
    wts[4 * L] = np.asarray(xx, dtype=np.float32)
    wts[4 * L + 1] = np.asarray(xy, dtype=np.float32)
    rows_d = _upload(rows, dev)                             # two stream-ordered uploads for the whole pyramid
    wts_d = _upload(wts, dev)                               # [4L+2, n]
    if dev.type == "cuda":
        # one gather launch per feature list (and one scatter launch in its backward) instead of ~11 torch ops per map:
        # the host-side launch cost of the composed form bounded the whole configs[3] iteration.  Device tensors ALWAYS take
        # this route (ops raises if the library lacks the entry point); the composed form below is the reference's own
        # arithmetic, kept for CPU tensors only -- it is what tests/test_style_loss.py pins to the reference's goldens and
        # what tests/test_kernels_gpu.py compares the kernel with, bit for bit.
        from . import ops
        a = ops.hypercolumns(feat_a, rows_d, wts_d)
        b = ops.hypercolumns(feat_b, rows_d, wts_d)
        return a.t()[None, :, :, None], b.t()[None, :, :, None]
    wts_d = wts_d.unsqueeze(2)                              # [4L+2, n, 1]
    ca, cb = [], []
    for i in range(L):
        r, w = rows_d[i], wts_d[4 * i:4 * i + 4]

        def gather(f):
            m = f.reshape(-1, f.shape[3])                         # [h*w, C], rows contiguous
            return (m.index_select(0, r[0]) * w[0] + m.index_select(0, r[1]) * w[1]
                    + m.index_select(0, r[2]) * w[2] + m.index_select(0, r[3]) * w[3])
        ca.append(gather(feat_a[i]))
        cb.append(gather(feat_b[i]))
    cx, cy = wts_d[4 * L], wts_d[4 * L + 1]
    a = torch.cat(ca + [cx, cy], 1).t()[None, :, :, None]
    b = torch.cat(cb + [cx, cy], 1).t()[None, :, :, None]
    return a, b


# ---------------------------------------------------------------------------------------------- distances and losses
def _cos_dist(x, y):
    """`pairwise_distances_cos` (StyleLoss.py:225-230)"""
    xn = torch.sqrt((x ** 2).sum(1).view(-1, 1))
    yn = torch.sqrt((y ** 2).sum(1).view(1, -1))
    return 1. - torch.mm(x, y.t()) / xn / yn


def _l2_dist(x, y):
    """sqrt of `pairwise_distances_sq_l2` (StyleLoss.py:232-237, 243)"""
    d = (x ** 2).sum(1).view(-1, 1) + (y ** 2).sum(1).view(1, -1) - 2.0 * torch.mm(x, y.t())
    return torch.sqrt(torch.clamp(d, 1e-5, 1e5) / x.size(1))


def _columns(t):
    """[1, d, n, 1] -> [n, d]"""
    return t[0, :, :, 0].t()


def _self_similarity_loss(a, b):
    """`content_loss` (StyleLoss.py:246-265): mean |cosine self-distance matrix of a - that of b|, coordinates dropped"""
    X = _columns(a)[:, :-2]
    Y = _columns(b)[:, :-2]
    return torch.abs(_cos_dist(X, X) - _cos_dist(Y, Y)).mean()


_YUV = ((0.577350, 0.577350, 0.577350), (-0.577350, 0.788675, -0.211325), (-0.577350, -0.211325, 0.788675))


def _remd(a, b):
    """`style_loss` (StyleLoss.py:272-293): relaxed EMD = max of the two mean nearest-neighbour distances; 3-channel
    inputs are compared in a YUV-like space with cosine + L2 distance"""
    d = a.shape[1]
    X, Y = _columns(a), _columns(b)
    if d == 3:
        C = _const("yuv", _YUV, X.device).to(X.dtype)
        X, Y = torch.mm(C, X.t()).t(), torch.mm(C, Y.t()).t()
    M = _cos_dist(X, Y)
    if d == 3:
        M = M + _l2_dist(X, Y)
    return torch.max(M.min(1)[0].mean(), M.min(0)[0].mean())


def _moment_loss(a, b):
    """`moment_loss` with moments [1, 2] (StyleLoss.py:295-323): mean |difference of means| + mean |difference of covariances|"""
    X, Y = _columns(a), _columns(b)
    mu_x, mu_y = X.mean(0, keepdim=True), Y.mean(0, keepdim=True)
    loss = torch.abs(mu_x - mu_y).mean()
    Xc, Yc = X - mu_x, Y - mu_y
    cov_x = torch.mm(Xc.t(), Xc) / (X.shape[0] - 1)
    cov_y = torch.mm(Yc.t(), Yc) / (Y.shape[0] - 1)
    return loss + torch.abs(cov_x - cov_y).mean()


def _pair_loss(feat_result, feat_content, feat_style, xx, xy, content_weight, moment_weight=1.0):
    """`calculate_loss` (StyleLoss.py:325-347)"""
    n = 1024
    res, con = _bilinear_columns(feat_result, feat_content, xx[:n], xy[:n])
    loss_content = _self_similarity_loss(res, con)
    sty = feat_style.view(1, feat_style.shape[1], -1, 1)
    loss_remd = _remd(res[:, :N_FEATURE_CHANNELS], sty[:, :N_FEATURE_CHANNELS])
    loss_moment = _moment_loss(res[:, :-2], sty)
    loss_moment = loss_moment + (1. / max(content_weight, 1.)) * _remd(res[:, :3], sty[:, :3])
    loss_style = loss_remd + moment_weight * loss_moment
    return (content_weight * loss_content + loss_style) / (content_weight + 1.0 + moment_weight)


def _scale_loss(result, content, style, content_weight, lr, extractor, style_maps=None, recompute=False):
    """`scale_loss` (StyleLoss.py:349-389): 5 x 1000 style hyper-columns, one sampling grid, three evaluations of the
    refolded image (the grid is reshuffled before the 2nd and 3rd).  The reference runs VGG16 on the (frozen) style image
    for each of the 5 draws; the maps are the same every time, so they are computed once (`style_maps`: the caller's copy
    from an earlier iteration) and only the sampling is repeated.  `recompute=True` is the reference's schedule verbatim (one
    VGG pass per draw): what bench.py's CPU-baseline leg times."""
    pyramid = _laplace_pyramid(result, 5)
    feat_content = extractor(content)
    if recompute:
        with torch.no_grad():
            feat_style = torch.cat([extractor.forward_samples_hypercolumn(style, samps=1000) for _ in range(5)], dim=2)
    else:
        if style_maps is None:
            with torch.no_grad():
                style_maps = [f.detach() for f in extractor(style)]
        feat_style = torch.cat([sample_hypercolumns(style_maps, 1000) for _ in range(5)], dim=2)
    xx, xy = _sample_grid(feat_content[0].shape[1], feat_content[0].shape[2])
    total = 0.0
    for it in range(3):
        stylized = _fold_pyramid(pyramid)
        if it != 0:
            np.random.shuffle(xx)
            np.random.shuffle(xy)
        total = total + _pair_loss(extractor(stylized), feat_content, feat_style, xx, xy, content_weight) * lr
    return total


def strotss_loss(out_tensor, style_tensor, content_weight=16.0, extractor=None, style_cache=None, recompute=False):
    """`strotss_loss` (StyleLoss.py:392-431): coarse-to-fine over the scales whose short side is >= 33 px; the running
    `result` image is the upsampled previous result plus the Laplacian of the content at that scale; only the finest scale
    carries weight 1 (the others 2e-3), the content weight halves per scale"""
    H, W = out_tensor.shape[2], out_tensor.shape[3]
    scales = [2 ** s for s in range(10) if min(H, W) // (2 ** s) >= 33][::-1]
    total, lr, result = 0.0, 2e-3, None
    for scale in scales:
        content = _resample(out_tensor, [H // scale, W // scale])
        style = _resample(style_tensor, [style_tensor.shape[2] // scale, style_tensor.shape[3] // scale])
        if scale == scales[0]:
            result = _laplacian(content) + style.mean(2, keepdim=True).mean(3, keepdim=True)
        elif scale == scales[-1]:
            result = _resample(result, [content.shape[2], content.shape[3]])
            lr = 1
        else:
            result = _resample(result, [content.shape[2], content.shape[3]]) + _laplacian(content)
        maps = None
        if style_cache is not None and not recompute:          # the style image and the VGG are frozen: its feature maps per scale are too
            maps = style_cache.get(scale)
            if maps is None:
                with torch.no_grad():
                    maps = style_cache[scale] = [f.detach() for f in extractor(style)]
        total = total + _scale_loss(result, content, style, content_weight, lr, extractor, maps, recompute)
        content_weight /= 2.0
    return total


# ---------------------------------------------------------------------------------------------- the plugin
class StyleLoss(LossInterface):
    """`StyleLoss` (StyleLoss.py:458-500).  Extra constructor arguments (not in the reference): `extractor` (anything with the
    `Vgg16Extractor` call surface), `vgg_params` (torchvision VGG16 state dict), `style_image` ([1,3,h,w] tensor in [0,1]),
    `reference_schedule` (recompute the frozen style maps per draw like the reference instead of caching them per scale)."""

    def __init__(self, extractor=None, vgg_params=None, style_image=None, reference_schedule=False, **kwargs):
        self.reference_schedule = reference_schedule   # True: one VGG pass per style draw, every iteration (StyleLoss.py:357-362)
        self.resized = None
        self._style_cache = {}          # scale -> VGG16 maps of the resized style image (frozen inputs: computed once)
        self.extractor = extractor
        self.vgg_params = vgg_params
        self.style = style_image
        super().__init__(**kwargs)

    @staticmethod
    def add_settings(parser):
        parser.add_argument("--style_file", type=str, default="", dest='style_file')
        parser.add_argument("--styleloss_content_weight", type=float, default=32, dest='styleloss_content_weight')
        parser.add_argument("--styleloss_ospace", type=str, default="uniform", dest='styleloss_ospace')
        parser.add_argument("--styleloss_skip", type=int, default=100, dest='styleloss_skip')
        parser.add_argument("--styleloss_every", type=int, default=1, dest='styleloss_every')
        return parser

    def parse_settings(self, args):
        if getattr(args, "style_file", ""):
            import glob
            from PIL import Image
            if "http" in args.style_file:
                raise RuntimeError("StyleLoss: remote style files are not fetched; download the image and pass its path")
            files = sorted(glob.glob(args.style_file))
            if not files:
                raise ValueError(f"StyleLoss: no file matches {args.style_file!r}")
            img = np.asarray(Image.open(files[0]).convert("RGB"), dtype=np.float32) / 255.0
            self.style = torch.from_numpy(img).permute(2, 0, 1).unsqueeze(0)
        if self.extractor is None:
            self.extractor = Vgg16Extractor(space=getattr(args, "styleloss_ospace", "uniform"), params=self.vgg_params,
                                            device=self.device)
        return args

    def get_loss(self, cur_cutouts, out, args, globals=None, lossGlobals=None):
        if self.resized is None:
            if self.style is None:
                raise ValueError("StyleLoss: no style image (set --style_file or pass style_image=)")
            self.resized = F.interpolate(self.style.to(out.device, torch.float32), out.size()[2:4], mode="bicubic",
                                         align_corners=False)
            self._style_cache = {}
        if globals["cur_iteration"] < args.styleloss_skip:
            return torch.tensor(0.0)
        if globals["cur_iteration"] % args.styleloss_every != 0:
            return torch.tensor(0.0)
        return strotss_loss(out, self.resized, args.styleloss_content_weight, extractor=self.extractor,
                            style_cache=self._style_cache, recompute=self.reference_schedule)
