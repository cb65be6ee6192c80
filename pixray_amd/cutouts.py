"""MakeCutouts (reference: /root/reference/pixray.py:400-511) on the HIP cutout kernels.

The reference draws its augmentation parameters inside kornia's parameter generators
(kornia==0.6.2, un-vendored) from torch's global RNG.  Here the draws are made on the host
from an explicit `torch.Generator` (`sample_cutout_params`), turned into per-cutout
descriptors (`build_descriptors`: two 3x3 pixel-space sampling matrices + modes + colour
jitter factors), uploaded, and consumed by one HIP kernel pipeline.  The same `params` dict
is what the CPU oracle consumes, which is how parity is defined (SURVEY.md §7 "Parity
definition").
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import numpy as np
import torch
from torch import nn

from . import ops

DESC_WORDS = 36
MODE_IDENT, MODE_ZEROS, MODE_BORDER, MODE_REFLECT, MODE_FILL, MODE_REFLECT_AC = 0, 1, 2, 3, 4, 5
GRID_MESH, GRID_AFFINE, GRID_AFFINE_AC, GRID_MESH_AC = 0, 1, 2, 3

# The align_corners flag each kornia 0.6.2 call of the reference's augmentation stack hands to F.grid_sample (kornia is not
# vendored: from the published 0.6.2 sources, parity unpinned -- the table is restated in oracle/cutouts_ref.py with the
# kornia definitions it follows).  The flag is DATA here: it selects the grid flavour written into the descriptor, and the
# kernel follows whatever the descriptor says (tests/test_path_gpu.py::test_cutout_align_corners_convention_is_a_descriptor_field).
KORNIA_062_CONVENTIONS = {
    "perspective_align_corners": False,   # K.RandomPerspective(align_corners=False) -> pixray.py:333-334, 363 pass the flag on
    "affine_align_corners": False,        # K.RandomAffine(align_corners=False)      -> pixray.py:349
    "crop_align_corners": True,           # K.RandomResizedCrop / K.CenterCrop(align_corners=True): crop_by_transform_mat -> warp_affine
    "cached_align_corners": True,         # kornia.geometry.transform.warp_perspective(..., align_corners=True) default, pixray.py:482-485
}


def _grid(kind: str, align_corners: bool) -> int:
    if kind == "mesh":
        return GRID_MESH_AC if align_corners else GRID_MESH
    return GRID_AFFINE_AC if align_corners else GRID_AFFINE


def _uniform(gen, shape, lo=0.0, hi=1.0):
    return torch.rand(shape, generator=gen, dtype=torch.float64) * (hi - lo) + lo


def base_size(S: int, aspect: float = 1.0):
    """(Hb, Wb) of the pooled cutout after the reference's aspect rescale (pixray.py:468-472; kornia `rescale` =
    F.interpolate to `int(size * factor)`): width * aspect on a wide canvas, height / aspect on a tall one."""
    if aspect == 1:
        return S, S
    if aspect > 1:
        return S, int(S * aspect)
    return int(S * (1 / aspect)), S


def sample_cutout_params(cutn: int, S: int, gen: torch.Generator, iteration: int = 0, noise_fac: float = 0.1,
                         fill: Optional[float] = None, aspect: float = 1.0) -> Dict[str, torch.Tensor]:
    """Host-side mirror of the reference's per-iteration draws.

    zoom set (first int(0.6*cutn), pixray.py:407): RandomPerspective(0.4, p=.7) corner offsets,
    RandomResizedCrop(scale .25-.95, ratio .85-1.2) box (10 tries then whole image), ColorJitter(p=.8)
    saturation U(.9,1.1) / hue U(-.1,.1) and the order of the two; wide set: RandomAffine
    translate U(+-2.5%), RandomPerspective(0.2, p=.7), ColorJitter; per-cutout noise factor
    U(0, noise_fac) (pixray.py:508-510); padding mode by iteration parity (1250-1253); gray fill
    U(0,1) (1255-1258).

    `aspect` = canvas width / height (pixray.py:1931 `global_aspect_width`).  When it is not 1 the augmentations run on
    the aspect-rescaled cutout (Hb x Wb, `base_size`): crop boxes are positioned inside it, and the wide set's RandomAffine
    becomes scale U(0.9 n_s, n_s), n_s = min(aspect, 1/aspect), translated by up to +-(1-n_s)/2 of the size along the canvas' short axis only
    (pixray.py:420-431)."""
    Hb, Wb = base_size(S, aspect)
    nz = int(0.6 * cutn)
    nw = cutn - nz
    p: Dict[str, torch.Tensor] = {"cutn": torch.tensor(cutn)}
    p["reflect"] = torch.tensor(1 if iteration % 2 == 0 else 0)
    p["fill"] = torch.tensor(float(_uniform(gen, ()).item()) if fill is None else float(fill), dtype=torch.float64)
    # ---- zoom
    p["z_persp_apply"] = _uniform(gen, (nz,)) < 0.7
    p["z_persp_rand"] = _uniform(gen, (nz, 4, 2))
    area = _uniform(gen, (nz, 10), 0.25, 0.95) * S * S
    log_ratio = _uniform(gen, (nz, 10), math.log(0.85), math.log(1.2))
    crop_ratio = torch.exp(log_ratio)
    w = torch.sqrt(area * crop_ratio).round().floor()
    h = torch.sqrt(area / crop_ratio).round().floor()
    ok = (w > 0) & (w < S) & (h > 0) & (h < S)
    first = torch.where(ok.any(1), ok.float().argmax(1), torch.zeros(nz, dtype=torch.long))
    ar = torch.arange(nz)
    cw = torch.where(ok.any(1), w[ar, first], torch.full((nz,), float(S), dtype=torch.float64))
    chh = torch.where(ok.any(1), h[ar, first], torch.full((nz,), float(S), dtype=torch.float64))
    xs = (_uniform(gen, (nz,)) * (Wb - cw + 1)).floor()
    ys = (_uniform(gen, (nz,)) * (Hb - chh + 1)).floor()
    p["z_crop"] = torch.stack([xs, ys, cw, chh], dim=1)
    p["z_jit_apply"] = _uniform(gen, (nz,)) < 0.8
    p["z_sat"] = _uniform(gen, (nz,), 0.9, 1.1).float()
    p["z_hue"] = _uniform(gen, (nz,), -0.1, 0.1).float()
    p["z_sat_first"] = torch.tensor(bool(_uniform(gen, ()).item() < 0.5))
    # ---- wide
    p["w_trans"] = _uniform(gen, (nw, 2), -0.025 * S, 0.025 * S)
    p["w_persp_apply"] = _uniform(gen, (nw,)) < 0.7
    p["w_persp_rand"] = _uniform(gen, (nw, 4, 2))
    p["w_jit_apply"] = _uniform(gen, (nw,)) < 0.8
    p["w_sat"] = _uniform(gen, (nw,), 0.9, 1.1).float()
    p["w_hue"] = _uniform(gen, (nw,), -0.1, 0.1).float()
    p["w_sat_first"] = torch.tensor(bool(_uniform(gen, ()).item() < 0.5))
    p["noise_fac"] = _uniform(gen, (cutn,), 0.0, noise_fac).float()
    p["noise"] = None
    p["aspect"] = torch.tensor(float(aspect), dtype=torch.float64)
    p["w_scale"] = torch.full((nw,), 0.95, dtype=torch.float64)
    if aspect != 1:         # drawn last so that the square-canvas streams are unchanged
        n_s = (1.0 / aspect) if aspect > 1 else aspect
        n_t = (1.0 - n_s) / 2
        p["w_scale"] = _uniform(gen, (nw,), 0.9 * n_s, n_s)
        t = _uniform(gen, (nw,), -1.0, 1.0)
        zeros = torch.zeros(nw, dtype=torch.float64)
        # kornia RandomAffine translate=(tx, ty): |dx| <= tx * width, |dy| <= ty * height
        p["w_trans"] = torch.stack([zeros, t * n_t * Hb], 1) if aspect > 1 else torch.stack([t * n_t * Wb, zeros], 1)
    return p


def _dlt(src: np.ndarray, dst: np.ndarray) -> np.ndarray:
    """4-point homographies (float64, batched) with M @ src ~ dst."""
    B = src.shape[0]
    A = np.zeros((B, 8, 8))
    x, y, u, v = src[:, :, 0], src[:, :, 1], dst[:, :, 0], dst[:, :, 1]
    A[:, 0::2, 0], A[:, 0::2, 1], A[:, 0::2, 2] = x, y, 1.0
    A[:, 0::2, 6], A[:, 0::2, 7] = -x * u, -y * u
    A[:, 1::2, 3], A[:, 1::2, 4], A[:, 1::2, 5] = x, y, 1.0
    A[:, 1::2, 6], A[:, 1::2, 7] = -x * v, -y * v
    b = np.empty((B, 8, 1))
    b[:, 0::2, 0], b[:, 1::2, 0] = u, v
    X = np.linalg.solve(A, b)[:, :, 0]
    return np.concatenate([X, np.ones((B, 1))], axis=1).reshape(B, 3, 3)


def _corners(S: int, B: int, W: int = None) -> np.ndarray:
    """corner points of an S x W image (W defaults to S), kornia order"""
    W = S if W is None else W
    c = np.array([[0.0, 0.0], [W - 1.0, 0.0], [W - 1.0, S - 1.0], [0.0, S - 1.0]])
    return np.repeat(c[None], B, axis=0)


def _norm_pixel(h: int, w: int) -> np.ndarray:
    """kornia normal_transform_pixel: pixel [0, w-1] x [0, h-1] -> [-1, 1]^2"""
    return np.array([[2.0 / (w - 1), 0.0, -1.0], [0.0, 2.0 / (h - 1), -1.0], [0.0, 0.0, 1.0]])


def _src_norm_from_dst_norm(M: np.ndarray, S, dst=None) -> np.ndarray:
    """inverse of kornia normalize_homography(M, src_hw, dst_hw): maps normalised destination coords to normalised source
    coords.  `S` / `dst`: an int (square) or (h, w); dst defaults to the source size."""
    src = (S, S) if isinstance(S, int) else S
    dst = src if dst is None else ((dst, dst) if isinstance(dst, int) else dst)
    return np.linalg.inv(_norm_pixel(*dst) @ (M @ np.linalg.inv(_norm_pixel(*src))))


_PTS_NORM = np.array([[1.0, 1.0], [-1.0, 1.0], [-1.0, -1.0], [1.0, -1.0]])


def _np(t):
    return t.detach().cpu().numpy().astype(np.float64) if isinstance(t, torch.Tensor) else np.asarray(t, dtype=np.float64)


def build_descriptors(p: Dict[str, torch.Tensor], S: int, conventions: Optional[Dict[str, bool]] = None) -> torch.Tensor:
    """[cutn, 36] fp64 descriptor table for prx_cutouts_forward (layout: include/prx.h).

    Each stage carries kornia's `src_norm_trans_dst_norm` 3x3 (the inverse of normalize_homography(M) with
    the [0, W-1] -> [-1, 1] convention) plus the grid flavour that produced the sampling grid in kornia 0.6.2:
      GRID_MESH[_AC]   (warp_perspective): create_meshgrid(normalized) + transform_points, then F.grid_sample
      GRID_AFFINE[_AC] (warp_affine)     : F.affine_grid(theta = first two rows, rounded to fp32), then F.grid_sample
    where _AC = that call's align_corners flag was True (`conventions`, default KORNIA_062_CONVENTIONS: the perspective and
    affine augmentations pass False, the resized / centre crops True).  The kernel evaluates the grid with the same
    precision steps, so tap positions round like the oracle's.  (numpy float64 on the host: ~0.3 ms for 64 cutouts.)"""
    cv = dict(KORNIA_062_CONVENTIONS, **(conventions or p.get("conventions") or {}))
    ac_p, ac_a, ac_c = bool(cv["perspective_align_corners"]), bool(cv["affine_align_corners"]), bool(cv["crop_align_corners"])
    cutn = int(p["cutn"])
    nz = int(0.6 * cutn)
    nw = cutn - nz
    aspect = float(p["aspect"]) if "aspect" in p else 1.0
    Hb, Wb = base_size(S, aspect)                 # the aspect-rescaled pooled image = source of stage A = size of stage A
    desc = np.zeros((cutn, DESC_WORDS))
    desc[:, 20] = float(p["fill"])
    eye = np.eye(3).reshape(9)
    desc[:, 0:9] = eye
    desc[:, 9:18] = eye
    desc[:, 28:32] = (0.0, 0.0, float(Wb), float(Hb))          # stage B reads the whole stage-A image ...

    def affine_theta(M, src, dst=None):
        t = _src_norm_from_dst_norm(M, src, dst)
        t[:, :2, :] = t[:, :2, :].astype(np.float32).astype(np.float64)   # F.affine_grid receives theta in fp32
        t[:, 2, :] = (0.0, 0.0, 1.0)
        return t.reshape(-1, 9)

    def persp_offsets(rand, dscale, h, w):
        return np.array([dscale * w / 2, dscale * h / 2])[None, None, :] * _np(rand) * _PTS_NORM[None]

    if nz > 0:
        start = _corners(Hb, nz, Wb)
        end = start + persp_offsets(p["z_persp_rand"], 0.4, Hb, Wb)
        app = _np(p["z_persp_apply"]) != 0
        desc[:nz, 0:9] = np.where(app[:, None], _src_norm_from_dst_norm(_dlt(start, end), (Hb, Wb)).reshape(nz, 9), eye[None])
        pad = (MODE_REFLECT_AC if ac_p else MODE_REFLECT) if int(p["reflect"]) else MODE_BORDER
        desc[:nz, 18] = np.where(app, float(pad), float(MODE_IDENT))
        desc[:nz, 26] = _grid("mesh", ac_p)
        crop = _np(p["z_crop"])
        xs, ys, w, h = crop[:, 0], crop[:, 1], crop[:, 2], crop[:, 3]
        src = np.stack([np.stack([xs, ys], 1), np.stack([xs + w - 1, ys], 1),
                        np.stack([xs + w - 1, ys + h - 1], 1), np.stack([xs, ys + h - 1], 1)], axis=1)
        Mc = _dlt(src, _corners(S, nz))
        Mc[:, 2, :] = (0.0, 0.0, 1.0)                                      # warp_affine drops the last row
        desc[:nz, 9:18] = affine_theta(Mc, (Hb, Wb), (S, S))
        desc[:nz, 19] = MODE_ZEROS
        desc[:nz, 27] = _grid("affine", ac_c)       # RandomResizedCrop(cropping_mode='resample') -> crop_by_transform_mat
        desc[:nz, 21] = _np(p["z_jit_apply"])
        desc[:nz, 22] = _np(p["z_sat"])
        desc[:nz, 23] = (p["z_hue"].float() * (2.0 * math.pi)).double().numpy()   # kornia: hue_factor * 2*pi in fp32
        desc[:nz, 24] = float(bool(p["z_sat_first"]))
    if nw > 0:
        sc = _np(p["w_scale"]) if "w_scale" in p else np.full(nw, 0.95)
        cx, cy = Wb / 2.0 - 0.5, Hb / 2.0 - 0.5
        tr = _np(p["w_trans"])
        Ma = np.zeros((nw, 3, 3))
        Ma[:, 0, 0] = sc; Ma[:, 1, 1] = sc; Ma[:, 2, 2] = 1.0
        Ma[:, 0, 2] = (1 - sc) * cx + tr[:, 0]
        Ma[:, 1, 2] = (1 - sc) * cy + tr[:, 1]
        desc[nz:, 0:9] = affine_theta(Ma, (Hb, Wb))
        desc[nz:, 18] = MODE_FILL
        desc[nz:, 26] = _grid("affine", ac_a)
        # ... except the wide set on a non-square canvas: CenterCrop(S) (pixray.py:433) = the centred S x S window
        desc[nz:, 28:32] = (float((Wb - S) // 2), float((Hb - S) // 2), float(S), float(S))
        start = _corners(S, nw)
        end = start + persp_offsets(p["w_persp_rand"], 0.2, S, S)
        app = _np(p["w_persp_apply"]) != 0
        desc[nz:, 9:18] = np.where(app[:, None], _src_norm_from_dst_norm(_dlt(start, end), S).reshape(nw, 9), eye[None])
        desc[nz:, 19] = np.where(app, float(MODE_FILL), float(MODE_IDENT))
        desc[nz:, 27] = _grid("mesh", ac_p)
        desc[nz:, 21] = _np(p["w_jit_apply"])
        desc[nz:, 22] = _np(p["w_sat"])
        desc[nz:, 23] = (p["w_hue"].float() * (2.0 * math.pi)).double().numpy()
        desc[nz:, 24] = float(bool(p["w_sat_first"]))
    desc[:, 25] = _np(p["noise_fac"])
    if "noise_seed" in p:                 # in-kernel N(0,1) draws (csrc/cutouts.hip philox_normal3) when no explicit noise tensor is handed over
        desc[:, 32] = _np(p["noise_seed"])
    return torch.from_numpy(desc)


def build_cached_descriptors(live: torch.Tensor, cutn: int, S: int, reflect: bool, fill: float, noise_fac: torch.Tensor,
                             aspect: float = 1.0, conventions: Optional[Dict[str, bool]] = None) -> torch.Tensor:
    """Descriptor table of the reference's CACHED-transform path (pixray.py:480-486): when `.transforms` is set (second
    and later calls inside one iteration: image prompts, pixray.py:1318-1333) the (aspect-rescaled) pooled image is
    warped ONCE with the composed 3x3 of the augmentation stages, `kornia.warp_perspective(x, T, (S,S),
    padding_mode=...)` with kornia 0.6.2's function default `align_corners=True` [UPSTREAM] (`conventions[
    "cached_align_corners"]`), zoom set padded by the iteration's reflection/border mode, wide set filled with the
    iteration's gray; no ColorJitter; fresh noise.

    `live` = the live descriptor table of this iteration.  Stage A degenerates to a copy of the base image and stage B
    carries the composed map: M1 @ C @ M2 with C = the change of normalised coordinates between the stage-B source
    window and the stage-A image, sampled with the GRID_MESH flavour of the cached call's align_corners flag."""
    cv = dict(KORNIA_062_CONVENTIONS, **(conventions or {}))
    ac = bool(cv["cached_align_corners"])
    t = live.double().numpy()
    Hb, Wb = base_size(S, aspect)
    M1 = t[:, 0:9].reshape(-1, 3, 3)
    M2 = t[:, 9:18].reshape(-1, 3, 3)
    win = t[:, 28:32]
    C = np.zeros((cutn, 3, 3))
    for i in range(cutn):
        ox, oy, ww, wh = win[i]
        shift = np.array([[1.0, 0.0, ox], [0.0, 1.0, oy], [0.0, 0.0, 1.0]])
        C[i] = _norm_pixel(Hb, Wb) @ shift @ np.linalg.inv(_norm_pixel(int(wh), int(ww)))
    M = M1 @ C @ M2
    nz = int(0.6 * cutn)
    desc = np.zeros((cutn, DESC_WORDS))
    desc[:, 0:9] = np.eye(3).reshape(9)
    desc[:, 18] = MODE_IDENT
    desc[:, 9:18] = M.reshape(-1, 9)
    desc[:nz, 19] = (MODE_REFLECT_AC if ac else MODE_REFLECT) if reflect else MODE_BORDER
    desc[nz:, 19] = MODE_FILL
    desc[:, 20] = float(fill)
    desc[:, 25] = _np(noise_fac)
    desc[:, 26] = GRID_MESH
    desc[:, 27] = _grid("mesh", ac)
    desc[:, 28:32] = (0.0, 0.0, float(Wb), float(Hb))
    return torch.from_numpy(desc)


class PinnedRing:
    """Host staging for values the host redraws every iteration and the device reads from ONE fixed buffer (so the
    iteration can be replayed from a hipGraph): a small ring of pinned buffers, each guarded by the event recorded after
    its last H2D copy.  The host may run several iterations ahead of the GPU; without the ring it would overwrite a
    pinned buffer whose copy is still queued, and an earlier iteration would see a later iteration's values."""

    def __init__(self, shape, dtype, device, slots: int = 4):
        self.host = [torch.empty(shape, dtype=dtype).pin_memory() for _ in range(slots)]
        self.events = [None] * slots
        self.dev = torch.empty(shape, dtype=dtype, device=device)
        self._i = 0

    def stage(self, value: torch.Tensor) -> torch.Tensor:
        """value (CPU) -> the fixed device buffer, stream-ordered on the current stream"""
        i = self._i
        self._i = (i + 1) % len(self.host)
        if self.events[i] is not None:
            self.events[i].synchronize()          # that slot's previous copy has executed: safe to overwrite
        self.host[i].copy_(value)
        self.dev.copy_(self.host[i], non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(self.dev.device))
        self.events[i] = ev
        return self.dev


class MakeCutouts(nn.Module):
    """Drop-in for the reference's `MakeCutouts(cut_size, cutn, cut_pow=1.)` (pixray.py:400-511):
    `forward(input[1,3,H,W]) -> [cutn,3,S,S]`, autograd-connected to `input`.

    Extra (optional) knobs the reference keeps in module globals: `iteration` (padding-mode parity,
    pixray.py:1250-1253), the RNG `generator`, and `shard=(lo, hi)` to produce only a slice of the
    cutout batch (multi-GPU sharding).  `last_params` / `transforms` hold the draws of
    the latest call (the reference's `.transforms` cache, pixray.py:480-498, is cleared the same way).
    `aspect_width` != 1 selects the reference's non-square-canvas augmentation family (pixray.py:420-431, 468-472)."""

    def __init__(self, cut_size, cutn, cut_pow=1., generator: Optional[torch.Generator] = None, noise_fac: float = 0.1,
                 aspect_width: float = 1.0):
        super().__init__()
        self.aspect_width = aspect_width   # canvas width / height: the reference's module global `global_aspect_width` (pixray.py:1931)
        self.cut_size = cut_size
        self.cutn = cutn
        self.cutn_zoom = int(0.6 * cutn)
        self.cut_pow = cut_pow
        self.noise_fac = noise_fac
        self.transforms = None
        self.generator = generator if generator is not None else torch.Generator().manual_seed(torch.initial_seed() % (2 ** 31))
        self.iteration = 0
        self.shard = None
        self.fill = None           # gray fill for this call (pixray.py:1255-1258); drawn if None
        self.last_params = None
        self.fixed_params = None   # tests / parity: use these draws instead of sampling
        self.conventions = None    # overrides of KORNIA_062_CONVENTIONS (which align_corners flag each kornia call passes)

    def _noise_keys(self):
        if getattr(self, "_noise_gen", None) is None:
            self._noise_gen = torch.Generator().manual_seed((self.generator.initial_seed() * 2654435761 + 0x5bd1e995) % (2 ** 63))
        return self._noise_gen

    # -- host part: draw this iteration's parameters and stage the descriptor table ---------------------------------
    def enable_static_buffers(self, device):
        """Keep the descriptor table in a fixed device buffer (fed from a pinned host buffer) so the device part of
        the iteration can be captured in a hipGraph and replayed."""
        lo, hi = (0, self.cutn) if self.shard is None else self.shard
        self._ring = PinnedRing((hi - lo, DESC_WORDS), torch.float64, device)
        self._static_desc = self._ring.dev

    def prepare(self, iteration=None, fill=None, device=None):
        if iteration is not None:
            self.iteration = iteration
        if fill is not None:
            self.fill = fill
        S = self.cut_size
        prm = self.fixed_params if self.fixed_params is not None else sample_cutout_params(
            self.cutn, S, self.generator, self.iteration, self.noise_fac, fill=self.fill, aspect=self.aspect_width)
        if self.noise_fac and prm.get("noise") is None and "noise_seed" not in prm:
            # the reference's randn_like (pixray.py:510) is drawn inside the stage-B kernel: one Philox key per cutout and
            # iteration, from a stream of its own so that the augmentation draws above stay where they were.  53-bit keys (what a
            # float64 descriptor word holds exactly; both Philox key words are used): no birthday collisions over a run's
            # ~1e5 (cutout, iteration) draws.  The draws do not follow torch's device RNG state the way the reference's
            # randn_like does: same seed -> same run here, but not the reference's noise field (statistically equivalent)
            prm = dict(prm, noise_seed=torch.randint(1, 2 ** 53 - 1, (self.cutn,), generator=self._noise_keys()))
        self.last_params = prm
        desc = build_descriptors(prm, S, self.conventions)
        self.transforms = desc          # this iteration's geometry (opaque, like the reference's composed 3x3 cache)
        lo, hi = (0, self.cutn) if self.shard is None else self.shard
        if getattr(self, "_static_desc", None) is not None:
            self._desc_dev = self._ring.stage(desc[lo:hi])                  # stream-ordered before the replay
        else:
            self._desc_dev = desc[lo:hi].contiguous()
        self._prepared = True

    # -- device part -----------------------------------------------------------------------------------------------------
    def forward(self, input, spot=None):
        spot_mask = None
        if spot is not None:
            # pixray.py:453-458: spot == 0 blanks the pooled pixels OUTSIDE the spot image's bright region, anything else
            # the pixels inside it.  `spot_masks` = (inside, outside) bool [3,S,S] tensors (fetch_spot_indexes, pixray.py:370-394;
            # loading / resizing the spot image is I/O and left to the caller)
            masks = getattr(self, "spot_masks", None)
            if masks is None:
                raise ValueError("spot prompts need MakeCutouts.spot_masks = (inside_mask, outside_mask), bool [3,S,S]")
            spot_mask = (masks[1] if spot == 0 else masks[0]).to(input.device)
        S = self.cut_size
        lo, hi = (0, self.cutn) if self.shard is None else self.shard
        if self.transforms is not None and not getattr(self, "_prepared", False):
            # cached path (pixray.py:480-486): a further call inside the same iteration re-uses this iteration's geometry
            prm = self.last_params
            asp = float(prm["aspect"]) if "aspect" in prm else 1.0
            facs = _uniform(self.generator, (self.cutn,), 0.0, self.noise_fac).float()
            desc = build_cached_descriptors(self.transforms, self.cutn, S, bool(prm["reflect"]), float(prm["fill"]), facs, asp,
                                            self.conventions)
            if self.noise_fac:
                desc[:, 32] = torch.randint(1, 2 ** 53 - 1, (self.cutn,), generator=self._noise_keys()).double()    # noise drawn in the kernel
            return ops.make_cutouts(input, desc[lo:hi].contiguous().to(input.device), None, S, base_size(S, asp), spot_mask)
        if not getattr(self, "_prepared", False):
            self.prepare()
        self._prepared = False
        prm = self.last_params
        desc_dev = self._desc_dev
        if desc_dev.device != input.device:
            desc_dev = desc_dev.to(input.device, non_blocking=True)
        noise = prm.get("noise")          # explicit draws (parity tests); otherwise the kernel draws them (descriptor word 32)
        if noise is not None:
            noise = noise[lo:hi].to(input.device, dtype=torch.float32).contiguous()
        return ops.make_cutouts(input, desc_dev, noise, S, base_size(S, float(prm["aspect"]) if "aspect" in prm else 1.0), spot_mask)
