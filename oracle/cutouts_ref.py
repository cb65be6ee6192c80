"""ORACLE (test infrastructure only -- never imported by the product package).

CPU fp32 restatement of `MakeCutouts.forward` (/root/reference/pixray.py:445-511, ctor 401-443,
kornia overrides 326-366) with every random draw made an explicit input (`params`), because
the reference's RNG streams (kornia parameter generators, device randn) cannot be reproduced.

The geometric and colour operators are kornia==0.6.2 (/root/reference/requirements.txt:18),
which is NOT vendored; their published algorithms are restated here the way kornia builds
them -- normalised homographies + `F.grid_sample` / `F.affine_grid` -- following
SURVEY.md Appendix A.3.  The product computes the same maps in pixel space inside one HIP
kernel, so agreement between the two is a real check of both.  Parity status: the CONVENTIONS
(which align_corners flag / padding each kornia 0.6.2 call passes: table below) are **unpinned** --
from the published 0.6.2 sources, kornia is not installable offline and the reference's tests hold
no fixture for this path.  The REALISATION of those conventions is pinned end to end (value, each
flag flipped, image gradient by finite differences) by a second implementation that shares nothing
with this file: numpy float64 closed-form pixel maps + scipy.ndimage.map_coordinates + colorsys
(tests/_independent_cutouts.py, tests/test_oracle_pins.py).

Pipeline (pixray.py):
  pooled = (AdaptiveAvgPool2d(S)(img) + AdaptiveMaxPool2d(S)(img)) / 2            461-463 (same for every cutout)
  zoom set, first int(0.6*cutn) cutouts                                            414-417, 493
     MyRandomPerspective(0.4, p=.7)  [padding_mode = reflection / border by iteration parity, 1250-1253]
     RandomResizedCrop(S, scale=(.25,.95), ratio=(.85,1.2), cropping_mode='resample')
     ColorJitter(hue=.1, saturation=.1, p=.8)
  wide set, the rest                                                               419-437, 494
     MyRandomAffine(translate=±2.5%, scale=.95, padding 'fill' gray)
     CenterCrop(S) on an S x S image (identity resample, align_corners=True)
     MyRandomPerspectivePadded(0.2, p=.7, padding 'fill' gray)
     ColorJitter(hue=.1, saturation=.1, p=.8)
  batch + U(0, 0.1) * N(0,1)                                                       508-510

kornia 0.6.2 defaults this restatement relies on (from the published 0.6.2 sources, `kornia/augmentation/augmentation.py`
and `kornia/geometry/transform/{imgwarp,crop2d}.py`; NOT checkable offline -> **parity unpinned**).  pixray passes none of
these flags itself (pixray.py:414-436), so each call runs on its constructor / function default:

  call in pixray.py                               kornia 0.6.2 definition                                     align_corners  resample   padding
  ----------------------------------------------  ----------------------------------------------------------  -------------  ---------  ---------------------------
  MyRandomPerspective (326-337, ctor 414)          RandomPerspective(distortion_scale, resample=BILINEAR,      False          bilinear   global_padding_mode:
                                                   align_corners=False, p) -> warp_perspective(input, T,                                'reflection' (even it) /
                                                   (h,w), mode, align_corners=flags['align_corners'], ...)                              'border' (odd it), 1250-1253
  K.RandomResizedCrop(cropping_mode='resample')    RandomResizedCrop(size, scale, ratio, resample=BILINEAR,    **True**       bilinear   'zeros' (hard-coded in
    (415)                                          align_corners=True, p=1., cropping_mode) ->                                           crop_by_transform_mat call)
                                                   crop_by_transform_mat(input, T, size, mode, 'zeros',
                                                   align_corners=flags['align_corners']) -> warp_affine
  K.ColorJitter(hue, saturation, p) (416, 436)     brightness / contrast factors sampled from [1,1] -> a       -              -          -
                                                   clamp to [0,1] each (identity on [0,1] images); saturation
                                                   U(.9,1.1), hue U(-.1,.1)*2pi in HSV, all four in a random
                                                   order drawn once per call
  MyRandomAffine (340-353, ctor 420-431)           RandomAffine(degrees, translate, scale, resample=BILINEAR,  False          bilinear   'fill', fill_value =
                                                   align_corners=False, padding_mode=ZEROS, p) ->                                        global_fill_color
                                                   warp_affine(input, T[:, :2], (h,w), mode, align_corners=
                                                   flags['align_corners'], padding_mode='fill', fill_value)
  K.CenterCrop(size, cropping_mode='resample')     CenterCrop(size, align_corners=True, resample=BILINEAR,     **True**       bilinear   'zeros'
    (433)                                          p=1., cropping_mode) -> crop_by_transform_mat(...);
                                                   center_crop_generator truncates the window origin to an
                                                   integer, so with align_corners=True the resample is an
                                                   exact copy of that window
  MyRandomPerspectivePadded (355-366, ctor 434)    RandomPerspective(... align_corners=False ...)              False          bilinear   'fill', fill_value
  kornia.geometry.transform.warp_perspective       warp_perspective(src, M, dsize, mode='bilinear',            **True**       bilinear   global_padding_mode /
    on the cached transforms (482-485)             padding_mode='zeros', align_corners=True, fill_value)       (fn default)              'fill'
  kornia.geometry.transform.rescale (469-472)      rescale(input, factor, interpolation='bilinear',            None (=False)  bilinear   -
                                                   align_corners=None) -> F.interpolate

Both warps normalise the 3x3 with `normalize_homography` ([0, W-1] -> [-1, 1], i.e. the align_corners=True geometry)
whatever the flag, build the destination grid with `create_meshgrid(normalized_coordinates=True)` (warp_perspective: always
linspace(-1, 1)) or `F.affine_grid(theta, size, align_corners=flag)` (warp_affine), and hand the flag to `F.grid_sample`.
With the flag False the map is therefore NOT the exact pixel map of the matrix (a half-pixel-class scale/shift that the
reference really has); with the flag True it is.  `CONVENTIONS` below carries the flags; `make_cutouts(..., conventions=)`
and the product's descriptor builder (`pixray_amd.cutouts.build_descriptors(..., conventions=)`) take the same dict, so a
test can flip one and watch the HIP kernel follow (tests/test_path_gpu.py::test_cutout_align_corners_convention_is_a_
descriptor_field).  Round 1 of this repository had `crop_align_corners` False on both sides; that was a misreading of the
RandomResizedCrop constructor default.
"""
from __future__ import annotations

import math
from typing import Dict

import torch
import torch.nn.functional as F

TWO_PI = 2.0 * math.pi

# the align_corners flag each kornia 0.6.2 call passes to F.grid_sample (table in the module docstring)
CONVENTIONS = {"perspective_align_corners": False, "affine_align_corners": False, "crop_align_corners": True,
               "cached_align_corners": True}


# ---------------------------------------------------------------- kornia geometry [UPSTREAM]
def get_perspective_transform(src: torch.Tensor, dst: torch.Tensor) -> torch.Tensor:
    """kornia.geometry.transform.get_perspective_transform: 3x3 M with M @ src_i ~ dst_i (DLT, 8x8 solve)."""
    B = src.shape[0]
    rows = []
    for i in range(4):
        x, y = src[:, i, 0], src[:, i, 1]
        u, v = dst[:, i, 0], dst[:, i, 1]
        o, z = torch.ones_like(x), torch.zeros_like(x)
        rows.append(torch.stack([x, y, o, z, z, z, -x * u, -y * u], dim=1))
        rows.append(torch.stack([z, z, z, x, y, o, -x * v, -y * v], dim=1))
    A = torch.stack(rows, dim=1)
    b = dst.reshape(B, 8, 1)
    X = torch.linalg.solve(A, b)
    M = torch.cat([X[:, :, 0], torch.ones(B, 1, dtype=src.dtype)], dim=1).reshape(B, 3, 3)
    return M


def _normal_transform_pixel(h: int, w: int, dtype) -> torch.Tensor:
    """kornia normal_transform_pixel: pixel [0, W-1] -> [-1, 1]."""
    t = torch.tensor([[1.0, 0.0, -1.0], [0.0, 1.0, -1.0], [0.0, 0.0, 1.0]], dtype=dtype)
    wd = 1e-14 if w == 1 else w - 1.0
    hd = 1e-14 if h == 1 else h - 1.0
    t[0, 0] = t[0, 0] * 2.0 / wd
    t[1, 1] = t[1, 1] * 2.0 / hd
    return t


def normalize_homography(M, src_hw, dst_hw):
    """kornia normalize_homography: dst_norm <- src_norm."""
    sn = _normal_transform_pixel(src_hw[0], src_hw[1], M.dtype)
    dn = _normal_transform_pixel(dst_hw[0], dst_hw[1], M.dtype)
    return dn @ (M @ torch.linalg.inv(sn))


def _meshgrid_norm(h, w, dtype):
    """kornia create_meshgrid(normalized_coordinates=True): linspace(-1, 1) (the [0, W-1] convention)."""
    xs = (torch.linspace(0, w - 1, w, dtype=dtype) / (w - 1) - 0.5) * 2
    ys = (torch.linspace(0, h - 1, h, dtype=dtype) / (h - 1) - 0.5) * 2
    gy, gx = torch.meshgrid(ys, xs, indexing="ij")
    return torch.stack([gx, gy], dim=-1)  # [h, w, 2]


def _fill_and_warp(src, grid, mode, align_corners, fill_value):
    """kornia 0.6.2 _fill_and_warp (padding_mode='fill')."""
    ones = torch.ones_like(src)
    fv = fill_value.to(src)[None, :, None, None]
    inv_ones = 1 - F.grid_sample(ones, grid, align_corners=align_corners, mode=mode, padding_mode="zeros")
    return F.grid_sample(src, grid, align_corners=align_corners, mode=mode, padding_mode="zeros") + inv_ones * fv


def warp_perspective(src, M, dsize, padding_mode, align_corners, fill_value=None):
    """kornia warp_perspective (bilinear). M maps src pixels -> dst pixels. Geometry in float64, sampling fp32."""
    B, _, H, W = src.shape
    Md = M.double()
    dn_sn = normalize_homography(Md, (H, W), dsize)
    sn_dn = torch.linalg.inv(dn_sn)
    grid = _meshgrid_norm(dsize[0], dsize[1], torch.float64)[None].expand(B, -1, -1, -1)
    hom = torch.cat([grid, torch.ones_like(grid[..., :1])], dim=-1)           # [B,h,w,3]
    pts = torch.einsum("bij,bhwj->bhwi", sn_dn, hom)
    z = pts[..., 2:3]
    scale = torch.where(z.abs() > 1e-8, 1.0 / z, torch.ones_like(z))
    g = (pts[..., :2] * scale).to(src.dtype)
    if padding_mode == "fill":
        return _fill_and_warp(src, g, "bilinear", align_corners, fill_value)
    return F.grid_sample(src, g, align_corners=align_corners, mode="bilinear", padding_mode=padding_mode)


def warp_affine(src, M2x3, dsize, padding_mode, align_corners, fill_value=None):
    """kornia warp_affine (bilinear): affine_grid + grid_sample."""
    B, C, H, W = src.shape
    M = torch.cat([M2x3.double(), torch.tensor([[[0.0, 0.0, 1.0]]], dtype=torch.float64).expand(B, 1, 3)], dim=1)
    dn_sn = normalize_homography(M, (H, W), dsize)
    sn_dn = torch.linalg.inv(dn_sn)
    g = F.affine_grid(sn_dn[:, :2, :].to(src.dtype), [B, C, dsize[0], dsize[1]], align_corners=align_corners)
    if padding_mode == "fill":
        return _fill_and_warp(src, g, "bilinear", align_corners, fill_value)
    return F.grid_sample(src, g, align_corners=align_corners, mode="bilinear", padding_mode=padding_mode)


# ---------------------------------------------------------------- kornia colour [UPSTREAM]
def rgb_to_hsv(image, eps=1e-8):
    max_rgb, argmax_rgb = image.max(-3)
    min_rgb, _ = image.min(-3)
    deltac = max_rgb - min_rgb
    v = max_rgb
    s = deltac / (max_rgb + eps)
    deltac = torch.where(deltac == 0, torch.ones_like(deltac), deltac)
    rc, gc, bc = torch.unbind(max_rgb.unsqueeze(-3) - image, dim=-3)
    h1 = bc - gc
    h2 = (rc - bc) + 2.0 * deltac
    h3 = (gc - rc) + 4.0 * deltac
    h = torch.stack((h1, h2, h3), dim=-3) / deltac.unsqueeze(-3)
    h = torch.gather(h, dim=-3, index=argmax_rgb.unsqueeze(-3)).squeeze(-3)
    h = (h / 6.0) % 1.0
    h = TWO_PI * h
    return torch.stack((h, s, v), dim=-3)


def hsv_to_rgb(image):
    h = image[..., 0, :, :] / TWO_PI
    s = image[..., 1, :, :]
    v = image[..., 2, :, :]
    hi = torch.floor(h * 6) % 6
    f = ((h * 6) % 6) - hi
    one = torch.tensor(1.0, dtype=image.dtype)
    p = v * (one - s)
    q = v * (one - f * s)
    t = v * (one - (one - f) * s)
    hi = hi.long()
    indices = torch.stack([hi, hi + 6, hi + 12], dim=-3)
    out = torch.stack((v, q, p, p, t, v, t, v, v, q, p, p, p, p, t, v, v, q), dim=-3)
    return torch.gather(out, -3, indices)


def adjust_saturation(img, factor):
    hsv = rgb_to_hsv(img)
    h, s, v = hsv[:, 0:1], hsv[:, 1:2], hsv[:, 2:3]
    s = torch.clamp(s * factor.view(-1, 1, 1, 1), 0.0, 1.0)
    return hsv_to_rgb(torch.cat([h, s, v], dim=1))


def adjust_hue(img, factor_rad):
    hsv = rgb_to_hsv(img)
    h, s, v = hsv[:, 0:1], hsv[:, 1:2], hsv[:, 2:3]
    h = torch.fmod(h + factor_rad.view(-1, 1, 1, 1), TWO_PI)
    return hsv_to_rgb(torch.cat([h, s, v], dim=1))


def color_jitter(img, apply, sat, hue, sat_first: bool):
    """kornia ColorJitter(hue=.1, saturation=.1): brightness/contrast factors are neutral (their code path
    is a clamp to [0,1], an identity here); saturation and hue applied in the batch's random order."""
    x = img
    if sat_first:
        x = adjust_hue(adjust_saturation(x, sat), hue * TWO_PI)
    else:
        x = adjust_saturation(adjust_hue(x, hue * TWO_PI), sat)
    return torch.where(apply.view(-1, 1, 1, 1), x, img)


# ---------------------------------------------------------------- MakeCutouts
def pooled_image(img, S):
    return (F.adaptive_avg_pool2d(img, (S, S)) + F.adaptive_max_pool2d(img, (S, S))) / 2


def _persp_matrix(rand, dscale, S, W=None):
    """kornia random_perspective_generator on an S x W image (W defaults to S): corner i moves inwards by
    rand * distortion_scale * (W/2, S/2)"""
    W = S if W is None else W
    B = rand.shape[0]
    start = torch.tensor([[0.0, 0.0], [W - 1.0, 0.0], [W - 1.0, S - 1.0], [0.0, S - 1.0]], dtype=torch.float64)
    start = start[None].expand(B, 4, 2)
    f = torch.tensor([dscale * W / 2, dscale * S / 2], dtype=torch.float64)
    pts_norm = torch.tensor([[1.0, 1.0], [-1.0, 1.0], [-1.0, -1.0], [1.0, -1.0]], dtype=torch.float64)
    end = start + f * rand.double() * pts_norm[None]
    return get_perspective_transform(start, end)


def base_size(S, aspect):
    """size of the pooled cutout after the aspect rescale (pixray.py:468-472; kornia rescale -> int(size * factor))"""
    if aspect == 1:
        return S, S
    return (S, int(S * aspect)) if aspect > 1 else (int(S * (1 / aspect)), S)


def base_image(img, S, aspect, spot_mask=None):
    """pooled cutout (pixray.py:463), blanked where the spot mask is set (pixray.py:465-466), rescaled to the canvas aspect
    when it is not square (pixray.py:468-472: kornia.geometry.transform.rescale = bilinear F.interpolate,
    align_corners=False)"""
    base = pooled_image(img, S)
    if spot_mask is not None:
        base = torch.where(spot_mask[None], torch.zeros_like(base), base)
    Hb, Wb = base_size(S, aspect)
    if (Hb, Wb) != (S, S):
        base = F.interpolate(base, size=(Hb, Wb), mode="bilinear", align_corners=False)
    return base


def _wide_affine(prm, nw, Hb, Wb):
    """MyRandomAffine (pixray.py:420-431): isotropic scale about the image centre (w/2 - 0.5, h/2 - 0.5) + translation"""
    sc = prm["w_scale"].double() if "w_scale" in prm else torch.full((nw,), 0.95, dtype=torch.float64)
    cx, cy = Wb / 2.0 - 0.5, Hb / 2.0 - 0.5
    Ma = torch.zeros(nw, 3, 3, dtype=torch.float64)
    Ma[:, 0, 0] = sc
    Ma[:, 1, 1] = sc
    Ma[:, 2, 2] = 1.0
    Ma[:, 0, 2] = (1 - sc) * cx + prm["w_trans"][:, 0].double()
    Ma[:, 1, 2] = (1 - sc) * cy + prm["w_trans"][:, 1].double()
    return Ma


def make_cutouts(img: torch.Tensor, prm: Dict[str, torch.Tensor], S: int, spot_mask=None, conventions=None) -> torch.Tensor:
    """MakeCutouts.forward(img[1,3,H,W]) -> [cutn,3,S,S] with explicit randomness `prm`
    (see pixray_amd.cutouts.sample_cutout_params for the fields; prm["aspect"] = canvas width / height).
    `conventions`: overrides of CONVENTIONS (which align_corners flag each kornia call passes)."""
    cv = dict(CONVENTIONS, **(conventions or prm.get("conventions") or {}))
    ac_p, ac_a, ac_c = cv["perspective_align_corners"], cv["affine_align_corners"], cv["crop_align_corners"]
    cutn = int(prm["cutn"])
    nz = int(0.6 * cutn)                                   # pixray.py:407
    nw = cutn - nz
    aspect = float(prm["aspect"]) if "aspect" in prm else 1.0
    base = base_image(img, S, aspect, spot_mask)           # pixray.py:463-472
    Hb, Wb = base.shape[-2:]
    pad_mode = "reflection" if int(prm["reflect"]) else "border"   # pixray.py:1250-1253
    fill = torch.full((3,), float(prm["fill"]), dtype=img.dtype)   # pixray.py:1255-1258
    outs = []
    if nz > 0:
        x = base.expand(nz, -1, -1, -1)
        Mp = _persp_matrix(prm["z_persp_rand"], 0.4, Hb, Wb)
        warped = warp_perspective(x, Mp, (Hb, Wb), pad_mode, align_corners=ac_p)
        x = torch.where(prm["z_persp_apply"].view(-1, 1, 1, 1), warped, x)
        xs, ys, w, h = [prm["z_crop"][:, i].double() for i in range(4)]
        src = torch.stack([torch.stack([xs, ys], 1), torch.stack([xs + w - 1, ys], 1),
                           torch.stack([xs + w - 1, ys + h - 1], 1), torch.stack([xs, ys + h - 1], 1)], dim=1)
        dst = torch.tensor([[0.0, 0.0], [S - 1.0, 0.0], [S - 1.0, S - 1.0], [0.0, S - 1.0]], dtype=torch.float64)
        Mc = get_perspective_transform(src, dst[None].expand(nz, 4, 2))
        x = warp_affine(x, Mc[:, :2, :], (S, S), "zeros", align_corners=ac_c)     # crop_by_transform_mat, RandomResizedCrop's flag
        x = color_jitter(x, prm["z_jit_apply"], prm["z_sat"], prm["z_hue"], bool(prm["z_sat_first"]))
        outs.append(x)
    if nw > 0:
        x = base.expand(nw, -1, -1, -1)
        Ma = _wide_affine(prm, nw, Hb, Wb)
        x = warp_affine(x, Ma[:, :2, :], (Hb, Wb), "fill", align_corners=ac_a, fill_value=fill)
        # CenterCrop(S) (pixray.py:433): center_crop_generator truncates the window origin to an integer and the crop is
        # resampled with align_corners=True -> an exact copy of the centred S x S window
        oy, ox = (Hb - S) // 2, (Wb - S) // 2
        x = x[:, :, oy:oy + S, ox:ox + S]
        Mp = _persp_matrix(prm["w_persp_rand"], 0.2, S)
        warped = warp_perspective(x, Mp, (S, S), "fill", align_corners=ac_p, fill_value=fill)
        x = torch.where(prm["w_persp_apply"].view(-1, 1, 1, 1), warped, x)
        x = color_jitter(x, prm["w_jit_apply"], prm["w_sat"], prm["w_hue"], bool(prm["w_sat_first"]))
        outs.append(x)
    batch = torch.cat(outs, dim=0)
    if "noise" in prm and prm["noise"] is not None:
        batch = batch + prm["noise_fac"].view(-1, 1, 1, 1) * prm["noise"]   # pixray.py:508-510
    return batch


# ---- cached-transform path (pixray.py:480-486) ------------------------------------------------
def composed_transforms(prm: Dict[str, torch.Tensor], S: int) -> torch.Tensor:
    """`self.transforms` (pixray.py:498): kornia's composed pixel-space 3x3 (aspect-rescaled cutout -> output) of the
    geometric augmentations of each cutout, identity for a stage whose apply-mask is off.  [cutn,3,3] float64."""
    cutn = int(prm["cutn"])
    nz = int(0.6 * cutn)
    nw = cutn - nz
    aspect = float(prm["aspect"]) if "aspect" in prm else 1.0
    Hb, Wb = base_size(S, aspect)
    eye = torch.eye(3, dtype=torch.float64)
    out = []
    if nz > 0:
        Mp = _persp_matrix(prm["z_persp_rand"], 0.4, Hb, Wb)
        Mp = torch.where(prm["z_persp_apply"].view(-1, 1, 1), Mp, eye[None].expand(nz, 3, 3))
        xs, ys, w, h = [prm["z_crop"][:, i].double() for i in range(4)]
        src = torch.stack([torch.stack([xs, ys], 1), torch.stack([xs + w - 1, ys], 1),
                           torch.stack([xs + w - 1, ys + h - 1], 1), torch.stack([xs, ys + h - 1], 1)], dim=1)
        dst = torch.tensor([[0.0, 0.0], [S - 1.0, 0.0], [S - 1.0, S - 1.0], [0.0, S - 1.0]], dtype=torch.float64)
        Mc = get_perspective_transform(src, dst[None].expand(nz, 4, 2)).clone()
        Mc[:, 2, :] = torch.tensor([0.0, 0.0, 1.0], dtype=torch.float64)          # the crop is applied as an affine map
        out.append(Mc @ Mp)
    if nw > 0:
        Ma = _wide_affine(prm, nw, Hb, Wb)
        crop = eye[None].repeat(nw, 1, 1)                                          # CenterCrop: shift by the window origin
        crop[:, 0, 2] = -float((Wb - S) // 2)
        crop[:, 1, 2] = -float((Hb - S) // 2)
        Mp = _persp_matrix(prm["w_persp_rand"], 0.2, S)
        Mp = torch.where(prm["w_persp_apply"].view(-1, 1, 1), Mp, eye[None].expand(nw, 3, 3))
        out.append(Mp @ crop @ Ma)
    return torch.cat(out, dim=0)


def make_cutouts_cached(img: torch.Tensor, prm: Dict[str, torch.Tensor], S: int, noise_fac=None, noise=None, spot_mask=None,
                        conventions=None) -> torch.Tensor:
    """MakeCutouts.forward when `.transforms` is cached (pixray.py:480-486; image prompts, pixray.py:1318-1333): ONE
    `kornia.warp_perspective(cutout, T, (S,S), padding_mode=...)` per set -- kornia 0.6.2's default align_corners=True
    [UPSTREAM, from knowledge: parity unpinned], zoom set with the iteration's reflection/border padding, wide set filled
    with the iteration's gray -- no ColorJitter, then fresh noise."""
    ac = dict(CONVENTIONS, **(conventions or {}))["cached_align_corners"]
    cutn = int(prm["cutn"])
    nz = int(0.6 * cutn)
    aspect = float(prm["aspect"]) if "aspect" in prm else 1.0
    T = composed_transforms(prm, S)
    base = base_image(img, S, aspect, spot_mask)
    pad_mode = "reflection" if int(prm["reflect"]) else "border"
    fill = torch.full((3,), float(prm["fill"]), dtype=img.dtype)
    outs = []
    if nz > 0:
        outs.append(warp_perspective(base.expand(nz, -1, -1, -1), T[:nz], (S, S), pad_mode, align_corners=ac))
    if cutn - nz > 0:
        outs.append(warp_perspective(base.expand(cutn - nz, -1, -1, -1), T[nz:], (S, S), "fill", align_corners=ac, fill_value=fill))
    batch = torch.cat(outs, dim=0)
    if noise is not None:
        batch = batch + noise_fac.view(-1, 1, 1, 1).to(batch) * noise
    return batch
