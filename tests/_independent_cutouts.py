"""TEST INFRASTRUCTURE: a second, independently derived implementation of `MakeCutouts.forward`
(/root/reference/pixray.py:445-511) used to pin `oracle/cutouts_ref.py` end to end (missing item 3).

Nothing here shares code, or a construction, with the oracle:

  oracle/cutouts_ref.py                                   this file
  ------------------------------------------------------  -----------------------------------------------------------------
  torch, fp32 sampling                                    numpy float64 + scipy.ndimage + the standard library's colorsys
  3x3 from an 8x8 linear solve                            3x3 = null vector of the 8x9 DLT system (SVD)
  normalised homographies, F.affine_grid / linspace grid  closed-form PIXEL maps derived from the kornia 0.6.2 conventions:
  and F.grid_sample(align_corners=flag, padding_mode)      flag True : src = M^-1 dst
                                                           flag False: the destination lattice of `affine_grid` is
                                                             d' = (2d+1)(Wd-1)/(2Wd) (warp_affine) or d (warp_perspective,
                                                             whose grid is always linspace(-1,1)); src = M^-1 d'; the sample
                                                             position grid_sample reads is  src * Ws/(Ws-1) - 1/2
                                                          sampled with scipy.ndimage.map_coordinates(order=1): 'grid-constant'
                                                          (zeros), 'nearest' (border), 'reflect' = half-sample symmetric
                                                          (grid_sample's reflection when align_corners is False), 'mirror'
                                                          (when it is True)
  F.adaptive_avg_pool2d / F.adaptive_max_pool2d           windows [floor(i H/S), ceil((i+1) H/S)) written out
  F.interpolate(bilinear, align_corners=False)            src = (d + 1/2) Hs/Hd - 1/2, clamped at 0, edge-replicated
  kornia's tensor HSV round trip                          colorsys.rgb_to_hsv / hsv_to_rgb per pixel

What it can NOT pin is the conventions themselves (which flag each kornia 0.6.2 call passes: the table in the oracle's
docstring) -- those come from the published 0.6.2 sources and stay "from knowledge".  What it does pin is that the oracle
realises those stated conventions correctly through every stage and their composition.
"""
import colorsys

import numpy as np
from scipy import ndimage

CONVENTIONS = {"perspective_align_corners": False, "affine_align_corners": False, "crop_align_corners": True,
               "cached_align_corners": True}


def homography_from_points(src, dst):
    """3x3 H with H @ (x, y, 1) ~ (u, v, 1) for four point pairs: the null vector of the 8x9 DLT system."""
    rows = []
    for (x, y), (u, v) in zip(src, dst):
        rows.append([-x, -y, -1, 0, 0, 0, u * x, u * y, u])
        rows.append([0, 0, 0, -x, -y, -1, v * x, v * y, v])
    _, _, vt = np.linalg.svd(np.asarray(rows, dtype=np.float64))
    H = vt[-1].reshape(3, 3)
    return H / H[2, 2]


def adaptive_pool_pair(img, S):
    """(adaptive average + adaptive max) / 2 of img[C,H,W] to S x S (pixray.py:463)."""
    C, H, W = img.shape
    out = np.zeros((C, S, S))
    for i in range(S):
        y0, y1 = (i * H) // S, -((-(i + 1) * H) // S)
        for j in range(S):
            x0, x1 = (j * W) // S, -((-(j + 1) * W) // S)
            win = img[:, y0:y1, x0:x1].reshape(C, -1)
            out[:, i, j] = (win.mean(1) + win.max(1)) / 2
    return out


def resize_bilinear(img, Hd, Wd):
    """F.interpolate(mode='bilinear', align_corners=False): src = (d + 1/2) * Hs/Hd - 1/2, negative positions clamped to 0."""
    C, Hs, Ws = img.shape
    ys = np.maximum((np.arange(Hd) + 0.5) * Hs / Hd - 0.5, 0.0)
    xs = np.maximum((np.arange(Wd) + 0.5) * Ws / Wd - 0.5, 0.0)
    yy, xx = np.meshgrid(ys, xs, indexing="ij")
    return np.stack([ndimage.map_coordinates(img[c], [yy, xx], order=1, mode="nearest") for c in range(C)])


def _sample(img, xs, ys, padding, align_corners, fill=None):
    mode = {"zeros": "grid-constant", "border": "nearest", "reflection": "mirror" if align_corners else "reflect"}
    if padding == "fill":           # kornia 0.6.2 _fill_and_warp: zero-padded sample + (1 - zero-padded sample of ones) * fill
        cover = ndimage.map_coordinates(np.ones(img.shape[1:]), [ys, xs], order=1, mode="grid-constant", cval=0.0)
        return np.stack([ndimage.map_coordinates(img[c], [ys, xs], order=1, mode="grid-constant", cval=0.0)
                         + (1 - cover) * fill for c in range(img.shape[0])])
    return np.stack([ndimage.map_coordinates(img[c], [ys, xs], order=1, mode=mode[padding], cval=0.0)
                     for c in range(img.shape[0])])


def warp(img, M, dsize, padding, align_corners, kind, fill=None):
    """kornia 0.6.2 warp_perspective (kind 'perspective') / warp_affine (kind 'affine') of img[C,H,W] by the pixel-space 3x3
    M (source -> destination), in closed-form pixel coordinates (module docstring)."""
    C, Hs, Ws = img.shape
    Hd, Wd = dsize
    dy, dx = np.meshgrid(np.arange(Hd, dtype=np.float64), np.arange(Wd, dtype=np.float64), indexing="ij")
    if kind == "affine" and not align_corners:          # affine_grid(align_corners=False) lattice, read in the [0, W-1] convention
        dx = (2 * dx + 1) * (Wd - 1) / (2 * Wd)
        dy = (2 * dy + 1) * (Hd - 1) / (2 * Hd)
    Mi = np.linalg.inv(M)
    den = Mi[2, 0] * dx + Mi[2, 1] * dy + Mi[2, 2]
    sx = (Mi[0, 0] * dx + Mi[0, 1] * dy + Mi[0, 2]) / den
    sy = (Mi[1, 0] * dx + Mi[1, 1] * dy + Mi[1, 2]) / den
    if not align_corners:                               # grid_sample un-normalises with ((g + 1) W - 1) / 2
        sx = sx * Ws / (Ws - 1) - 0.5
        sy = sy * Hs / (Hs - 1) - 0.5
    return _sample(img, sx, sy, padding, align_corners, fill)


def _jitter(img, sat, hue, sat_first):
    """ColorJitter(saturation, hue) through HSV (brightness / contrast neutral): s <- clamp(s * sat, 0, 1); h <- h + hue turns."""
    C, H, W = img.shape
    out = np.empty_like(img)
    for y in range(H):
        for x in range(W):
            r, g, b = img[:, y, x]
            for op in (("s", "h") if sat_first else ("h", "s")):
                h, s, v = colorsys.rgb_to_hsv(r, g, b)
                if op == "s":
                    s = min(max(s * sat, 0.0), 1.0)
                else:
                    h = (h + hue) % 1.0
                r, g, b = colorsys.hsv_to_rgb(h, s, v)
            out[:, y, x] = (r, g, b)
    return out


def _persp(rand, dscale, H, W):
    start = np.array([[0.0, 0.0], [W - 1.0, 0.0], [W - 1.0, H - 1.0], [0.0, H - 1.0]])
    inward = np.array([[1.0, 1.0], [-1.0, 1.0], [-1.0, -1.0], [1.0, -1.0]])
    end = start + np.asarray(rand, dtype=np.float64) * inward * np.array([dscale * W / 2, dscale * H / 2])
    return homography_from_points(start, end)


def base_size(S, aspect):
    if aspect == 1:
        return S, S
    return (S, int(S * aspect)) if aspect > 1 else (int(S * (1 / aspect)), S)


def make_cutouts(img, prm, S, conventions=None):
    """img [3,H,W] float64 in [0,1]; prm: the dict of pixray_amd.cutouts.sample_cutout_params converted to numpy / python values
    (see `params_to_numpy`).  Returns [cutn,3,S,S] float64."""
    cv = dict(CONVENTIONS, **(conventions or {}))
    cutn = int(prm["cutn"])
    nz = int(0.6 * cutn)
    aspect = float(prm.get("aspect", 1.0))
    base = adaptive_pool_pair(img, S)
    Hb, Wb = base_size(S, aspect)
    if (Hb, Wb) != (S, S):
        base = resize_bilinear(base, Hb, Wb)
    pad = "reflection" if int(prm["reflect"]) else "border"
    fill = float(prm["fill"])
    out = []
    for i in range(cutn):
        x = base
        if i < nz:
            if prm["z_persp_apply"][i]:
                x = warp(x, _persp(prm["z_persp_rand"][i], 0.4, Hb, Wb), (Hb, Wb), pad, cv["perspective_align_corners"],
                         "perspective")
            x0, y0, w, h = [float(t) for t in prm["z_crop"][i]]
            box = np.array([[x0, y0], [x0 + w - 1, y0], [x0 + w - 1, y0 + h - 1], [x0, y0 + h - 1]])
            full = np.array([[0.0, 0.0], [S - 1.0, 0.0], [S - 1.0, S - 1.0], [0.0, S - 1.0]])
            Mc = homography_from_points(box, full)
            Mc[2] = (0.0, 0.0, 1.0)                   # crop_by_transform_mat applies it with warp_affine: the first two rows
            x = warp(x, Mc, (S, S), "zeros", cv["crop_align_corners"], "affine")
            key, j = "z", i
        else:
            j = i - nz
            sc = float(prm["w_scale"][j]) if "w_scale" in prm else 0.95
            cx, cy = Wb / 2.0 - 0.5, Hb / 2.0 - 0.5
            Ma = np.array([[sc, 0.0, (1 - sc) * cx + float(prm["w_trans"][j][0])],
                           [0.0, sc, (1 - sc) * cy + float(prm["w_trans"][j][1])], [0.0, 0.0, 1.0]])
            x = warp(x, Ma, (Hb, Wb), "fill", cv["affine_align_corners"], "affine", fill)
            oy, ox = (Hb - S) // 2, (Wb - S) // 2
            x = x[:, oy:oy + S, ox:ox + S]
            if prm["w_persp_apply"][j]:
                x = warp(x, _persp(prm["w_persp_rand"][j], 0.2, S, S), (S, S), "fill", cv["perspective_align_corners"],
                         "perspective", fill)
            key = "w"
        if prm[key + "_jit_apply"][j]:
            x = _jitter(x, float(prm[key + "_sat"][j]), float(prm[key + "_hue"][j]), bool(prm[key + "_sat_first"]))
        if prm.get("noise") is not None:
            x = x + float(prm["noise_fac"][i]) * prm["noise"][i]
        out.append(x)
    return np.stack(out)


def params_to_numpy(p):
    out = {}
    for k, v in p.items():
        out[k] = v.detach().cpu().double().numpy() if hasattr(v, "detach") and v.dtype.is_floating_point else \
            (v.detach().cpu().numpy() if hasattr(v, "detach") else v)
    return out


def make_cutouts_cached(img, prm, S, conventions=None):
    """the cached-transform path (pixray.py:480-486): ONE warp_perspective per cutout by the composed pixel-space 3x3 of its
    geometric stages (identity for a stage whose apply mask is off), no ColorJitter; fresh noise added by the caller."""
    cv = dict(CONVENTIONS, **(conventions or {}))
    cutn = int(prm["cutn"])
    nz = int(0.6 * cutn)
    aspect = float(prm.get("aspect", 1.0))
    base = adaptive_pool_pair(img, S)
    Hb, Wb = base_size(S, aspect)
    if (Hb, Wb) != (S, S):
        base = resize_bilinear(base, Hb, Wb)
    pad = "reflection" if int(prm["reflect"]) else "border"
    fill = float(prm["fill"])
    out = []
    for i in range(cutn):
        T = np.eye(3)
        if i < nz:
            if prm["z_persp_apply"][i]:
                T = _persp(prm["z_persp_rand"][i], 0.4, Hb, Wb)
            x0, y0, w, h = [float(t) for t in prm["z_crop"][i]]
            # an axis-aligned box onto the full S x S frame is a scale + shift per axis
            Mc = np.array([[(S - 1) / (w - 1), 0.0, -x0 * (S - 1) / (w - 1)], [0.0, (S - 1) / (h - 1), -y0 * (S - 1) / (h - 1)],
                           [0.0, 0.0, 1.0]])
            T = Mc @ T
            out.append(warp(base, T, (S, S), pad, cv["cached_align_corners"], "perspective"))
        else:
            j = i - nz
            sc = float(prm["w_scale"][j]) if "w_scale" in prm else 0.95
            cx, cy = Wb / 2.0 - 0.5, Hb / 2.0 - 0.5
            T = np.array([[sc, 0.0, (1 - sc) * cx + float(prm["w_trans"][j][0])],
                          [0.0, sc, (1 - sc) * cy + float(prm["w_trans"][j][1])], [0.0, 0.0, 1.0]])
            T = np.array([[1.0, 0.0, -float((Wb - S) // 2)], [0.0, 1.0, -float((Hb - S) // 2)], [0.0, 0.0, 1.0]]) @ T
            if prm["w_persp_apply"][j]:
                T = _persp(prm["w_persp_rand"][j], 0.2, S, S) @ T
            out.append(warp(base, T, (S, S), "fill", cv["cached_align_corners"], "perspective", fill))
    return np.stack(out)
