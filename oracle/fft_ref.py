"""ORACLE (test infrastructure only -- never imported by the product package).

CPU restatement of the fft drawer's spectrum -> image map (BASELINE.json configs[3]; /root/reference/fftdrawer.py:45-62
`init_from_tensor`, 79-86 `synth`: `fft_image(shape, sd=0.01, decay_power)` wrapped by `to_valid_rgb(image_f, colors=1.5)`
and evaluated as `image_f(contrast=0.9)`).

The arithmetic lives in a dependency that is NOT in /root/reference: `aphantasia.image` of eps696/aphantasia @7e6b3bb
(/root/reference/requirements.txt).  Its published algorithm, restated:

    params  [1, 3, H, Wf, 2]   real / imaginary parts, N(0, sd = 0.01);  Wf = W//2 + 1 (W even) or W//2 + 2 (W odd: the lucid
                               frequency helper keeps one surplus column, which the inverse transform of width W drops)
    f[u,v]  = sqrt(fftfreq(H)[u]^2 + fftfreq(W)[v]^2)                                   (rfft2d_freqs)
    scale   = sqrt(W*H) / max(f, 1/max(W,H))^decay
    image   = irfft2(scale * params as complex, s = (H, W), norm = "ortho")
    image   = image * contrast / image.std()                                            (unbiased std over all elements)
    rgb     = sigmoid( image projected through  M / max column norm,  M = color_correlation_svd_sqrt / (colors, 1, 1) )

No FFT library is used here: the inverse real transform is written out as its two DFT sums in float64 (matrix products with
explicit twiddle matrices), which is what `irfft2` is defined to compute -- complex inverse DFT along H, then the
complex-to-real inverse DFT along W in which the half spectrum stands for its Hermitian extension, so that the imaginary
parts of the DC column and (W even) the Nyquist column do not contribute:

    Y[h,v] = sum_u S[u,v] e^{+2 pi i u h / H}
    x[h,w] = Re( Y[h,0] + 2 sum_{0 < v < W/2} Y[h,v] e^{+2 pi i v w / W} + [W even] Y[h,W/2] (-1)^w ) / sqrt(H W)

so the product's `torch.fft.irfftn` (pocketfft on the CPU, rocFFT on MI355X) is checked against the definition, not against
another FFT.  Parity status: the map is pinned to the definition of the transform and to the reference's call site
(fftdrawer.py:57,62,84: sd 0.01, decay from --fft_decay, colors 1.5, contrast 0.9); the aphantasia constants (the colour
matrix, the 1/max(W,H) floor) are from the published source, not checkable offline: **parity unpinned** for those.
"""
import math

import numpy as np
import torch

COLOR_CORRELATION_SVD_SQRT = ((0.26, 0.09, 0.02), (0.27, 0.00, -0.05), (0.27, -0.09, 0.03))


def n_freq_columns(w: int) -> int:
    return w // 2 + (2 if w % 2 == 1 else 1)


def radial_freqs(h: int, w: int) -> np.ndarray:
    fy = np.array([(u if u < (h + 1) // 2 else u - h) / h for u in range(h)], dtype=np.float64)          # fftfreq(h)
    fx = np.array([(v if v < (w + 1) // 2 else v - w) / w for v in range(w)], dtype=np.float64)[:n_freq_columns(w)]
    return np.sqrt(fx[None, :] ** 2 + fy[:, None] ** 2)


def rand_init(size, seed: int) -> torch.Tensor:
    """fftdrawer.py:57 `fft_image(shape, sd=0.01, ...)` with no resume image: the spectrum the optimiser owns (seeded the way
    the product's drawer seeds it, so that both start from the same tensor)"""
    w, h = size
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(1, 3, h, n_freq_columns(w), 2, generator=g) * 0.01).requires_grad_(True)


def colour_matrix(colors: float) -> torch.Tensor:
    m = np.asarray(COLOR_CORRELATION_SVD_SQRT, dtype=np.float64) / np.asarray([colors, 1.0, 1.0])
    m = m / np.linalg.norm(m, axis=0).max()
    return torch.tensor(m.T)               # [c, d]: rgb_d = sum_c image_c * M[d, c]


def inverse_real_dft2(spec: torch.Tensor, h: int, w: int) -> torch.Tensor:
    """spec [..., H, >= W//2+1, 2] (float64) -> [..., H, W]: the two DFT sums of the module docstring, "ortho" scaling"""
    wh = w // 2 + 1
    re, im = spec[..., :wh, 0], spec[..., :wh, 1]
    u = torch.arange(h, dtype=torch.float64)
    ang_h = 2 * math.pi * torch.outer(u, u) / h                              # [h_out, u]
    ch, sh = torch.cos(ang_h), torch.sin(ang_h)
    yr = ch @ re - sh @ im                                                   # Y = E_H S
    yi = sh @ re + ch @ im
    v = torch.arange(wh, dtype=torch.float64)
    ang_w = 2 * math.pi * torch.outer(v, torch.arange(w, dtype=torch.float64)) / w          # [v, w_out]
    weight = torch.full((wh,), 2.0, dtype=torch.float64)
    weight[0] = 1.0
    if w % 2 == 0:
        weight[-1] = 1.0
    cw, sw = torch.cos(ang_w) * weight[:, None], torch.sin(ang_w) * weight[:, None]
    return (yr @ cw - yi @ sw) / math.sqrt(h * w)                            # Re(Y e^{i ang})


def synth(params: torch.Tensor, size, decay: float = 1.5, contrast: float = 0.9, colors: float = 1.5) -> torch.Tensor:
    """spectrum [1,3,H,Wf,2] -> image [1,3,H,W] in (0,1) (fftdrawer.py:79-86), float32 out, differentiable"""
    w, h = size
    freqs = radial_freqs(h, w)
    scale = math.sqrt(w * h) / np.maximum(freqs, 1.0 / max(w, h)) ** decay
    spec = params.double() * torch.tensor(scale)[None, None, :, :, None]
    image = inverse_real_dft2(spec, h, w)
    image = image * contrast / image.std()
    rgb = torch.einsum("nchw,cd->ndhw", image, colour_matrix(colors))
    return torch.sigmoid(rgb).float()
