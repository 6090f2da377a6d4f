"""Slice-profile kernels (host side, float64 closed forms).

Counterpart of ``nitorch.core.kernels.smooth`` as called at
unires/_project.py:277: profile (dirac | rect | tri | gauss) convolved with the
linear-interpolation basis, sampled at integer offsets, normalised to sum 1.
nitorch is absent from the build container, so support lengths follow its
published behaviour; the Gaussian truncation is a parameter.
"""
import math

import numpy as np


def _tri_cdf(t):
    """Integral of the unit triangle max(0, 1-|s|) from -inf to t."""
    t = np.clip(t, -1.0, 1.0)
    return np.where(t < 0, 0.5 * (t + 1.0) ** 2, 1.0 - 0.5 * (1.0 - t) ** 2)


def _tri_w_second_antiderivative(t, w):
    """G with G'' = (1/w) max(0, 1-|t|/w)  (unit-area triangle of FWHM w)."""
    t = np.asarray(t, dtype=np.float64)
    lo = np.clip(t, -w, 0.0)
    g_neg = (lo + w) ** 3 / (6.0 * w * w)
    mid = np.clip(t, 0.0, w)
    g_mid = mid - (w ** 3 - (w - mid) ** 3) / (6.0 * w * w)
    g_hi = np.maximum(t - w, 0.0)
    return np.where(t <= 0, g_neg, w / 6.0 + g_mid + g_hi)


def smooth1d(kind, fwhm, gauss_lim=None):
    """1-D kernel as a float64 numpy vector. kind: -1 dirac, 0 rect, 1 tri, 2 gauss.
    (Gaussian support: `floor((4 fwhm + 2) / 2)` unless ``gauss_lim`` says otherwise - the truncation nitorch
    applies is unpinned here, see oracle/nitorch_restated.py:smooth1d; the taps beyond the narrower candidate
    are ~1e-8 of the sum.)"""
    w = float(fwhm)
    if kind == -1:
        return np.ones(1)
    if kind == 0:
        lim = int(math.floor((w + 2.0) / 2.0))
        x = np.arange(-lim, lim + 1, dtype=np.float64)
        ker = (_tri_cdf(x + w / 2.0) - _tri_cdf(x - w / 2.0)) / w
    elif kind == 1:
        lim = int(math.floor((2.0 * w + 2.0) / 2.0))
        x = np.arange(-lim, lim + 1, dtype=np.float64)
        G = lambda t: _tri_w_second_antiderivative(t, w)
        ker = G(x + 1.0) - 2.0 * G(x) + G(x - 1.0)
    elif kind == 2:
        lim = int(math.floor((4.0 * w + 2.0) / 2.0)) if gauss_lim is None else int(gauss_lim)
        x = np.arange(-lim, lim + 1, dtype=np.float64)
        s = (w / math.sqrt(8.0 * math.log(2.0))) ** 2 + 1e-12
        w1, w2, w3 = 0.5 * math.sqrt(2.0 / s), -0.5 / s, math.sqrt(s / (2.0 * math.pi))
        erf = np.vectorize(math.erf)
        ker = 0.5 * (erf(w1 * (x + 1)) * (x + 1) + erf(w1 * (x - 1)) * (x - 1)
                     - 2.0 * erf(w1 * x) * x) \
            + w3 * (np.exp(w2 * (x + 1) ** 2) + np.exp(w2 * (x - 1) ** 2)
                    - 2.0 * np.exp(w2 * x ** 2))
        ker = np.maximum(ker, 0.0)
    else:
        raise ValueError('Undefined slice profile')
    return ker / ker.sum()


def factorise(smo_ker, rtol=1e-5):
    """Separable factors of a dense (…,kx,ky,kz) kernel (the reference stores
    smooth(..., sep=False), an outer product).  Raises if it is not rank one."""
    k = np.asarray(smo_ker, dtype=np.float64).reshape(smo_ker.shape[-3:])
    tot = k.sum()
    if tot == 0:
        raise ValueError('smo_ker sums to zero')
    fx = k.sum(axis=(1, 2)) / tot
    fy = k.sum(axis=(0, 2)) / tot
    fz = k.sum(axis=(0, 1)) / tot * tot
    rec = fx[:, None, None] * fy[None, :, None] * fz[None, None, :]
    if np.abs(rec - k).max() > rtol * np.abs(k).max():
        raise NotImplementedError('smo_ker is not separable (rank one)')
    return [fx.astype(np.float32), fy.astype(np.float32), fz.astype(np.float32)]
