"""CPU oracle: Pillow's Image.resize restated (ImageNet-S resize operators).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Reference: RobustART/noise/utils/imagenet_s_gen.py:19-34
(pil_resize_mode_dict), :127-146 ('val': resize to 8/7 * 224 = 256 on both sides, centre crop 224), :151-166
(PIL_resize).  The arithmetic is Pillow's libImaging/Resample.c (8 bits per channel: 22-bit fixed-point
coefficients, horizontal pass to a uint8 intermediate, then vertical) and Geometry.c (NEAREST = 16.16 fixed-point
affine).  Pinned bit-exact against the Pillow wheel in this image (tests/test_oracle_golden.py)."""
import math

import numpy as np

FILTERS = ('nearest', 'bilinear', 'bicubic', 'box', 'hamming', 'lanczos')
SUPPORT = {'box': 0.5, 'bilinear': 1.0, 'hamming': 1.0, 'bicubic': 2.0, 'lanczos': 3.0}


def _sinc(x):
    if x == 0.0:
        return 1.0
    x = x * math.pi
    return math.sin(x) / x


def filter_value(name, x):
    if name == 'box':
        return 1.0 if -0.5 < x <= 0.5 else 0.0
    if name == 'bilinear':
        x = abs(x)
        return 1.0 - x if x < 1.0 else 0.0
    if name == 'hamming':
        x = abs(x)
        if x == 0.0:
            return 1.0
        if x >= 1.0:
            return 0.0
        x = x * math.pi
        return math.sin(x) / x * (0.54 + 0.46 * math.cos(x))
    if name == 'bicubic':
        a = -0.5
        x = abs(x)
        if x < 1.0:
            return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
        if x < 2.0:
            return (((x - 5) * x + 8) * x - 4) * a
        return 0.0
    if name == 'lanczos':
        if -3.0 <= x < 3.0:
            return _sinc(x) * _sinc(x / 3)
        return 0.0
    raise KeyError(name)


def precompute_coeffs(in_size, out_size, name):
    """Resample.c precompute_coeffs + normalize_coeffs_8bpc -> (bounds [(xmin, n)], integer coefficient rows)."""
    scale = in_size / out_size
    fs = max(scale, 1.0)
    support = SUPPORT[name] * fs
    ss = 1.0 / fs
    bounds, coeffs = [], []
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = max(int(center - support + 0.5), 0)
        xmax = min(int(center + support + 0.5), in_size) - xmin
        k = [filter_value(name, (x + xmin - center + 0.5) * ss) for x in range(xmax)]
        ww = 0.0
        for v in k:
            ww += v
        if ww != 0.0:
            k = [v / ww for v in k]
        ki = [int(-0.5 + v * (1 << 22)) if v < 0 else int(0.5 + v * (1 << 22)) for v in k]
        bounds.append((xmin, xmax))
        coeffs.append(np.array(ki, dtype=np.int64))
    return bounds, coeffs


def _clip8(v):
    return np.clip(v >> 22, 0, 255)


def pil_resize_u8(img, out_h, out_w, name):
    """Image.fromarray(img).resize((out_w, out_h), FILTER) on HxWx3 uint8."""
    img = np.asarray(img, dtype=np.uint8)
    h, w, c = img.shape
    if name == 'nearest':
        # Geometry.c affine_fixed: 16.16 fixed point, FIX(v) = floor(v*65536 + 0.5)
        def fix(v):
            return int(math.floor(v * 65536.0 + 0.5))
        a0, a4 = w / out_w, h / out_h
        xs = (fix(a0 * 0.5) + np.arange(out_w, dtype=np.int64) * fix(a0)) >> 16
        ys = (fix(a4 * 0.5) + np.arange(out_h, dtype=np.int64) * fix(a4)) >> 16
        return img[np.clip(ys, 0, h - 1)][:, np.clip(xs, 0, w - 1)]
    tmp = img
    if out_w != w:
        bounds, coeffs = precompute_coeffs(w, out_w, name)
        tmp = np.empty((h, out_w, c), dtype=np.uint8)
        for xx in range(out_w):
            xmin, n = bounds[xx]
            acc = (img[:, xmin:xmin + n, :].astype(np.int64) * coeffs[xx][None, :, None]).sum(1) + (1 << 21)
            tmp[:, xx, :] = _clip8(acc)
    out = tmp
    if out_h != h:
        bounds, coeffs = precompute_coeffs(h, out_h, name)
        out = np.empty((out_h, tmp.shape[1], c), dtype=np.uint8)
        for yy in range(out_h):
            ymin, n = bounds[yy]
            acc = (tmp[ymin:ymin + n, :, :].astype(np.int64) * coeffs[yy][:, None, None]).sum(0) + (1 << 21)
            out[yy] = _clip8(acc)
    return out


def imagenet_s_val(img, resize_type, size=224):
    """imagenet_s_gen.py:127-137 with a 'pil-*' resize_type: resize both sides to size*8/7, centre crop."""
    name = {'pil-bilinear': 'bilinear', 'pil-nearest': 'nearest', 'pil-box': 'box', 'pil-hamming': 'hamming',
            'pil-cubic': 'bicubic', 'pil-lanczos': 'lanczos'}[resize_type]
    first = int(size * 8 / 7)
    r = pil_resize_u8(img, first, first, name)
    i = int(round((first - size) / 2.))
    return r[i:i + size, i:i + size]
