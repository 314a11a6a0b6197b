"""A documented stand-in for the three scikit-image 0.17 entry points that five of the reference's corruptions call
(RobustART/noise/utils/imagenet_c/corruptions.py: impulse_noise :136-140, gaussian_blur :162-166, glass_blur :169-184,
spatter's mud branch :329-343, brightness :353-361, saturate :364-372).  scikit-image is not installed in this image, so those
corruptions cannot run unmodified; with ONLY these functions supplied, the reference's own code around them -- the pixel-swap
loop order and randint bounds of glass_blur, the thresholds and blends of spatter, the clipping and scaling everywhere -- runs
as written and its outputs become golden vectors "pinned modulo shim" (tests/golden/make_golden_shim.py).

This is NOT scikit-image and pins nothing about scikit-image itself: each function restates the published 0.17 behaviour,
    filters.gaussian      -> scipy.ndimage.gaussian_filter(image, sigma per axis (0 on the channel axis when multichannel),
                             mode='nearest', cval=0, truncate=4.0), float32 / float64 inputs keep their dtype (img_as_float),
                             which is literally what skimage/filters/_gaussian.py does;
    util.random_noise     -> mode 's&p' only: two np.random.choice([True, False], size, p=[p, 1-p]) draws (flipped, then salted
                             with p = 0.5); out[flipped & salted] = 1, out[flipped & ~salted] = 0 (unsigned input);
    color.rgb2hsv/hsv2rgb -> the array formulas of skimage/color/colorconv.py.
Used only by the golden generator in the build container; nothing on the GPU box imports it.
"""
import types

import numpy as np
from scipy import ndimage as ndi


def _as_float(image):
    image = np.asarray(image)
    if image.dtype in (np.float32, np.float64):
        return image
    if image.dtype == np.uint8:
        return image.astype(np.float64) / 255.0
    return image.astype(np.float64)


def gaussian(image, sigma=1, output=None, mode='nearest', cval=0, multichannel=None, preserve_range=False, truncate=4.0):
    image = _as_float(image)
    if multichannel:
        sigma = [sigma] * (image.ndim - 1) + [0]
    if output is None:
        output = np.empty_like(image)
    ndi.gaussian_filter(image, sigma, output=output, mode=mode, cval=cval, truncate=truncate)
    return output


def random_noise(image, mode='gaussian', seed=None, clip=True, **kwargs):
    if mode != 's&p':
        raise NotImplementedError('skimage_shim.random_noise: only the mode the reference uses (s&p)')
    image = _as_float(image)
    low_clip = -1.0 if image.min() < 0 else 0.0
    out = image.copy()
    p = kwargs.get('amount', 0.05)
    q = kwargs.get('salt_vs_pepper', 0.5)
    flipped = np.random.choice([True, False], size=image.shape, p=[p, 1 - p])
    salted = np.random.choice([True, False], size=image.shape, p=[q, 1 - q])
    peppered = ~salted
    out[flipped & salted] = 1
    out[flipped & peppered] = low_clip
    return out


def rgb2hsv(rgb):
    arr = _as_float(rgb)
    out = np.empty_like(arr)
    out_v = arr.max(-1)
    delta = np.ptp(arr, -1)
    old = np.seterr(invalid='ignore', divide='ignore')
    out_s = delta / out_v
    out_s[delta == 0.] = 0.
    idx = (arr[:, :, 0] == out_v)                       # red is max
    out[idx, 0] = (arr[idx, 1] - arr[idx, 2]) / delta[idx]
    idx = (arr[:, :, 1] == out_v)                       # green is max
    out[idx, 0] = 2. + (arr[idx, 2] - arr[idx, 0]) / delta[idx]
    idx = (arr[:, :, 2] == out_v)                       # blue is max
    out[idx, 0] = 4. + (arr[idx, 0] - arr[idx, 1]) / delta[idx]
    out_h = (out[:, :, 0] / 6.) % 1.
    out_h[delta == 0.] = 0.
    np.seterr(**old)
    out[:, :, 0] = out_h
    out[:, :, 1] = out_s
    out[:, :, 2] = out_v
    out[np.isnan(out)] = 0
    return out


def hsv2rgb(hsv):
    arr = _as_float(hsv)
    hi = np.floor(arr[:, :, 0] * 6)
    f = arr[:, :, 0] * 6 - hi
    p = arr[:, :, 2] * (1 - arr[:, :, 1])
    q = arr[:, :, 2] * (1 - f * arr[:, :, 1])
    t = arr[:, :, 2] * (1 - (1 - f) * arr[:, :, 1])
    v = arr[:, :, 2]
    hi = np.dstack([hi, hi, hi]).astype(np.uint8) % 6
    return np.choose(hi, [np.dstack((v, t, p)), np.dstack((q, v, p)), np.dstack((p, v, t)), np.dstack((p, q, v)),
                          np.dstack((t, p, v)), np.dstack((v, p, q))])


def install(sys_modules):
    """Register `skimage`, `skimage.filters`, `skimage.util`, `skimage.color` stand-ins (before the reference is imported)."""
    sk = types.ModuleType('skimage')
    sk.__doc__ = __doc__
    filters, util, color = types.ModuleType('skimage.filters'), types.ModuleType('skimage.util'), types.ModuleType('skimage.color')
    filters.gaussian = gaussian
    util.random_noise = random_noise
    color.rgb2hsv, color.hsv2rgb = rgb2hsv, hsv2rgb
    sk.filters, sk.util, sk.color = filters, util, color
    sys_modules.update({'skimage': sk, 'skimage.filters': filters, 'skimage.util': util, 'skimage.color': color})
    return sk
