"""CPU oracle: numpy restatement of RobustART's ImageNet-C corruptions.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Each function cites the
reference lines it follows (paths relative to /root/reference/RobustART/noise/utils/).

Shape of the API
----------------
The reference pulls randomness from the global ``np.random`` stream inside each
corruption.  A parallel machine cannot reproduce that stream, so the oracle (and
the HIP kernels) split every corruption into

    draws = draw(name, x_u8_hwc, severity, rng)     # what np.random returned, in reference order
    y_u8  = corrupt(name, x_u8_hwc, severity, draws) # deterministic part, given the draws

``draw`` with ``rng = np.random.RandomState(k)`` consumes the stream exactly as the
reference does after ``np.random.seed(k)``, which is how tests/test_oracle_golden.py
pins the oracle bit-for-bit against the unmodified reference.  The HIP kernels take
the same ``draws`` through the C-ABI's ``injected`` pointer for the parity tests.

``corrupt`` returns the array after the reference's ``np.uint8(.)`` truncation
(imagenet_c/__init__.py:35).
"""
import math

import numpy as np
from scipy import ndimage as ndi

# imagenet_c/__init__.py:5-8 -- order is part of the API (corruption_number indexes it)
CORRUPTION_NAMES = (
    'gaussian_noise', 'shot_noise', 'impulse_noise', 'defocus_blur',
    'glass_blur', 'motion_blur', 'zoom_blur', 'snow', 'frost', 'fog',
    'brightness', 'contrast', 'elastic_transform', 'pixelate', 'jpeg_compression',
    'speckle_noise', 'gaussian_blur', 'spatter', 'saturate')

# corruptions.py per-severity tables
PARAMS = {
    'gaussian_noise': [.08, .12, 0.18, 0.26, 0.38],                       # :123
    'shot_noise': [60, 25, 12, 5, 3],                                     # :130
    'impulse_noise': [.03, .06, .09, 0.17, 0.27],                         # :137
    'speckle_noise': [.15, .2, 0.35, 0.45, 0.6],                          # :144
    'gaussian_blur': [1, 2, 3, 4, 6],                                     # :163
    'glass_blur': [(0.7, 1, 2), (0.9, 2, 1), (1, 2, 3), (1.1, 3, 2), (1.5, 4, 2)],   # :171
    'defocus_blur': [(3, 0.1), (4, 0.5), (6, 0.5), (8, 0.5), (10, 0.5)],  # :188
    'motion_blur': [(10, 3), (15, 5), (15, 8), (15, 12), (20, 15)],       # :202
    'zoom_blur': [np.arange(1, 1.11, 0.01), np.arange(1, 1.16, 0.01), np.arange(1, 1.21, 0.02),
                  np.arange(1, 1.26, 0.02), np.arange(1, 1.31, 0.03)],    # :220-224
    'fog': [(1.5, 2), (2., 2), (2.5, 1.7), (2.5, 1.5), (3., 1.4)],        # :236
    'frost': [(1, 0.4), (0.8, 0.6), (0.7, 0.7), (0.65, 0.7), (0.6, 0.75)],  # :245-249
    'snow': [(0.1, 0.3, 3, 0.5, 10, 4, 0.8), (0.2, 0.3, 2, 0.5, 12, 4, 0.7),
             (0.55, 0.3, 4, 0.9, 12, 8, 0.7), (0.55, 0.3, 4.5, 0.85, 12, 8, 0.65),
             (0.55, 0.3, 2.5, 0.85, 12, 12, 0.55)],                       # :266-270
    'spatter': [(0.65, 0.3, 4, 0.69, 0.6, 0), (0.65, 0.3, 3, 0.68, 0.6, 0), (0.65, 0.3, 2, 0.68, 0.5, 0),
                (0.65, 0.3, 1, 0.65, 1.5, 1), (0.67, 0.4, 1, 0.65, 1.5, 1)],  # :294-298
    'contrast': [0.4, .3, .2, .1, .05],                                   # :346
    'brightness': [.1, .2, .3, .4, .5],                                   # :354
    'saturate': [(0.3, 0), (0.1, 0), (2, 0), (5, 0.1), (20, 0.2)],        # :365
    'jpeg_compression': [25, 18, 15, 10, 7],                              # :376
    'pixelate': [0.6, 0.5, 0.4, 0.3, 0.25],                               # :386
    'elastic_transform': [(244 * 2, 244 * 0.7, 244 * 0.1), (244 * 2, 244 * 0.08, 244 * 0.2),
                          (244 * 0.05, 244 * 0.01, 244 * 0.02), (244 * 0.07, 244 * 0.01, 244 * 0.02),
                          (244 * 0.12, 244 * 0.01, 244 * 0.02)],          # :396-400 (244 sic)
}


def _u8(y):
    """imagenet_c/__init__.py:35 -- np.uint8() on the float result = truncation toward zero."""
    return np.uint8(y)


# --------------------------------------------------------------------------------------
# third-party stand-ins (semantics: SURVEY.md Appendix A/B)
# --------------------------------------------------------------------------------------

def sk_gaussian(image, sigma, multichannel=False, mode='nearest', truncate=4.0):
    """skimage 0.17.2 filters.gaussian == scipy.ndimage.gaussian_filter (skimage calls it).
    multichannel=True => sigma 0 on the channel axis.  float32 in -> float32 out."""
    image = np.asarray(image)
    if not np.issubdtype(image.dtype, np.floating):
        image = image.astype(np.float64)
    sig = (sigma, sigma, 0) if multichannel else sigma
    return ndi.gaussian_filter(image, sig, mode=mode, truncate=truncate)


def gaussian_kernel1d(sigma, radius):
    """scipy.ndimage._filters._gaussian_kernel1d (order 0)."""
    x = np.arange(-radius, radius + 1)
    phi = np.exp(-0.5 / (sigma * sigma) * x ** 2)
    return phi / phi.sum()


def _reflect101(i, n):
    """OpenCV BORDER_REFLECT_101 index map (gfedcb|abcdefgh|gfedcba)."""
    i = np.asarray(i)
    if n == 1:
        return np.zeros_like(i)
    period = 2 * (n - 1)
    i = np.mod(i, period)
    return np.where(i >= n, period - i, i)


def cv_gaussian_kernel1d_f32(ksize, sigma):
    """cv2.getGaussianKernel(ksize, sigma, CV_32F) for sigma > 0."""
    x = np.arange(ksize, dtype=np.float64) - (ksize - 1) * 0.5
    k = np.exp(-(x * x) / (2.0 * sigma * sigma))
    k /= k.sum()
    return k.astype(np.float32)


def cv_gaussian_blur_f32(img, ksize, sigma):
    """cv2.GaussianBlur(float32 2-D, (ksize,ksize), sigmaX=sigma), BORDER_REFLECT_101,
    separable row-then-column passes with float32 intermediates."""
    img = np.asarray(img, dtype=np.float32)
    k = cv_gaussian_kernel1d_f32(ksize, sigma)
    r = ksize // 2
    h, w = img.shape
    cols = _reflect101(np.arange(-r, w + r), w)
    tmp = np.zeros_like(img)
    pad = img[:, cols]
    for t in range(ksize):
        tmp += pad[:, t:t + w] * k[t]
    rows = _reflect101(np.arange(-r, h + r), h)
    pad = tmp[rows, :]
    out = np.zeros_like(img)
    for t in range(ksize):
        out += pad[t:t + h, :] * k[t]
    return out


def disk_kernel(radius, alias_blur):
    """imagenet_c/corruptions.py:26-38 (disk)."""
    if radius <= 8:
        L = np.arange(-8, 8 + 1)
        ksize = 3
    else:
        L = np.arange(-radius, radius + 1)
        ksize = 5
    X, Y = np.meshgrid(L, L)
    aliased = np.array((X ** 2 + Y ** 2) <= radius ** 2, dtype=np.float32)
    aliased /= np.sum(aliased)
    return cv_gaussian_blur_f32(aliased, ksize, alias_blur)


def cv_filter2d_reflect101(plane, kernel):
    """cv2.filter2D(plane, -1, kernel): correlation, centre anchor, BORDER_REFLECT_101, fp64."""
    kh, kw = kernel.shape
    ry, rx = kh // 2, kw // 2
    h, w = plane.shape
    rows = _reflect101(np.arange(-ry, h + ry), h)
    cols = _reflect101(np.arange(-rx, w + rx), w)
    pad = plane[rows][:, cols]
    out = np.zeros((h, w), dtype=np.float64)
    k = kernel.astype(np.float64)
    for a in range(kh):
        for b in range(kw):
            out += pad[a:a + h, b:b + w] * k[a, b]
    return out


def im_motion_blur_u8(img_u8, radius, sigma, angle_deg):
    """ImageMagick MotionBlurImage(radius, sigma, angle) on an 8-bit image (HxW or HxWxC),
    edge virtual pixels, re-quantised to 8 bit (corruptions.py:42-51 MotionImage).
    Kernel: width = 2*ceil(radius)+1 one-sided gaussian taps i = 0..width-1, normalised;
    tap i samples (x + ceil(i*cos(a) - 0.5), y + ceil(i*sin(a) - 0.5))."""
    img = np.asarray(img_u8)
    squeeze = img.ndim == 2
    if squeeze:
        img = img[..., None]
    h, w, _ = img.shape
    width = int(2.0 * math.ceil(radius) + 1.0)
    i = np.arange(width, dtype=np.float64)
    k = np.exp(-(i * i) / (2.0 * sigma * sigma)) / (math.sqrt(2.0 * math.pi) * sigma)
    k = k / k.sum()
    a = math.radians(angle_deg)
    px, py = width * math.sin(a), width * math.cos(a)
    hyp = math.hypot(px, py)
    offx = np.ceil(i * py / hyp - 0.5).astype(np.int64)
    offy = np.ceil(i * px / hyp - 0.5).astype(np.int64)
    ys = np.arange(h)[:, None]
    xs = np.arange(w)[None, :]
    acc = np.zeros(img.shape, dtype=np.float64)
    for t in range(width):
        yy = np.clip(ys + offy[t], 0, h - 1)
        xx = np.clip(xs + offx[t], 0, w - 1)
        acc += k[t] * img[yy, xx].astype(np.float64)
    out = np.clip(np.floor(acc + 0.5), 0, 255).astype(np.uint8)
    return out[..., 0] if squeeze else out


def clipped_zoom(img, zoom_factor):
    """corruptions.py:104-114; scipy.ndimage.zoom(order=1) restated per SURVEY Appendix A.3:
    output o samples source (o + trim)*(ch-1)/(out-1), separable linear interpolation.
    Result dtype follows the input (scipy computes in double and casts)."""
    img = np.asarray(img)
    h = img.shape[0]
    ch = int(np.ceil(h / float(zoom_factor)))
    top = (h - ch) // 2
    out_n = int(round(ch * zoom_factor))
    trim = (out_n - h) // 2
    crop = img[top:top + ch, top:top + ch].astype(np.float64)
    o = np.arange(h, dtype=np.float64) + trim
    s = o * (ch - 1) / (out_n - 1) if out_n > 1 else np.zeros_like(o)
    i0 = np.floor(s).astype(np.int64)
    i0 = np.clip(i0, 0, ch - 1)
    i1 = np.minimum(i0 + 1, ch - 1)
    t = s - i0
    # rows then columns (weights multiply; order-1 spline = bilinear)
    r = crop[i0] * (1.0 - t)[:, None, None] + crop[i1] * t[:, None, None]
    c = r[:, i0] * (1.0 - t)[None, :, None] + r[:, i1] * t[None, :, None]
    return c.astype(img.dtype)


def plasma_fractal(uniform_draws, mapsize=256, wibbledecay=3):
    """corruptions.py:55-101 diamond-square, with the np.random.uniform(-wibble, wibble, shape)
    results supplied as a list in call order."""
    maparray = np.empty((mapsize, mapsize), dtype=np.float64)
    maparray[0, 0] = 0
    stepsize = mapsize
    wibble = 100
    it = iter(uniform_draws)

    def wibbledmean(array):
        return array / 4 + wibble * next(it)

    while stepsize >= 2:
        # fillsquares (:70-77)
        cornerref = maparray[0:mapsize:stepsize, 0:mapsize:stepsize]
        squareaccum = cornerref + np.roll(cornerref, shift=-1, axis=0)
        squareaccum += np.roll(squareaccum, shift=-1, axis=1)
        maparray[stepsize // 2:mapsize:stepsize, stepsize // 2:mapsize:stepsize] = wibbledmean(squareaccum)
        # filldiamonds (:79-92)
        drgrid = maparray[stepsize // 2:mapsize:stepsize, stepsize // 2:mapsize:stepsize]
        ulgrid = maparray[0:mapsize:stepsize, 0:mapsize:stepsize]
        ldrsum = drgrid + np.roll(drgrid, 1, axis=0)
        lulsum = ulgrid + np.roll(ulgrid, -1, axis=1)
        ltsum = ldrsum + lulsum
        maparray[0:mapsize:stepsize, stepsize // 2:mapsize:stepsize] = wibbledmean(ltsum)
        tdrsum = drgrid + np.roll(drgrid, 1, axis=1)
        tulsum = ulgrid + np.roll(ulgrid, -1, axis=0)
        ttsum = tdrsum + tulsum
        maparray[stepsize // 2:mapsize:stepsize, 0:mapsize:stepsize] = wibbledmean(ttsum)
        stepsize //= 2
        wibble /= wibbledecay

    maparray -= maparray.min()
    return maparray / maparray.max()


def plasma_draw_shapes(mapsize=256):
    """Shapes of the successive np.random.uniform calls inside plasma_fractal."""
    shapes = []
    stepsize = mapsize
    while stepsize >= 2:
        n = mapsize // stepsize
        shapes += [(n, n), (n, n), (n, n)]
        stepsize //= 2
    return shapes


def plasma_draw_wibbles(wibbledecay, mapsize=256):
    out = []
    stepsize = mapsize
    wibble = 100
    while stepsize >= 2:
        out += [wibble, wibble, wibble]
        stepsize //= 2
        wibble /= wibbledecay
    return out


def rgb2hsv(arr):
    """skimage 0.17.2 color.rgb2hsv, fp64."""
    arr = np.asarray(arr, dtype=np.float64)
    out = np.empty_like(arr)
    out_v = arr.max(-1)
    delta = np.ptp(arr, -1)
    with np.errstate(invalid='ignore', divide='ignore'):
        out_s = delta / out_v
        out_s[delta == 0.] = 0.
        idx = (arr[:, :, 0] == out_v)
        out[idx, 0] = (arr[idx, 1] - arr[idx, 2]) / delta[idx]
        idx = (arr[:, :, 1] == out_v)
        out[idx, 0] = 2. + (arr[idx, 2] - arr[idx, 0]) / delta[idx]
        idx = (arr[:, :, 2] == out_v)
        out[idx, 0] = 4. + (arr[idx, 0] - arr[idx, 1]) / delta[idx]
        out_h = (out[:, :, 0] / 6.) % 1.
    out_h[delta == 0.] = 0.
    out[:, :, 0] = out_h
    out[:, :, 1] = out_s
    out[:, :, 2] = out_v
    out[np.isnan(out)] = 0
    return out


def hsv2rgb(arr):
    """skimage 0.17.2 color.hsv2rgb, fp64."""
    arr = np.asarray(arr, dtype=np.float64)
    hi = np.floor(arr[:, :, 0] * 6)
    f = arr[:, :, 0] * 6 - hi
    p = arr[:, :, 2] * (1 - arr[:, :, 1])
    q = arr[:, :, 2] * (1 - f * arr[:, :, 1])
    t = arr[:, :, 2] * (1 - (1 - f) * arr[:, :, 1])
    v = arr[:, :, 2]
    hi = np.dstack([hi, hi, hi]).astype(np.uint8) % 6
    return np.choose(hi, [np.dstack((v, t, p)), np.dstack((q, v, p)), np.dstack((p, v, t)),
                          np.dstack((p, q, v)), np.dstack((t, p, v)), np.dstack((v, p, q))])


# ---- Pillow BOX resize (pinned bit-exact vs Pillow in tests) ----

def _box_coeffs(in_size, out_size):
    scale = in_size / out_size
    fs = max(scale, 1.0)
    support = 0.5 * fs
    bounds, coeffs = [], []
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = max(int(center - support + 0.5), 0)
        xmax = min(int(center + support + 0.5), in_size) - xmin
        w = np.zeros(xmax, dtype=np.float64)
        for x in range(xmax):
            t = (x + xmin - center + 0.5) / fs
            w[x] = 1.0 if (-0.5 < t <= 0.5) else 0.0
        ww = w.sum()
        if ww != 0.0:
            w = w / ww
        k = np.array([int(0.5 + v * (1 << 22)) if v >= 0 else int(-0.5 + v * (1 << 22)) for v in w],
                     dtype=np.int64)
        bounds.append((xmin, xmax))
        coeffs.append(k)
    return bounds, coeffs


def pil_box_resize_u8(img, out_h, out_w):
    """Pillow Image.resize((out_w,out_h), BOX) on an HxWx3 uint8 image: horizontal pass to a
    uint8 intermediate, then vertical; 22-bit fixed-point coefficients."""
    img = np.asarray(img, dtype=np.uint8)
    h, w, c = img.shape
    if out_w != w:
        bounds, coeffs = _box_coeffs(w, out_w)
        tmp = np.empty((h, out_w, c), dtype=np.uint8)
        for xx in range(out_w):
            xmin, n = bounds[xx]
            acc = (img[:, xmin:xmin + n, :].astype(np.int64) * coeffs[xx][None, :, None]).sum(1)
            tmp[:, xx, :] = np.clip((acc + (1 << 21)) >> 22, 0, 255)
    else:
        tmp = img
    if out_h != h:
        bounds, coeffs = _box_coeffs(h, out_h)
        out = np.empty((out_h, tmp.shape[1], c), dtype=np.uint8)
        for yy in range(out_h):
            ymin, n = bounds[yy]
            acc = (tmp[ymin:ymin + n, :, :].astype(np.int64) * coeffs[yy][:, None, None]).sum(0)
            out[yy] = np.clip((acc + (1 << 21)) >> 22, 0, 255)
    else:
        out = tmp
    return out


# ---- libjpeg(-turbo) baseline 4:2:0 ISLOW round trip ----

JPEG_STD_LUMA = np.array([
    16, 11, 10, 16, 24, 40, 51, 61, 12, 12, 14, 19, 26, 58, 60, 55,
    14, 13, 16, 24, 40, 57, 69, 56, 14, 17, 22, 29, 51, 87, 80, 62,
    18, 22, 37, 56, 68, 109, 103, 77, 24, 35, 55, 64, 81, 104, 113, 92,
    49, 64, 78, 87, 103, 121, 120, 101, 72, 92, 95, 98, 112, 100, 103, 99], dtype=np.int64).reshape(8, 8)
JPEG_STD_CHROMA = np.array([
    17, 18, 24, 47, 99, 99, 99, 99, 18, 21, 26, 66, 99, 99, 99, 99,
    24, 26, 56, 99, 99, 99, 99, 99, 47, 66, 99, 99, 99, 99, 99, 99,
    99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99,
    99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99], dtype=np.int64).reshape(8, 8)


def jpeg_quant_tables(quality):
    quality = max(1, min(100, int(quality)))
    scale = 5000 // quality if quality < 50 else 200 - quality * 2
    ty = np.clip((JPEG_STD_LUMA * scale + 50) // 100, 1, 255)
    tc = np.clip((JPEG_STD_CHROMA * scale + 50) // 100, 1, 255)
    return ty, tc


def _FIX(x):
    return int(x * 65536 + 0.5)


def _ds(x, n):
    return (x + (1 << (n - 1))) >> n


_F = dict(f0_298=2446, f0_390=3196, f0_541=4433, f0_765=6270, f0_899=7373, f1_175=9633,
          f1_501=12299, f1_847=15137, f1_961=16069, f2_053=16819, f2_562=20995, f3_072=25172)


def _fdct_1d(d, pass1):
    """jfdctint.c one pass over the last axis of d (...,8) int64."""
    d0, d1, d2, d3, d4, d5, d6, d7 = [d[..., i] for i in range(8)]
    t0, t7 = d0 + d7, d0 - d7
    t1, t6 = d1 + d6, d1 - d6
    t2, t5 = d2 + d5, d2 - d5
    t3, t4 = d3 + d4, d3 - d4
    t10, t13 = t0 + t3, t0 - t3
    t11, t12 = t1 + t2, t1 - t2
    sh = 11 if pass1 else 15
    if pass1:
        o0 = (t10 + t11) << 2
        o4 = (t10 - t11) << 2
    else:
        o0 = _ds(t10 + t11, 2)
        o4 = _ds(t10 - t11, 2)
    z1 = (t12 + t13) * _F['f0_541']
    o2 = _ds(z1 + t13 * _F['f0_765'], sh)
    o6 = _ds(z1 - t12 * _F['f1_847'], sh)
    z1 = t4 + t7
    z2 = t5 + t6
    z3 = t4 + t6
    z4 = t5 + t7
    z5 = (z3 + z4) * _F['f1_175']
    t4 = t4 * _F['f0_298']
    t5 = t5 * _F['f2_053']
    t6 = t6 * _F['f3_072']
    t7 = t7 * _F['f1_501']
    z1 = z1 * -_F['f0_899']
    z2 = z2 * -_F['f2_562']
    z3 = z3 * -_F['f1_961'] + z5
    z4 = z4 * -_F['f0_390'] + z5
    o7 = _ds(t4 + z1 + z3, sh)
    o5 = _ds(t5 + z2 + z4, sh)
    o3 = _ds(t6 + z2 + z3, sh)
    o1 = _ds(t7 + z1 + z4, sh)
    return np.stack([o0, o1, o2, o3, o4, o5, o6, o7], axis=-1)


def _idct_1d(x, sh):
    """jidctint.c one pass over the last axis of x (...,8) int64, descale by sh."""
    x0, x1, x2, x3, x4, x5, x6, x7 = [x[..., i] for i in range(8)]
    z2, z3 = x2, x6
    z1 = (z2 + z3) * _F['f0_541']
    t2 = z1 - z3 * _F['f1_847']
    t3 = z1 + z2 * _F['f0_765']
    t0 = (x0 + x4) << 13
    t1 = (x0 - x4) << 13
    t10, t13 = t0 + t3, t0 - t3
    t11, t12 = t1 + t2, t1 - t2
    a0, a1, a2, a3 = x7, x5, x3, x1
    z1 = a0 + a3
    z2 = a1 + a2
    z3 = a0 + a2
    z4 = a1 + a3
    z5 = (z3 + z4) * _F['f1_175']
    a0 = a0 * _F['f0_298']
    a1 = a1 * _F['f2_053']
    a2 = a2 * _F['f3_072']
    a3 = a3 * _F['f1_501']
    z1 = z1 * -_F['f0_899']
    z2 = z2 * -_F['f2_562']
    z3 = z3 * -_F['f1_961'] + z5
    z4 = z4 * -_F['f0_390'] + z5
    a0 = a0 + z1 + z3
    a1 = a1 + z2 + z4
    a2 = a2 + z2 + z3
    a3 = a3 + z1 + z4
    o = [_ds(t10 + a3, sh), _ds(t11 + a2, sh), _ds(t12 + a1, sh), _ds(t13 + a0, sh),
         _ds(t13 - a0, sh), _ds(t12 - a1, sh), _ds(t11 - a2, sh), _ds(t10 - a3, sh)]
    return np.stack(o, axis=-1)


def _blocks(plane):
    h, w = plane.shape
    return plane.reshape(h // 8, 8, w // 8, 8).transpose(0, 2, 1, 3)


def _unblocks(b):
    nh, nw = b.shape[:2]
    return b.transpose(0, 2, 1, 3).reshape(nh * 8, nw * 8)


def _codec_plane(plane_u8like, qt):
    """FDCT -> quantise -> dequantise -> IDCT of one plane (int64 samples 0..255)."""
    b = _blocks(plane_u8like.astype(np.int64) - 128)
    # pass 1 rows, pass 2 columns
    b = _fdct_1d(b, True)
    b = _fdct_1d(b.transpose(0, 1, 3, 2), False).transpose(0, 1, 3, 2)
    d = qt << 3
    v = (np.abs(b) + (d >> 1)) // d
    v = np.where(b < 0, -v, v)
    c = v * qt
    # IDCT pass 1 columns (descale 11), pass 2 rows (descale 18)
    c = _idct_1d(c.transpose(0, 1, 3, 2), 11).transpose(0, 1, 3, 2)
    c = _idct_1d(c, 18)
    return np.clip(_unblocks(c) + 128, 0, 255)


def jpeg_roundtrip_u8(img, quality):
    """Pillow save(JPEG, quality) -> open, restated (baseline, 4:2:0, ISLOW, fancy upsampling).
    HxWx3 uint8, H and W multiples of 16."""
    img = np.asarray(img, dtype=np.uint8)
    h, w, _ = img.shape
    assert h % 16 == 0 and w % 16 == 0, "restatement covers MCU-aligned sizes (224 = 14*16)"
    r, g, b = [img[..., i].astype(np.int64) for i in range(3)]
    H = 32768
    y = (_FIX(0.29900) * r + _FIX(0.58700) * g + _FIX(0.11400) * b + H) >> 16
    cb = (-_FIX(0.16874) * r - _FIX(0.33126) * g + _FIX(0.50000) * b + (128 << 16) + H - 1) >> 16
    cr = (_FIX(0.50000) * r - _FIX(0.41869) * g - _FIX(0.08131) * b + (128 << 16) + H - 1) >> 16

    def down(p):
        s = p[0::2, 0::2] + p[0::2, 1::2] + p[1::2, 0::2] + p[1::2, 1::2]
        bias = np.where(np.arange(s.shape[1]) % 2 == 0, 1, 2)[None, :]
        return (s + bias) >> 2

    ty, tc = jpeg_quant_tables(quality)
    y2 = _codec_plane(y, ty)
    cb2 = _codec_plane(down(cb), tc)
    cr2 = _codec_plane(down(cr), tc)

    def up(p):
        ch, cw = p.shape
        above = np.vstack([p[:1], p[:-1]])
        below = np.vstack([p[1:], p[-1:]])
        rows = np.empty((ch * 2, cw), dtype=np.int64)
        rows[0::2] = 3 * p + above
        rows[1::2] = 3 * p + below
        left = np.hstack([rows[:, :1], rows[:, :-1]])
        right = np.hstack([rows[:, 1:], rows[:, -1:]])
        out = np.empty((ch * 2, cw * 2), dtype=np.int64)
        out[:, 0::2] = (3 * rows + left + 8) >> 4
        out[:, 1::2] = (3 * rows + right + 7) >> 4
        out[:, 0] = (4 * rows[:, 0] + 8) >> 4
        out[:, -1] = (4 * rows[:, -1] + 7) >> 4
        return out

    cbu = up(cb2) - 128
    cru = up(cr2) - 128
    rr = y2 + ((_FIX(1.40200) * cru + H) >> 16)
    bb = y2 + ((_FIX(1.77200) * cbu + H) >> 16)
    gg = y2 + ((-_FIX(0.34414) * cbu + H - _FIX(0.71414) * cru) >> 16)
    return np.clip(np.stack([rr, gg, bb], -1), 0, 255).astype(np.uint8)


# ---- OpenCV warpAffine (INTER_LINEAR, fixed-point coordinates), elastic_transform ----

def cv_get_affine_transform(src, dst):
    """cv2.getAffineTransform: solve the 6x6 system in double."""
    a = np.zeros((6, 6), dtype=np.float64)
    b = np.zeros(6, dtype=np.float64)
    for i in range(3):
        a[i, 0:3] = [src[i][0], src[i][1], 1]
        a[i + 3, 3:6] = [src[i][0], src[i][1], 1]
        b[i] = dst[i][0]
        b[i + 3] = dst[i][1]
    return np.linalg.solve(a, b).reshape(2, 3)


def cv_warp_affine_linear_reflect101(img, M):
    """cv2.warpAffine(img float32 HxWxC, M, (W,H), INTER_LINEAR, BORDER_REFLECT_101):
    M inverted, source coordinates in 1/32-pixel fixed point (AB_BITS=10), float32 weights."""
    img = np.asarray(img, dtype=np.float32)
    h, w = img.shape[:2]
    M = np.asarray(M, dtype=np.float64)
    D = M[0, 0] * M[1, 1] - M[0, 1] * M[1, 0]
    D = 1.0 / D if D != 0 else 0.0
    A11, A22 = M[1, 1] * D, M[0, 0] * D
    A12, A21 = -M[0, 1] * D, -M[1, 0] * D
    b1 = -A11 * M[0, 2] - A12 * M[1, 2]
    b2 = -A21 * M[0, 2] - A22 * M[1, 2]
    AB = 1024
    xs = np.arange(w, dtype=np.float64)
    adelta = np.rint(A11 * xs * AB).astype(np.int64)
    bdelta = np.rint(A21 * xs * AB).astype(np.int64)
    ys = np.arange(h, dtype=np.float64)
    X0 = np.rint((A12 * ys + b1) * AB).astype(np.int64) + 16
    Y0 = np.rint((A22 * ys + b2) * AB).astype(np.int64) + 16
    X = (X0[:, None] + adelta[None, :]) >> 5
    Y = (Y0[:, None] + bdelta[None, :]) >> 5
    sx, sy = X >> 5, Y >> 5
    fx = (X & 31).astype(np.float32) / np.float32(32)
    fy = (Y & 31).astype(np.float32) / np.float32(32)
    x0, x1 = _reflect101(sx, w), _reflect101(sx + 1, w)
    y0, y1 = _reflect101(sy, h), _reflect101(sy + 1, h)
    one = np.float32(1)
    w00 = ((one - fy) * (one - fx))[..., None]
    w01 = ((one - fy) * fx)[..., None]
    w10 = (fy * (one - fx))[..., None]
    w11 = (fy * fx)[..., None]
    return img[y0, x0] * w00 + img[y0, x1] * w01 + img[y1, x0] * w10 + img[y1, x1] * w11


# --------------------------------------------------------------------------------------
# draws: consume np.random exactly as the reference does
# --------------------------------------------------------------------------------------

def draw(name, x, severity, rng):
    """Return the random draws corruption `name` makes, in reference order, from `rng`
    (a np.random.RandomState; RandomState(k) == the stream after np.random.seed(k))."""
    x = np.asarray(x)
    shape = x.shape
    s = severity - 1
    if name == 'gaussian_noise':          # corruptions.py:126
        return {'noise': rng.normal(size=shape, scale=PARAMS[name][s])}
    if name == 'speckle_noise':           # :147
        return {'noise': rng.normal(size=shape, scale=PARAMS[name][s])}
    if name == 'shot_noise':              # :133
        return {'counts': rng.poisson(x / 255. * PARAMS[name][s])}
    if name == 'impulse_noise':           # :139 -> skimage random_noise('s&p'): two np.random.choice calls
        u_flip = rng.random_sample(shape)
        u_salt = rng.random_sample(shape)
        amount = PARAMS[name][s]
        code = np.zeros(shape, dtype=np.uint8)          # 0 keep, 1 salt, 2 pepper
        flipped = u_flip < amount
        salted = u_salt < 0.5
        code[flipped & salted] = 1
        code[flipped & ~salted] = 2
        return {'code': code}
    if name == 'glass_blur':              # :176-179
        _, delta, iters = PARAMS[name][s]
        n = 224 - 2 * delta
        d = np.empty((iters, n, n, 2), dtype=np.int8)
        for i in range(iters):
            for a in range(n):
                for b in range(n):
                    d[i, a, b] = rng.randint(-delta, delta, size=(2,))   # (dx, dy)
        return {'dxdy': d}
    if name == 'motion_blur':             # :208
        return {'angle': float(rng.uniform(-45, 45))}
    if name == 'snow':                    # :273, :283
        c = PARAMS[name][s]
        layer = rng.normal(size=shape[:2], loc=c[0], scale=c[1])
        return {'layer': layer, 'angle': float(rng.uniform(-135, -45))}
    if name == 'frost':                   # :250, :259 (texture shape must be supplied by the caller)
        raise ValueError("frost draws depend on the texture: use draw_frost(rng, tex_shapes)")
    if name == 'fog':                     # :68 via plasma_fractal
        wib = plasma_draw_wibbles(PARAMS[name][s][1])
        return {'uniform': [rng.uniform(-w_, w_, shp) for w_, shp in zip(wib, plasma_draw_shapes())]}
    if name == 'elastic_transform':       # :412, :416, :418
        c = PARAMS[name][s]
        jitter = rng.uniform(-c[2], c[2], size=(3, 2)).astype(np.float32)
        fx = rng.uniform(-1, 1, size=shape[:2])
        fy = rng.uniform(-1, 1, size=shape[:2])
        return {'jitter': jitter, 'field_x': fx, 'field_y': fy}
    if name == 'spatter':                 # :301
        c = PARAMS[name][s]
        return {'layer': rng.normal(size=shape[:2], loc=c[0], scale=c[1])}
    return {}


def draw_frost(rng, tex_shapes):
    """corruptions.py:250,259: idx = randint(5) (over a 6-entry list), then the crop origin."""
    idx = int(rng.randint(5))
    th, tw = tex_shapes[idx]
    x_start, y_start = int(rng.randint(0, th - 224)), int(rng.randint(0, tw - 224))
    return {'idx': idx, 'x_start': x_start, 'y_start': y_start}


# --------------------------------------------------------------------------------------
# deterministic parts
# --------------------------------------------------------------------------------------

def gaussian_noise(x, severity, draws):
    """corruptions.py:122-126."""
    x = np.array(x) / 255.
    return np.clip(x + draws['noise'], 0, 1) * 255


def shot_noise(x, severity, draws):
    """corruptions.py:129-133."""
    c = PARAMS['shot_noise'][severity - 1]
    return np.clip(draws['counts'] / float(c), 0, 1) * 255


def impulse_noise(x, severity, draws):
    """corruptions.py:136-140 (skimage random_noise s&p, salt_vs_pepper 0.5)."""
    x = np.array(x) / 255.
    code = draws['code']
    x = np.where(code == 1, 1.0, np.where(code == 2, 0.0, x))
    return np.clip(x, 0, 1) * 255


def speckle_noise(x, severity, draws):
    """corruptions.py:143-147."""
    x = np.array(x) / 255.
    return np.clip(x + x * draws['noise'], 0, 1) * 255


def gaussian_blur(x, severity, draws=None):
    """corruptions.py:162-166."""
    c = PARAMS['gaussian_blur'][severity - 1]
    x = sk_gaussian(np.array(x) / 255., sigma=c, multichannel=True)
    return np.clip(x, 0, 1) * 255


def glass_blur(x, severity, draws):
    """corruptions.py:169-184."""
    sigma, delta, iters = PARAMS['glass_blur'][severity - 1]
    x = np.uint8(sk_gaussian(np.array(x) / 255., sigma=sigma, multichannel=True) * 255)
    d = draws['dxdy']
    for i in range(iters):
        for a, h in enumerate(range(224 - delta, delta, -1)):
            for b, w in enumerate(range(224 - delta, delta, -1)):
                dx, dy = int(d[i, a, b, 0]), int(d[i, a, b, 1])
                hp, wp = h + dy, w + dx
                # the reference writes `x[h, w], x[h_prime, w_prime] = x[h_prime, w_prime], x[h, w]` (:181-182) on a numpy
                # array: both right-hand sides are VIEWS, so after x[h, w] has received the neighbour's pixel the second
                # assignment copies that same pixel back onto the neighbour.  What executes is a COPY x[h, w] <- x[h', w'],
                # not a swap (found by the pinned-modulo-shim golden, tests/golden/make_golden_shim.py)
                x[h, w] = x[hp, wp]
    return np.clip(sk_gaussian(x / 255., sigma=sigma, multichannel=True), 0, 1) * 255


def defocus_blur(x, severity, draws=None):
    """corruptions.py:187-198."""
    radius, alias = PARAMS['defocus_blur'][severity - 1]
    x = np.array(x) / 255.
    kernel = disk_kernel(radius, alias)
    ch = [cv_filter2d_reflect101(x[:, :, d], kernel) for d in range(3)]
    return np.clip(np.array(ch).transpose((1, 2, 0)), 0, 1) * 255


def motion_blur(x, severity, draws):
    """corruptions.py:201-216."""
    radius, sigma = PARAMS['motion_blur'][severity - 1]
    return np.clip(im_motion_blur_u8(np.array(x), radius, sigma, draws['angle']), 0, 255)


def zoom_blur(x, severity, draws=None):
    """corruptions.py:219-232 (fp32 accumulate)."""
    c = PARAMS['zoom_blur'][severity - 1]
    x = (np.array(x) / 255.).astype(np.float32)
    out = np.zeros_like(x)
    for z in c:
        out += clipped_zoom(x, z)
    x = (x + out) / (len(c) + 1)
    return np.clip(x, 0, 1) * 255


def snow(x, severity, draws):
    """corruptions.py:265-290."""
    c = PARAMS['snow'][severity - 1]
    x = np.array(x, dtype=np.float32) / 255.
    layer = clipped_zoom(draws['layer'][..., np.newaxis], c[2])
    layer[layer < c[3]] = 0
    layer_u8 = (np.clip(layer.squeeze(), 0, 1) * 255).astype(np.uint8)
    layer_u8 = im_motion_blur_u8(layer_u8, c[4], c[5], draws['angle'])
    layer = (layer_u8 / 255.)[..., np.newaxis]
    gray = (np.float32(0.299) * x[..., 0] + np.float32(0.587) * x[..., 1] + np.float32(0.114) * x[..., 2])
    x = c[6] * x + (1 - c[6]) * np.maximum(x, gray.reshape(224, 224, 1) * 1.5 + 0.5)
    return np.clip(x + layer + np.rot90(layer, k=2), 0, 1) * 255


def frost(x, severity, draws):
    """corruptions.py:244-262; draws['texture'] = the RGB 224x224x3 crop the reference would have
    cut from its (absent) frost photo."""
    a, b = PARAMS['frost'][severity - 1]
    return np.clip(a * np.array(x) + b * np.asarray(draws['texture']), 0, 255)


def fog(x, severity, draws):
    """corruptions.py:235-241."""
    c = PARAMS['fog'][severity - 1]
    x = np.array(x) / 255.
    max_val = x.max()
    x = x + c[0] * plasma_fractal(draws['uniform'], wibbledecay=c[1])[:224, :224][..., np.newaxis]
    return np.clip(x * max_val / (max_val + c[0]), 0, 1) * 255


def brightness(x, severity, draws=None):
    """corruptions.py:353-361."""
    c = PARAMS['brightness'][severity - 1]
    x = rgb2hsv(np.array(x) / 255.)
    x[:, :, 2] = np.clip(x[:, :, 2] + c, 0, 1)
    return np.clip(hsv2rgb(x), 0, 1) * 255


def saturate(x, severity, draws=None):
    """corruptions.py:364-372."""
    c = PARAMS['saturate'][severity - 1]
    x = rgb2hsv(np.array(x) / 255.)
    x[:, :, 1] = np.clip(x[:, :, 1] * c[0] + c[1], 0, 1)
    return np.clip(hsv2rgb(x), 0, 1) * 255


def contrast(x, severity, draws=None):
    """corruptions.py:345-350."""
    c = PARAMS['contrast'][severity - 1]
    x = np.array(x) / 255.
    means = np.mean(x, axis=(0, 1), keepdims=True)
    return np.clip((x - means) * c + means, 0, 1) * 255


def pixelate(x, severity, draws=None):
    """corruptions.py:385-391."""
    c = PARAMS['pixelate'][severity - 1]
    s = int(224 * c)
    return pil_box_resize_u8(pil_box_resize_u8(np.array(x), s, s), 224, 224)


def jpeg_compression(x, severity, draws=None):
    """corruptions.py:375-382."""
    return jpeg_roundtrip_u8(np.array(x), PARAMS['jpeg_compression'][severity - 1])


def elastic_transform(x, severity, draws):
    """corruptions.py:395-424."""
    c = PARAMS['elastic_transform'][severity - 1]
    image = np.array(x, dtype=np.float32) / 255.
    shape = image.shape
    shape_size = shape[:2]
    center_square = np.float32(shape_size) // 2
    square_size = min(shape_size) // 3
    pts1 = np.float32([center_square + square_size,
                       [center_square[0] + square_size, center_square[1] - square_size],
                       center_square - square_size])
    pts2 = pts1 + draws['jitter']
    M = cv_get_affine_transform(pts1, pts2)
    image = cv_warp_affine_linear_reflect101(image, M)
    dx = (sk_gaussian(draws['field_x'], c[1], mode='reflect', truncate=3) * c[0]).astype(np.float32)
    dy = (sk_gaussian(draws['field_y'], c[1], mode='reflect', truncate=3) * c[0]).astype(np.float32)
    dx, dy = dx[..., np.newaxis], dy[..., np.newaxis]
    xx, yy, zz = np.meshgrid(np.arange(shape[1]), np.arange(shape[0]), np.arange(shape[2]))
    indices = np.reshape(yy + dy, (-1, 1)), np.reshape(xx + dx, (-1, 1)), np.reshape(zz, (-1, 1))
    return np.clip(ndi.map_coordinates(image, indices, order=1, mode='reflect').reshape(shape), 0, 1) * 255


# ---- OpenCV pieces of spatter's water branch, restated (PARITY UNPINNED: cv2 is absent from this image and the
#      reference has no fixture for them; the restatements follow OpenCV 4.5's imgproc sources as recalled in SURVEY.md
#      Appendix B -- canny.cpp, distransform.cpp, thresh.cpp, box_filter, histogram.cpp, filter.dispatch.cpp).  Every
#      stage is integer / fixed-point arithmetic, so the HIP kernels reproduce this file bit for bit.

def _reflect101_pad(a, r):
    return np.pad(a, r, mode='reflect')


def cv_canny_u8(img, low, high):
    """cv2.Canny(img, low, high): aperture 3, L1 gradient magnitude, BORDER_REPLICATE Sobel, non-maximum suppression with
    the TG22 fixed-point sector test, hysteresis over 8-neighbours.  Returns 0 / 255."""
    p = np.pad(img.astype(np.int32), 1, mode='edge')
    dx = (p[:-2, 2:] + 2 * p[1:-1, 2:] + p[2:, 2:]) - (p[:-2, :-2] + 2 * p[1:-1, :-2] + p[2:, :-2])
    dy = (p[2:, :-2] + 2 * p[2:, 1:-1] + p[2:, 2:]) - (p[:-2, :-2] + 2 * p[:-2, 1:-1] + p[:-2, 2:])
    mag = np.abs(dx) + np.abs(dy)
    mp = np.pad(mag, 1, mode='constant')                 # zero border rows / columns (canny.cpp mag_buf)
    m = mag
    left, right = mp[1:-1, :-2], mp[1:-1, 2:]
    up, down = mp[:-2, 1:-1], mp[2:, 1:-1]
    ax, ay = np.abs(dx).astype(np.int64), np.abs(dy).astype(np.int64) << 15
    tg22x = ax * 13573                                   # TG22 = (int)(0.4142135623730950488016887242097 * (1 << 15) + 0.5)
    tg67x = tg22x + (ax << 16)
    horiz = ay < tg22x
    vert = (~horiz) & (ay > tg67x)
    diag = ~(horiz | vert)
    s_neg = (dx < 0) != (dy < 0)                         # (xs ^ ys) < 0  ->  s = -1
    # s = 1: compare with (i-1, j-1) and (i+1, j+1);  s = -1: with (i-1, j+1) and (i+1, j-1)
    ul, dr = mp[:-2, :-2], mp[2:, 2:]
    ur, dl = mp[:-2, 2:], mp[2:, :-2]
    c_h = (m > left) & (m >= right)
    c_v = (m > up) & (m >= down)
    c_d = np.where(s_neg, (m > ur) & (m > dl), (m > ul) & (m > dr))
    local_max = (horiz & c_h) | (vert & c_v) | (diag & c_d)
    cand = (m > low) & local_max
    strong = cand & (m > high)
    # hysteresis: candidates 8-connected to a strong pixel
    import scipy.ndimage as ndi
    grown = ndi.binary_propagation(strong, structure=np.ones((3, 3), bool), mask=cand)
    return np.where(grown, 255, 0).astype(np.uint8)


CV_DIST_A, CV_DIST_B, CV_DIST_C = 65536, 91750, 143976      # cvRound({1, 1.4, 2.1969} * 2^16): DIST_L2, 5x5 mask
CV_DIST_INIT = (2 ** 31 - 1) >> 2


def cv_distance_transform_l2_5(src):
    """cv2.distanceTransform(src, cv2.DIST_L2, 5): two-pass 5x5 chamfer distance to the nearest zero pixel in 16.16 fixed
    point (distransform.cpp distanceTransform_5x5).  Returns the int64 fixed-point plane (float result = plane / 65536)."""
    h, w = src.shape
    a, b, c = CV_DIST_A, CV_DIST_B, CV_DIST_C
    T = np.full((h + 4, w + 4), CV_DIST_INIT, dtype=np.int64)
    jj = np.arange(w, dtype=np.int64)
    for i in range(h):                                   # forward
        r1, r2 = T[i + 1], T[i]                          # rows i-1, i-2 (padded by 2)
        cand = np.minimum.reduce([r2[1:w + 1] + c, r2[3:w + 3] + c, r1[0:w] + c, r1[1:w + 1] + b, r1[2:w + 2] + a,
                                  r1[3:w + 3] + b, r1[4:w + 4] + c])
        cand = np.where(src[i] == 0, 0, cand)
        cand[0] = min(cand[0], T[i + 2, 1] + a) if src[i, 0] != 0 else 0        # left border neighbour (INIT)
        # tmp[j] = min(cand[j], tmp[j-1] + a)  ==  a*j + running min of (cand[k] - a*k)
        T[i + 2, 2:w + 2] = a * jj + np.minimum.accumulate(cand - a * jj)
    for i in range(h - 1, -1, -1):                       # backward
        r1, r2 = T[i + 3], T[i + 4]                      # rows i+1, i+2
        cur = T[i + 2, 2:w + 2]
        cand = np.minimum.reduce([cur, r2[3:w + 3] + c, r2[1:w + 1] + c, r1[4:w + 4] + c, r1[3:w + 3] + b, r1[2:w + 2] + a,
                                  r1[1:w + 1] + b, r1[0:w] + c])
        rev = cand[::-1]
        T[i + 2, 2:w + 2] = (a * jj + np.minimum.accumulate(rev - a * jj))[::-1]
    return np.minimum(T[2:h + 2, 2:w + 2], CV_DIST_INIT)


def cv_blur3_f32_to_u8_fixed(fixed):
    """cv2.blur(float32 plane, (3, 3)).astype(np.uint8) on values that are multiples of 2^-16 below 21: the 9-term sums are
    exact in float32, the result is float32(sum * (1/9)) truncated."""
    p = _reflect101_pad(fixed.astype(np.int64), 1)
    s9 = sum(p[dy:dy + fixed.shape[0], dx:dx + fixed.shape[1]] for dy in range(3) for dx in range(3))
    val = ((s9.astype(np.float64) / 65536.0) * (1.0 / 9.0)).astype(np.float32)
    return val.astype(np.uint8)


def cv_equalize_hist_u8(img):
    """cv2.equalizeHist (histogram.cpp): lut[i] = saturate(cvRound(cumsum_after_first_bin * 255.f / (total - hist[first])))."""
    hist = np.bincount(img.ravel(), minlength=256)
    i0 = int(np.nonzero(hist)[0][0])
    total = img.size
    if hist[i0] == total:
        return np.full_like(img, i0)
    scale = np.float32(255.0) / np.float32(total - hist[i0])
    lut = np.zeros(256, dtype=np.uint8)
    acc = 0
    for i in range(i0 + 1, 256):
        acc += int(hist[i])
        lut[i] = np.uint8(min(255, max(0, int(np.rint(np.float32(acc) * scale)))))
    return lut[img]


def cv_filter2d_u8_int3(img, ker):
    """cv2.filter2D(u8, cv2.CV_8U, 3x3 integer kernel): correlation, BORDER_REFLECT_101, saturated."""
    p = _reflect101_pad(img.astype(np.int64), 1)
    h, w = img.shape
    acc = sum(int(ker[dy][dx]) * p[dy:dy + h, dx:dx + w] for dy in range(3) for dx in range(3))
    return np.clip(acc, 0, 255).astype(np.uint8)


def cv_blur3_u8(img):
    """cv2.blur(u8, (3, 3)): round(sum / 9) (9 is odd: no ties)."""
    p = _reflect101_pad(img.astype(np.int64), 1)
    h, w = img.shape
    s9 = sum(p[dy:dy + h, dx:dx + w] for dy in range(3) for dx in range(3))
    return ((2 * s9 + 9) // 18).astype(np.uint8)


def spatter_water_stages(liquid, c):
    """The integer pipeline of corruptions.py:305-318 on the thresholded liquid layer; returns (liquid_u8, dist_u8_final)."""
    liquid_u8 = (liquid * 255).astype(np.uint8)
    dist = 255 - cv_canny_u8(liquid_u8, 50, 150)
    fixed = np.minimum(cv_distance_transform_l2_5(dist), 20 * 65536)          # threshold(dist, 20, 20, THRESH_TRUNC)
    d = cv_blur3_f32_to_u8_fixed(fixed)
    d = cv_equalize_hist_u8(d)
    d = cv_filter2d_u8_int3(d, [[-2, -1, 0], [-1, 1, 1], [0, 1, 2]])
    d = cv_blur3_u8(d)
    return liquid_u8, d


def spatter(x, severity, draws):
    """corruptions.py:293-342.  Mud branch (severity 4-5): scipy / numpy only.  Water branch (severity 1-3): the OpenCV
    stages above -- PARITY UNPINNED."""
    c = PARAMS['spatter'][severity - 1]
    x = np.array(x, dtype=np.float32) / 255.
    liquid = sk_gaussian(draws['layer'], sigma=c[2])
    liquid[liquid < c[3]] = 0
    if c[5] == 0:
        liquid_u8, d = spatter_water_stages(liquid, c)
        m = liquid_u8 * d.astype(np.float32)                                  # uint8 * float32 -> float32, exact
        m = np.repeat(m[..., None], 4, axis=2)                                # cvtColor GRAY2BGRA (alpha of a float image: 1)
        m[..., 3] = 1.0
        m /= np.max(m, axis=(0, 1))
        m *= c[4]
        color = np.concatenate((175 / 255. * np.ones_like(m[..., :1]), 238 / 255. * np.ones_like(m[..., :1]),
                                238 / 255. * np.ones_like(m[..., :1])), axis=2)
        color = np.concatenate((color, np.ones_like(color[..., :1])), axis=2)  # BGR2BGRA
        xa = np.concatenate((x, np.ones_like(x[..., :1])), axis=2)
        return np.clip(xa + m * color, 0, 1)[..., :3] * 255                    # BGRA2BGR
    m = np.where(liquid > c[3], 1, 0)
    m = sk_gaussian(m.astype(np.float32), sigma=c[4])
    m[m < 0.8] = 0
    color = np.concatenate((63 / 255. * np.ones_like(x[..., :1]), 42 / 255. * np.ones_like(x[..., :1]),
                            20 / 255. * np.ones_like(x[..., :1])), axis=2)
    color *= m[..., np.newaxis]
    x *= (1 - m[..., np.newaxis])
    return np.clip(x + color, 0, 1) * 255


_FUNCS = {
    'gaussian_noise': gaussian_noise, 'shot_noise': shot_noise, 'impulse_noise': impulse_noise,
    'defocus_blur': defocus_blur, 'glass_blur': glass_blur, 'motion_blur': motion_blur,
    'zoom_blur': zoom_blur, 'snow': snow, 'frost': frost, 'fog': fog, 'brightness': brightness,
    'contrast': contrast, 'elastic_transform': elastic_transform, 'pixelate': pixelate,
    'jpeg_compression': jpeg_compression, 'speckle_noise': speckle_noise,
    'gaussian_blur': gaussian_blur, 'spatter': spatter, 'saturate': saturate}


def corrupt_float(name, x, severity, draws=None):
    """The corruption function's return value before the np.uint8 cast."""
    return _FUNCS[name](x, severity, draws)


def corrupt(name, x, severity, draws=None):
    """imagenet_c/__init__.py:13-35: dispatch + np.uint8 truncation."""
    return _u8(corrupt_float(name, x, severity, draws))


def corrupt_batch(name, batch, severity, rng):
    """add_noise_utils.py:27-31: per-image loop in index order, draws pulled from one stream."""
    out = np.empty_like(batch)
    for i in range(batch.shape[0]):
        out[i] = corrupt(name, batch[i], severity, draw(name, batch[i], severity, rng))
    return out
