"""TEST INFRASTRUCTURE ONLY -- numpy restatement of cv2.resize for 8-bit 3-channel images
(INTER_NEAREST / LINEAR / CUBIC / AREA / LANCZOS4), the five 'opencv-*' ImageNet-S resize operators
(RobustART/noise/utils/imagenet_s_gen.py:27-33, 138-146).

PARITY UNPINNED: OpenCV is an unvendored, unpinned dependency of the reference (requirements.txt: opencv-python)
and is not installed in the build container, so there is nothing to run and no golden vector to capture.  This
restates the published algorithm of opencv/modules/imgproc/src/resize.cpp (4.x series) from its source:
float coordinates `fx = (float)((dx+0.5)*scale - 0.5)`, int16 coefficients `saturate_cast<short>(c*2048)`,
int32 horizontal rows with replicated edge taps, the two vertical fixed-point casts, INTER_LINEAR -> INTER_AREA for
exact 2x2 decimation, integer / DecimateAlpha area paths, area-mode coordinates when an axis is up-scaled.
Only tests/ may import this module.
"""
import math

import numpy as np

f32 = np.float32
NEAREST, LINEAR, CUBIC, AREA, LANCZOS4 = 0, 1, 2, 3, 4
FLT_EPSILON = 1.1920929e-07
DBL_EPSILON = 2.220446049250313e-16


def _sat_short(v):
    return int(np.clip(np.rint(f32(v)), -32768, 32767))          # cvRound: half to even


def _cubic(x):
    A = f32(-0.75)
    x = f32(x)
    one = f32(1)
    c0 = ((A * (x + one) - f32(5) * A) * (x + one) + f32(8) * A) * (x + one) - f32(4) * A
    c1 = ((A + f32(2)) * x - (A + f32(3))) * x * x + one
    c2 = ((A + f32(2)) * (one - x) - (A + f32(3))) * (one - x) * (one - x) + one
    c3 = one - c0 - c1 - c2
    return [f32(c0), f32(c1), f32(c2), f32(c3)]


def _lanczos4(x):
    x = f32(x)
    if x < FLT_EPSILON:
        c = [f32(0)] * 8
        c[3] = f32(1)
        return c
    s45 = 0.70710678118654752440084436210485
    cs = [(1, 0), (-s45, -s45), (0, 1), (s45, -s45), (-1, 0), (s45, s45), (0, -1), (-s45, s45)]
    y0 = -(float(x) + 3) * math.pi * 0.25
    s0, c0 = math.sin(y0), math.cos(y0)
    c, total = [], f32(0)
    for i in range(8):
        y = -(float(x) + 3 - i) * math.pi * 0.25
        v = f32((cs[i][0] * s0 + cs[i][1] * c0) / (y * y))
        c.append(v)
        total = f32(total + v)
    inv = f32(f32(1) / total)
    return [f32(v * inv) for v in c]


def _axis_table(ssize, dsize, interp, is_x):
    inv_scale = dsize / ssize
    scale = 1.0 / inv_scale
    ksize = 4 if interp == CUBIC else (8 if interp == LANCZOS4 else 2)
    first, coef = [], []
    for d in range(dsize):
        if interp != AREA:
            f = f32((d + 0.5) * scale - 0.5)
            s = int(math.floor(f))
            f = f32(f - f32(s))
        else:
            s = int(math.floor(d * scale))
            f = f32((d + 1) - (s + 1) * inv_scale)
            f = f32(0) if f <= 0 else f32(f - f32(math.floor(f)))
        if is_x and ksize == 2:
            if s < 0:
                f, s = f32(0), 0
            if s >= ssize - 1:
                f, s = f32(0), ssize - 1
        if interp == CUBIC:
            c = _cubic(f)
        elif interp == LANCZOS4:
            c = _lanczos4(f)
        else:
            c = [f32(f32(1) - f), f]
        first.append(s - (ksize // 2 - 1))
        coef.append([_sat_short(f32(v * f32(2048))) for v in c])
    return np.array(first), np.array(coef, dtype=np.int64)


def _area_table(ssize, dsize):
    scale = ssize / dsize
    rows = []
    for dx in range(dsize):
        fsx1 = dx * scale
        fsx2 = fsx1 + scale
        cell = min(scale, ssize - fsx1)
        sx1, sx2 = int(math.ceil(fsx1)), int(math.floor(fsx2))
        sx2 = min(sx2, ssize - 1)
        sx1 = min(sx1, sx2)
        r = []
        if sx1 - fsx1 > 1e-3:
            r.append((sx1 - 1, f32((sx1 - fsx1) / cell)))
        for sx in range(sx1, sx2):
            r.append((sx, f32(1.0 / cell)))
        if fsx2 - sx2 > 1e-3:
            r.append((sx2, f32(min(min(fsx2 - sx2, 1.0), cell) / cell)))
        rows.append(r)
    return rows


def resize(img, dsize_wh, interp):
    """img: (h, w, 3) uint8; dsize_wh = (width, height) as in cv2.resize -> (height, width, 3) uint8."""
    h, w, _ = img.shape
    rw, rh = int(dsize_wh[0]), int(dsize_wh[1])
    scale_x, scale_y = 1.0 / (rw / w), 1.0 / (rh / h)
    if interp == NEAREST:
        xs = np.minimum(np.floor(np.arange(rw) * scale_x).astype(np.int64), w - 1)
        ys = np.minimum(np.floor(np.arange(rh) * scale_y).astype(np.int64), h - 1)
        return img[ys][:, xs]
    isx, isy = int(np.rint(scale_x)), int(np.rint(scale_y))
    area_fast = abs(scale_x - isx) < DBL_EPSILON and abs(scale_y - isy) < DBL_EPSILON
    if interp == LINEAR and area_fast and isx == 2 and isy == 2:
        interp = AREA
    if interp == AREA and scale_x >= 1 and scale_y >= 1:
        if area_fast:
            s = img[:rh * isy, :rw * isx].astype(np.int64).reshape(rh, isy, rw, isx, 3).sum((1, 3))
            if isx == 2 and isy == 2:
                return ((s + 2) >> 2).astype(np.uint8)
            v = np.rint(s.astype(f32) * f32(f32(1) / f32(isx * isy)))
            return np.clip(v, 0, 255).astype(np.uint8)
        xt, yt = _area_table(w, rw), _area_table(h, rh)
        out = np.zeros((rh, rw, 3), np.uint8)
        src = img.astype(f32)
        # buf[sy][dx] = sum_k S[sy][sx_k] * alpha_k, accumulated in table order (float32)
        buf = np.zeros((h, rw, 3), f32)
        for dx, r in enumerate(xt):
            acc = np.zeros((h, 3), f32)
            for sx, a in r:
                acc = (acc + src[:, sx, :] * a).astype(f32)
            buf[:, dx, :] = acc
        for dy, r in enumerate(yt):
            acc = None
            for sy, b in r:
                t = (buf[sy] * b).astype(f32)
                acc = t if acc is None else (acc + t).astype(f32)
            out[dy] = np.clip(np.rint(acc), 0, 255).astype(np.uint8)
        return out
    fx, cx = _axis_table(w, rw, interp, True)
    fy, cyc = _axis_table(h, rh, interp, False)
    k = cx.shape[1]
    src = img.astype(np.int64)
    rows = np.zeros((h, rw, 3), np.int64)
    for j in range(k):
        idx = np.clip(fx + j, 0, w - 1)
        rows += src[:, idx, :] * cx[:, j][None, :, None]
    out = np.zeros((rh, rw, 3), np.int64)
    if k == 2:
        s0 = rows[np.clip(fy, 0, h - 1)]
        s1 = rows[np.clip(fy + 1, 0, h - 1)]
        b0, b1 = cyc[:, 0][:, None, None], cyc[:, 1][:, None, None]
        out = (((b0 * (s0 >> 4)) >> 16) + ((b1 * (s1 >> 4)) >> 16) + 2) >> 2
    else:
        for j in range(k):
            out += rows[np.clip(fy + j, 0, h - 1)] * cyc[:, j][:, None, None]
        out = (out + (1 << 21)) >> 22
    return np.clip(out, 0, 255).astype(np.uint8)


def imagenet_s_val(img, interp, resize=224):
    """imagenet_s_gen.py:138-146: cv2.resize to (8/7*resize, 8/7*resize), centre crop resize x resize."""
    big = int(resize * 8 / 7)
    r = globals()['resize'](img, (big, big), interp)
    o = int(round((big - resize) / 2.))
    return r[o:o + resize, o:o + resize]
