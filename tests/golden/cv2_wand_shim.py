"""Stand-ins for the OpenCV and ImageMagick (wand) entry points that five more of the reference's corruptions call
(RobustART/noise/utils/imagenet_c/corruptions.py: defocus_blur :187-198 with disk :26-38, motion_blur :201-216, snow :265-290,
spatter severities 1-3 :293-328, elastic_transform :395-424).  Neither library is installed, so these corruptions cannot run
unmodified; with ONLY the library calls supplied, the reference's own code around them -- parameter tables, the order and shapes of
its np.random draws, thresholds, channel loops, BGR / RGB flips, blends, rot90, clipping, the final uint8 cast -- runs as written
and its outputs become golden vectors "pinned modulo shim" (tests/golden/make_golden_shim.py).

Every function here simply forwards to the oracle's own restatement of that primitive (oracle/corruptions_np.py: cv_* / im_*
functions, written from the libraries' published semantics).  So these goldens pin the COMPOSITION, not the
primitives: a wrong tap order inside cv2.filter2D would be invisible here, a wrong random-draw order or a swapped channel in the
reference's glue would not.  The rows stay "parity unpinned" in DESIGN.md.
Used only by the golden generator in the build container; nothing on the GPU box imports it.
"""
import io
import types

import numpy as np
from PIL import Image as _PILImage

from oracle import corruptions_np as O

BORDER_REFLECT_101, IMREAD_UNCHANGED, CV_8U = 4, -1, 0
COLOR_RGB2GRAY, COLOR_GRAY2BGRA, COLOR_BGR2BGRA, COLOR_BGRA2BGR = 7, 9, 0, 1
DIST_L2, THRESH_TRUNC = 2, 2
# constants other reference modules read at import time (imagenet_s_gen.py:27-33); no function here implements them
INTER_NEAREST, INTER_LINEAR, INTER_CUBIC, INTER_AREA, INTER_LANCZOS4 = 0, 1, 2, 3, 4
IMREAD_COLOR, COLOR_BGR2RGB = 1, 4


def GaussianBlur(src, ksize, sigmaX):
    assert ksize[0] == ksize[1]
    return O.cv_gaussian_blur_f32(src, ksize[0], sigmaX)


def filter2D(src, ddepth, kernel):
    if ddepth == CV_8U:                                     # spatter: uint8 plane, 3 x 3 integer kernel, saturated
        return O.cv_filter2d_u8_int3(src, kernel)
    return O.cv_filter2d_reflect101(src, np.asarray(kernel))     # defocus_blur: fp64 plane, fp32 disk


def getAffineTransform(src, dst):
    return O.cv_get_affine_transform(src, dst)


def warpAffine(src, M, dsize, borderMode=None):
    assert borderMode == BORDER_REFLECT_101 and tuple(dsize) == (src.shape[1], src.shape[0])
    return O.cv_warp_affine_linear_reflect101(src, M)


def imdecode(buf, flags):
    with _PILImage.open(io.BytesIO(np.asarray(buf, dtype=np.uint8).tobytes())) as im:
        arr = np.array(im)
    return arr[..., ::-1].copy() if arr.ndim == 3 else arr   # OpenCV hands colour images back as BGR


def cvtColor(src, code):
    if code == COLOR_RGB2GRAY:                              # float32 HxWx3 -> HxW (the weights OpenCV documents)
        return (np.float32(0.299) * src[..., 0] + np.float32(0.587) * src[..., 1] + np.float32(0.114) * src[..., 2])
    if code == COLOR_GRAY2BGRA:                             # float image: replicate, alpha = 1
        out = np.repeat(src[..., None], 4, axis=2)
        out[..., 3] = 1.0
        return out
    if code == COLOR_BGR2BGRA:
        return np.concatenate((src, np.ones_like(src[..., :1])), axis=2)
    if code == COLOR_BGRA2BGR:
        return src[..., :3]
    raise NotImplementedError(code)


def Canny(image, threshold1, threshold2):
    return O.cv_canny_u8(image, threshold1, threshold2)


def distanceTransform(src, distanceType, maskSize):
    assert distanceType == DIST_L2 and maskSize == 5
    return (O.cv_distance_transform_l2_5(src).astype(np.float64) / 65536.0).astype(np.float32)


def threshold(src, thresh, maxval, type_):
    assert type_ == THRESH_TRUNC
    return float(thresh), np.minimum(src, np.float32(thresh))


def blur(src, ksize):
    assert tuple(ksize) == (3, 3)
    if src.dtype == np.uint8:
        return O.cv_blur3_u8(src)
    # float32 plane (the truncated distance, multiples of 2^-16 below 21): 9-term sum, times 1/9, in float32
    p = np.pad(src.astype(np.float64), 1, mode='reflect')
    h, w = src.shape
    s9 = sum(p[dy:dy + h, dx:dx + w] for dy in range(3) for dx in range(3))
    return (s9 * (1.0 / 9.0)).astype(np.float32)


def equalizeHist(src):
    return O.cv_equalize_hist_u8(src)


class WandImage(object):
    """wand.image.Image(blob=png bytes): just enough for the reference's MotionImage subclass."""

    def __init__(self, blob=None, **_):
        with _PILImage.open(io.BytesIO(blob)) as im:
            self.arr = np.array(im)
        self.wand = self                                      # the handle the reference passes to MagickMotionBlurImage

    def make_blob(self):
        out = io.BytesIO()
        _PILImage.fromarray(self.arr).save(out, format='PNG')
        return out.getvalue()


class _MotionBlurFn(object):
    argtypes = None                                           # the reference assigns the ctypes signature here

    def __call__(self, handle, radius, sigma, angle):
        handle.arr = O.im_motion_blur_u8(handle.arr, radius, sigma, angle)


def install(sys_modules):
    """Register `cv2`, `wand`, `wand.image`, `wand.api`, `wand.color` stand-ins (before the reference is imported)."""
    cv2 = types.ModuleType('cv2')
    for k, v in globals().items():
        if k[0].isupper() or k in ('filter2D', 'getAffineTransform', 'warpAffine', 'imdecode', 'cvtColor', 'distanceTransform',
                                   'threshold', 'blur', 'equalizeHist'):
            if k not in ('WandImage', 'O'):
                setattr(cv2, k, v)
    wand, wimg, wapi, wcol = (types.ModuleType(n) for n in ('wand', 'wand.image', 'wand.api', 'wand.color'))
    wimg.Image = WandImage
    lib = types.SimpleNamespace(MagickMotionBlurImage=_MotionBlurFn())
    wapi.library = lib
    wand.image, wand.api, wand.color = wimg, wapi, wcol
    sys_modules.update({'cv2': cv2, 'wand': wand, 'wand.image': wimg, 'wand.api': wapi, 'wand.color': wcol})
    return cv2
