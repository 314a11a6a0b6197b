"""ImageNet-S ("system noise": decoder x resize operator) -- drop-in for add_noise_for_imagenet_s /
ImageTransfer(return_online=True) (RobustART/noise/utils/add_noise_utils.py:34-38, imagenet_s_gen.py:38-279).

Decoding a file is host I/O (PIL, as the reference's 'pil' decoder); the resize operator -- the part that differs
between the ImageNet-S variants and the part that is arithmetic -- runs on the GPU, bit-exact with Pillow, through
rart_pil_resize_u8; the five 'opencv-*' operators run through rart_cv_resize_u8, a restatement of OpenCV's resize.cpp
(OpenCV is not installed here, so those five are parity-unpinned -- see oracle/resize_cv_np.py).  Not available in this
build (raise NotImplementedError, loudly): the 'opencv' and 'ffmpeg' decoders and the reference's memcached reader.  Unlike the reference (which passes float sizes to Image.resize and only works on
Python <= 3.9, imagenet_s_gen.py:129,167) sizes are converted with int()."""
import random

import numpy as np

from .. import _lib

PIL_FILTERS = {'pil-nearest': 0, 'pil-bilinear': 1, 'pil-cubic': 2, 'pil-box': 3, 'pil-hamming': 4, 'pil-lanczos': 5}
# imagenet_s_gen.py:27-33 -> cv2.INTER_* constants
CV_MODES = {'opencv-nearest': 0, 'opencv-bilinear': 1, 'opencv-cubic': 2, 'opencv-area': 3, 'opencv-lanczos': 4}


def pil_resize(batch_u8, resize_hw, filter_id, crop=None):
    """batch_u8: CUDA uint8 (n,h,w,3) -> CUDA uint8 (n, ch, cw, 3) = Image.resize((rw, rh), filter) then crop
    (cy, cx, ch, cw); crop=None keeps the whole resized image."""
    torch = _lib.require_gpu()
    lib = _lib.load()
    n, h, w, _ = batch_u8.shape
    rh, rw = int(resize_hw[0]), int(resize_hw[1])
    cy, cx, ch, cw = crop if crop is not None else (0, 0, rh, rw)
    out = torch.empty(n, ch, cw, 3, dtype=torch.uint8, device=batch_u8.device)
    nb = lib.rart_pil_resize_workspace_bytes(n, h, w, rh, rw, filter_id, cy, cx, ch, cw)
    ws = _lib.workspace(nb, batch_u8.device)
    _lib.check(lib.rart_pil_resize_u8(_lib.ptr(batch_u8), _lib.ptr(out), n, h, w, rh, rw, filter_id, cy, cx, ch, cw,
                                      _lib.ptr(ws), nb, _lib.stream_ptr()))
    return out


def cv_resize(batch_u8, resize_hw, interpolation, crop=None):
    """batch_u8: CUDA uint8 (n,h,w,3) -> cv2.resize(img, (rw, rh), interpolation) then crop (cy, cx, ch, cw)."""
    torch = _lib.require_gpu()
    lib = _lib.load()
    n, h, w, _ = batch_u8.shape
    rh, rw = int(resize_hw[0]), int(resize_hw[1])
    cy, cx, ch, cw = crop if crop is not None else (0, 0, rh, rw)
    out = torch.empty(n, ch, cw, 3, dtype=torch.uint8, device=batch_u8.device)
    nb = lib.rart_cv_resize_workspace_bytes(n, h, w, rh, rw, interpolation, cy, cx, ch, cw)
    ws = _lib.workspace(nb, batch_u8.device)
    _lib.check(lib.rart_cv_resize_u8(_lib.ptr(batch_u8), _lib.ptr(out), n, h, w, rh, rw, interpolation, cy, cx, ch, cw,
                                     _lib.ptr(ws), nb, _lib.stream_ptr()))
    return out


def decode(path_or_bytes, decoder_type='pil'):
    """imagenet_s_gen.py:177-196 ('pil' branch): RGB uint8 array."""
    if decoder_type != 'pil':
        raise NotImplementedError("decoder_type %r needs OpenCV / ffmpeg, which this build does not have; only 'pil' is "
                                  "available" % (decoder_type,))
    import io
    from PIL import Image
    src = io.BytesIO(path_or_bytes) if isinstance(path_or_bytes, (bytes, bytearray)) else path_or_bytes
    with Image.open(src) as img:
        return np.array(img.convert('RGB'))


def _train_params(img_hw, rng):
    """imagenet_s_gen.py:222-263 (random-resized-crop box), with an explicit random.Random."""
    import math
    h, w = img_hw
    area = h * w
    for _ in range(10):
        target_area = rng.uniform(0.08, 1.0) * area
        log_ratio = (math.log(3. / 4.), math.log(4. / 3.))
        ar = math.exp(rng.uniform(*log_ratio))
        cw, chh = int(round(math.sqrt(target_area * ar))), int(round(math.sqrt(target_area / ar)))
        if 0 < cw <= w and 0 < chh <= h:
            return rng.randint(0, h - chh), rng.randint(0, w - cw), chh, cw
    in_ratio = w / h
    if in_ratio < 3. / 4.:
        cw = w
        chh = int(round(cw / (3. / 4.)))
    elif in_ratio > 4. / 3.:
        chh = h
        cw = int(round(chh * (4. / 3.)))
    else:
        cw, chh = w, h
    return (h - chh) // 2, (w - cw) // 2, chh, cw


def image_transfer(image, decoder_type='pil', resize_type='pil-bilinear', transform_type='val', resize=224, seed=None):
    """ImageTransfer(file_path=image, ..., return_online=True).getimage(): (224,224,3) uint8 ndarray.  `image`: a file
    path / encoded bytes, or an already decoded HxWx3 uint8 array."""
    torch = _lib.require_gpu()
    if resize_type not in PIL_FILTERS and resize_type not in CV_MODES:
        raise NotImplementedError(resize_type)
    arr = image if isinstance(image, np.ndarray) else decode(image, decoder_type)
    size = (resize, resize) if not isinstance(resize, tuple) else resize
    is_cv = resize_type in CV_MODES
    op = cv_resize if is_cv else pil_resize
    f = CV_MODES[resize_type] if is_cv else PIL_FILTERS[resize_type]
    if transform_type == 'val':
        first = tuple(int(s * 8 / 7) for s in size)                    # imagenet_s_gen.py:129,139
        th, tw = size
        i, j = int(round((first[0] - th) / 2.)), int(round((first[1] - tw) / 2.))
        dev = torch.from_numpy(np.ascontiguousarray(arr)[None]).cuda()
        return op(dev, first, f, crop=(i, j, th, tw))[0].cpu().numpy()
    if transform_type == 'train':
        # seed None: the GLOBAL random module, like the reference's get_params (imagenet_s_gen.py:222-246), so a user's
        # random.seed(...) makes AddNoise('imagenet-s') reproducible; an explicit seed gets its own generator
        y, x, h, w = _train_params(arr.shape[:2], random if seed is None else random.Random(seed))
        dev = torch.from_numpy(np.ascontiguousarray(arr[y:y + h, x:x + w])[None]).cuda()
        return op(dev, size, f)[0].cpu().numpy()
    raise NotImplementedError(transform_type)
