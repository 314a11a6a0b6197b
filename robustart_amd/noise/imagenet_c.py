"""ImageNet-C corruptions on MI355X -- host side of rart_corrupt_u8.

Mirrors RobustART/noise/utils/imagenet_c/__init__.py:5-35: the same `corruption_tuple` order,
`corruption_dict`, and `corrupt(x, severity, corruption_name, corruption_number)` with the same
ValueError when neither a name nor a number is given.  Differences (documented in DESIGN.md):
  * works on whole batches (uint8 NHWC) resident in HBM instead of one PIL image at a time;
  * randomness is counter based (seed, global sample index, element) -- see noise/rng.py -- or
    injected through `draws` for bit-exact parity with the reference's np.random draws.
"""
import ctypes

import numpy as np

from .. import _lib
from . import rng as _rng

CORRUPTION_NAMES = (
    'gaussian_noise', 'shot_noise', 'impulse_noise', 'defocus_blur',
    'glass_blur', 'motion_blur', 'zoom_blur', 'snow', 'frost', 'fog',
    'brightness', 'contrast', 'elastic_transform', 'pixelate', 'jpeg_compression',
    'speckle_noise', 'gaussian_blur', 'spatter', 'saturate')

# per-corruption: ordered (draw key, numpy dtype) of the `injected` device arrays (include/robustart_hip.h)
_INJECT_LAYOUT = {
    'gaussian_noise': (('noise', np.float64),),
    'speckle_noise': (('noise', np.float64),),
    'shot_noise': (('counts', np.int32),),
    'impulse_noise': (('code', np.uint8),),
    'glass_blur': (('dxdy', np.int8),),
    'motion_blur': (('angle', np.float64),),
    'snow': (('layer', np.float64), ('angle', np.float64)),
    'frost': (('texture', np.uint8),),
    'fog': (('uniform', np.float64),),
    'elastic_transform': (('jitter', np.float32), ('field_x', np.float64), ('field_y', np.float64)),
    'spatter': (('layer', np.float64),),
}

_frost_textures = []
_frost_dev = {}          # device (str) -> stacked textures on that device


def set_frost_textures(textures):
    """Register the frost photographs (uint8 HxWx3 RGB arrays, each at least 225x225).  The
    reference expects six files frost/frost{1..6}.{png,jpg} that are not in its repository
    (corruptions.py:251-256); without textures `frost` raises FileNotFoundError here."""
    global _frost_textures, _frost_dev
    _frost_textures = [np.ascontiguousarray(t, dtype=np.uint8) for t in textures]
    _frost_dev = {}


def _frost_stack(torch, device=None):
    """The registered photographs as one uint8 [k][Hmax][Wmax][3] tensor (zero padded) on `device` (default: the current
    device), built on first use and cached PER DEVICE: a process driving several GPUs gets one copy on each."""
    dev = torch.device(device) if device is not None else torch.device('cuda', torch.cuda.current_device())
    key = str(dev)
    if key not in _frost_dev:
        hm = max(t.shape[0] for t in _frost_textures)
        wm = max(t.shape[1] for t in _frost_textures)
        st = np.zeros((len(_frost_textures), hm, wm, 3), dtype=np.uint8)
        for i, t in enumerate(_frost_textures):
            st[i, :t.shape[0], :t.shape[1]] = t
        _frost_dev[key] = torch.from_numpy(st).to(dev)
    return _frost_dev[key]


def _as_device_batch(x):
    """-> (uint8 cuda tensor NHWC, kind, original) ; kind in {'tensor','ndarray','image'}."""
    torch = _lib.require_gpu()
    if isinstance(x, torch.Tensor):
        if x.dtype != torch.uint8 or not x.is_cuda or x.dim() != 4 or x.shape[-1] != 3:
            raise TypeError('tensor input must be a CUDA uint8 tensor shaped (n, h, w, 3)')
        if not x.is_contiguous():
            raise TypeError('tensor input must be contiguous (the kernels write in place)')
        return x, 'tensor', x
    arr = np.asarray(x)
    if arr.dtype != np.uint8:
        raise TypeError('image data must be uint8, got %s' % arr.dtype)
    if arr.ndim == 3:
        if arr.shape[-1] != 3:
            raise TypeError('expected an HxWx3 image')
        return torch.from_numpy(np.ascontiguousarray(arr)[None]).cuda(), 'image', x
    if arr.ndim == 4 and arr.shape[-1] == 3:
        return torch.from_numpy(np.ascontiguousarray(arr)).cuda(), 'ndarray', x
    raise TypeError('expected (h, w, 3) or (n, h, w, 3) uint8 data')


def _host_draws(name, n, severity, seed, offset, device=None):
    """Native-mode per-image scalar draws made on the host (counter based, stream ids >= 8)."""
    torch = _lib.require_gpu()
    if name == 'frost':
        if not _frost_textures:
            raise FileNotFoundError(
                "frost needs texture photographs: the reference's frost/frost{1..6} files are not part "
                "of its repository.  Call robustart_amd.noise.imagenet_c.set_frost_textures([...]).")
        # corruptions.py:250,259: randint(5) over the 6-entry list, then the crop origin.  The photographs live on the device (uploaded
        # once, padded into one [k][H][W][3] stack); the n crops are ONE gather there -- the host loop used to copy 38 MB per 256
        # images through pageable memory (6.8 ms of a 7 ms frost call)
        stack = _frost_stack(torch, device)          # on the BATCH's device (multi-GPU processes)
        k = min(5, len(_frost_textures))
        samples = offset + np.arange(n, dtype=np.int64)
        idx = (_rng.host_uniform_many(seed, samples, 8) * k).astype(np.int64)
        th = np.array([t.shape[0] for t in _frost_textures], dtype=np.int64)[idx]
        tw = np.array([t.shape[1] for t in _frost_textures], dtype=np.int64)[idx]
        xs = (_rng.host_uniform_many(seed, samples, 9) * (th - 224)).astype(np.int64)
        ys = (_rng.host_uniform_many(seed, samples, 10) * (tw - 224)).astype(np.int64)
        meta = torch.from_numpy(np.stack([idx, xs, ys])).to(stack.device, non_blocking=False)
        ar = torch.arange(224, device=stack.device)
        crops = stack[meta[0][:, None, None], meta[1][:, None, None] + ar[None, :, None], meta[2][:, None, None] + ar[None, None, :]]
        return {'texture': crops.contiguous()}
    return None


def corrupt_batch_(batch, corruption_id, severity, seed=None, sample_offset=None, draws=None, out=None):
    """Corruption of a CUDA uint8 NHWC tensor through the C-ABI, in place unless `out` (same shape)
    is given.  `draws`: None, or a dict (keys as oracle/corruptions_np.draw) of per-image arrays
    stacked on axis 0."""
    torch = _lib.require_gpu()
    lib = _lib.load()
    n, h, w, _ = batch.shape
    name = CORRUPTION_NAMES[corruption_id]
    if seed is None:
        seed = _rng.current_seed()
    if sample_offset is None:
        sample_offset = _rng.next_offset(n)
    if draws is None and name == 'frost' and (h, w) == (224, 224) and 1 <= len(_frost_textures) <= 8 and n <= 65535 \
            and batch.data_ptr() % 16 == 0 and (out is None or out.data_ptr() % 16 == 0):
        # the photographs are resident on the device: texture index, crop origin and crop read happen in the blend kernel
        stack = _frost_stack(torch, batch.device)
        dims = (ctypes.c_int * (2 * len(_frost_textures)))(*[int(v) for t in _frost_textures for v in t.shape[:2]])
        dst = batch if out is None else out
        _lib.check(lib.rart_frost_textures_u8(_lib.ptr(batch), _lib.ptr(dst), n, h, w, severity, _lib.ptr(stack), len(_frost_textures),
                                              stack.shape[1], stack.shape[2], dims, seed, sample_offset, _lib.stream_ptr()))
        return dst
    if draws is None:
        draws = _host_draws(name, n, severity, seed, sample_offset, batch.device)
    keep, held = [], []
    inj = None
    n_inj = 0
    if draws is not None and name in _INJECT_LAYOUT:
        arrs = []
        for key, dt in _INJECT_LAYOUT[name]:
            if key not in draws:
                break
            v = draws[key]
            if torch.is_tensor(v):                   # already on the device (frost crops gathered there): same-stream lifetime, no sync
                t = v.to(torch.uint8).contiguous().view(-1) if dt == np.uint8 else v.contiguous().view(-1)
                held.append(t)
                arrs.append(t.data_ptr())
                continue
            if isinstance(v, (list, tuple)) and len(v) and isinstance(v[0], (list, tuple)):
                # per-image list of arrays (fog: the successive np.random.uniform results) -> [n][flat]
                v = np.stack([np.concatenate([np.asarray(q).ravel() for q in per]) for per in v])
            a = np.ascontiguousarray(np.asarray(v), dtype=dt)
            if name == 'glass_blur' and key == 'dxdy' and a.size:
                # the overlapped copy chain packs dx + delta and dy + delta into nibbles and its schedule is proved for |dx|, |dy| <= delta
                # (tests/test_schedules_cpu.py): numpy's randint(-delta, delta) of the reference never leaves [-delta, delta), an injected
                # array that does is refused here, on the host, instead of silently addressing the wrong pixel
                delta = (1, 2, 2, 3, 4)[severity - 1]
                if int(a.min()) < -delta or int(a.max()) >= delta:
                    raise ValueError('glass_blur draws must lie in [-%d, %d) at severity %d, got [%d, %d]' % (delta, delta, severity, int(a.min()), int(a.max())))
            t = torch.from_numpy(a.reshape(-1).view(np.uint8) if a.dtype != np.uint8 else a.reshape(-1)).cuda()
            keep.append(t)
            arrs.append(t.data_ptr())
        n_inj = len(arrs)
        if n_inj:
            inj = (ctypes.c_void_p * n_inj)(*arrs)
    ws_bytes = lib.rart_corrupt_workspace_bytes(corruption_id, severity, n, h, w)
    ws = _lib.workspace(ws_bytes, batch.device)
    dst = batch if out is None else out
    _lib.check(lib.rart_corrupt_u8(_lib.ptr(batch), _lib.ptr(dst), n, h, w, corruption_id, severity,
                                   seed, sample_offset, inj, n_inj, _lib.ptr(ws), ws_bytes if ws is not None else 0,
                                   _lib.stream_ptr()))
    if keep:
        torch.cuda.current_stream().synchronize()   # injected buffers must outlive the kernels
    return dst


def to_unit_nchw(batch_u8, out=None):
    """uint8 NHWC device batch -> fp32 NCHW in [0,1] (x / 255), one kernel (rart_u8_to_unit_f32_nchw): the input of the attacks
    (adv/attack.py:20-23); bit-identical to batch.permute(0, 3, 1, 2).float().div(255) (a multiplication by fp32 1/255)."""
    torch = _lib.require_gpu()
    n, h, w, _ = batch_u8.shape
    if out is None:
        out = torch.empty(n, 3, h, w, dtype=torch.float32, device=batch_u8.device)
    if (h * w) % 4:
        out.copy_(batch_u8.permute(0, 3, 1, 2).float().div_(255.0))
        return out
    _lib.check(_lib.load().rart_u8_to_unit_f32_nchw(_lib.ptr(batch_u8), _lib.ptr(out), n, h, w, _lib.stream_ptr()))
    return out


def noise_severities_(batch, outs, severities=(1, 2, 3, 4, 5), seeds=None, sample_offset=None, corruption_id=0):
    """gaussian_noise (corruption_id 0) or speckle_noise (15) of ONE source batch at several severities in ONE launch
    (rart_noise_multi_u8): the generation loop of ImageNet-C corrupts every image at all five severities
    (imagenet_c/__init__.py:13-35, one call per (image, severity)); five launches read the source five times, this one once.
    outs[i] receives severity severities[i] drawn with seeds[i] (default: the current seed + severity, i.e. an independent field per
    severity, as the reference's successive np.random draws are) -- bit-identical to corrupt_batch_(batch, corruption_id,
    severities[i], seeds[i], sample_offset, out=outs[i]).  Falls back to exactly those calls for sizes the fused kernel does not take."""
    torch = _lib.require_gpu()
    lib = _lib.load()
    n, h, w, _ = batch.shape
    ns = len(severities)
    assert len(outs) == ns and 1 <= ns <= 8
    if seeds is None:
        seeds = [(_rng.current_seed() + int(sv)) & 0xFFFFFFFFFFFFFFFF for sv in severities]
    if sample_offset is None:
        sample_offset = _rng.next_offset(n)
    ptrs = (ctypes.c_void_p * ns)(*[o.data_ptr() for o in outs])
    rc = lib.rart_noise_multi_u8(_lib.ptr(batch), ptrs, ns, n, h, w, corruption_id, (ctypes.c_int * ns)(*[int(v) for v in severities]),
                                 (ctypes.c_uint64 * ns)(*[int(v) for v in seeds]), sample_offset, _lib.stream_ptr())
    if rc == 2:                                   # RART_ERR_UNSUPPORTED: size / alignment / generator -> the per-severity launches
        for o, sv, sd in zip(outs, severities, seeds):
            corrupt_batch_(batch, corruption_id, int(sv), int(sd), sample_offset, out=o)
        return outs
    _lib.check(rc)
    return outs


def _make(name, cid):
    def f(x, severity=1, **kw):
        return corrupt(x, severity=severity, corruption_number=cid, **kw)
    f.__name__ = name
    f.__qualname__ = name
    f.__doc__ = 'ImageNet-C %s (RobustART/noise/utils/imagenet_c/corruptions.py) on MI355X.' % name
    return f


corruption_tuple = tuple(_make(nm, i) for i, nm in enumerate(CORRUPTION_NAMES))
corruption_dict = {f.__name__: f for f in corruption_tuple}
globals().update(corruption_dict)


def corrupt(x, severity=1, corruption_name=None, corruption_number=-1, seed=None, sample_offset=None,
            draws=None):
    """imagenet_c/__init__.py:13-35.  x: PIL image / HxWx3 uint8 array (returns a new HxWx3 array, like
    the reference), an (n,h,w,3) uint8 ndarray or CUDA uint8 tensor (corrupted IN PLACE and returned,
    like add_noise_utils.py:27-31)."""
    if corruption_name:
        cid = CORRUPTION_NAMES.index(corruption_name) if corruption_name in CORRUPTION_NAMES else None
        if cid is None:
            raise KeyError(corruption_name)
    elif corruption_number != -1:
        cid = range(len(CORRUPTION_NAMES))[corruption_number]   # IndexError like tuple indexing
    else:
        raise ValueError("Either corruption_name or corruption_number must be passed")
    dev, kind, orig = _as_device_batch(x)
    if dev.shape[0] == 0:
        return orig
    corrupt_batch_(dev, cid, severity, seed, sample_offset, draws)
    if kind == 'tensor':
        return orig
    host = dev.cpu().numpy()
    if kind == 'image':
        return host[0]
    np.copyto(orig, host)      # in place, same object returned (add_noise_utils.py:27-31)
    return orig
