"""Deterministic synthetic inputs shared by the golden generator and the tests."""
import zlib

import numpy as np

# the 8 corruptions whose reference code runs in the build container without missing wheels
RUNNABLE = ('gaussian_noise', 'shot_noise', 'speckle_noise', 'contrast', 'fog', 'zoom_blur',
            'pixelate', 'jpeg_compression')


def make_image(seed, h=224, w=224):
    """Smooth structure + noise, uint8 HxWx3 (exercises blur/JPEG paths better than white noise)."""
    rs = np.random.RandomState(seed)
    yy, xx = np.mgrid[0:h, 0:w]
    base = np.stack([127 + 100 * np.sin(xx / 17. + seed), 127 + 100 * np.cos(yy / 23.),
                     (xx + yy) / float(h + w) * 255], -1)
    img = base + rs.normal(0, 20, base.shape)
    return np.clip(img, 0, 255).astype(np.uint8)


def make_batch_u8(n, seed=1234, h=224, w=224):
    return np.stack([make_image(seed + i, h, w) for i in range(n)])


def case_seed(name, severity):
    return (zlib.crc32(name.encode()) + severity) % (2 ** 31)
