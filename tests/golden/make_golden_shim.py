"""Golden vectors "pinned modulo shim": the UNMODIFIED reference functions impulse_noise, gaussian_blur, glass_blur, spatter,
brightness, saturate, defocus_blur, motion_blur, snow and elastic_transform (RobustART/noise/utils/imagenet_c/corruptions.py) run
with the scikit-image entry points they call supplied by tests/golden/skimage_shim.py and the OpenCV / ImageMagick entry points by
tests/golden/cv2_wand_shim.py (none of the three libraries is installed).

Run in the build container only:   python tests/golden/make_golden_shim.py
Output: tests/golden/corruptions_shim_ref.npz -- sha256 of the full uint8 output + a 64 x 64 crop per (name, severity),
inputs regenerated from seeds by tests/_inputs.py (glass_blur additionally on three images at severity 3).
What this pins: the reference's own loop order, np.random consumption, thresholds, blends, clipping and the final
np.uint8 cast of corrupt() -- not scikit-image's arithmetic, which the shim restates (see its header).
"""
import hashlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))

sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import skimage_shim  # noqa: E402
import cv2_wand_shim  # noqa: E402

skimage_shim.install(sys.modules)
cv2_wand_shim.install(sys.modules)
if True:                                                  # numpy 2 made the binary mode of fromstring an error-level warning;
    np.fromstring = lambda s, dtype=float, **k: np.frombuffer(s, dtype=dtype)      # noqa: E731  the reference decodes PNG blobs with it
from _ref_import import import_reference_noise  # noqa: E402

import_reference_noise()
from PIL import Image  # noqa: E402
from RobustART.noise.utils.imagenet_c import corrupt as ref_corrupt  # noqa: E402
from _inputs import make_image, case_seed  # noqa: E402

CASES = [('impulse_noise', (1, 2, 3, 4, 5)), ('gaussian_blur', (1, 2, 3, 4, 5)), ('glass_blur', (1, 2, 3, 4, 5)),
         ('spatter', (1, 2, 3, 4, 5)), ('brightness', (1, 2, 3, 4, 5)), ('saturate', (1, 2, 3, 4, 5)),
         # with the OpenCV / ImageMagick calls supplied by cv2_wand_shim.py (which forwards to the oracle's restatements):
         ('defocus_blur', (1, 2, 3, 4, 5)), ('motion_blur', (1, 2, 3, 4, 5)), ('snow', (1, 2, 3, 4, 5)),
         ('elastic_transform', (1, 2, 3, 4, 5))]


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def main():
    out = {}
    for name, sevs in CASES:
        for sev in sevs:
            x = make_image(sev)
            np.random.seed(case_seed(name, sev))
            y = np.asarray(ref_corrupt(Image.fromarray(x), severity=sev, corruption_name=name))
            assert y.dtype == np.uint8 and y.shape == x.shape
            out['%s/%d/sha' % (name, sev)] = np.array(sha(y))
            out['%s/%d/crop' % (name, sev)] = y[80:144, 80:144].copy()
            print(name, sev, out['%s/%d/sha' % (name, sev)])
    for img_seed in (11, 12, 13):                       # glass_blur on three more images (the swap chain is the risky part)
        x = make_image(img_seed)
        np.random.seed(case_seed('glass_blur', 3) + img_seed)
        y = np.asarray(ref_corrupt(Image.fromarray(x), severity=3, corruption_name='glass_blur'))
        out['glass_blur/3/img%d/sha' % img_seed] = np.array(sha(y))
        out['glass_blur/3/img%d/crop' % img_seed] = y[80:144, 80:144].copy()
    np.savez_compressed(os.path.join(HERE, 'corruptions_shim_ref.npz'), **out)
    print('corruptions_shim_ref.npz', len(out), 'entries')


if __name__ == '__main__':
    main()
