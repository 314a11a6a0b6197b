"""Import the read-only reference package (/root/reference) with stubs for the
third-party wheels that are absent in this container.

Only ever used by tests/golden/make_golden.py, which runs in the build container
to emit the committed fixtures. Nothing in tests/, bench.py or smoke() imports
this at run time: /root/reference does not exist on the GPU box.
"""
import sys
import types
import unittest.mock

REFERENCE_ROOT = "/root/reference"


def import_reference_noise():
    import numpy
    import torch

    sys.dont_write_bytecode = True
    stub = ['skimage', 'skimage.filters', 'wand', 'wand.image', 'wand.api', 'wand.color', 'cv2',
            'foolbox', 'art', 'art.estimators', 'art.estimators.classification', 'art.attacks',
            'art.attacks.evasion', 'torchvision', 'torchvision.datasets', 'torchvision.transforms',
            'ffmpeg']
    for m in stub:
        if m not in sys.modules:
            sys.modules[m] = unittest.mock.MagicMock()
    if isinstance(sys.modules['wand.image'], unittest.mock.MagicMock):      # (a shim installed before this call keeps its Image class)
        sys.modules['wand.image'].Image = type('W', (), {})
    if not hasattr(numpy, 'float_'):
        numpy.float_ = numpy.float64
    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.cuda.random.manual_seed = lambda s: None
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    import RobustART.noise as ref_noise  # noqa
    return ref_noise
