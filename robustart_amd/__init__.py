"""robustart_amd -- MI355X-native implementation of RobustART's AddNoise hot path.

Drop-in surface (mirrors RobustART/noise/__init__.py:1 and RobustART/noise/add_noise.py):

    from robustart_amd.noise import AddNoise
    AddNoise('imagenet-c').set_config(corruption_name='gaussian_noise', severity=3)

The arithmetic runs in hand-written HIP kernels behind the C-ABI declared in
include/robustart_hip.h (robustart_amd/lib/librobustart_hip.so).  There is no CPU fallback:
if the library is missing or no GPU is present the product path raises.
"""
__version__ = '0.1.0'
