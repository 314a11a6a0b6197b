from .add_noise import AddNoise  # noqa: F401  (RobustART/noise/__init__.py:1)
from .rng import manual_seed  # noqa: F401
