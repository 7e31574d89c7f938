"""2048_amd — MI355X-native batched 2048 environment + n-tuple TD(0) learner.

The hot path lives in hand-written HIP kernels (csrc/g2048.hip) behind the C ABI of
include/g2048.h; this package is the thin Python host side that keeps the reference's `Game` /
`QAgent` surface.  The directory name is not a Python identifier: import it with
`importlib.import_module('2048_amd')`, or through the drop-in alias package `game2048`.
"""
from . import rng  # noqa: F401  (pure-Python spec of the device RNG)

__all__ = ['rng', 'Engine', 'load_library']


def load_library():
    from . import _lib
    return _lib.load()


def __getattr__(name):
    if name == 'Engine':
        from .engine import Engine
        return Engine
    raise AttributeError(name)
