"""Re-export a module of the 2048_amd package (whose name is not a Python identifier) under game2048.*"""
import importlib
import os
import sys

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if _ROOT not in sys.path:
    sys.path.insert(0, _ROOT)


def reexport(target, into, claim=()):
    mod = importlib.import_module(f'2048_amd.{target}')
    names = getattr(mod, '__all__', [n for n in dir(mod) if not n.startswith('_')])
    for n in names:
        into[n] = getattr(mod, n)
    for cls in claim:                      # pickles name classes by module path: keep the reference's
        getattr(mod, cls).__module__ = into['__name__']
    return mod
