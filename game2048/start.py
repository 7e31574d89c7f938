"""game2048.start — drop-in for the reference's game2048/start.py (implemented in 2048_amd/start.py)."""
from ._alias import reexport
reexport('start', globals())
