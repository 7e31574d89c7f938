"""game2048.r_learning — drop-in for the reference's game2048/r_learning.py (implemented in 2048_amd/agent.py);
`from game2048.r_learning import *` is what the reference's show.py does (show.py:4)."""
from . import game_logic as _gl  # noqa: F401  (Game claims its module path first)
from ._alias import reexport
reexport('agent', globals(), claim=('QAgent',))
