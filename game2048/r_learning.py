"""game2048.r_learning — drop-in for the reference's game2048/r_learning.py (implemented in 2048_amd/agent.py);
`from game2048.r_learning import *` is what the reference's show.py does (show.py:4)."""
from . import game_logic as _gl  # noqa: F401  (Game claims its module path first)
from ._alias import reexport
# (a pickled agent names its `features` function too: f_n must resolve to game2048.r_learning.f_n for the reference to load it)
reexport('agent', globals(), claim=('QAgent', 'f_2', 'f_3', 'f_4', 'f_5', 'f_6'))
