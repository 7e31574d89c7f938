"""game2048.game_logic — drop-in for the reference's game2048/game_logic.py (implemented in 2048_amd/game.py)."""
from ._alias import reexport
reexport('game', globals(), claim=('Game',))
