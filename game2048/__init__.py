"""Drop-in alias: `from game2048.r_learning import *` (what the reference's show.py does) resolves to the
MI355X-native implementation in 2048_amd/."""
