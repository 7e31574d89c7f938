"""A headless stand-in for `pygame` / `pygame.locals`, just wide enough for the reference's show.py (show.py:1-2, 34-149).

It draws nothing and RECORDS what show.py asks it to draw: every `font.render(text)` call.  One `Show.display()` call is one
frame = one status message ("score = ...", "score ..., moves ..., now = left", "Over! score ...") followed by sixteen cell
numbers in the order show.py paints them (column by column: `self.game.row[j, i]` for i, then j — show.py:53-56).
`event.get()` stays empty until an "Over!" frame has been drawn, then delivers one QUIT, which is how show.py's endless
closing loops are left (show.py:120-125, 143-148).

Test infrastructure only.  tests/golden/make_show_transcript.py registers it as `pygame` and runs the reference's own show.py
on top of this repository's `game2048` package.
"""
import re
import sys
import types

QUIT, KEYDOWN = 12, 2
K_LEFT, K_UP, K_RIGHT, K_DOWN, K_r = 276, 273, 275, 274, 114


class Recorder:
    def __init__(self):
        self.texts = []             # every rendered string, in order
        self.over_drawn = False
        self.quits_sent = 0
        self.waits = []

    def reset(self):
        self.__init__()

    def frames(self):
        """[(score, moves, move or -1, over 0/1, sixteen face values in ROW-MAJOR order)] parsed from the rendered strings."""
        out, i, t = [], 0, self.texts
        names = {'left': 0, 'up': 1, 'right': 2, 'down': 3}
        while i < len(t):
            msg = t[i]
            m = (re.fullmatch(r'Over! score (-?\d+), moves (\d+)', msg) or re.fullmatch(r'score = (-?\d+) after (\d+) moves', msg)
                 or re.fullmatch(r'score (-?\d+), moves (\d+), now = (\w+)', msg))
            assert m, f'not a status message: {msg!r}'
            cells = [int(x) for x in t[i + 1:i + 17]]
            assert len(cells) == 16
            board = [cells[c * 4 + r] for r in range(4) for c in range(4)]       # painted column by column -> row-major
            move = names[m.group(3)] if m.lastindex == 3 else -1
            out.append([int(m.group(1)), int(m.group(2)), move, int(msg.startswith('Over!'))] + board)
            i += 17
        return out


REC = Recorder()


class _Surface:
    def fill(self, colour):
        pass

    def blit(self, what, where):
        pass


class _Font:
    def render(self, text, antialias, colour):
        REC.texts.append(text)
        if text.startswith('Over!'):
            REC.over_drawn = True
        return _Surface()


class _Event:
    def __init__(self, type_):
        self.type, self.key = type_, None


def _event_get():
    if REC.over_drawn:
        REC.quits_sent += 1
        return [_Event(QUIT)]
    return []


def install():
    """Register the stub as `pygame` and `pygame.locals`; returns the recorder."""
    pg = types.ModuleType('pygame')
    pg.init = lambda: None
    pg.quit = lambda: None
    pg.QUIT, pg.KEYDOWN = QUIT, KEYDOWN
    pg.K_LEFT, pg.K_UP, pg.K_RIGHT, pg.K_DOWN, pg.K_r = K_LEFT, K_UP, K_RIGHT, K_DOWN, K_r
    pg.display = types.SimpleNamespace(set_caption=lambda s: None, set_mode=lambda size, flags=0, depth=0: _Surface(), update=lambda: None)
    pg.font = types.SimpleNamespace(SysFont=lambda name, size: _Font())
    pg.draw = types.SimpleNamespace(rect=lambda surface, colour, rect: None)
    pg.event = types.SimpleNamespace(get=_event_get)
    pg.time = types.SimpleNamespace(wait=lambda ms: REC.waits.append(ms))
    loc = types.ModuleType('pygame.locals')
    loc.QUIT, loc.KEYDOWN = QUIT, KEYDOWN
    loc.__all__ = ['QUIT', 'KEYDOWN']
    pg.locals = loc
    sys.modules['pygame'] = pg
    sys.modules['pygame.locals'] = loc
    return REC
