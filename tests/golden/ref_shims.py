"""Functional stand-ins for `pygame` and `gymnasium` so that the *logic* of the reference
package (/root/reference/memory_gym) can be imported and run in the survey container, where
neither third-party dependency is installed.

THIS FILE IS FIXTURE TOOLING, NOT PRODUCT AND NOT ORACLE.  It is only used by
`tests/golden/make_golden.py`, which runs in the build container (where /root/reference exists)
to capture golden logic trajectories.  Nothing on the GPU box imports it.

What is emulated:
  * pygame.Rect            -- integer rect; `center` setter rounds floats half-away-from-zero
                              (evidence: docs/assets/emm_0.gif)
  * pygame.math.Vector2    -- double precision; rotate/lerp/normalize/distance_to as in pygame's C source
  * pygame.Surface/draw/transform/surfarray/display/time/event -- inert (pixels are NOT produced here)
  * gymnasium.Env          -- reset(seed) -> Generator(PCG64(SeedSequence(seed))), lazily created np_random
  * gymnasium.spaces, gymnasium.envs.registration.register, gymnasium.utils.seeding.np_random
"""
import math
import sys
import types

import numpy as np


# ----------------------------------------------------------------------------------------------
# pygame.math.Vector2
# ----------------------------------------------------------------------------------------------
class Vector2:
    __slots__ = ("x", "y")

    def __init__(self, x=0.0, y=None):
        if y is None:
            if isinstance(x, (int, float, np.integer, np.floating)):
                self.x = float(x)
                self.y = float(x)
            else:
                self.x = float(x[0])
                self.y = float(x[1])
        else:
            self.x = float(x)
            self.y = float(y)

    def __len__(self):
        return 2

    def __getitem__(self, i):
        return (self.x, self.y)[i]

    def __iter__(self):
        yield self.x
        yield self.y

    def __add__(self, o):
        return Vector2(self.x + float(o[0]), self.y + float(o[1]))

    __radd__ = __add__

    def __sub__(self, o):
        return Vector2(self.x - float(o[0]), self.y - float(o[1]))

    def __rsub__(self, o):
        return Vector2(float(o[0]) - self.x, float(o[1]) - self.y)

    def __mul__(self, s):
        return Vector2(self.x * float(s), self.y * float(s))

    __rmul__ = __mul__

    def __eq__(self, o):
        try:
            return self.x == o[0] and self.y == o[1]
        except Exception:
            return False

    def __repr__(self):
        return "Vector2(%r, %r)" % (self.x, self.y)

    def length(self):
        return math.sqrt(self.x * self.x + self.y * self.y)

    def normalize(self):
        l = self.length()
        return Vector2(self.x / l, self.y / l)

    def _rot(self, angle):
        # pygame src_c/math.c:_vector2_rotate_helper
        eps = 1e-6
        angle = math.fmod(angle, 360.0)
        if angle < 0:
            angle += 360.0
        if math.fmod(angle + eps, 90.0) < 2 * eps:
            k = int((angle + eps) / 90.0)
            if k in (0, 4):
                return self.x, self.y
            if k == 1:
                return -self.y, self.x
            if k == 2:
                return -self.x, -self.y
            return self.y, -self.x
        rad = angle * math.pi / 180.0
        s, c = math.sin(rad), math.cos(rad)
        return c * self.x - s * self.y, s * self.x + c * self.y

    def rotate(self, angle):
        x, y = self._rot(angle)
        return Vector2(x, y)

    def rotate_ip(self, angle):
        self.x, self.y = self._rot(angle)

    def lerp(self, o, t):
        return Vector2(self.x * (1 - t) + float(o[0]) * t, self.y * (1 - t) + float(o[1]) * t)

    def distance_to(self, o):
        dx = float(o[0]) - self.x
        dy = float(o[1]) - self.y
        return math.sqrt(dx * dx + dy * dy)


# ----------------------------------------------------------------------------------------------
# pygame.Rect
# ----------------------------------------------------------------------------------------------
def _round_haz(v):
    """float -> int, half away from zero (what the pygame build behind the v1.0 GIFs did)."""
    if isinstance(v, (int, np.integer)):
        return int(v)
    v = float(v)
    return int(math.floor(v + 0.5)) if v >= 0 else -int(math.floor(-v + 0.5))


class Rect:
    def __init__(self, *args):
        if len(args) == 1:
            args = tuple(args[0])
        if len(args) == 2:
            args = (args[0][0], args[0][1], args[1][0], args[1][1])
        self.x, self.y, self.w, self.h = (int(a) for a in args)

    def copy(self):
        return Rect(self.x, self.y, self.w, self.h)

    def __getitem__(self, i):
        return (self.x, self.y, self.w, self.h)[i]

    def __iter__(self):
        return iter((self.x, self.y, self.w, self.h))

    def __len__(self):
        return 4

    @property
    def center(self):
        return (self.x + self.w // 2, self.y + self.h // 2)

    @center.setter
    def center(self, v):
        cx, cy = _round_haz(v[0]), _round_haz(v[1])
        self.x += cx - (self.x + (self.w >> 1))
        self.y += cy - (self.y + (self.h >> 1))

    @property
    def topleft(self):
        return (self.x, self.y)

    @property
    def bottomright(self):
        return (self.x + self.w, self.y + self.h)

    @property
    def width(self):
        return self.w

    @property
    def height(self):
        return self.h

    def __repr__(self):
        return "Rect(%d,%d,%d,%d)" % (self.x, self.y, self.w, self.h)


# ----------------------------------------------------------------------------------------------
# pygame.Surface & friends (inert)
# ----------------------------------------------------------------------------------------------
class Surface:
    _display = None

    def __init__(self, size, *a, **k):
        self._w, self._h = int(size[0]), int(size[1])
        self._alpha = 255

    def get_rect(self, **kw):
        r = Rect(0, 0, self._w, self._h)
        if "center" in kw:
            r.center = kw["center"]
        return r

    def get_size(self):
        return (self._w, self._h)

    def get_width(self):
        return self._w

    def get_height(self):
        return self._h

    def fill(self, *a, **k):
        pass

    def set_colorkey(self, *a, **k):
        pass

    def blit(self, *a, **k):
        pass

    def set_alpha(self, v):
        self._alpha = max(0, min(255, int(v)))

    def get_alpha(self):
        return self._alpha


def _rotated_size(w, h, angle):
    if angle % 90 == 0:
        return (w, h) if (angle // 90) % 2 == 0 else (h, w)
    rad = angle * 0.01745329251994329
    sa, ca = math.sin(rad), math.cos(rad)
    cx, cy, sx, sy = ca * w, ca * h, sa * w, sa * h
    nx = int(max(abs(cx + sy), abs(cx - sy), abs(-cx + sy), abs(-cx - sy)))
    ny = int(max(abs(sx + cy), abs(sx - cy), abs(-sx + cy), abs(-sx - cy)))
    return nx, ny


def install():
    """Install fake `pygame` and `gymnasium` packages into sys.modules."""
    pg = types.ModuleType("pygame")
    pg.__path__ = []
    pg.NOFRAME = 0
    pg.init = lambda *a, **k: None
    pg.quit = lambda *a, **k: None
    pg.Surface = Surface
    pg.Rect = Rect

    pgmath = types.ModuleType("pygame.math")
    pgmath.Vector2 = Vector2
    pg.math = pgmath

    disp = types.ModuleType("pygame.display")

    def set_mode(size, *a, **k):
        Surface._display = Surface(size)
        return Surface._display

    disp.set_mode = set_mode
    disp.set_caption = lambda *a, **k: None
    disp.flip = lambda *a, **k: None
    disp.get_surface = lambda: Surface._display
    pg.display = disp

    draw = types.ModuleType("pygame.draw")
    draw.circle = draw.rect = draw.line = lambda *a, **k: None
    pg.draw = draw

    tr = types.ModuleType("pygame.transform")
    tr.rotate = lambda s, angle: Surface(_rotated_size(s._w, s._h, angle))
    tr.scale = lambda s, size: Surface(size)
    pg.transform = tr

    sa = types.ModuleType("pygame.surfarray")
    sa.array3d = lambda s: np.zeros((s._w, s._h, 3), dtype=np.uint8)
    pg.surfarray = sa

    tm = types.ModuleType("pygame.time")

    class Clock:
        def tick(self, *a):
            return 0

    tm.Clock = Clock
    pg.time = tm

    ev = types.ModuleType("pygame.event")
    ev.set_allowed = lambda *a, **k: None
    ev.get = lambda *a, **k: []
    pg.event = ev

    img = types.ModuleType("pygame.image")
    img.save = lambda *a, **k: None
    pg.image = img

    sdl2 = types.ModuleType("pygame._sdl2")
    sdl2.Window = sdl2.Texture = sdl2.Renderer = type("Dummy", (), {})
    pg._sdl2 = sdl2

    for name, mod in [("pygame", pg), ("pygame.math", pgmath), ("pygame.display", disp), ("pygame.draw", draw),
                      ("pygame.transform", tr), ("pygame.surfarray", sa), ("pygame.time", tm), ("pygame.event", ev),
                      ("pygame.image", img), ("pygame._sdl2", sdl2)]:
        sys.modules[name] = mod

    # ------------------------------------------------------------------ gymnasium
    gym = types.ModuleType("gymnasium")
    gym.__path__ = []

    class Env:
        _np_random = None

        def reset(self, seed=None, options=None):
            if seed is not None:
                self._np_random = np.random.Generator(np.random.PCG64(np.random.SeedSequence(seed)))

        @property
        def np_random(self):
            if self._np_random is None:
                self._np_random = np.random.Generator(np.random.PCG64(np.random.SeedSequence()))
            return self._np_random

        @np_random.setter
        def np_random(self, v):
            self._np_random = v

    gym.Env = Env
    spaces = types.ModuleType("gymnasium.spaces")

    class _Space:
        def __init__(self, *a, **k):
            self.args, self.kwargs = a, k

    spaces.Discrete = type("Discrete", (_Space,), {})
    spaces.MultiDiscrete = type("MultiDiscrete", (_Space,), {})
    spaces.Box = type("Box", (_Space,), {})
    spaces.Dict = type("Dict", (_Space,), {})
    gym.spaces = spaces

    envs = types.ModuleType("gymnasium.envs")
    envs.__path__ = []
    reg = types.ModuleType("gymnasium.envs.registration")
    reg.registry = {}

    def register(id, entry_point, **kw):
        reg.registry[id] = entry_point

    reg.register = register
    envs.registration = reg
    gym.envs = envs

    utils = types.ModuleType("gymnasium.utils")
    utils.__path__ = []
    seeding = types.ModuleType("gymnasium.utils.seeding")

    def np_random(seed=None):
        ss = np.random.SeedSequence(seed)
        return np.random.Generator(np.random.PCG64(ss)), ss.entropy

    seeding.np_random = np_random
    utils.seeding = seeding
    gym.utils = utils

    for name, mod in [("gymnasium", gym), ("gymnasium.spaces", spaces), ("gymnasium.envs", envs),
                      ("gymnasium.envs.registration", reg), ("gymnasium.utils", utils),
                      ("gymnasium.utils.seeding", seeding)]:
        sys.modules[name] = mod
    return pg, gym
