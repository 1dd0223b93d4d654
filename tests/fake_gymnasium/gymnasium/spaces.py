"""TEST INFRASTRUCTURE (see gymnasium/__init__.py of this stand-in): the four space types the reference uses."""
import numpy as np


class Space:
    def contains(self, x):
        raise NotImplementedError

    def __contains__(self, x):
        return self.contains(x)


class Discrete(Space):
    def __init__(self, n):
        self.n = int(n)
        self.shape = ()
        self.dtype = np.int64

    def contains(self, x):
        if isinstance(x, (np.generic, np.ndarray)) and np.asarray(x).shape == () and np.issubdtype(np.asarray(x).dtype, np.integer):
            x = int(x)
        return isinstance(x, int) and 0 <= x < self.n


class MultiDiscrete(Space):
    def __init__(self, nvec):
        self.nvec = np.asarray(nvec, dtype=np.int64)
        self.shape = self.nvec.shape
        self.dtype = np.int64

    def contains(self, x):
        x = np.asarray(x)
        return x.shape == self.shape and np.issubdtype(x.dtype, np.integer) and bool(((0 <= x) & (x < self.nvec)).all())


class Box(Space):
    def __init__(self, low, high, shape=None, dtype=np.float32):
        self.dtype = np.dtype(dtype)
        self.shape = tuple(shape) if shape is not None else np.asarray(low).shape
        self.low = np.broadcast_to(np.asarray(low, dtype=self.dtype), self.shape)
        self.high = np.broadcast_to(np.asarray(high, dtype=self.dtype), self.shape)

    def contains(self, x):
        if not isinstance(x, np.ndarray):
            return False
        return x.shape == self.shape and np.can_cast(x.dtype, self.dtype) and bool((x >= self.low).all() and (x <= self.high).all())


class Dict(Space):
    def __init__(self, spaces):
        self.spaces = dict(spaces)

    def contains(self, x):
        return isinstance(x, dict) and x.keys() == self.spaces.keys() and all(self.spaces[k].contains(v) for k, v in x.items())

    def __getitem__(self, k):
        return self.spaces[k]
