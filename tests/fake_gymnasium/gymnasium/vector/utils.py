"""TEST INFRASTRUCTURE: batch_space for the four space types of the stand-in."""
import numpy as np

from .. import spaces


def batch_space(space, n=1):
    if isinstance(space, spaces.Box):
        return spaces.Box(np.stack([space.low] * n), np.stack([space.high] * n), dtype=space.dtype)
    if isinstance(space, spaces.Discrete):
        return spaces.MultiDiscrete(np.full((n,), space.n))
    if isinstance(space, spaces.MultiDiscrete):
        return spaces.MultiDiscrete(np.stack([space.nvec] * n))
    if isinstance(space, spaces.Dict):
        return spaces.Dict({k: batch_space(v, n) for k, v in space.spaces.items()})
    raise TypeError(space)
