"""TEST INFRASTRUCTURE: a minimal stand-in for gymnasium 0.29 (the version the reference pins, setup.py:35), written for
tests/test_gymnasium_protocol.py because gymnasium itself is not installed on the build or GPU boxes.  It restates the
SEQUENCE that `gymnasium.make(id)` runs on an environment -- resolve the `module:Class` entry point, construct, set
`env.unwrapped.spec`, wrap in a passive checker (space / type assertions on reset and step) and an order enforcer (step
before reset raises) -- and the small part of `Env`, `Wrapper`, `spaces` and `vector.VectorEnv` that sequence touches.
Nothing here is used by the product."""
from . import spaces  # noqa: F401
from .envs.registration import make, register, registry, spec  # noqa: F401

__version__ = "0.29.1-standin"


from .core import Env, Wrapper  # noqa: E402,F401
from . import vector  # noqa: E402,F401
