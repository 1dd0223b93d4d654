"""ctypes binding of the CPU oracle (oracle/_build/libmemgym_oracle.so) -- test infrastructure only."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB_PATH = os.path.join(ORACLE_DIR, "_build", "libmemgym_oracle.so")

_lib = None


def build():
    subprocess.check_call(["make", "-s", "-C", ORACLE_DIR])


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        L = C.CDLL(LIB_PATH)
        L.mgo_create.restype = C.c_void_p
        L.mgo_create.argtypes = [C.c_char_p, C.c_double]
        L.mgo_destroy.argtypes = [C.c_void_p]
        L.mgo_set_option.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(C.c_double), C.c_int]
        L.mgo_is_discrete.argtypes = [C.c_void_p]
        L.mgo_gt_dim.argtypes = [C.c_void_p]
        L.mgo_screen_dim.argtypes = [C.c_void_p]
        L.mgo_reset.argtypes = [C.c_void_p, C.c_int64, C.c_void_p]
        L.mgo_step.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int)]
        L.mgo_get.restype = C.c_double
        L.mgo_get.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(C.c_int)]
        L.mgo_get_list.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.c_int]
        L.mgo_get_gt.argtypes = [C.c_void_p, C.c_void_p]
        L.mgo_rng_words.argtypes = [C.c_void_p, C.c_void_p]
        L.mgo_render_debug.argtypes = [C.c_void_p, C.c_void_p]
        L.mgo_scene.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        L.mgo_test_rng.argtypes = [C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        L.mgo_batch_create.restype = C.c_void_p
        L.mgo_batch_create.argtypes = [C.c_char_p, C.c_int, C.c_double]
        L.mgo_batch_destroy.argtypes = [C.c_void_p]
        L.mgo_batch_env.restype = C.c_void_p
        L.mgo_batch_env.argtypes = [C.c_void_p, C.c_int]
        L.mgo_batch_set_option.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(C.c_double), C.c_int]
        L.mgo_batch_reset.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.mgo_batch_step.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.mgo_batch_expert.argtypes = [C.c_void_p, C.c_double, C.c_uint64, C.c_uint64, C.c_void_p]
        L.mgo_batch_get.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p]
        _lib = L
    return _lib


def _opt_values(v):
    if isinstance(v, (list, tuple, np.ndarray)):
        return [float(x) for x in v]
    return [float(v)]


class OracleEnv:
    """One reference-shaped environment instance (reset(seed, options) / step(action))."""

    def __init__(self, env_id, scale=0.25, _handle=None):
        self.L = lib()
        self.env_id = env_id
        self.scale = scale
        self._owned = _handle is None
        self.h = self.L.mgo_create(env_id.encode(), scale) if _handle is None else _handle
        if not self.h:
            raise ValueError("oracle: unknown env id " + env_id)
        self.dim = self.L.mgo_screen_dim(self.h)
        self.discrete = bool(self.L.mgo_is_discrete(self.h))
        self.gt_dim = self.L.mgo_gt_dim(self.h)

    def close(self):
        if self.h and self._owned:
            self.L.mgo_destroy(self.h)
        self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_options(self, options):
        for k, v in (options or {}).items():
            vals = _opt_values(v)
            arr = (C.c_double * len(vals))(*vals)
            if self.L.mgo_set_option(self.h, k.encode(), arr, len(vals)) != 0:
                raise AssertionError("Provided reset parameter (" + str(k) + ") is not valid. Check spelling.")

    def reset(self, seed=None, options=None, want_obs=True):
        self.set_options(options)
        obs = np.empty((self.dim, self.dim, 3), np.uint8) if want_obs else None
        rc = self.L.mgo_reset(self.h, -1 if seed is None else int(seed), obs.ctypes.data if want_obs else None)
        assert rc == 0
        return obs

    def step(self, action, want_obs=True):
        a = np.atleast_1d(np.asarray(action)).astype(np.int32)
        arr = (C.c_int * 2)(int(a[0]), int(a[1]) if a.size > 1 else 0)
        obs = np.empty((self.dim, self.dim, 3), np.uint8) if want_obs else None
        r, d = C.c_double(), C.c_int()
        self.L.mgo_step(self.h, arr, obs.ctypes.data if want_obs else None, C.byref(r), C.byref(d))
        return obs, r.value, bool(d.value)

    def get(self, field):
        ok = C.c_int()
        v = self.L.mgo_get(self.h, field.encode(), C.byref(ok))
        return v if ok.value else None

    def get_list(self, name, cap=4096):
        buf = np.empty(cap, np.float64)
        n = self.L.mgo_get_list(self.h, name.encode(), buf.ctypes.data, cap)
        return None if n < 0 else buf[:min(n, cap)].copy()

    def gt(self):
        buf = np.zeros(4)
        self.L.mgo_get_gt(self.h, buf.ctypes.data)
        return buf[:self.gt_dim]

    def rng_words(self):
        w = np.zeros(6, np.uint64)
        self.L.mgo_rng_words(self.h, w.ctypes.data)
        return w

    def scene(self, values):
        """Test hook (oracle/mgo_env.h mgo_vtbl.scene): put the instance into the scene described by the family-specific
        vector and draw it with the family's own drawing code; returns the observation [x][y][c]."""
        v = np.ascontiguousarray(values, dtype=np.float64)
        obs = np.empty((self.dim, self.dim, 3), np.uint8)
        assert self.L.mgo_scene(self.h, v.ctypes.data, len(v), obs.ctypes.data) == 0, "scene refused"
        return obs

    def debug_view(self):
        """render() with render_mode "debug_rgb_array": uint8 [336][336][3], image order"""
        out = np.zeros((336, 336, 3), np.uint8)
        assert self.L.mgo_render_debug(self.h, out.ctypes.data) == 0
        return out


class OracleBatch:
    """N independent instances; env i is seeded seeds[i]; optional same-step auto-reset."""

    def __init__(self, env_id, n, scale=0.25, options=None):
        self.L = lib()
        self.n = n
        self.h = self.L.mgo_batch_create(env_id.encode(), n, scale)
        if not self.h:
            raise ValueError("oracle: unknown env id " + env_id)
        self.envs = [OracleEnv(env_id, scale, _handle=self.L.mgo_batch_env(self.h, i)) for i in range(n)]
        self.dim = self.envs[0].dim
        self.discrete = self.envs[0].discrete
        self.set_options(options)

    def set_options(self, options):
        """Like the reference's reset(options=...): the caller passes the complete dictionary (defaults included) if
        earlier values are to be forgotten."""
        for k, v in (options or {}).items():
            vals = _opt_values(v)
            arr = (C.c_double * len(vals))(*vals)
            if self.L.mgo_batch_set_option(self.h, k.encode(), arr, len(vals)) != 0:
                raise AssertionError("Provided reset parameter (" + str(k) + ") is not valid. Check spelling.")

    def close(self):
        if self.h:
            self.L.mgo_batch_destroy(self.h)
            self.h = None

    def expert_actions(self, eps, seed, step, out=None):
        """The test's policy (oracle/mgo_api.c mgo_batch_expert): every instance's competent action for its CURRENT state, a
        uniformly random one with probability eps; int32 [n] or [n, 2]."""
        a = np.empty((self.n,) if self.discrete else (self.n, 2), np.int32) if out is None else out
        assert self.L.mgo_batch_expert(self.h, float(eps), int(seed), int(step), a.ctypes.data) == 0, "this family has no expert"
        return a

    def get_all(self, field):
        """float64 [n]: one state field of every instance (NaN where it does not apply)"""
        out = np.empty(self.n, np.float64)
        self.L.mgo_batch_get(self.h, field.encode(), out.ctypes.data)
        return out

    def reset(self, seeds=None, out=None):
        """`out`: a preallocated uint8 [n, dim, dim, 3] array to write the frames into (no allocation per call)."""
        obs = np.empty((self.n, self.dim, self.dim, 3), np.uint8) if out is None else out
        s = None if seeds is None else np.ascontiguousarray(seeds, dtype=np.int64)
        self.L.mgo_batch_reset(self.h, None if s is None else s.ctypes.data, obs.ctypes.data)
        return obs

    def step(self, actions, autoreset=True, want_obs=True, out=None):
        """`out` = (obs uint8 [n, dim, dim, 3], reward float64 [n], done uint8 [n]): preallocated outputs, nothing is
        allocated per call (bench.py's CPU baseline; a fresh 347-MB frame array per step is mostly page faults)."""
        a = np.ascontiguousarray(actions, dtype=np.int32)
        if out is not None:
            obs, rew, done = out
        else:
            obs = np.empty((self.n, self.dim, self.dim, 3), np.uint8) if want_obs else None
            rew = np.empty(self.n, np.float64)
            done = np.empty(self.n, np.uint8)
        self.L.mgo_batch_step(self.h, a.ctypes.data, int(autoreset), obs.ctypes.data if want_obs else None,
                              rew.ctypes.data, done.ctypes.data)
        return obs, rew, done
