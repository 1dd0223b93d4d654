"""ctypes binding of libmemgym_hip.so (C ABI: include/memgym.h).

The HIP library is the product: there is NO CPU or PyTorch fallback -- if the shared object is missing or does
not load, importing this module raises.  `import torch` happens first on purpose: torch ships its own
libamdhip64.so.7 and the dynamic loader must resolve our DT_NEEDED entry to that already-loaded runtime so that
device pointers and streams are shared with torch tensors.
"""
import ctypes as C
import os

import torch  # noqa: F401  (must be loaded before the native library, see above)

PKG_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB_PATH = os.environ.get("MEMGYM_HIP_LIB", os.path.join(PKG_ROOT, "lib", "libmemgym_hip.so"))

MG_INFO_SLOTS = 8
MG_MAX_OPTION_SETS = 8


class InfoBuffers(C.Structure):
    _fields_ = [("struct_size", C.c_size_t), ("ep_reward_dev", C.c_void_p), ("ep_length_dev", C.c_void_p), ("aux_dev", C.c_void_p * MG_INFO_SLOTS),
                ("final_obs_dev", C.c_void_p), ("reward64_dev", C.c_void_p), ("gt64_dev", C.c_void_p), ("capacity_dev", C.c_void_p)]


class SingleIO(C.Structure):  # include/memgym.h: mg_single_io (host addresses of pinned, device-mapped buffers)
    _fields_ = [("struct_size", C.c_size_t), ("obs", C.c_void_p), ("vec", C.c_void_p), ("reward", C.c_void_p), ("done", C.c_void_p),
                ("gt", C.c_void_p), ("ep_reward", C.c_void_p), ("ep_length", C.c_void_p), ("aux", C.c_void_p * MG_INFO_SLOTS)]


class ObsAllocInfo(C.Structure):
    _fields_ = [("zones", C.c_int), ("pieces", C.c_int), ("piece_bytes", C.c_size_t), ("searched_bytes", C.c_size_t),
                ("probe_same_tbps", C.c_double), ("probe_cross_tbps", C.c_double), ("search_ms", C.c_double)]


MG_OBS_SEARCH_DEFAULT = (1 << 64) - 1


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "memory_gym_amd: native library %s not found. Build it with `python __graft_entry__.py` "
            "(hipcc --offload-arch=gfx950). There is no fallback path." % LIB_PATH)
    L = C.CDLL(LIB_PATH)
    L.mg_last_error.restype = C.c_char_p
    L.mg_create.argtypes = [C.c_char_p, C.c_int32, C.c_int, C.POINTER(C.c_void_p)]
    L.mg_destroy.argtypes = [C.c_void_p]
    L.mg_destroy.restype = None
    for f in ("mg_num_envs", "mg_action_dim", "mg_gt_dim"):
        getattr(L, f).argtypes = [C.c_void_p]
        getattr(L, f).restype = C.c_int32
    L.mg_info_name.argtypes = [C.c_void_p, C.c_int]
    L.mg_info_name.restype = C.c_char_p
    L.mg_set_option.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(C.c_double), C.c_int]
    L.mg_vec_dim.argtypes = [C.c_void_p]
    L.mg_vec_dim.restype = C.c_int32
    L.mg_bind_vector_obs.argtypes = [C.c_void_p, C.c_void_p]
    L.mg_set_obs_format.argtypes = [C.c_void_p, C.c_int]
    L.mg_obs_bytes.argtypes = [C.c_void_p]
    L.mg_obs_bytes.restype = C.c_size_t
    L.mg_reset.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.mg_step.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(InfoBuffers),
                          C.c_int, C.c_void_p]
    L.mg_render.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    L.mg_render_debug.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    L.mg_state_size.argtypes = [C.c_void_p]
    L.mg_state_size.restype = C.c_size_t
    L.mg_get_state.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    L.mg_set_state.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    L.mg_set_profiling.argtypes = [C.c_void_p, C.c_int]
    L.mg_get_profile.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_int64)]
    L.mg_poll_errors.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
    L.mg_peek_errors.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
    L.mg_obs_alloc.argtypes = [C.c_int, C.c_size_t, C.c_size_t, C.POINTER(C.c_void_p), C.POINTER(ObsAllocInfo)]
    if hasattr(L, "mg_obs_alloc_for"):  # round 6 (absent from the older builds tools/ A/B against through MEMGYM_HIP_LIB)
        L.mg_obs_alloc_for.argtypes = [C.c_int, C.c_size_t, C.c_size_t, C.c_size_t, C.POINTER(C.c_void_p), C.POINTER(ObsAllocInfo)]
        L.mg_obs_plan.argtypes = [C.c_size_t, C.c_size_t, C.c_int, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t), C.POINTER(C.c_int), C.c_int]
    L.mg_obs_free.argtypes = [C.c_void_p]
    L.mg_obs_debug_stats.argtypes = [C.POINTER(C.c_size_t), C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
    if hasattr(L, "mg_single_open"):  # round 5
        L.mg_ground_truth64.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.mg_single_open.argtypes = [C.c_void_p, C.POINTER(SingleIO)]
        L.mg_single_reset.argtypes = [C.c_void_p, C.c_int64, C.c_int, C.c_void_p]
        L.mg_single_step.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]
    if hasattr(L, "mg_obs_set_search_ms"):  # (absent from builds of earlier rounds that tools/ A/B against through MEMGYM_HIP_LIB)
        L.mg_obs_set_search_ms.argtypes = [C.c_double]
        L.mg_store_probe.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
    L.mg_enable_peer_access.argtypes = [C.c_int, C.c_int]
    L.mg_debug_rng.argtypes = [C.c_void_p, C.c_int32, C.c_void_p]
    if hasattr(L, "mg_set_option_set"):
        L.mg_set_option_set.argtypes = [C.c_void_p, C.c_int, C.c_char_p, C.POINTER(C.c_double), C.c_int]
        L.mg_bind_option_sets.argtypes = [C.c_void_p, C.c_void_p]
    L.mg_set_capacity.argtypes = [C.c_void_p, C.c_char_p, C.c_int64]
    L.mg_capacity.argtypes = [C.c_void_p, C.c_char_p]
    L.mg_capacity.restype = C.c_int64
    if hasattr(L, "mg_debug_counter"):  # (absent from builds of earlier rounds that tools/ A/B against through MEMGYM_HIP_LIB)
        L.mg_debug_counter.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(C.c_int64)]
    return L


LIB = _load()


def last_error():
    return (LIB.mg_last_error() or b"").decode()


def check(rc, what):
    if rc == -3:  # an option value this build does not support (include/memgym.h), e.g. found when the geometry is rebuilt
        raise NotImplementedError("%s: %s" % (what, last_error()))
    if rc != 0:
        raise RuntimeError("%s failed (%d): %s" % (what, rc, last_error()))
