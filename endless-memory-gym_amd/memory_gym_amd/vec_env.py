"""Batched, device-resident Memory Gym environments behind the reference's env ids and reset-options dict.

`VecMemoryGym(env_id, num_envs, device)` holds `num_envs` independent instances of one of the reference's
environments (memory_gym/__init__.py:13-61) on one MI355X.  reset()/step() enqueue HIP kernels on torch's current
stream and return torch tensors that alias the kernels' output buffers (no copies, no host synchronisation):

    obs      uint8  [N, 84, 84, 3]  indexed [env][x][y][c], exactly the reference's array3d layout
    reward   float32 [N]
    done     bool   [N]             (`truncation` is always False in the reference)
    info     dict of tensors: "done_mask", end-of-episode records ("reward", "length", "success", ...) that are valid
             where done_mask is set, and "ground_truth" [N, G] for the endless envs.

`MemoryGymEnv` is the num_envs == 1 adapter with the reference's exact single-instance signature
(numpy obs, Python float reward, bool done, False, dict info).
"""
import ctypes as C
import os
import weakref

import numpy as np
import torch

from . import _native
from .reset_params import DEFAULTS, calc_max_episode_steps, process_reset_params

ENV_IDS = list(DEFAULTS.keys())


class _Space:
    def __init__(self, kind, **kw):
        self.kind = kind
        self.__dict__.update(kw)

    def __repr__(self):
        return "%s(%s)" % (self.kind, ", ".join("%s=%r" % kv for kv in self.__dict__.items() if kv[0] != "kind"))


def _spaces(action_dim, gt_dim, vec_dim=0):
    try:  # use real gymnasium spaces when the host has them
        from gymnasium import spaces
        act = spaces.Discrete(4) if action_dim == 1 else spaces.MultiDiscrete([3, 3])
        obs = spaces.Box(low=0, high=255, shape=[84, 84, 3], dtype=np.uint8)
        if vec_dim:  # MortarMayhemB*: mortar_mayhem_b_grid.py:83-96
            obs = spaces.Dict({"visual_observation": obs,
                               "vector_observation": spaces.Box(low=np.zeros(vec_dim, np.float32), high=np.ones(vec_dim, np.float32),
                                                                shape=(vec_dim,), dtype=np.float32)})
        gt = None if gt_dim == 0 else spaces.Box(low=np.zeros(gt_dim, np.float32), high=np.ones(gt_dim, np.float32),
                                                  shape=(gt_dim,), dtype=np.float32)
    except Exception:
        act = _Space("Discrete", n=4) if action_dim == 1 else _Space("MultiDiscrete", nvec=[3, 3])
        obs = _Space("Box", low=0, high=255, shape=[84, 84, 3], dtype=np.uint8)
        if vec_dim:
            obs = _Space("Dict", spaces={"visual_observation": obs, "vector_observation": _Space(
                "Box", low=0.0, high=1.0, shape=(vec_dim,), dtype=np.float32)})
        gt = None if gt_dim == 0 else _Space("Box", low=0.0, high=1.0, shape=(gt_dim,), dtype=np.float32)
    return act, obs, gt


_BALANCED = weakref.WeakSet()  # live mg_obs_alloc buffers of this process


def is_balanced_buffer(tensor):
    """True if `tensor` lies in memory obtained from mg_obs_alloc (HIP virtual memory: physical pieces mapped into a reserved
    range).  Such memory has no IPC handle (hipIpcGetMemHandle needs a hipMalloc allocation): memory_gym_amd.dist.PeerObsBuffer
    refuses to export it; collectives (RCCL) and peer access work on it like on any device memory."""
    p = tensor.data_ptr()
    return any(o.ptr is not None and o.ptr <= p < o.ptr + o.nbytes for o in _BALANCED)


class _ObsMemory:
    """Owner of a device buffer obtained from mg_obs_alloc; tensors made from it keep it alive (CUDA array interface)."""

    def __init__(self, ptr, shape, typestr, info):
        self.ptr, self.info = ptr, info
        self.nbytes = int(np.prod(shape))
        _BALANCED.add(self)
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": typestr, "data": (ptr, False), "version": 2, "strides": None}

    def __del__(self):
        try:
            if self.ptr:
                _native.LIB.mg_obs_free(C.c_void_p(self.ptr))
                self.ptr = None
        except Exception:
            pass


def alloc_obs_buffer(shape, dtype, device, search_budget_bytes=None, frame_bytes=None):
    """A tensor of `shape` / `dtype` on `device` whose memory comes from mg_obs_alloc_for (pieces from two HBM zones, see
    include/memgym.h); returns (tensor, info dict).  The memory is released when the last tensor viewing it goes away.
    `frame_bytes`: bytes of one instance's observation (default: what `shape[1:]` and `dtype` say) -- the pieces are dealt to the zones
    by the raster launch's windows of 14,336 observations.
    MEMGYM_OBS_SEARCH_MS bounds the search in time (default 1,500 ms), MEMGYM_OBS_SEARCH_GB its transient filler allocations
    (default: half of the free memory, at
    most 128 GiB -- VRAM that other processes on the GPU cannot have for the few milliseconds the search lasts; 0 = do not
    search).  The buffer is accessible from this device and from every device with peer access to it."""
    device = torch.device(device)
    elem = torch.empty((), dtype=dtype).element_size()
    nbytes = int(np.prod(shape)) * elem
    if search_budget_bytes is None:
        e = os.environ.get("MEMGYM_OBS_SEARCH_GB")
        search_budget_bytes = _native.MG_OBS_SEARCH_DEFAULT if e is None else int(float(e) * (1 << 30))
    ms = os.environ.get("MEMGYM_OBS_SEARCH_MS")  # time bound of the search (default 1,500 ms; include/memgym.h: mg_obs_set_search_ms)
    if ms is not None and hasattr(_native.LIB, "mg_obs_set_search_ms"):
        _native.check(_native.LIB.mg_obs_set_search_ms(float(ms)), "mg_obs_set_search_ms")
    ptr, info = C.c_void_p(), _native.ObsAllocInfo()
    with torch.cuda.device(device):
        torch.cuda.current_stream().synchronize()
        if frame_bytes is None:
            frame_bytes = int(np.prod(shape[1:])) * elem if len(shape) > 1 else 84 * 84 * 3
        lab = os.environ.get("MEMGYM_OBS_FRAME_BYTES")  # tools/fmt_ab.sh: the piece order of another format (A/B of the window rule)
        if lab:
            frame_bytes = int(lab)
        if hasattr(_native.LIB, "mg_obs_alloc_for"):
            _native.check(_native.LIB.mg_obs_alloc_for(device.index or 0, nbytes, C.c_size_t(frame_bytes), C.c_size_t(search_budget_bytes), C.byref(ptr),
                                                       C.byref(info)), "mg_obs_alloc_for")
        else:  # (a build of an earlier round, loaded through MEMGYM_HIP_LIB by tools/)
            _native.check(_native.LIB.mg_obs_alloc(device.index or 0, nbytes, C.c_size_t(search_budget_bytes), C.byref(ptr), C.byref(info)), "mg_obs_alloc")
    # bfloat16 has no array-interface typestr: expose the bytes and view them
    owner = _ObsMemory(ptr.value, (nbytes,), "|u1", {k: getattr(info, k) for k, _ in info._fields_})
    t = torch.as_tensor(owner, device=device).view(dtype).view(tuple(shape))
    return t, owner.info


class VecMemoryGym:
    metadata = {"render_modes": ["rgb_array", "debug_rgb_array"], "render_fps": 25}

    OBS_FORMATS = {"u8_xyc": (0, torch.uint8, (84, 84, 3)), "f32_chw": (1, torch.float32, (3, 84, 84)),
                   "f16_chw": (2, torch.float16, (3, 84, 84)), "bf16_chw": (3, torch.bfloat16, (3, 84, 84))}

    def __init__(self, env_id, num_envs=1, device=None, render_mode=None, obs_format="u8_xyc", final_observation=False,
                 obs_buffer=None, obs_placement=None, ground_truth64=False, on_capacity="raise", capacity=None):
        if env_id not in DEFAULTS:
            raise ValueError("unknown env id %r" % (env_id,))
        if obs_format not in self.OBS_FORMATS:
            raise ValueError("obs_format must be one of %s" % sorted(self.OBS_FORMATS))
        if not torch.cuda.is_available():
            raise RuntimeError("memory_gym_amd needs a ROCm GPU (MI355X); no CPU fallback exists")
        self.env_id = env_id
        self.num_envs = int(num_envs)
        self.device = torch.device("cuda", torch.cuda.current_device() if device is None else
                                   (device if isinstance(device, int) else torch.device(device).index or 0))
        self.render_mode = render_mode
        h = C.c_void_p()
        _native.check(_native.LIB.mg_create(env_id.encode(), self.num_envs, self.device.index, C.byref(h)), "mg_create")
        self._h = h
        # capacity={"path_segments": 1024} / {"commands": 2048}: room for the lists the reference grows without limit (include/memgym.h:
        # mg_set_capacity; the defaults -- 128 segments = 1,024 tiles, 512 commands -- are what every measured number of this repo uses)
        self.capacity = {}
        for what, value in (capacity or {}).items():
            with torch.cuda.device(self.device):
                rc = _native.LIB.mg_set_capacity(h, what.encode(), int(value))
            if rc != 0:
                raise ValueError("capacity[%r] = %r: %s" % (what, value, _native.last_error()))
            self.capacity[what] = int(value)
        self.action_dim = _native.LIB.mg_action_dim(h)
        self.gt_dim = _native.LIB.mg_gt_dim(h)
        self.vec_dim = _native.LIB.mg_vec_dim(h)
        self.has_ground_truth_info = self.gt_dim > 0
        self.action_space, self.observation_space, self.ground_truth_space = _spaces(self.action_dim, self.gt_dim, self.vec_dim)
        N, dev = self.num_envs, self.device
        # "u8_xyc" is the reference's observation; "f32_chw"/"f16_chw" are obs/255 in [c][y][x] order, converted inside
        # the raster kernel's stream-out (what a trainer would otherwise compute from the uint8 frame every step)
        self.obs_format = obs_format
        code, dt, shape = self.OBS_FORMATS[obs_format]
        _native.check(_native.LIB.mg_set_obs_format(h, code), "mg_set_obs_format")
        assert _native.LIB.mg_obs_bytes(h) == 84 * 84 * 3 * torch.empty((), dtype=dt).element_size()
        self.obs_placement_info = None
        if obs_placement is None:
            obs_placement = os.environ.get("MEMGYM_OBS_PLACEMENT", "balanced")
        if obs_placement not in ("balanced", "plain"):
            raise ValueError("obs_placement must be 'balanced' or 'plain'")
        if obs_buffer is None:
            # "balanced": physical pages from two HBM zones (include/memgym.h: mg_obs_alloc; 12-15 % on the raster kernel);
            # buffers of 304 MiB or less and "plain" are ordinary allocations
            nbytes = N * 84 * 84 * 3 * torch.empty((), dtype=dt).element_size()
            self.obs = None
            if obs_placement == "balanced" and nbytes > (304 << 20):
                try:
                    self.obs, self.obs_placement_info = alloc_obs_buffer((N,) + shape, dt, dev)
                except RuntimeError as e:  # e.g. out of memory during the search: an ordinary allocation works as well
                    import warnings
                    warnings.warn("memory_gym_amd: balanced observation buffer not available (%s); using a plain allocation" % (e,))
            if self.obs is None:
                self.obs = torch.empty((N,) + shape, dtype=dt, device=dev)
        else:
            # caller-owned observation memory: the raster kernel writes the frames there (any device-accessible address,
            # e.g. this rank's rows of a peer-mapped buffer on another GPU: memory_gym_amd.dist.PeerObsBuffer)
            if tuple(obs_buffer.shape) != (N,) + shape or obs_buffer.dtype != dt or not obs_buffer.is_cuda or not obs_buffer.is_contiguous():
                raise ValueError("obs_buffer must be a contiguous CUDA tensor of shape %s and dtype %s" % ((N,) + shape, dt))
            self.obs = obs_buffer
        # gymnasium-0.29 vector convention: keep the terminal frame of instances that finish (and auto-reset) in a step
        self.final_obs = torch.zeros((N,) + shape, dtype=dt, device=dev) if final_observation else None
        # MortarMayhemB*: obs is the reference's Dict; `vector_obs` is written by the library whenever an instance resets
        self.vector_obs = None
        if self.vec_dim:
            self.vector_obs = torch.zeros((N, self.vec_dim), dtype=torch.float32, device=dev)
            _native.check(_native.LIB.mg_bind_vector_obs(h, self.vector_obs.data_ptr()), "mg_bind_vector_obs")
        self.reward = torch.zeros(N, dtype=torch.float32, device=dev)
        self.reward64 = torch.zeros(N, dtype=torch.float64, device=dev)  # the reference's Python float, unrounded
        self.done_u8 = torch.zeros(N, dtype=torch.uint8, device=dev)
        self.gt = torch.zeros((N, max(self.gt_dim, 1)), dtype=torch.float32, device=dev)
        # ground_truth64=True: info["ground_truth"] is the reference's float64 array (endless_mortar_mayhem.py:259,358: 0.6, not
        # float32's 0.60000002), computed by one more small launch per call (include/memgym.h: gt64_dev); default: float32
        self.gt64 = torch.zeros((N, max(self.gt_dim, 1)), dtype=torch.float64, device=dev) if (ground_truth64 and self.gt_dim) else None
        self.ep_reward = torch.zeros(N, dtype=torch.float64, device=dev)
        self.ep_length = torch.zeros(N, dtype=torch.int32, device=dev)
        self.info_names = []
        for k in range(_native.MG_INFO_SLOTS):
            nm = _native.LIB.mg_info_name(h, k)
            if nm is None:
                break
            self.info_names.append(nm.decode())
        self.aux = [torch.zeros(N, dtype=torch.float32, device=dev) for _ in self.info_names]
        self._info = _native.InfoBuffers()
        self._info.struct_size = C.sizeof(_native.InfoBuffers)
        self._err = C.c_int(0)
        self._info.ep_reward_dev = self.ep_reward.data_ptr()
        self._info.ep_length_dev = self.ep_length.data_ptr()
        for k, t in enumerate(self.aux):
            self._info.aux_dev[k] = t.data_ptr()
        self._info.final_obs_dev = self.final_obs.data_ptr() if self.final_obs is not None else None
        self._info.reward64_dev = self.reward64.data_ptr()
        self._info.gt64_dev = self.gt64.data_ptr() if self.gt64 is not None else None
        # What happens when ONE instance reaches a capacity of this build (128 path segments, 128 fall-off cells, 512 commands, 16 live
        # spotlights; the reference's lists are unbounded).  The kernels end that instance's episode and raise a sticky error bit either
        # way.  "raise" (default): step() turns the bit into a RuntimeError -- nothing that differs from the reference goes unnoticed.
        # "truncate": step() reports the instance as truncated (`truncated[i]`, info["capacity_exceeded"][i]) and the batch goes on: one
        # instance of 262,144 that outlives a list does not abort a rollout (missing #1).
        if on_capacity not in ("raise", "truncate"):
            raise ValueError("on_capacity must be 'raise' or 'truncate'")
        self.on_capacity = on_capacity
        self.capacity_u8 = torch.zeros(N, dtype=torch.uint8, device=dev) if on_capacity == "truncate" else None
        self._info.capacity_dev = self.capacity_u8.data_ptr() if self.capacity_u8 is not None else None
        self._tolerated = self.CAPACITY_BITS if on_capacity == "truncate" else 0
        self.reset_params = process_reset_params(env_id, None)
        self._applied = dict(DEFAULTS[env_id])
        self._set_params = [self._applied]  # what each option set of the handle holds (set 0 = the handle-wide one)
        self._set_of = None                 # int32 [N]: the set instance i runs under, once a masked reset has used other options
        self.max_episode_steps = None
        self.autoreset = True
        self._seeded = False  # no instance has an RNG stream before the first reset (or load_state_dict)
        self._swapped = False  # use_obs_buffer() since the last call that wrote every row
        # `truncation` is always False in the reference; on_capacity="truncate": True where the episode ended on a capacity of this build
        self._truncated = self.capacity_u8.view(torch.bool) if self.capacity_u8 is not None else torch.zeros(N, dtype=torch.bool, device=dev)
        self._n_actions = self.num_envs * self.action_dim
        self._p_err = C.byref(self._err)
        self._bind_step()

    # ------------------------------------------------------------------ plumbing
    def _stream(self):
        return C.c_void_p(self._raw_stream())

    def _raw_stream(self):
        """torch's CURRENT stream of the handle's device as a raw hipStream_t (it is looked up at every call: the caller may be inside
        `with torch.cuda.stream(...)`)."""
        try:
            return torch._C._cuda_getCurrentRawStream(self.device.index)  # (~0.2 us; the public accessor builds a Stream object: ~2 us)
        except AttributeError:
            return torch.cuda.current_stream(self.device).cuda_stream

    def _bind_step(self):
        """The arguments of mg_step that do not change from call to call, resolved once (pointers of the output tensors, the info
        dictionary): at a few thousand instances a step lasts 20-30 us on the GPU and the host side of step() was as long
        (round 5).  use_obs_buffer / use_step_buffers call it again."""
        self._p_reward = self.reward.data_ptr()
        self._p_done = self.done_u8.data_ptr()
        self._p_gt = self.gt.data_ptr() if self.gt_dim else None
        self._p_info = C.byref(self._info)
        self._done_bool = self.done_u8.view(torch.bool)
        info = {"done_mask": self._done_bool, "reward": self.ep_reward, "length": self.ep_length}
        for nm, t in zip(self.info_names, self.aux):
            info[nm] = t
        if self.gt_dim:
            info["ground_truth"] = self.gt if self.gt64 is None else self.gt64
        if self.capacity_u8 is not None:
            info["capacity_exceeded"] = self._truncated
        self._info_dict = info

    def _write_set(self, set_id, params):
        """Bring option set `set_id` of the handle to `params` (only the keys that differ from what it holds are sent)."""
        have = self._set_params[set_id]
        for k, v in params.items():
            if k in have and have[k] == v:
                continue
            vals = [float(x) for x in v] if isinstance(v, (list, tuple, np.ndarray)) else [float(v)]
            arr = (C.c_double * len(vals))(*vals)
            if set_id == 0:
                rc = _native.LIB.mg_set_option(self._h, k.encode(), arr, len(vals))
            else:
                rc = _native.LIB.mg_set_option_set(self._h, set_id, k.encode(), arr, len(vals))
            if rc == -2:
                raise AssertionError("Provided reset parameter (" + str(k) + ") is not valid. Check spelling.")
            if rc == -4:
                raise AssertionError(_native.last_error())
            if rc != 0:
                raise NotImplementedError("reset parameter %s=%r%s: %s" % (k, v, "" if set_id == 0 else " for a subset of the instances", _native.last_error()))
            have[k] = v

    def _apply_options_masked(self, options, mask):
        """reset(options=..., mask=...): like the reference's reset(seed, options) of ONE instance (mortar_mayhem_grid.py:213-236),
        the options belong to the instances that are being reset and to nobody else.  The handle keeps up to
        MG_MAX_OPTION_SETS parameter sets (include/memgym.h: mg_set_option_set); the masked instances are moved to the set that
        holds these options (a free one is filled if none does).  options=None keeps every masked instance under the options it
        has (a masked reset is this library's extension -- the re-start of finished instances; it is reset(mask=None, options=None)
        that means "the defaults for everybody", like the reference's)."""
        if options is None:
            return
        params = process_reset_params(self.env_id, options)
        k = next((j for j, p in enumerate(self._set_params) if p == params), None)
        if k == 0 and self._set_of is None:
            return  # everybody runs under these options already
        if k is None:
            used = set([0])
            if self._set_of is not None:
                used |= set(torch.unique(self._set_of).cpu().tolist())  # (synchronises: only when a new set is needed)
            free = [j for j in range(1, _native.MG_MAX_OPTION_SETS) if j not in used]
            if not free:
                raise NotImplementedError("more than %d different option sets alive in one handle" % _native.MG_MAX_OPTION_SETS)
            k = free[0]
            while len(self._set_params) <= k:
                self._set_params.append({})
            # EVERY key is sent (an empty `have`): the library then checks each geometry key against the handle's geometry and
            # refuses a set whose agent_scale / arena_size / ... differs from it -- also when the value asked for is the
            # reference's default and the handle's is not (diffing against the defaults sent nothing in that case and
            # the instances silently ran under the handle's geometry)
            self._set_params[k] = {}
            try:
                self._write_set(k, params)
            except Exception:
                self._set_params[k] = {}  # half-written: never matched by a later reset, rewritten in full when it is reused
                raise
        if self._set_of is None:
            self._set_of = torch.zeros(self.num_envs, dtype=torch.int32, device=self.device)
            _native.check(_native.LIB.mg_bind_option_sets(self._h, self._set_of.data_ptr()), "mg_bind_option_sets")
        self._set_of[mask.to(device=self.device, dtype=torch.bool)] = k

    def _apply_options(self, options):
        params = process_reset_params(self.env_id, options)
        self._write_set(0, params)
        if self._set_of is not None:
            # a reset of every instance: all of them run under these options.  The per-instance index is unbound again, so the
            # handle is back in its one-set launch arrangement (one-launch mortar step, fused resets, lazy segments), and the other
            # sets are forgotten: their geometry entries may be stale now, a later masked reset writes the set it needs in full.
            _native.check(_native.LIB.mg_bind_option_sets(self._h, None), "mg_bind_option_sets")
            self._set_of = None
        del self._set_params[1:]
        self.reset_params = params
        if self.env_id in ("MortarMayhemB-Grid-v0", "MortarMayhemB-v0"):  # mortar_mayhem_b_grid.py:149-153
            self.max_episode_steps = calc_max_episode_steps(
                max(params["command_count"]), 0, 0, max(params["explosion_delay"]), max(params["explosion_duration"]))
        elif self.env_id in ("MortarMayhem-Grid-v0", "MortarMayhem-v0"):
            self.max_episode_steps = calc_max_episode_steps(
                max(params["command_count"]), max(params["command_show_duration"]), max(params["command_show_delay"]),
                max(params["explosion_delay"]), max(params["explosion_duration"]))
        else:
            self.max_episode_steps = params.get("max_steps", None)

    def _seed_tensor(self, seed):
        if seed is None:
            if self._seeded:
                return None
            # first reset without a seed: like gymnasium's np_random(None), every instance starts from OS entropy
            seed = np.random.SeedSequence().generate_state(self.num_envs, np.uint64) >> np.uint64(1)
            seed = seed.astype(np.int64)
        if isinstance(seed, torch.Tensor):
            s = seed.to(device=self.device, dtype=torch.int64)
        elif np.isscalar(seed):
            # one integer seeds instance i with seed + i (instance 0 == the reference's reset(seed=seed))
            s = torch.arange(self.num_envs, device=self.device, dtype=torch.int64) + int(seed)
        else:
            s = torch.as_tensor(np.asarray(seed, dtype=np.int64), device=self.device)
        assert s.numel() == self.num_envs
        return s.contiguous()

    # ------------------------------------------------------------------ API
    def reset(self, seed=None, return_info=True, options=None, mask=None):
        """Env.reset(seed, options) for all instances, or for those selected by the bool/uint8 tensor `mask`: the options then belong
        to those instances only (per-instance option sets, include/memgym.h: mg_set_option_set); mask with options=None re-starts
        them under the options each of them has.  `reset_params` and `max_episode_steps` describe the options of the latest FULL
        reset (option set 0); a masked reset with options does not change them."""
        with torch.cuda.device(self.device):
            if mask is not None and seed is None and not self._seeded:  # (checked before any option state changes)
                raise RuntimeError("a masked reset(seed=None) needs an earlier full reset: the other instances have no RNG stream yet")
            if mask is None:
                self._apply_options(options)
            else:
                self._apply_options_masked(options, mask)
            s = self._seed_tensor(seed)
            m = None if mask is None else mask.to(device=self.device, dtype=torch.uint8).contiguous()
            if m is not None and self._swapped:
                # a masked reset writes the rows of the masked instances only; after use_obs_buffer() the other rows of the new
                # buffer hold whatever was there: draw every instance's CURRENT frame into it first (one raster launch, this
                # sequence only).  (Rows a masked reset had left untouched right before the swap stay as they are: mg_render.)
                _native.check(_native.LIB.mg_render(self._h, self.obs.data_ptr(), self._stream()), "mg_render")
            self._swapped = False
            _native.check(_native.LIB.mg_reset(self._h, None if s is None else s.data_ptr(),
                                               None if m is None else m.data_ptr(), self.obs.data_ptr(),
                                               self.gt.data_ptr() if self.gt_dim else None, self._stream()), "mg_reset")
            if self.gt64 is not None:
                _native.check(_native.LIB.mg_ground_truth64(self._h, self.gt64.data_ptr(), self._stream()), "mg_ground_truth64")
            self._seeded = True
        info = {"ground_truth": self.gt if self.gt64 is None else self.gt64} if self.gt_dim else {}
        return self._obs(), info

    def step(self, actions):
        a = actions
        if not (isinstance(a, torch.Tensor) and a.dtype == torch.int32 and a.device == self.device and a.is_contiguous()):
            a = a if isinstance(a, torch.Tensor) else torch.as_tensor(np.asarray(a))
            a = a.to(device=self.device, dtype=torch.int32).contiguous()
        assert a.numel() == self._n_actions, "actions must have shape [N] or [N, 2]"
        if torch.cuda.current_device() != self.device.index:
            with torch.cuda.device(self.device):
                self._launch_step(a)
        else:
            self._launch_step(a)
        self._swapped = False  # a step writes every row
        _native.LIB.mg_peek_errors(self._h, self._p_err)  # host-mapped word: no synchronisation
        if self._err.value & ~self._tolerated:
            self.check_errors()
        info = dict(self._info_dict)
        if self.final_obs is not None and self.autoreset:  # rows valid where done_mask is set
            info["final_observation"] = self.final_obs
        return self._obs(), self.reward, self._done_bool, self._truncated, info

    def _launch_step(self, a):
        rc = _native.LIB.mg_step(self._h, a.data_ptr(), self.obs.data_ptr(), self._p_reward, self._p_done, self._p_gt, self._p_info,
                                 1 if self.autoreset else 0, self._raw_stream())
        if rc != 0:
            _native.check(rc, "mg_step")

    def new_obs_buffer(self):
        """A second observation buffer like `obs` (same shape, dtype, device and -- if `obs` came from mg_obs_alloc -- the same
        zone-balanced placement), for consumers that double-buffer (use_obs_buffer)."""
        if self.obs_placement_info is not None:
            try:
                t, _ = alloc_obs_buffer(tuple(self.obs.shape), self.obs.dtype, self.device)
                return t
            except RuntimeError:
                pass
        return torch.empty_like(self.obs)

    def use_obs_buffer(self, tensor):
        """Make `tensor` (same shape / dtype / device as `obs`) the buffer the NEXT reset / step writes its observations to
        (mg_step takes the buffer per call).  A consumer that still reads the previous buffer -- e.g. a gather to another
        rank running beside the next step (memory_gym_amd.dist.ObsGatherer) -- alternates between two buffers this way.
        A step() or a full reset() writes every row of the new buffer; a MASKED reset right after a swap first draws the
        current frame of every instance into it (the rows it does not reset would otherwise be stale)."""
        if tuple(tensor.shape) != tuple(self.obs.shape) or tensor.dtype != self.obs.dtype or tensor.device != self.obs.device or not tensor.is_contiguous():
            raise ValueError("use_obs_buffer: need a contiguous tensor like env.obs")
        self.obs = tensor
        self._swapped = True
        self._bind_step()

    def use_step_buffers(self, reward, done_u8):
        """Make `reward` (float32 [N]) and `done_u8` (uint8 [N]) the tensors the NEXT step stores its rewards / dones into (mg_step
        takes both per call) -- e.g. two views of one packed buffer that is shipped to another rank with the frames
        (memory_gym_amd.dist.ObsGatherer: BASELINE config 5's "obs (+reward, done)")."""
        for t, dt in ((reward, torch.float32), (done_u8, torch.uint8)):
            if tuple(t.shape) != (self.num_envs,) or t.dtype != dt or t.device != self.device or not t.is_contiguous():
                raise ValueError("use_step_buffers: need contiguous float32 [N] / uint8 [N] tensors on the handle's device")
        if reward.data_ptr() % 4:
            raise ValueError("use_step_buffers: the reward tensor must be 4-byte aligned")
        self.reward, self.done_u8 = reward, done_u8
        self._bind_step()

    def _obs(self):
        if self.vector_obs is None:
            return self.obs
        return {"visual_observation": self.obs, "vector_observation": self.vector_obs}

    def render(self):
        """rgb_array mode of the reference: fliplr(rot90(obs, 3)) == transpose to [y][x][c] (mortar_mayhem_grid.py:401-402);
        debug_rgb_array mode: the ground-truth view of every instance, uint8 [N, 336, 336, 3] (:403-405, include/memgym.h
        mg_render_debug)."""
        if self.render_mode == "debug_rgb_array":
            return self.render_debug()
        if self.obs_format != "u8_xyc":
            return (self.obs.permute(0, 2, 3, 1).float() * 255.0).round().to(torch.uint8)
        return self.obs.permute(0, 2, 1, 3)

    def render_debug(self):
        out = torch.empty((self.num_envs, 336, 336, 3), dtype=torch.uint8, device=self.device)
        with torch.cuda.device(self.device):
            _native.check(_native.LIB.mg_render_debug(self._h, out.data_ptr(), self._stream()), "mg_render_debug")
        return out

    def state_dict(self):
        n = _native.LIB.mg_state_size(self._h)
        buf = np.empty(n, np.uint8)
        with torch.cuda.device(self.device):
            _native.check(_native.LIB.mg_get_state(self._h, buf.ctypes.data, n), "mg_get_state")
        # the reset options in force belong to the state: geometry, schedules and limits are derived from them
        sd = {"env_id": self.env_id, "num_envs": self.num_envs, "blob": buf, "options": dict(self._applied), "seeded": self._seeded,
              "capacity": dict(self.capacity)}
        if self._set_of is not None:  # per-instance option sets in use
            sd["option_sets"] = [dict(p) for p in self._set_params]
            sd["set_of"] = self._set_of.cpu().numpy()
        return sd

    def load_state_dict(self, sd):
        """Restore a checkpoint, also into a handle that was never reset: the options in force when it was taken are
        applied first (a rebuild of the geometry happens at the restore, not at some later reset), then the state."""
        assert sd["env_id"] == self.env_id and sd["num_envs"] == self.num_envs
        if sd.get("capacity", {}) != self.capacity:
            raise ValueError("the checkpoint was taken with capacity=%r, this handle was made with %r" % (sd.get("capacity", {}), self.capacity))
        with torch.cuda.device(self.device):
            opts = sd.get("options")
            if opts is not None and any(self._applied.get(k) != v for k, v in opts.items()):
                # options only take effect at a reset: do one (its frames and RNG consumption are overwritten right below)
                self.reset(seed=0, options={k: v for k, v in opts.items() if k in DEFAULTS[self.env_id]})
                # whatever that throw-away reset flagged (e.g. use_exit=False on instances that have no exit YET: the restored
                # ones bring theirs) says nothing about the restored episodes
                _native.LIB.mg_poll_errors(self._h, C.byref(C.c_int()))
            if sd.get("option_sets") is not None:
                for k, p in enumerate(sd["option_sets"]):
                    while len(self._set_params) <= k:
                        self._set_params.append({})
                    if k > 0 and p:  # (an empty entry: a slot that was never filled or half-written when the checkpoint was taken)
                        self._set_params[k] = {}
                        self._write_set(k, process_reset_params(self.env_id, {kk: vv for kk, vv in p.items() if kk in DEFAULTS[self.env_id]}))
                if self._set_of is None:
                    self._set_of = torch.zeros(self.num_envs, dtype=torch.int32, device=self.device)
                    _native.check(_native.LIB.mg_bind_option_sets(self._h, self._set_of.data_ptr()), "mg_bind_option_sets")
                self._set_of.copy_(torch.as_tensor(np.asarray(sd["set_of"], dtype=np.int32), device=self.device))
            elif self._set_of is not None:  # the checkpoint was taken with one option set: back to that arrangement
                _native.check(_native.LIB.mg_bind_option_sets(self._h, None), "mg_bind_option_sets")
                self._set_of = None
                del self._set_params[1:]
            buf = np.ascontiguousarray(sd["blob"], dtype=np.uint8)
            _native.check(_native.LIB.mg_set_state(self._h, buf.ctypes.data, buf.size), "mg_set_state")
            self._seeded = bool(sd.get("seeded", True))  # (env.obs shows the restored episodes from the next step on)

    def set_profiling(self, every):
        """Bracket the kernels of every `every`-th step with HIP events (0/False = off, 1/True = every step)."""
        _native.check(_native.LIB.mg_set_profiling(self._h, int(every)), "mg_set_profiling")

    def get_profile(self, kind):
        """(total_ms, launches) of the logic (kind 0) or raster (kind 1) kernel since the last call."""
        ms, n = C.c_double(), C.c_int64()
        _native.check(_native.LIB.mg_get_profile(self._h, kind, C.byref(ms), C.byref(n)), "mg_get_profile")
        return ms.value, n.value

    CAPACITY_BITS = 1 | 4 | 8 | 32  # the bits that say "an instance reached a capacity of this build and its episode was ended"

    ERROR_BITS = {1: "more than 16 live spotlights in one instance (raise spawn_interval / spot speeds or lower initial_spawns)",
                  2: "path generation found no valid path (pygame_assets.py:723-724 raises here too)",
                  4: "endless path longer than 128 segments", 8: "more than 128 distinct fall-off cells",
                  16: "past-path window wider than 16 columns",
                  32: "Endless Mortar Mayhem command list reached its 512-entry capacity (the episode was ended)",
                  64: "a deferred-reset queue overflowed (a previous fused launch did not drain it)",
                  256: "use_exit=False for an instance that never had an exit (the reference raises AttributeError at this reset)"}

    def check_errors(self):
        """Raise if a kernel flagged a capacity/failure condition since the last call (synchronises the device).
        step() looks at the same bits after every call without synchronising and ends up here when one is set."""
        f = C.c_int()
        _native.check(_native.LIB.mg_poll_errors(self._h, C.byref(f)), "mg_poll_errors")
        self.capacity_events = getattr(self, "capacity_events", 0) | (f.value & self._tolerated)  # (kinds seen so far, on_capacity="truncate")
        f.value &= ~self._tolerated
        if f.value:
            what = "; ".join(m for b, m in self.ERROR_BITS.items() if f.value & b)
            hint = " (make(..., on_capacity='truncate') reports such instances as truncated instead of raising)" if f.value & self.CAPACITY_BITS else ""
            raise RuntimeError("memory_gym_amd: device error flags 0x%x: %s -- the frames of the affected instances are "
                               "no longer the reference's (include/memgym.h: mg_poll_errors)%s" % (f.value, what, hint))

    def rng_words(self, i):
        w = np.zeros(6, np.uint64)
        _native.check(_native.LIB.mg_debug_rng(self._h, int(i), w.ctypes.data), "mg_debug_rng")
        return w

    def debug_counter(self, name):
        """Named test / telemetry counter of the handle (include/memgym.h: mg_debug_counter), e.g. "one_launch_rescues"."""
        v = C.c_int64()
        _native.check(_native.LIB.mg_debug_counter(self._h, name.encode(), C.byref(v)), "mg_debug_counter")
        return v.value

    def close(self):
        if getattr(self, "_h", None):
            _native.LIB.mg_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


try:  # a real gymnasium.Env when the host has gymnasium (the reference's CustomEnv(gym.Env), environment.py:3-7)
    import gymnasium as _gym
    _EnvBase = _gym.Env
except Exception:  # gymnasium is not a dependency of the hot path
    _gym = None
    _EnvBase = object


class MemoryGymEnv(_EnvBase):
    """Single-instance environment with the reference's exact signatures (reset -> (obs, info); step -> 5-tuple of numpy
    obs, Python float reward, bool, False, dict) -- what `gymnasium.make(id)` returns.  Subclasses gymnasium.Env when
    gymnasium is importable; without it the same protocol surface (`unwrapped`, `spec`, `np_random`, `metadata`,
    `render_mode`, spaces) is provided here, so code written against the reference runs either way.

    Round 5: the single-instance fast path of the C ABI (include/memgym.h: mg_single_open / mg_single_reset / mg_single_step).
    Observation, reward, done, ground truth and the end-of-episode record live in pinned host memory that is mapped into the
    device; the kernels read the action from it and store their results straight into it, so step() is ONE native call (store the
    action, enqueue the step's launches, wait for the stream) and a copy of the 21-KB frame -- no torch operation.  (Rounds 1-4: an
    H2D action copy, two indexing kernels, three D2H copies and a stream synchronisation per step: 14 k steps/s.)"""

    metadata = {"render_modes": ["rgb_array", "debug_rgb_array"], "render_fps": 25}
    spec = None
    env_id = None  # set by the per-id subclasses in memory_gym_amd.envs

    def __init__(self, env_id=None, device=None, render_mode=None):
        env_id = env_id or self.env_id
        self.vec = VecMemoryGym(env_id, 1, device, render_mode)
        self.vec.autoreset = False
        self.action_space = self.vec.action_space
        self.observation_space = self.vec.observation_space
        self.has_ground_truth_info = self.vec.has_ground_truth_info
        if self.has_ground_truth_info:
            self.ground_truth_space = self.vec.ground_truth_space
        self.render_mode = render_mode
        self._np_random = None
        io = _native.SingleIO()
        io.struct_size = C.sizeof(_native.SingleIO)
        with torch.cuda.device(self.vec.device):
            _native.check(_native.LIB.mg_single_open(self.vec._h, C.byref(io)), "mg_single_open")
        self._io = io

        def view(ptr, ctype, shape):
            return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(ctype)), shape=shape)
        self._obs_host = view(io.obs, C.c_uint8, (84, 84, 3))
        self._vec_host = view(io.vec, C.c_float, (self.vec.vec_dim,)) if self.vec.vec_dim else None
        self._reward_host = view(io.reward, C.c_double, (1,))
        self._done_host = view(io.done, C.c_uint8, (1,))
        self._gt_host = view(io.gt, C.c_double, (max(self.vec.gt_dim, 1),))
        self._ep_reward_host = view(io.ep_reward, C.c_double, (1,))
        self._ep_length_host = view(io.ep_length, C.c_int32, (1,))
        self._aux_host = [view(io.aux[k], C.c_float, (1,)) for k in range(len(self.vec.info_names))]
        self._two_actions = self.vec.action_dim == 2
        # (the launch stream is torch's CURRENT stream at every call, like the batched path: a handle cached at construction or at the last
        # reset may be another stream than the caller's by now, or a released one -- ADVICE r5)
        self._raw_stream = self.vec._raw_stream
        self._step_fn, self._h, self._errword = _native.LIB.mg_single_step, self.vec._h, self.vec._err
        if self.vec.vector_obs is not None:
            # mg_single_open re-bound the handle's vector observation to the mapped block: `vec.vector_obs` (the batched tensor) is no
            # longer written; this adapter's observations carry the vector from the mapped block
            self.vec.vector_obs_stale = True
        self._peek = _native.LIB.mg_peek_errors

    # ---- gymnasium.Env protocol surface that exists with or without gymnasium
    @property
    def unwrapped(self):
        return self

    @property
    def np_random(self):
        """A numpy Generator like gymnasium.Env.np_random (the environment's own draws happen on the device, from the
        PCG64 stream that reset(seed) seeds exactly like gymnasium does; this host-side generator is seeded alike)."""
        if self._np_random is None:
            self._np_random = np.random.Generator(np.random.PCG64(np.random.SeedSequence()))
        return self._np_random

    @np_random.setter
    def np_random(self, value):
        self._np_random = value

    def _observation(self):
        o = self._obs_host.copy()
        if self._vec_host is not None:
            o = {"visual_observation": o, "vector_observation": self._vec_host.copy()}
        return o

    @property
    def max_episode_steps(self):
        return self.vec.max_episode_steps

    def reset(self, seed=None, return_info=True, options=None):
        if seed is not None:  # gymnasium.Env.reset(seed): the host-side generator follows the same seed
            self._np_random = np.random.Generator(np.random.PCG64(np.random.SeedSequence(int(seed))))
        v = self.vec
        with torch.cuda.device(v.device):
            v._apply_options(options)
            if seed is None and not v._seeded:  # first reset without a seed: OS entropy, like gymnasium's np_random(None)
                seed = int(np.random.SeedSequence().generate_state(1, np.uint64)[0] >> np.uint64(1))
            _native.check(_native.LIB.mg_single_reset(self._h, 0 if seed is None else int(seed), 0 if seed is None else 1, C.c_void_p(self._raw_stream())), "mg_single_reset")
            v._seeded = True
        out = {}
        if v.gt_dim:
            out["ground_truth"] = self._gt_host[:v.gt_dim].copy()
        return self._observation(), out

    def step(self, action):
        if self._two_actions:
            a0, a1 = int(action[0]), int(action[1])
        else:
            a0 = a1 = int(action)
        rc = self._step_fn(self._h, a0, a1, self._raw_stream())
        if rc != 0:  # negative: the call failed; positive: device error bits as they stand (include/memgym.h)
            if rc < 0:
                _native.check(rc, "mg_single_step")
            if rc & ~self.vec._tolerated:
                self.vec.check_errors()
        r, d = float(self._reward_host[0]), bool(self._done_host[0])  # r: the reference's Python float, bit for bit
        out = {}
        if d:  # end of episode (rare): the reference's terminal info dict
            out["reward"] = float(self._ep_reward_host[0])
            out["length"] = int(self._ep_length_host[0])
            for nm, a in zip(self.vec.info_names, self._aux_host):
                out[nm] = float(a[0])
        if self.vec.gt_dim:
            out["ground_truth"] = self._gt_host[:self.vec.gt_dim].copy()
        return self._observation(), r, d, False, out

    def render(self):
        if self.render_mode == "debug_rgb_array":
            return self.vec.render_debug()[0].cpu().numpy()
        return self._obs_host.transpose(1, 0, 2).copy()  # mortar_mayhem_grid.py:401-402

    def state_dict(self):
        return self.vec.state_dict()

    def load_state_dict(self, sd):
        self.vec.load_state_dict(sd)

    def close(self):
        self.vec.close()
