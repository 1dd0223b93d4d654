"""Multi-GPU plumbing: one process per GPU (torch.distributed; backend "nccl" == RCCL over xGMI on ROCm, "gloo" in
the CPU tests).  Environment instances are independent, so the path shards with NO data-path collective:

    rank r of W owns global instances [lo, hi) = shard_range(N_total, r, W), and instance i is always seeded
    base_seed + i whatever W is, so results do not depend on the world size.

The only (optional) exchange is BASELINE config 5's gather of observations/rewards/dones to rank 0 for a
single-learner rollout.  Two forms with the same result on rank 0:

    gather_to_rank0   a torch.distributed gather (RCCL send/recv): every rank writes its frames locally, then ships them;
    ObsGatherer       the same collective, double-buffered: the gather of step t (frames, and the step's rewards / dones as
                      one packed 5-B-per-instance tensor) runs beside step t + 1;
    PeerObsBuffer     rank 0's [N_total, ...] observation tensor is mapped into every rank (HIP IPC, peer access over
                      xGMI) and handed to VecMemoryGym as `obs_buffer`: the raster kernels store their frames straight
                      into rank 0's HBM -- no second copy, no collective on the data path (only a barrier per step).
"""
import torch
import torch.distributed as dist


def shard_range(n_total, rank, world):
    """Contiguous block of instance indices owned by `rank` (blocks differ by at most one instance)."""
    base, rem = divmod(int(n_total), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_seeds(n_total, rank, world, base_seed=0, device=None):
    lo, hi = shard_range(n_total, rank, world)
    return torch.arange(lo, hi, dtype=torch.int64, device=device) + int(base_seed)


def gather_to_rank0(tensor, dst=0, group=None):
    """Gather equally-shaped per-rank tensors to `dst`; returns the concatenation on dst, None elsewhere.
    Ragged shards (N_total % W != 0) are padded to the largest shard and trimmed on dst."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    n_local = torch.tensor([tensor.shape[0]], device=tensor.device, dtype=torch.int64)
    sizes = [torch.zeros_like(n_local) for _ in range(world)]
    dist.all_gather(sizes, n_local, group=group)
    sizes = [int(s.item()) for s in sizes]
    n_max = max(sizes)
    if tensor.shape[0] < n_max:
        pad = torch.zeros((n_max - tensor.shape[0],) + tuple(tensor.shape[1:]), dtype=tensor.dtype, device=tensor.device)
        tensor = torch.cat([tensor, pad], 0)
    bufs = [torch.empty_like(tensor) for _ in range(world)] if rank == dst else None
    dist.gather(tensor.contiguous(), bufs, dst=dst, group=group)
    if rank != dst:
        return None
    return torch.cat([b[:s] for b, s in zip(bufs, sizes)], 0)


def packed_scalars(n, device):
    """The 5 B / instance that travel with a step's frames ("gather of obs (+reward, done)"): ONE flat
    uint8 tensor of 4 n + n bytes (padded to 16) and two views of it -- `reward` float32 [n] over the first 4 n bytes, `done`
    uint8 [n] behind it.  The step kernels store into the views (VecMemoryGym.use_step_buffers), one collective ships the flat
    tensor.  Returns (flat, reward, done)."""
    n = int(n)
    flat = torch.zeros((5 * n + 15) // 16 * 16, dtype=torch.uint8, device=device)
    return flat, flat[:4 * n].view(torch.float32), flat[4 * n:5 * n]


def unpack_scalars(flat, n):
    """(reward float32 [n], done bool [n]) views of a flat tensor laid out by packed_scalars()."""
    n = int(n)
    return flat[:4 * n].view(torch.float32), flat[4 * n:5 * n].view(torch.bool)


class ObsGatherer:
    """BASELINE config 5's optional exchange, double-buffered: the gather of step t's observations AND of its rewards / dones
    (5 B per instance, one small collective behind the frames') to rank `dst` runs BESIDE step t + 1 (hard
    part 6; section 8e: "gather obs (+ reward/done, 5 B/env) to rank 0").

    The environment alternates between two observation buffers (VecMemoryGym.use_obs_buffer) and two packed reward / done
    buffers (use_step_buffers); after step t has been
    enqueued, the gather of its buffers is issued as asynchronous collectives (RCCL runs them on its own stream, ordered
    behind the step's kernels), and only the step that is about to OVERWRITE those buffers -- step t + 2 -- waits for them.
    Rank dst keeps two sets of receive buffers as well; `gathered()` joins the gather of the latest step and returns its W
    per-rank observation tensors, `gathered_step()` those plus the W reward and done tensors (rank dst; None elsewhere).
    Shards must be equal-sized (N_total % W == 0, what the bench uses).

        g = ObsGatherer(env)                 # env: VecMemoryGym, after reset()
        for t in range(T):
            obs, rew, done, _, info = g.step(actions[t])      # gather of step t starts, step t - 1's may still run
            frames, rewards, dones = g.gathered_step()        # rank dst: three lists of W tensors of the LATEST step; else None
    """

    def __init__(self, env, dst=0, group=None, scalars=True):
        self.env, self.dst, self.group = env, dst, group
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        self.bufs = [env.obs, env.new_obs_buffer() if hasattr(env, "new_obs_buffer") else torch.empty_like(env.obs)]
        self.recv = [[torch.empty_like(env.obs) for _ in range(self.world)] for _ in range(2)] if self.rank == dst else [None, None]
        self.work = [None, None]
        # reward + done: two packed buffers the step kernels store into, two sets of receive buffers on dst
        self.scalars = bool(scalars) and hasattr(env, "use_step_buffers")
        self.n = int(env.obs.shape[0])
        if self.scalars:
            self.packed = [packed_scalars(self.n, env.obs.device) for _ in range(2)]
            self.recv_packed = ([[torch.empty_like(self.packed[0][0]) for _ in range(self.world)] for _ in range(2)]
                                if self.rank == dst else [None, None])
        self.work_packed = [None, None]
        self.t = 0
        # what the environment stored into before this gatherer re-routed it (restored by close())
        self._own = (env.obs, getattr(env, "reward", None), getattr(env, "done_u8", None))

    def close(self):
        """Join the outstanding gathers and hand the environment its OWN observation / reward / done tensors back: from the next
        step on the env stores into them again and nothing returned by it aliases this gatherer's double buffers."""
        self.drain()
        obs, rew, done = self._own
        self.env.use_obs_buffer(obs)
        if self.scalars and rew is not None:
            self.env.use_step_buffers(rew, done)

    def _join(self, k):
        for w in (self.work, self.work_packed):
            if w[k] is not None:
                w[k].wait()
                w[k] = None

    def step(self, actions):
        """env.step(actions) into this step's half of the double buffers, then the asynchronous gathers of that half.
        LIFETIME of what it returns: obs, reward and done are views of the double buffers -- valid until the step AFTER NEXT
        overwrites them (like gathered()); a trainer that keeps them longer copies them.  close() ends the arrangement."""
        k = self.t & 1
        self._join(k)  # the gathers that still read these buffers (step t - 2): the launch stream waits for them
        self.env.use_obs_buffer(self.bufs[k])
        if self.scalars:
            self.env.use_step_buffers(self.packed[k][1], self.packed[k][2])
        out = self.env.step(actions)
        self.work[k] = dist.gather(self.bufs[k], self.recv[k], dst=self.dst, group=self.group, async_op=True)
        if self.scalars:
            self.work_packed[k] = dist.gather(self.packed[k][0], self.recv_packed[k], dst=self.dst, group=self.group, async_op=True)
        self.t += 1
        return out

    def gathered(self):
        """Join the gather of the latest step; rank dst gets its W per-rank observation tensors (valid until the step
        after next), the other ranks None."""
        if self.t == 0:
            return None
        k = (self.t - 1) & 1
        self._join(k)
        return self.recv[k]

    def gathered_step(self):
        """Like gathered(), with the step's rewards and dones: rank dst gets (frames, rewards, dones), three lists of W per-rank
        tensors (uint8 frames, float32 [n], bool [n]; valid until the step after next), the other ranks None."""
        frames = self.gathered()
        if frames is None or not self.scalars:
            return None if frames is None else (frames, None, None)
        k = (self.t - 1) & 1
        pairs = [unpack_scalars(f, self.n) for f in self.recv_packed[k]]
        return frames, [p[0] for p in pairs], [p[1] for p in pairs]

    def drain(self):
        for k in range(2):
            self._join(k)


class PeerObsBuffer:
    """Rank `dst`'s observation tensor [N_total, *frame_shape], shared with every rank of `group`.

    `full` is the whole tensor (memory on dst's GPU, mapped into this process), `local` this rank's rows
    `shard_range(N_total, rank, world)` -- pass it as `obs_buffer=` to `memory_gym_amd.make`.  `fence()` is the per-step
    synchronisation: after it returns on dst, every rank's frames of the step are in `full`.
    `ok` is False (on every rank alike, `why` says which rank and why, a warning is issued) when some rank's GPU has no peer
    access to dst's GPU; `full` / `local` are None then and the caller falls back to `gather_to_rank0`.

    The mapping uses torch's CUDA-IPC tensor sharing (`hipIpcGetMemHandle` / `hipIpcOpenMemHandle`, dmabuf mode:
    keep HSA_ENABLE_IPC_MODE_LEGACY=0); dst must keep the object alive while the others use it."""

    def __init__(self, n_total, frame_shape=(84, 84, 3), dtype=torch.uint8, device=None, dst=0, group=None):
        from torch.multiprocessing.reductions import reduce_tensor

        from . import _native
        self.rank, self.world, self.dst, self.group = dist.get_rank(group), dist.get_world_size(group), dst, group
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self.ok, self.why, self.full, self.local = True, "", None, None
        # 1. can this rank's GPU store into dst's GPU?  (peer access is checked and switched on explicitly: the raster kernels
        #    run on THIS device with a pointer into dst's memory; nothing else on the path would enable it for them)
        idx = [None]
        if self.rank == dst:
            idx[0] = self.device.index
        dist.broadcast_object_list(idx, src=dst, group=group)
        dst_index = idx[0]
        if self.rank != dst and torch.cuda.device_count() > 1 and dst_index != self.device.index:
            if dst_index >= torch.cuda.device_count():
                self.ok, self.why = False, "rank %d does not see device %d" % (self.rank, dst_index)
            elif _native.LIB.mg_enable_peer_access(self.device.index, dst_index) != 0:
                self.ok, self.why = False, "rank %d: %s" % (self.rank, _native.last_error())
        # every rank must take the same path
        flags = [None] * self.world
        dist.all_gather_object(flags, (self.ok, self.why), group=group)
        bad = [w for ok, w in flags if not ok]
        if bad:
            self.ok, self.why = False, "; ".join(bad)
            import warnings
            warnings.warn("memory_gym_amd.dist.PeerObsBuffer: peer-mapped observation stores are not available (%s); "
                          "use gather_to_rank0 instead" % self.why)
            return
        # 2. share dst's tensor (HIP IPC, dmabuf mode)
        payload = [None]
        if self.rank == dst:
            self.full = torch.empty((int(n_total),) + tuple(frame_shape), dtype=dtype, device=self.device)
            self._check_exportable(self.full)
            payload[0] = reduce_tensor(self.full)  # (rebuild function, picklable arguments incl. the IPC handle)
        dist.broadcast_object_list(payload, src=dst, group=group)
        if self.rank != dst:
            rebuild, args = payload[0]
            self.full = rebuild(*args)
        lo, hi = shard_range(n_total, self.rank, self.world)
        self.local = self.full[lo:hi]
        self._token = torch.zeros(1, device=self.device)

    @staticmethod
    def _check_exportable(tensor):
        """hipIpcGetMemHandle works on hipMalloc allocations only; a zone-balanced buffer (mg_obs_alloc: hipMemCreate pieces in a
        reserved virtual range) would fail late and cryptically inside torch's reduce_tensor.  Refuse it here, by name."""
        from .vec_env import is_balanced_buffer
        if is_balanced_buffer(tensor):
            raise ValueError("PeerObsBuffer: a buffer from mg_obs_alloc (obs_placement='balanced') cannot be exported through HIP IPC; "
                             "share an ordinary allocation (torch.empty) -- its rows are filled over xGMI at a fifth of one memory "
                             "zone's rate, placement does not matter for it (DESIGN.md section 6)")

    def fence(self):
        """Stream-ordered on the nccl backend (a 4-byte all-reduce enqueued behind this rank's kernels); on gloo the
        device is synchronised first."""
        if dist.get_backend(self.group) != "nccl":
            torch.cuda.synchronize(self.device)
            dist.barrier(group=self.group)
        else:
            dist.all_reduce(self._token, group=self.group)

    def bind_scalars(self, env):
        """Route `env`'s step rewards and dones (5 B per instance) into a packed buffer that fence_with_scalars() ships to dst:
        the frames travel as the raster kernels' own stores over xGMI, the scalars as ONE small collective that orders the
        streams at the same time (it takes the place of fence()'s 4-byte all-reduce)."""
        self.n_local = int(env.num_envs)
        self._packed = packed_scalars(self.n_local, self.device)
        self._scalars_env, self._scalars_own = env, (env.reward, env.done_u8)
        env.use_step_buffers(self._packed[1], self._packed[2])  # (from now on env.step()'s reward / done are views of the packed buffer: valid until the next step)
        self._recv_packed = [torch.empty_like(self._packed[0]) for _ in range(self.world)] if self.rank == self.dst else None

    def unbind_scalars(self):
        """the environment stores its rewards / dones into its own tensors again"""
        if getattr(self, "_scalars_env", None) is not None:
            self._scalars_env.use_step_buffers(*self._scalars_own)
            self._scalars_env = None

    def fence_with_scalars(self):
        """After it returns on dst (stream-ordered on nccl), every rank's frames of the step are in `full` and dst holds every
        rank's rewards and dones: returns (rewards, dones), two lists of W per-rank tensors, on dst; None elsewhere.  Equal-sized
        shards (N_total % W == 0)."""
        if dist.get_backend(self.group) != "nccl":  # (gloo gathers host tensors only: the two-process test on one GPU)
            torch.cuda.synchronize(self.device)
            host = [torch.empty(self._packed[0].shape, dtype=torch.uint8) for _ in range(self.world)] if self.rank == self.dst else None
            dist.gather(self._packed[0].cpu(), host, dst=self.dst, group=self.group)
            if self.rank == self.dst:
                for d, h in zip(self._recv_packed, host):
                    d.copy_(h)
        else:
            dist.gather(self._packed[0], self._recv_packed, dst=self.dst, group=self.group)
        if self.rank != self.dst:
            return None
        pairs = [unpack_scalars(f, self.n_local) for f in self._recv_packed]
        return [p[0] for p in pairs], [p[1] for p in pairs]
