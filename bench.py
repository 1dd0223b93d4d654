#!/usr/bin/env python3
"""bench.py -- aggregate env-steps/s of the batched Memory Gym hot path on N MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--env ID] [--envs-per-gpu M] [--gather [rccl|peer]]

A "step" is one mg_step() over every instance of the workload: logic kernel + raster kernel (one launch that holds both for
the mortar family), with
same-step auto-reset, 84x84x3 uint8 observations written to HBM.  Inputs (actions) are generated on the device
before the timed region.

N > 1: one process per GPU (torch.distributed, backend nccl == RCCL), instances sharded with no data-path collective
(weak scaling: per-GPU work fixed).  `python bench.py --gpus N` alone spawns the N ranks itself (one per GPU, rendezvous on
127.0.0.1); launched under torch.distributed.run it uses the ranks it is given.  After the headline workload an N > 1
run also measures BASELINE.json's config 5 (Endless-MortarMayhem-v0, 32,768 instances per GPU) without gather, with the
RCCL gather of observations to rank 0 and with the peer-mapped variant, reported under "config5".

Setup (not timed, not counted as warm-up): reset of every instance and `--settle` steps with random actions, so that the
episodes de-synchronise -- right after a reset all instances are in the same phase of their episode and the first ~40
raster launches are 10-20 % slower than the stationary mix.  Then W warm-up steps, then EXACTLY K timed steps between
barrier + synchronize on both sides (max over ranks).

The timed region is measured with a HIP event pair recorded on the launch stream right behind the opening fence and right
before the closing one (`timing` says so, the host clock around the same region is kept next to it as
`wall_ms_per_step`; with a gather on another stream the host clock between the fences is the one that counts).

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` (raster kernel, HIP events recorded by
the library on the launch stream: every 8th step of the timed region when K >= 64; for shorter runs the timed region stays
undisturbed and the 32 steps after it are all bracketed) and `cpu_baseline` (the CPU oracle = a port of the reference's
algorithm, bounded sample, OpenMP over instances, timed in a subprocess; rank 0, N = 1 only).  N = 1 also carries
`c1` (BASELINE config C1, the reference's own bench.py loop: one instance, reset(seed=1), actions from PCG64(12345), 1,000
episodes -- BASELINE's 1,000 since round 5 -- the HIP single-instance adapter and the oracle on one thread), the C3 entry's `reset_share` (time of the
path-generating resets over the step time) and, when rocprofv3 is on PATH, `roofline.traffic` MEASURED by two child passes
of this very command (--pmc WRITE_SIZE, --pmc FETCH_SIZE; /opt/skills/guides/MI355X_MICROARCH.md, HBM section).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "endless-memory-gym_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md); ~6.3 TB/s achievable
FRAME = 84 * 84 * 3
# algorithmic bytes of the RASTER kernel per env-step: the observation it writes + the 16-byte draw descriptor it reads.
# (Whole step incl. logic kernel state/action/reward traffic, SURVEY.md 8(d): MM-Grid 21,305 B.)
STEP_BYTES = {"MortarMayhem-Grid-v0": 21305, "MortarMayhem-v0": 21309, "Endless-MortarMayhem-v0": 21821,
              "MysteryPath-v0": 21437, "Endless-SearingSpotlights-v0": 22205}
# bytes of the frame descriptor the raster reads per frame
DESC_BYTES = {"MortarMayhem-Grid-v0": 16, "MortarMayhem-v0": 16, "Endless-MortarMayhem-v0": 16, "MortarMayhemB-Grid-v0": 16, "MortarMayhemB-v0": 16,
              "MysteryPath-v0": 64, "MysteryPath-Grid-v0": 64, "Endless-MysteryPath-v0": 64, "SearingSpotlights-v0": 128, "Endless-SearingSpotlights-v0": 128}
DEFAULT_ENVS = {"MortarMayhem-Grid-v0": 65536, "MortarMayhem-v0": 65536, "MortarMayhemB-Grid-v0": 65536, "MortarMayhemB-v0": 65536, "Endless-MortarMayhem-v0": 32768,
                "MysteryPath-v0": 32768, "MysteryPath-Grid-v0": 32768, "Endless-MysteryPath-v0": 32768, "SearingSpotlights-v0": 16384,
                "Endless-SearingSpotlights-v0": 16384}
OBS_ELEM = {"u8_xyc": 1, "f32_chw": 4, "f16_chw": 2, "bf16_chw": 2}


# ---------------------------------------------------------------------------------------------------------------- CPU baseline
def cpu_worker(env_id, threads, budget_s):
    """(subprocess) Time the CPU oracle -- oracle/: a restatement of the reference's per-instance algorithm incl. its
    software raster -- with `threads` OpenMP threads (the environment variables were set before this process started).
    Observation, reward and done buffers are allocated once; a step touches no fresh memory."""
    import numpy as np

    import oracle_lib

    n = 64 * threads if threads > 1 else 256
    b = oracle_lib.OracleBatch(env_id, n)
    disc = b.discrete
    obs = np.zeros((n, 84, 84, 3), np.uint8)
    rew, done = np.zeros(n, np.float64), np.zeros(n, np.uint8)
    b.reset(np.arange(n, dtype=np.int64), out=obs)
    g = np.random.Generator(np.random.PCG64(0))
    acts = [(g.integers(0, 4, n) if disc else g.integers(0, 3, (n, 2))).astype(np.int32) for _ in range(16)]
    for a in acts[:3]:
        b.step(a, autoreset=True, out=(obs, rew, done))
    t0 = time.perf_counter()
    steps = 0
    while time.perf_counter() - t0 < budget_s:
        b.step(acts[steps % 16], autoreset=True, out=(obs, rew, done))
        steps += 1
    dt = time.perf_counter() - t0
    b.close()
    print(json.dumps({"value": n * steps / dt, "instances": n, "steps": steps, "seconds": dt, "threads": threads}))


def usable_cpus():
    """CPUs this process may actually use: the affinity mask, capped by the cgroup CPU quota (a container that shows 256
    CPUs but may burn 16 CPU-seconds per second runs 256 busy threads at a sixteenth of their speed)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    quota = None
    try:
        q, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]  # cgroup v2
        if q != "max":
            quota = float(q) / float(period)
    except Exception:
        try:  # cgroup v1
            q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            period = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / period
        except Exception:
            pass
    if quota:
        n = max(1, min(n, int(quota)))
    return n


def cpu_model():
    """the host CPU's model string next to `cores`"""
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.lower().startswith("model name"):
                return ln.split(":", 1)[1].strip()
    except Exception:
        pass
    import platform
    return platform.processor() or platform.machine() or "unknown"


def cpu_baseline(env_id):
    """Single-thread and all-core rate of the CPU oracle on this host, each in its own subprocess so that the OpenMP
    runtime is configured before it loads (OMP_NUM_THREADS / OMP_PROC_BIND)."""
    cores = usable_cpus()
    res = {}
    for threads, budget in ((1, 4.0), (cores, 12.0)):
        env = dict(os.environ, OMP_NUM_THREADS=str(threads), OMP_PROC_BIND="false", OMP_WAIT_POLICY="passive")
        out = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-worker", env_id, str(threads), str(budget)],
                             env=env, capture_output=True, text=True, timeout=300)
        line = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
        if not line:
            raise RuntimeError("cpu worker failed: " + out.stderr[-300:])
        res[threads] = json.loads(line[-1])
        if cores == 1:
            break
    one, allc = res[1], res[cores]
    return {"value": allc["value"], "unit": "env steps/s", "cores": cores, "kind": "port", "cpu_model": cpu_model(),
            "cores_note": "usable CPUs = affinity mask (%d) capped by the cgroup CPU quota" % (len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else cores),
            "single_thread": one["value"], "scaling_efficiency": allc["value"] / (cores * one["value"]),
            "sample": "%s, CPU oracle (C, incl. software raster, preallocated buffers): %d instances x %d steps on %d OpenMP "
                      "threads in %.1f s; 1 thread: %d instances x %d steps in %.1f s"
                      % (env_id, allc["instances"], allc["steps"], cores, allc["seconds"], one["instances"], one["steps"], one["seconds"])}


def pygame_baseline(env_id, episodes=30):
    """time the real PyGame reference ONLY if the operator installed it on this host
    (PyPI package `memory-gym`); never substitute an estimate."""
    try:
        import gymnasium as gym
        import memory_gym  # noqa: F401
        import numpy as np
    except Exception:
        return "unavailable (memory-gym/pygame/gymnasium are not installed on this host)"
    env = gym.make(env_id)
    g = np.random.Generator(np.random.PCG64(12345))
    steps, t0 = 0, time.perf_counter()
    env.reset(seed=1)
    for _ in range(episodes):
        done = False
        while not done:
            a = int(g.integers(0, 4)) if hasattr(env.action_space, "n") else g.integers(0, 3, 2)
            _, _, done, _, _ = env.step(a)
            steps += 1
        env.reset()
    return {"value": steps / (time.perf_counter() - t0), "unit": "env steps/s", "cores": 1, "kind": "reference",
            "sample": "%d episodes of %s, PyGame reference, 1 process" % (episodes, env_id)}


# ---------------------------------------------------------------------------------------------------------------- one workload
def run_workload(env_id, n_local, K, W, settle, world, rank, dev, obs_format="u8_xyc", gather=None, events=True, event_stride=8,
                 count_done=False, leg="headline", long_window=0, policy=None):
    """Create the environments, settle, warm up, time K steps; returns a dict (identical on every rank).
    long_window: after everything else, one more window of that many steps timed the same way (BASELINE.md section 3 asks for
    >= 2,000 timed steps whatever K the caller chose) -> "value_long".
    policy: None = uniform random actions from 64 pre-generated buffers; "follower:EPS" (ids with a ground truth that names the way,
    Endless-MysteryPath-v0) = the action read off the info["ground_truth"] the previous step returned, a random one with probability
    EPS -- three small torch kernels per step on the launch stream, INSIDE the timed region like a trainer's policy would be."""
    if os.environ.get("MEMGYM_BENCH_FAKE"):
        return fake_workload(env_id, n_local, K, W, settle, world, rank, dev, obs_format, gather, events, event_stride, count_done, leg, long_window, policy)
    import torch
    import torch.distributed as dist

    import memory_gym_amd
    from memory_gym_amd.dist import ObsGatherer, PeerObsBuffer, shard_seeds

    n_total = n_local * world
    peer = None
    note = None
    if gather == "peer" and world > 1:
        code, dt_, shape = memory_gym_amd.VecMemoryGym.OBS_FORMATS[obs_format]
        peer = PeerObsBuffer(n_total, frame_shape=shape, dtype=dt_, device=dev)
        if not peer.ok:  # no peer access between this GPU and rank 0's: fall back to the collective
            note = "peer mapping unavailable (%s): fell back to the RCCL gather" % peer.why
            peer, gather = None, "rccl"
    own_buffer = None
    pad = int(os.environ.get("MEMGYM_BENCH_OBS_PAD_FRAMES", "0"))
    if pad and peer is None and obs_format == "u8_xyc":
        # measurement hook (library variants built with -DMG_LAB_OBS_STRIDE: frames at a padded stride need room behind the N-th frame)
        from memory_gym_amd.vec_env import alloc_obs_buffer
        big, _ = alloc_obs_buffer((n_local + pad, 84, 84, 3), torch.uint8, dev)
        own_buffer = big[:n_local]
    venv = None
    if policy == "vector_api":  # the gymnasium-0.29 vector convention over the same handle (terminal observations + same-call resets)
        from memory_gym_amd.vector import GymnasiumVectorEnv
        assert peer is None and own_buffer is None and gather is None
        venv = GymnasiumVectorEnv(env_id, n_local, device=dev.index, obs_format=obs_format)
        env, policy = venv.env, None
    else:
        env = memory_gym_amd.make(env_id, num_envs=n_local, device=dev.index, obs_format=obs_format,
                                  obs_buffer=peer.local if peer else own_buffer)
    # instance i (global index) is seeded i whatever the world size -> results are world-size invariant
    env.reset(seed=shard_seeds(n_total, rank, world, base_seed=0, device=dev))
    g = torch.Generator(device=dev).manual_seed(1234 + rank)
    n_act_bufs = 64
    shape, hi = ((n_local,), 4) if env.action_dim == 1 else ((n_local, 2), 3)
    acts = [torch.randint(0, hi, shape, device=dev, generator=g, dtype=torch.int32) for _ in range(n_act_bufs)]
    # the collective path is taken whenever a process group exists -- also at world size 1 (RCCL on one GPU: the plumbing
    # test of tests/test_gpu_rccl_world1.py)
    dist_on = dist.is_available() and dist.is_initialized()
    gatherer = ObsGatherer(env) if (gather == "rccl" and dist_on) else None  # double-buffered: gather t (frames + rewards / dones) beside step t + 1
    if peer is not None:  # the step's rewards / dones (5 B per instance) ride on the collective that orders the streams
        peer.bind_scalars(env)

    follow = None
    if policy:
        assert policy.startswith("follower:") and env.gt_dim == 3 and gatherer is None and peer is None, policy
        eps = float(policy.split(":")[1])
        way = torch.tensor([1.0, 2.0, 3.0], device=dev)  # ground truth one-hot (right, up, down) -> actions 1, 2, 3
        rnd = [torch.rand(n_local, device=dev, generator=g) < eps for _ in range(n_act_bufs)]
        follow = {"gt": env.gt.clone()}  # (reset() above filled env.gt)

    def one_step(k):
        if follow is not None:
            a = torch.where(rnd[k % n_act_bufs], acts[k % n_act_bufs], (follow["gt"].to(torch.float32) @ way).to(torch.int32))
            _, _, _, _, info = env.step(a)
            follow["gt"] = info["ground_truth"]
            return
        if gatherer is not None:
            gatherer.step(acts[k % n_act_bufs])
            return
        if venv is not None:
            venv.step(acts[k % n_act_bufs])
            return
        env.step(acts[k % n_act_bufs])
        if peer is not None:  # the frames are already in rank 0's memory; the gather of the packed rewards / dones orders the streams
            peer.fence_with_scalars()

    def fence():
        if gatherer is not None:
            gatherer.drain()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for k in range(settle):  # setup: de-synchronise the episodes
        one_step(k)
    _fault(leg, rank)
    # ... and keep stepping until the rate is flat (C4's first timed window was 5 % below the other five -- 200 steps
    # do not settle every workload): windows of 50 steps until two in a row are within 1 % of their predecessor, at most 20
    settle_windows = []
    if not (gather and dist_on):
        sa, sb = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        flat = 0
        for w in range(20):
            sa.record()
            for k in range(50):
                one_step(settle + 50 * w + k)
            sb.record()
            torch.cuda.synchronize()
            settle_windows.append(sa.elapsed_time(sb))
            if w and abs(settle_windows[-1] - settle_windows[-2]) <= 0.01 * settle_windows[-2]:
                flat += 1
                if flat >= 2:
                    break
            else:
                flat = 0
        settle += 50 * len(settle_windows)
    for k in range(W):
        one_step(settle + k)
    in_region = events and K >= 64
    if in_region:
        env.set_profiling(max(1, event_stride))
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    done_sum = torch.zeros((), dtype=torch.int64, device=dev) if count_done else None
    path0 = None
    if count_done:  # C3: the library's own telemetry of the A* path generation inside the step's launches (wave-ticks, paths)
        try:
            path0 = (env.debug_counter("path_gen_ticks"), env.debug_counter("path_gen_paths"))
        except Exception:
            path0 = None
    fence()
    t0 = time.perf_counter()
    ev0.record()  # hipEventRecord on the launch stream (torch's current stream is the one mg_step is given)
    for k in range(K):
        one_step(settle + W + k)
    ev1.record()
    fence()
    dt_wall = time.perf_counter() - t0
    dt_event = ev0.elapsed_time(ev1) * 1e-3
    # everything of a step is enqueued on the launch stream unless a collective runs beside it: then the fences decide
    dt = dt_wall if (gather and dist_on) else dt_event
    # spread of the headline: five more windows of K steps each, timed the same way right behind the timed region
    windows = []
    if not (gather and dist_on):
        wev = [torch.cuda.Event(enable_timing=True) for _ in range(6)]
        wev[0].record()
        for w in range(5):
            for k in range(K):
                one_step(settle + W + K * (w + 1) + k)
            wev[w + 1].record()
        torch.cuda.synchronize()
        windows = [n_total * K / (wev[w].elapsed_time(wev[w + 1]) * 1e-3) for w in range(5)]
    raster_ms = raster_n = logic_ms = logic_n = 0
    region = None
    if events:
        if in_region:
            region = "every %d-th step of the timed region" % max(1, event_stride)
        else:  # short run: keep the timed region undisturbed, bracket every launch of the 32 steps after it
            env.set_profiling(1)
            for k in range(32):
                one_step(settle + W + 6 * K + k)
            region = "32 steps right after the timed region and its five spread windows, every launch bracketed"
        raster_ms, raster_n = env.get_profile(1)
        logic_ms, logic_n = env.get_profile(0)
        env.set_profiling(False)
    # per-box control: pure store streams over the SAME observation buffer at the SAME launch size, right beside
    # the timed region -- a linear 16-byte fill (the memory system's store ceiling here) and the raster's store shape without
    # any compose work (include/memgym.h: mg_store_probe).  Best and median of 9 launches each after 3 warm-ups.
    box = None
    if obs_format == "u8_xyc" and peer is None and gatherer is None and hasattr(memory_gym_amd._native.LIB, "mg_store_probe"):  # (a gatherer's buffers are checked below)
        import ctypes as C
        box = {}
        stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        for pattern, name in ((0, "linear_fill"), (1, "frame_shaped")):
            evs = [torch.cuda.Event(enable_timing=True) for _ in range(13)]
            for q in range(12):
                evs[q].record()
                memory_gym_amd._native.check(memory_gym_amd._native.LIB.mg_store_probe(C.c_void_p(env.obs.data_ptr()), n_local, pattern, stream), "mg_store_probe")
            evs[12].record()
            torch.cuda.synchronize()
            ms = sorted(evs[q].elapsed_time(evs[q + 1]) for q in range(3, 12))
            box[name] = {"best_GBps": FRAME * n_local / (ms[0] * 1e-3) / 1e9, "median_GBps": FRAME * n_local / (ms[len(ms) // 2] * 1e-3) / 1e9,
                         "best_ms": ms[0], "median_ms": ms[len(ms) // 2]}
        env.step(acts[0])  # (the probes zeroed the observations: one step redraws every frame)
    # BASELINE.md section 3's stated timing, whatever --steps was: one window of >= 2,000 steps, same event pair, same stream
    dt_long = 0.0
    if long_window and not (gather and dist_on):
        la, lb = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        fence()
        la.record()
        for k in range(long_window):
            one_step(settle + W + 7 * K + 32 + k)
        lb.record()
        fence()
        dt_long = la.elapsed_time(lb) * 1e-3
    t = torch.tensor([dt, dt_long], device=dev, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt_max, dt_long = float(t[0].item()), float(t[1].item())
    extra = {}
    path_gen = None
    if path0 is not None:  # (read right behind the timed region: it covers the K timed steps and the five spread windows)
        ticks = env.debug_counter("path_gen_ticks") - path0[0]
        paths = env.debug_counter("path_gen_paths") - path0[1]
        steps_seen = K * (1 + len(windows)) + (32 if (events and not in_region) else 0)
        path_gen = {"paths_per_step": paths / steps_seen, "wave_us_per_step": ticks / 100.0 / steps_seen,
                    "wave_us_per_path": (ticks / 100.0 / paths) if paths else None, "steps": steps_seen}
    if count_done:  # after the timed region: how many instances finish per step, and what resetting that many costs
        for k in range(100):
            env.step(acts[k % n_act_bufs])
            done_sum += env.done_u8.sum()
        torch.cuda.synchronize()
        per_step = float(done_sum.item()) / 100.0
        mask = torch.ones(n_local, dtype=torch.bool, device=dev)
        e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        reset_ms, render_ms = [], []
        for _ in range(5):
            e0.record()
            env.reset(mask=mask)   # reset kernel (path generation for every instance) + raster
            e1.record()
            env.render()           # the raster alone
            e2.record()
            torch.cuda.synchronize()
            reset_ms.append(e0.elapsed_time(e1))
            render_ms.append(e1.elapsed_time(e2))
        per_reset_us = max(0.0, (min(reset_ms) - min(render_ms))) * 1e3 / n_local
        extra = {"resets_per_step": per_step, "reset_kernel_us_per_instance": per_reset_us,
                 "full_reset_kernel_ms": max(0.0, min(reset_ms) - min(render_ms))}
    out = {"env_id": env_id, "n_local": n_local, "n_total": n_total, "seconds": dt_max, "value": n_total * K / dt_max,
           "wall_ms_per_step": dt_wall / K * 1e3, "timing": ("host clock between the fences (a collective runs beside the launch stream)"
                                                               if (gather and dist_on) else "hipEvent pair on the launch stream around the K steps"),
           "extra": extra, "value_windows": windows, "path_gen": path_gen, "box": box, "setup_steps": settle,
           "settle_windows_ms": settle_windows,
           "ms_per_step": dt_max / K * 1e3, "raster_avg_ms": raster_ms / raster_n if raster_n else None, "raster_launches": raster_n,
           "logic_avg_ms": logic_ms / logic_n if logic_n else None, "event_region": region,
           "obs_placement": getattr(env, "obs_placement_info", None), "gather": gather if dist_on else None, "note": note,
           "value_long": ({"value": n_total * long_window / dt_long, "steps": long_window, "seconds": dt_long, "ms_per_step": dt_long / long_window * 1e3,
                           "timing": "hipEvent pair on the launch stream around the window, max over ranks; right behind the roofline's bracketed steps"}
                          if dt_long else None)}
    if policy:
        out["policy"] = policy
    if gatherer is not None and rank == 0:  # the frames rank 0 received in the last step equal the ranks' own (rank 0's: checked here)
        got, grew, gdone = gatherer.gathered_step()
        k_last = (gatherer.t - 1) & 1
        out["gather_check"] = bool(torch.equal(got[0], gatherer.bufs[k_last]) and torch.equal(grew[0], gatherer.packed[k_last][1])
                                   and torch.equal(gdone[0], gatherer.packed[k_last][2].view(torch.bool)))
        out["gathered"] = "obs + reward (f32) + done (u8): %d + 5 B per instance" % (got[0][0].numel() * got[0].element_size())
    env.close()
    del env, gatherer, peer
    torch.cuda.empty_cache()
    return out


def secondary_workloads(primary, dev, settle, traffic=True):
    """The other single-GPU BASELINE configs (C3, C4, the per-GPU shard of C5), measured the same way after the headline
    run so that one bench line carries them; informational (the contract's `value` is the headline workload's)."""
    out = []
    for env_id, label in (("MysteryPath-v0", "C3"), ("Endless-SearingSpotlights-v0", "C4"), ("Endless-MortarMayhem-v0", "C5 per-GPU shard")):
        if env_id == primary:
            continue
        r = run_workload(env_id, DEFAULT_ENVS[env_id], 200, 30, settle, 1, 0, dev, count_done=(label == "C3"))
        entry = {"config": label, "workload": "%s, %d envs" % (env_id, r["n_local"]), "value": r["value"], "unit": "env steps/s",
                 "ms_per_step": r["ms_per_step"], "raster_avg_ms": r["raster_avg_ms"], "logic_avg_ms": r["logic_avg_ms"],
                 "raster_GBps": (FRAME + DESC_BYTES[env_id]) * r["n_local"] / (r["raster_avg_ms"] * 1e-3) / 1e9 if r["raster_avg_ms"] else None,
                 "value_windows": r["value_windows"],
                 "obs_placement_zones": (r["obs_placement"] or {}).get("zones")}
        # the same roofline figures as the headline's: SURVEY.md 8(d) bytes per instance-step over the whole step, the dominant
        # launch's algorithmic bytes (frame + descriptor) over its own time, and its HBM traffic from two rocprofv3 child passes
        rl = {"bytes_per_step": STEP_BYTES[env_id], "peak": HBM_PEAK_GBPS, "unit": "GB/s",
              "frac_whole_step": STEP_BYTES[env_id] * r["n_local"] / (r["ms_per_step"] * 1e-3) / 1e9 / HBM_PEAK_GBPS,
              "dominant_kernel": "step + raster in one launch" if not r["logic_avg_ms"] else "raster",
              "frac_dominant_kernel": (entry["raster_GBps"] / HBM_PEAK_GBPS) if entry["raster_GBps"] else None, "traffic": None}
        if r.get("box") and entry["raster_GBps"]:
            # the number that bounds stores on this box is the LINEAR fill; the frame-shaped probe is a control (the same store shape
            # without compose work), not a ceiling: a raster with compose work can beat it
            rl["frac_of_linear_fill"] = entry["raster_GBps"] / r["box"]["linear_fill"]["median_GBps"]
            rl["box_probe_GBps"] = {"linear_fill": r["box"]["linear_fill"]["median_GBps"], "frame_probe": r["box"]["frame_shaped"]["median_GBps"]}
            rl["frac_of_frame_probe"] = entry["raster_GBps"] / r["box"]["frame_shaped"]["median_GBps"]
        if traffic:
            tb, meta = measure_traffic(env_id, r["n_local"])
            rl["traffic"], rl["traffic_source"] = tb, meta.get("source")
            rl["traffic_over_bytes_per_launch"] = tb / ((FRAME + DESC_BYTES[env_id]) * r["n_local"]) if tb else None
        entry["roofline"] = rl
        if label == "C3" and r["extra"]:
            # "report the reset (A* path-gen) kernel's share of time separately".  The path generation
            # of an auto-reset runs inside the step's logic kernel (served by the instance's wave), so its share is reported
            # as (instances that reset per step) x (cost of one reset, from a masked mg_reset of the whole batch: reset kernel
            # = mg_reset - raster) over the step time, next to the logic kernel's own time.
            x = r["extra"]
            est = x["resets_per_step"] * x["reset_kernel_us_per_instance"] * 1e-3 / r["ms_per_step"]
            entry["reset_share_detail"] = dict(x, standalone_estimate=est, method="estimate: resets per step x (masked full-batch mg_reset - mg_render) / instances, over ms_per_step",
                                               logic_share=(r["logic_avg_ms"] / r["ms_per_step"]) if r["logic_avg_ms"] else None)
            entry["reset_share"] = est
            if r.get("path_gen"):
                # MEASURED in the launches themselves: the path generator's waves stamp the real-time clock around every path
                # (mg_debug_counter "path_gen_ticks" / "path_gen_paths"); share = wave-time spent generating paths over the
                # wave-time the chip offers during a step (256 CUs x 7 workgroups x 4 waves resident)
                pg = dict(r["path_gen"])
                pg["chip_wave_us_per_step"] = r["ms_per_step"] * 1e3 * 256 * 7 * 4
                pg["share_of_wave_time"] = pg["wave_us_per_step"] / pg["chip_wave_us_per_step"]
                pg["method"] = "measured in the timed launches: per-wave real-time-clock stamps around every path generation"
                entry["reset_share"] = pg["share_of_wave_time"]
                entry["reset_share_measured"] = pg
        out.append(entry)
    return out


def other_workloads(primary, dev, settle):
    """Informational entries measured like the secondary workloads, without the PMC child passes (profiles/ holds those):
    the three env ids that are in no BASELINE config; Endless-MysteryPath-v0 once more under an agent that FOLLOWS its path (round 6:
    the regime a trained agent puts the library in -- segments appended every few steps, hardly a reset -- next to the random-action
    one, so that a store flavour or a generator is judged on both); the headline workload in the two fused float formats
    (value / 255 in CHW order written by the raster's stream-out instead of a second pass), priced at the bytes they
    write per instance; and the headline workload behind the gymnasium vector front end (terminal observations)."""
    out = []
    plan = [(env_id, "u8_xyc", None) for env_id in ("SearingSpotlights-v0", "Endless-MysteryPath-v0", "MysteryPath-Grid-v0") if env_id != primary]
    plan.append(("Endless-MysteryPath-v0", "u8_xyc", "follower:0.02"))
    plan += [("MortarMayhem-Grid-v0", "f32_chw", None), ("MortarMayhem-Grid-v0", "bf16_chw", None)]
    plan.append(("MortarMayhem-Grid-v0", "u8_xyc", "vector_api"))  # (round 6) the same handle behind GymnasiumVectorEnv.step
    plan.append(("Endless-MysteryPath-v0", "u8_xyc", "vector_api"))  # ... and the id the convention costs most (a sparse raster launch of terminal frames)
    for env_id, fmt, policy in plan:
        r = run_workload(env_id, DEFAULT_ENVS[env_id], 200, 30, settle, 1, 0, dev, obs_format=fmt, policy=policy)
        algo = FRAME * OBS_ELEM[fmt] + DESC_BYTES[env_id]  # algorithmic bytes of the dominant launch per instance-step (frame + descriptor)
        dom = algo * r["n_local"] / (r["raster_avg_ms"] * 1e-3) / 1e9 if r["raster_avg_ms"] else None
        rl = {"bytes_per_launch_per_instance": algo, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "dominant_kernel_GBps": dom,
              "frac_dominant_kernel": dom / HBM_PEAK_GBPS if dom else None,
              "frac_whole_step_frame_bytes_only": algo * r["n_local"] / (r["ms_per_step"] * 1e-3) / 1e9 / HBM_PEAK_GBPS,
              "traffic": None, "traffic_note": "not measured in this run: profiles/ holds the PMC passes"}
        if r.get("box") and dom:
            rl["frac_of_linear_fill"] = dom / r["box"]["linear_fill"]["median_GBps"]
            rl["box_probe_GBps"] = {"linear_fill": r["box"]["linear_fill"]["median_GBps"], "frame_probe": r["box"]["frame_shaped"]["median_GBps"]}
            rl["frac_of_frame_probe"] = dom / r["box"]["frame_shaped"]["median_GBps"]
        e = {"workload": "%s, %d envs%s" % (env_id, r["n_local"], "" if fmt == "u8_xyc" else ", observations as %s" % fmt), "obs_format": fmt,
             "policy": policy or "uniform random", "value": r["value"], "unit": "env steps/s", "ms_per_step": r["ms_per_step"],
             "raster_avg_ms": r["raster_avg_ms"], "logic_avg_ms": r["logic_avg_ms"], "value_windows": r["value_windows"],
             "obs_placement_zones": (r["obs_placement"] or {}).get("zones"), "roofline": rl}
        if policy == "vector_api":
            e["policy"] = "uniform random"
            e["api"] = "memory_gym_amd.vector.GymnasiumVectorEnv.step"
            e["workload"] += ", gymnasium vector convention"
            e["api_note"] = ("terminal observations kept in infos['final_observation'], finished instances reset in the same call: mg_step with "
                             "mg_info_buffers.final_obs_dev -- the step's own launches draw a finishing instance's terminal frame as well as its "
                             "reset frame (Endless-MysteryPath: the terminal frames by one sparse raster launch behind them); the dominant launch "
                             "is the step's, the rest of ms_per_step is what the convention adds")
        elif policy:
            e["policy_note"] = ("actions from the previous step's info['ground_truth'] (one-hot right / up / down), a random one with probability %s; "
                                "the policy's three small torch kernels per step run on the launch stream inside the timed region" % policy.split(":")[1])
        out.append(e)
    return out


def c1_leg(episodes=1000):
    """BASELINE config C1 (the reference's own loop, /root/reference/bench.py:12-30, restated in tests/c1_loop.py): one
    MortarMayhem-Grid-v0 instance, reset(seed=1), actions from Generator(PCG64(12345)), `episodes` episodes with the resets
    inside the timed region.  The HIP single-instance adapter here, the CPU oracle (one thread) in a subprocess; both walk
    the same episodes (`steps` must agree)."""
    import c1_loop

    hip = c1_loop.run("hip", "MortarMayhem-Grid-v0", episodes)
    out = {"recipe": "MortarMayhem-Grid-v0 x1, reset(seed=1), PCG64(12345) actions, %d episodes, resets inside the timed region" % episodes,
           "hip_adapter": hip,
           "note": "one instance is latency-bound on a GPU (one or two launches and one device->host round trip per step); the batched "
                   "path is the product, this leg is the plumbing check BASELINE.md asks for"}
    try:
        env = dict(os.environ, OMP_NUM_THREADS="1")
        p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "c1_loop.py"), "--backend", "oracle", "--episodes", str(episodes), "--json"],
                           env=env, capture_output=True, text=True, timeout=300)
        line = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
        out["cpu_oracle_1_thread"] = json.loads(line[-1])
        out["same_episodes"] = out["cpu_oracle_1_thread"]["steps"] == hip["steps"]
    except Exception as e:
        out["cpu_oracle_1_thread"] = "unavailable: %s" % e
    return out


def under_rocprof():
    return any(k.startswith("ROCPROF") for k in os.environ) or "rocprofiler" in os.environ.get("LD_PRELOAD", "")


def measure_traffic(argv_env, n_local):
    """roofline.traffic measured for THIS workload on THIS box: two child passes of bench.py under rocprofv3 (--pmc WRITE_SIZE
    and --pmc FETCH_SIZE need separate passes: TCC slots), --kernel-trace only next to them, as MI355X_MICROARCH.md's HBM
    section prescribes; counters are in KiB, FETCH_SIZE is doubled on gfx950.  Average over the raster launches of the child's
    timed region.  Returns (bytes per launch or None, description dict)."""
    import shutil
    import sqlite3
    import tempfile

    exe = shutil.which("rocprofv3")
    if not exe:
        return None, {"source": "rocprofv3 not on PATH"}
    if under_rocprof():
        return None, {"source": "this process already runs under rocprofv3: no nested passes"}
    meta, vals = {}, {}
    for counter in ("WRITE_SIZE", "FETCH_SIZE"):
        d = tempfile.mkdtemp(prefix="memgym_pmc_", dir="/tmp")
        cmd = [exe, "--pmc", counter, "--kernel-trace", "-d", d, "-o", "p", "--", sys.executable, os.path.abspath(__file__),
               "--env", argv_env, "--envs-per-gpu", str(n_local), "--steps", "24", "--warmup", "4", "--settle", "60", "--no-cpu-baseline",
               "--no-secondary", "--no-events", "--no-traffic", "--no-c1"]
        try:
            p = subprocess.run(cmd, capture_output=True, text=True, timeout=240, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"))
            db_path = None
            for root, _, files in os.walk(d):
                for f in files:
                    if f.endswith("_results.db"):
                        db_path = os.path.join(root, f)
            if db_path is None:
                raise RuntimeError("no results.db (rc %d): %s" % (p.returncode, p.stderr[-200:]))
            db = sqlite3.connect(db_path)

            def table(prefix):
                for (n,) in db.execute("select name from sqlite_master where type='table'"):
                    if n.startswith(prefix + "_0") or n == prefix:
                        return n
                raise KeyError(prefix)
            kd, ks, pe, pi = table("rocpd_kernel_dispatch"), table("rocpd_info_kernel_symbol"), table("rocpd_pmc_event"), table("rocpd_info_pmc")
            rows = db.execute("select e.value, d.end - d.start from %s e join %s q on e.pmc_id = q.id join %s d on d.event_id = e.event_id "
                              "join %s s on d.kernel_id = s.id where q.name = ? and s.display_name like '%%raster%%' order by d.start desc limit 20"
                              % (pe, pi, kd, ks), (counter,)).fetchall()
            if not rows:
                raise RuntimeError("no raster dispatches with %s in the trace" % counter)
            vals[counter] = sum(r[0] for r in rows) / len(rows) * 1024.0
            meta[counter + "_launches"] = len(rows)
            meta[counter + "_pass_raster_avg_us"] = sum(r[1] for r in rows) / len(rows) / 1e3
            line = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
            if line:  # where the child's observation buffer lay (the PMC passes are perturbed: see profiles/r03_pmc_passes.md)
                meta[counter + "_pass_obs_placement"] = json.loads(line[-1]).get("obs_placement")
        except Exception as e:
            meta[counter + "_error"] = str(e)[:300]
        finally:
            shutil.rmtree(d, ignore_errors=True)
    if "WRITE_SIZE" in vals and "FETCH_SIZE" in vals:
        meta.update(source="MEASURED in this run: two child passes of this command under rocprofv3 (--pmc WRITE_SIZE / --pmc FETCH_SIZE, "
                           "--kernel-trace), last 20 raster launches; counters in KiB, FETCH_SIZE doubled (gfx950, MI355X_MICROARCH.md)",
                    write_bytes=vals["WRITE_SIZE"], fetch_bytes_corrected_x2=2 * vals["FETCH_SIZE"])
        return vals["WRITE_SIZE"] + 2 * vals["FETCH_SIZE"], meta
    meta.setdefault("source", "rocprofv3 child passes failed")
    return None, meta



# ---------------------------------------------------------------------------------------------------------------- the one line
class LineGuard:
    """Rank 0 owes the driver exactly ONE JSON line, and an N-GPU run must not be able to end without it (the
    config-5 legs run behind the headline, and a rank that fails or hangs in one of them used to leave the others waiting in a
    collective until the driver's limit).  The line that WOULD be printed right now is kept as a string (`set_line`); it leaves
    the process exactly once -- at the normal end (`finish`) or from a watchdog thread that runs beside the main thread (which may
    sit in a collective or a device synchronisation for ever) and fires when
      * the armed deadline of the current phase passes (`arm`),
      * some rank announced a failure through the rendezvous store (`abort`; a host-side TCP store, nothing a hung GPU can block),
      * the process is told to end (SIGTERM / SIGINT: torch.distributed.run and `self_launch` send it to the surviving ranks when
        one rank dies) -- seen through signal.set_wakeup_fd, which the C-level handler writes whatever the main thread is doing.
    Ranks other than 0 print nothing; their watchdogs only end them."""

    def __init__(self, rank, world):
        import signal
        import threading

        self.rank, self.world = rank, world
        self.line = None            # JSON text of the line as it stands
        self.printed = False
        self.lock = threading.Lock()
        self.deadline, self.what = None, ""
        self.store = None
        self.done = False
        self.rfd, self.wfd = os.pipe()
        os.set_blocking(self.wfd, False)
        try:
            for sig in (signal.SIGTERM, signal.SIGINT):
                signal.signal(sig, lambda *_: None)  # (a Python-level handler must exist for the C-level one to write the wake-up byte)
            signal.set_wakeup_fd(self.wfd, warn_on_full_buffer=False)
        except ValueError:  # not the main thread (imported by a test)
            pass
        self.thread = threading.Thread(target=self._watch, name="bench-line-guard", daemon=True)
        self.thread.start()

    # -- what the main thread calls
    def set_line(self, out):
        text = json.dumps(out)
        with self.lock:
            self.line = text

    def arm(self, seconds, what):
        self.deadline, self.what = time.monotonic() + seconds, what

    def disarm(self):
        self.deadline = None

    def attach_store(self, store):
        self.store = store

    def abort(self, reason):
        """tell every rank (through the store) that this run cannot go on; the watchdogs end the ranks, rank 0 prints first"""
        try:
            if self.store is not None:
                self.store.set("memgym_bench/abort", "rank %d: %s" % (self.rank, reason))
        except Exception:
            pass

    def agree(self, key, ok, why="", timeout_s=120.0):
        """All ranks say ok / not ok under `key` through the store and read each other's word: (True, "") if every rank said ok.
        Host side only -- no collective, so a rank that could not even build its buffers is heard by ranks that could."""
        if self.store is None or self.world == 1:
            return ok, why
        from datetime import timedelta
        try:
            self.store.set("memgym_bench/%s/%d" % (key, self.rank), "1" if ok else ("0" + why[:200]))
            keys = ["memgym_bench/%s/%d" % (key, r) for r in range(self.world)]
            self.store.wait(keys, timedelta(seconds=timeout_s))
            words = [self.store.get(k).decode() for k in keys]
        except Exception as e:
            return False, "no agreement on %s within %.0f s: %s" % (key, timeout_s, str(e)[:120])
        bad = ["rank %d: %s" % (r, w[1:] or "failed") for r, w in enumerate(words) if not w.startswith("1")]
        return not bad, "; ".join(bad)

    def finish(self, out=None):
        if out is not None:
            self.set_line(out)
        self.done = True
        self._emit(None)

    # -- the watchdog
    def _emit(self, reason):
        with self.lock:
            if self.printed:
                return
            self.printed = True
            if self.rank != 0 or self.line is None:
                return
            text = self.line
            if reason:
                j = json.loads(text)
                j["ended_early"] = reason
                text = json.dumps(j)
            sys.stdout.write(text + "\n")
            sys.stdout.flush()

    def _bail(self, reason, rc):
        had_line = self.line is not None
        self._emit(reason)
        if self.rank == 0 or reason:
            sys.stderr.write("bench.py rank %d: ending early: %s\n" % (self.rank, reason))
            sys.stderr.flush()
        os._exit(0 if (had_line or self.rank != 0) and rc == 0 else (rc or 3))

    def _watch(self):
        import select
        last_store = 0.0
        while not self.done:
            r, _, _ = select.select([self.rfd], [], [], 0.25)
            if self.done:
                return
            if r:
                sig = os.read(self.rfd, 16)
                self._bail("signal %s while %s" % (",".join(str(b) for b in sig), self.what or "running"), 0)
            now = time.monotonic()
            if self.deadline is not None and now > self.deadline:
                self.abort("watchdog: %s did not finish in time" % self.what)
                self._bail("watchdog: %s did not finish within its limit (a rank hung or died)" % self.what, 0)
            if self.store is not None and now - last_store > 1.0:
                last_store = now
                try:
                    if self.store.check(["memgym_bench/abort"]):
                        self._bail("aborted: " + self.store.get("memgym_bench/abort").decode(), 0)
                except Exception as e:  # the store's server (rank 0 / the launcher) is gone
                    if self.rank != 0:
                        self._bail("the rendezvous store is gone (%s)" % str(e)[:80], 0)


def _fault(leg, rank):
    """Test hook (tests/test_bench_guard.py): MEMGYM_BENCH_TEST_FAULT=<rank>:<leg>:<exit|raise|hang> makes that rank fail inside that leg."""
    spec = os.environ.get("MEMGYM_BENCH_TEST_FAULT")
    if not spec:
        return
    r, lg, mode = spec.split(":")
    if int(r) != rank or lg != leg:
        return
    if mode == "exit":
        os._exit(17)
    if mode == "raise":
        raise RuntimeError("injected failure in leg " + leg)
    if mode == "hang":
        while True:
            time.sleep(1.0)


def fake_workload(env_id, n_local, K, W, settle, world, rank, dev, obs_format="u8_xyc", gather=None, events=True, event_stride=8,
                  count_done=False, leg="headline", long_window=0, policy=None):
    """MEMGYM_BENCH_FAKE=1 (tests/test_bench_guard.py, no GPU): the control flow of an N-rank run -- rendezvous, agreement, legs with
    collectives, the one line -- with a stand-in for the measurement: a few gloo all-reduces and a constant rate."""
    import torch
    import torch.distributed as dist

    t = torch.ones(1)
    for k in range(3):
        if world > 1:
            dist.all_reduce(t)
        if k == 1:
            _fault(leg, rank)
        time.sleep(0.05)
    n_total = n_local * world
    return {"env_id": env_id, "n_local": n_local, "n_total": n_total, "seconds": K * 1e-4, "value": n_total / 1e-4, "wall_ms_per_step": 0.1,
            "timing": "FAKE (MEMGYM_BENCH_FAKE=1): no measurement", "extra": {}, "value_windows": [], "path_gen": None, "box": None,
            "setup_steps": settle, "settle_windows_ms": [], "ms_per_step": 0.1, "raster_avg_ms": None, "raster_launches": 0, "logic_avg_ms": None,
            "event_region": None, "obs_placement": None, "gather": gather if world > 1 else None, "note": None, "value_long": None}


def rank_report(dev, backend):
    """What this rank runs on, for the line's `ranks` (gathered from every rank through the store)."""
    import torch

    d = {"rank": int(os.environ.get("RANK", "0")), "local_rank": int(os.environ.get("LOCAL_RANK", "0")), "pid": os.getpid(), "device": str(dev)}
    try:
        if dev.type == "cuda":
            p = torch.cuda.get_device_properties(dev)
            d.update(name=p.name, arch=getattr(p, "gcnArchName", None), hbm_GiB=round(p.total_memory / 2**30, 1), cus=p.multi_processor_count,
                     pci_bus_id=getattr(p, "pci_bus_id", None), uuid=str(getattr(p, "uuid", "")) or None)
            d["peer_access_to_rank0_device"] = bool(dev.index == 0 or torch.cuda.can_device_access_peer(dev.index, 0)) if torch.cuda.device_count() > 1 else None
    except Exception as e:
        d["error"] = str(e)[:120]
    return d


# ---------------------------------------------------------------------------------------------------------------- launch
def self_launch(args):
    """`python bench.py --gpus N` without a launcher: spawn the N ranks (one per GPU); rank 0 prints the JSON line.  All ranks are
    watched together: when one ends with an error the others get a few seconds to notice (the store's abort word) and are then told
    to end (SIGTERM -- their LineGuard prints rank 0's line first); nothing here waits for a hung rank longer than that."""
    import signal

    if os.environ.get("MEMGYM_BENCH_FAKE"):
        have = args.gpus
    else:
        import torch
        have = torch.cuda.device_count()
    if os.environ.get("MEMGYM_BENCH_ONE_DEVICE"):  # plumbing test: every rank on GPU 0 (tests/test_gpu_bench_ranks.py)
        have = args.gpus
    if have < args.gpus:
        print("bench.py: --gpus %d but this host shows %d GPU(s)" % (args.gpus, have), file=sys.stderr)
        return 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    first_bad = None
    try:
        while any(p.poll() is None for p in procs):
            time.sleep(0.2)
            bad = [p for p in procs if p.poll() not in (None, 0)]
            if bad and first_bad is None:
                first_bad = time.monotonic()
            if first_bad is not None and time.monotonic() - first_bad > 8.0:
                for p in procs:
                    if p.poll() is None:
                        p.send_signal(signal.SIGTERM)
                t_end = time.monotonic() + 10.0
                while any(p.poll() is None for p in procs) and time.monotonic() < t_end:
                    time.sleep(0.1)
                break
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    rc0 = procs[0].wait()
    return rc0  # rank 0's word: 0 when its line went out (a failed leg is IN the line: `ended_early` / `config5`)


def build_headline(args, r, env_id, n_local, K, W, world, ranks):
    """rank 0: the contract's line from the headline measurement `r` (everything the later legs add is optional)."""
    obs_elem = OBS_ELEM[args.obs_format]
    gather_txt = ""
    if r["gather"]:
        gather_txt = ", peer-mapped obs stores into rank 0's HBM" if r["gather"] == "peer" else ", RCCL obs gather to rank 0"
    out = {
        "metric": "env steps/sec (aggregate)", "value": r["value"], "unit": "env steps/s", "n_gpus": world, "steps": K,
        "warmup": W, "ms_per_step": r["ms_per_step"], "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "u8", "data": "synthetic", "obs_format": args.obs_format, "setup_steps": r["setup_steps"],
        "config": {"workload": "%s, %d envs/GPU x %d GPU, 84x84x3 obs (%s), same-step auto-reset, uniform random "
                               "actions generated on device%s" % (env_id, n_local, world, args.obs_format, gather_txt),
                   "env_id": env_id, "envs_per_gpu": n_local, "envs_total": r["n_total"],
                   "parallelism": "env-sharded x%d, no data-path collective" % world if not r["gather"] else
                   "env-sharded x%d + %s(obs + reward + done)->rank0" % (world, "peer-mapped stores / packed gather" if r["gather"] == "peer" else "gather")},
        "per_gpu_value": r["value"] / world, "timing": r["timing"], "wall_ms_per_step": r["wall_ms_per_step"],
        # five consecutive windows of K steps each right behind the timed one (this rank's share x N for N > 1): the spread a
        # K-step headline carries
        "value_windows": r["value_windows"],
        # BASELINE.md section 3's stated timing (>= 2,000 timed steps), whatever --steps was: one such window, timed the same way
        "value_2000": r.get("value_long"),
        "policy": r.get("policy") or "uniform random",
        # who ran: every rank's device as that rank reports it, the backend, the collective library's version
        "ranks": ranks,
    }
    if r["note"]:
        out["note"] = r["note"]
    if "gather_check" in r:
        out["gather_check"] = r["gather_check"]
        out["gathered"] = r.get("gathered")
    if r["raster_launches"]:
        avg_ms = r["raster_avg_ms"]
        # Algorithmic bytes of the timed launch per instance-step.  Mortar family: the step's workgroups ride in front of the
        # raster's in ONE launch, so that launch moves the whole step's bytes: SURVEY.md 8(d)'s figure (MortarMayhem-Grid
        # 21,305 B; rounds 3-4 priced it at the raster's 21,184 only).  Two-launch families: the raster's frame + descriptor.
        one_launch = not r["logic_avg_ms"]
        per_inst = (STEP_BYTES.get(env_id, FRAME + DESC_BYTES.get(env_id, 16)) + FRAME * (obs_elem - 1)) if one_launch \
            else (FRAME * obs_elem + DESC_BYTES.get(env_id, 16))
        rb = per_inst * n_local
        achieved = rb / (avg_ms * 1e-3) / 1e9
        traffic, traffic_source, traffic_meta = None, None, None
        if world == 1 and obs_elem == 1 and not args.no_traffic:
            traffic, traffic_meta = measure_traffic(env_id, n_local)
            traffic_source = traffic_meta.get("source")
        pmc = os.path.join(ROOT, "profiles", "pmc_latest.json")
        if traffic is None and os.path.exists(pmc):
            try:
                j = json.load(open(pmc))
                if j.get("env_id") == env_id and j.get("envs_per_gpu") == n_local and obs_elem == 1:
                    traffic = j.get("hbm_bytes_per_launch")
                    traffic_source = ("NOT measured in this run: rocprofv3 --pmc WRITE_SIZE / FETCH_SIZE passes of this workload, "
                                      "%s (profiles/pmc_latest.json, regenerated by tools/profile_round.sh)" % j.get("source", "committed profile"))
            except Exception:
                traffic = None
        # (mortar family: the step's workgroups ride in front of the raster's in ONE launch, so `avg_launch_ms` is the whole
        # step's launch and there is no separate logic kernel to time; the algorithmic bytes stay the raster's)
        out["roofline"] = {"bound": "hbm", "kernel": "step + raster in one launch (rank 0)" if one_launch else "raster (rank 0)", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                           "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic, "traffic_source": traffic_source,
                           "traffic_passes": traffic_meta,
                           "bytes_per_launch": rb, "bytes_per_instance_step": per_inst, "avg_launch_ms": avg_ms, "launches": r["raster_launches"],
                           "event_region": r["event_region"], "logic_kernel_avg_ms": r["logic_avg_ms"],
                           "whole_step_GBps": (STEP_BYTES.get(env_id, FRAME) + FRAME * (obs_elem - 1)) * r["n_total"] / (r["seconds"] / K) / 1e9}
        if r.get("box"):
            b = r["box"]
            # frac_of_linear_fill leads: the linear 16-byte fill is what bounds stores on this box and this placement.  The frame
            # probe (the raster's store shape without compose work) is a CONTROL, not a ceiling -- a raster with compose work can
            # beat it (round 5: 1.016 for C3), so "0.95 of it" does not mean "5 % left" (it was `frac_of_box_ceiling`).
            out["roofline"]["frac_of_linear_fill"] = achieved / b["linear_fill"]["median_GBps"]
            out["roofline"]["box_probe_GBps"] = {"linear_fill": b["linear_fill"]["median_GBps"], "frame_probe": b["frame_shaped"]["median_GBps"],
                                                 "linear_fill_best": b["linear_fill"]["best_GBps"], "frame_probe_best": b["frame_shaped"]["best_GBps"],
                                                 "how": "mg_store_probe over this run's observation buffer (%d frames), median / best of 9 launches each, right behind the timed region" % n_local}
            out["roofline"]["frac_of_frame_probe"] = achieved / b["frame_shaped"]["median_GBps"]
    if r["obs_placement"]:  # mg_obs_alloc: observation buffer assembled from pieces in different HBM zones
        out["obs_placement"] = r["obs_placement"]
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=30)
    ap.add_argument("--settle", type=int, default=200, help="setup steps after the reset (episodes de-synchronise); not timed, not warm-up")
    ap.add_argument("--env", default="MortarMayhem-Grid-v0")
    ap.add_argument("--envs-per-gpu", type=int, default=0)
    ap.add_argument("--obs-format", default="u8_xyc", choices=sorted(OBS_ELEM),
                    help="raster stream-out format; the BASELINE.json metric is quoted on the default (the reference's uint8 obs)")
    ap.add_argument("--gather", nargs="?", const="rccl", default=None, choices=["rccl", "peer"],
                    help="observation gather to rank 0 every step for the HEADLINE workload: 'rccl' = torch.distributed.gather, "
                         "'peer' = the raster kernels store straight into rank 0's HBM (memory_gym_amd.dist.PeerObsBuffer)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the informational C3 / C4 / C5 measurements")
    ap.add_argument("--no-other", action="store_true", help="skip the informational measurements of the env ids that are in no BASELINE config")
    ap.add_argument("--no-events", action="store_true", help="do not bracket kernels with HIP events")
    ap.add_argument("--no-traffic", action="store_true", help="do not measure roofline.traffic with rocprofv3 child passes")
    ap.add_argument("--no-c1", action="store_true", help="skip the single-instance C1 leg")
    ap.add_argument("--policy", default=os.environ.get("MEMGYM_BENCH_POLICY") or None,
                    help="actions of the HEADLINE workload: default uniform random (BASELINE's); 'follower:EPS' = read off info['ground_truth'] (Endless-MysteryPath-v0)")
    ap.add_argument("--long-window", type=int, default=2000, help="steps of the extra window behind the timed region (value_2000; 0 = none)")
    ap.add_argument("--config5-envs", type=int, default=32768, help="instances per GPU of the config-5 legs of an N > 1 run (BASELINE: 32,768)")
    ap.add_argument("--config5-steps", type=int, default=100)
    ap.add_argument("--leg-limit", type=float, default=240.0, help="seconds a config-5 leg (and the rendezvous) may take before the watchdog prints the line and ends the run")
    ap.add_argument("--event-stride", type=int, default=8, help="bracket every N-th step with HIP events (each bracketed step costs ~15 us)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend for N > 1 (nccl == RCCL; gloo only for plumbing tests with all ranks on one GPU)")
    ap.add_argument("--cpu-worker", nargs=3, metavar=("ENV", "THREADS", "SECONDS"), help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.cpu_worker:
        return cpu_worker(args.cpu_worker[0], int(args.cpu_worker[1]), float(args.cpu_worker[2]))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args))

    from datetime import timedelta

    # The observation buffer's placement search (include/memgym.h: mg_obs_alloc_for) is bounded at 1.5 s by default -- a trainer's start-up
    # should not hang on it -- and on a box whose memory earlier processes have dirtied (the driver wipes what it hands out, ~27 ms per GiB)
    # that bound can end the walk inside the first zone: the buffer is then a plain allocation, 0-20 % slower depending on where it lands
    # (round 6, eight processes in a row on one box: 282-289 M with two zones, 286.6 and 231.5 M on the two one-zone fallbacks).  A benchmark
    # run can afford the walk (setup, not timed): 8 s unless the caller has set a bound.  `obs_placement` in the line says what was found.
    os.environ.setdefault("MEMGYM_OBS_SEARCH_MS", "8000")

    import torch
    import torch.distributed as dist

    fake = bool(os.environ.get("MEMGYM_BENCH_FAKE"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("MEMGYM_BENCH_ONE_DEVICE"):
        local_rank = 0
    guard = LineGuard(rank, world)
    guard.arm(args.leg_limit, "the rendezvous")
    # a collective that never completes must not take rank 0 down before its line is out: the only thing that ends a hung rank of
    # this program is the LineGuard (which prints first).  c10d's own watchdog would abort the process at its timeout.
    os.environ.setdefault("TORCH_NCCL_ASYNC_ERROR_HANDLING", "0")
    pg_timeout = timedelta(seconds=120)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if not fake:
            torch.cuda.set_device(local_rank)
        if args.backend == "nccl" and not fake:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank), timeout=pg_timeout)
        else:
            dist.init_process_group("gloo", timeout=pg_timeout)
    else:
        if not fake:
            torch.cuda.set_device(0)
        if args.gather == "rccl" and "MASTER_PORT" in os.environ:  # a one-rank RCCL group: the gather path on a single GPU
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            dist.init_process_group(args.backend, rank=0, world_size=1, timeout=pg_timeout, **({"device_id": torch.device("cuda", 0)} if args.backend == "nccl" else {}))
    if args.gpus != world and rank == 0:
        print("warning: --gpus %d but WORLD_SIZE %d (the launcher decides)" % (args.gpus, world), file=sys.stderr)
    dev = torch.device("cpu") if fake else torch.device("cuda", local_rank)
    backend = (dist.get_backend() if dist.is_initialized() else None)

    # every rank's report, through the rendezvous store (host side)
    ranks = {"world": world, "backend": backend, "devices": [rank_report(dev, backend)]}
    if dist.is_initialized():
        try:
            from torch.distributed.distributed_c10d import _get_default_store
            store = _get_default_store()
            guard.attach_store(store)
            store.set("memgym_bench/rank/%d" % rank, json.dumps(ranks["devices"][0]))
            if rank == 0:
                keys = ["memgym_bench/rank/%d" % q for q in range(world)]
                store.wait(keys, timedelta(seconds=60))
                ranks["devices"] = [json.loads(store.get(k).decode()) for k in keys]
        except Exception as e:
            ranks["note"] = "rank reports unavailable: %s" % str(e)[:120]
        if backend == "nccl":
            try:
                ranks["collective_library"] = "RCCL %s (torch.cuda.nccl.version())" % ".".join(str(v) for v in torch.cuda.nccl.version())
            except Exception:
                pass
        ranks["process_group_timeout_s"] = pg_timeout.total_seconds()

    env_id = args.env
    n_local = args.envs_per_gpu or DEFAULT_ENVS[env_id]
    K, W = args.steps, args.warmup
    guard.arm(max(600.0, 2 * args.leg_limit), "the headline workload")
    r = run_workload(env_id, n_local, K, W, args.settle, world, rank, dev, args.obs_format, args.gather, not args.no_events, args.event_stride,
                     long_window=args.long_window, policy=args.policy)
    guard.disarm()

    out = None
    if rank == 0:
        out = build_headline(args, r, env_id, n_local, K, W, world, ranks)
        guard.set_line(out)  # from here on the run cannot end without its headline
        sys.stderr.write("bench.py: headline measured (%d GPU%s): %.4g env steps/s -- held by the line guard, printed once at the end\n" % (world, "s" if world > 1 else "", out["value"]))
        sys.stderr.flush()

    if world > 1 and not args.no_secondary and args.obs_format == "u8_xyc":
        # BASELINE.json config 5: Endless-MortarMayhem-v0, 32,768 instances per GPU, with and without the gather to rank 0.  Each leg
        # runs under the guard's deadline; the ranks agree on going in and on how it went through the store (no collective), and the
        # first failure anywhere ends the phase for everybody: a rank that raised cannot leave the others waiting in a collective.
        c5 = {}
        if rank == 0:
            out["config5"] = {"workload": "Endless-MortarMayhem-v0, %d envs/GPU x %d GPU (%d envs)" % (args.config5_envs, world, args.config5_envs * world)}
        for label, gm in (("no_gather", None), ("gather_rccl", "rccl"), ("gather_peer", "peer")):
            guard.arm(args.leg_limit, "config5 leg " + label)
            go, why = guard.agree("pre/" + label, True, timeout_s=args.leg_limit)
            if not go:
                c5[label] = "skipped: " + why
                break
            ok, err = True, ""
            try:
                q = run_workload("Endless-MortarMayhem-v0", args.config5_envs, args.config5_steps, 20, args.settle, world, rank, dev, "u8_xyc", gm,
                                 not args.no_events, args.event_stride, leg=label)
                c5[label] = {"value": q["value"], "unit": "env steps/s", "per_gpu_value": q["value"] / world, "ms_per_step": q["ms_per_step"],
                             "timing": q["timing"], "wall_ms_per_step": q["wall_ms_per_step"],
                             "raster_avg_ms_rank0": q["raster_avg_ms"], "note": q["note"],
                             # what rank 0 holds after a step of this leg ("obs (+reward, done)")
                             "gathered": (None if gm is None else q.get("gathered") if q["gather"] == "rccl" else
                                          "obs by peer-mapped stores + reward (f32) + done (u8) by one packed gather: 21168 + 5 B per instance"),
                             "gather_check": q.get("gather_check")}
            except Exception as e:  # keep the headline line alive -- and tell the others, who may be waiting for this rank in a collective
                ok, err = False, "%s: %s" % (type(e).__name__, str(e)[:200])
                c5[label] = "failed: " + err
            if rank == 0:
                out["config5"].update(c5)
                guard.set_line(out)
            if not ok:
                guard.abort("config5 leg %s failed: %s" % (label, err))
                break
            all_ok, why = guard.agree("post/" + label, ok, err, timeout_s=args.leg_limit)
            if not all_ok:  # (a rank that failed has raised the abort word already; this rank stops here too)
                if rank == 0:
                    out["config5"][label] = {"this_rank": c5[label], "failed_elsewhere": why}
                    guard.set_line(out)
                break
        guard.disarm()

    if rank == 0:
        if world == 1 and not args.no_cpu_baseline and not fake:
            try:
                out["cpu_baseline"] = cpu_baseline(env_id)
            except Exception as e:  # the oracle is optional equipment on a box without gcc
                out["cpu_baseline"] = {"value": None, "unit": "env steps/s", "cores": os.cpu_count(), "kind": "port", "cpu_model": cpu_model(),
                                       "sample": "unavailable: %s" % e}
            out["pygame_baseline"] = pygame_baseline(env_id)
        if world == 1 and not args.no_c1 and not fake:
            try:
                out["c1"] = c1_leg()
            except Exception as e:
                out["c1"] = "failed: %s" % e
        if world == 1 and not args.no_secondary and args.obs_format == "u8_xyc" and not fake:
            try:
                out["secondary_workloads"] = secondary_workloads(env_id, dev, args.settle, traffic=not args.no_traffic)
            except Exception as e:
                out["secondary_workloads"] = "failed: %s" % e
            if not args.no_other:
                try:
                    out["other_workloads"] = other_workloads(env_id, dev, args.settle)
                except Exception as e:
                    out["other_workloads"] = "failed: %s" % e
    guard.finish(out)  # the ONE line
    if dist.is_initialized():
        import threading
        try:
            aborted = world > 1 and guard.store is not None and guard.store.check(["memgym_bench/abort"])
        except Exception:
            aborted = True
        if aborted:
            os._exit(0)  # a failed phase: the process group is in no state to be torn down collectively
        t = threading.Timer(20.0, lambda: os._exit(0))  # the teardown is a collective too: bounded
        t.daemon = True
        t.start()
        try:
            dist.destroy_process_group()
        except Exception:
            pass
        t.cancel()


if __name__ == "__main__":
    main()
