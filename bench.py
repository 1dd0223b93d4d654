#!/usr/bin/env python3
"""bench.py -- aggregate env-steps/s of the batched Memory Gym hot path on N MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--env ID] [--envs-per-gpu M] [--gather]

A "step" is one mg_step() over every instance of the workload: logic kernel + raster kernel, with
same-step auto-reset, 84x84x3 uint8 observations written to HBM.  Inputs (actions) are generated on the device
before the timed region.  N > 1: one process per GPU (torch.distributed, backend nccl == RCCL), instances sharded
with no data-path collective (weak scaling: per-GPU work fixed); --gather adds the optional RCCL gather of
observations to rank 0 that BASELINE.json's config 5 names.

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` (raster kernel, HIP events on the
launch stream inside the timed region) and `cpu_baseline` (the CPU oracle = a port of the reference's algorithm,
bounded sample, OpenMP over instances; rank 0, N=1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "endless-memory-gym_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md); ~6.3 TB/s achievable
FRAME = 84 * 84 * 3
# algorithmic bytes of the RASTER kernel per env-step: the observation it writes + the 16-byte draw descriptor it reads.
# (Whole step incl. logic kernel state/action/reward traffic, SURVEY.md 8(d): MM-Grid 21,305 B.)
RASTER_BYTES = {"default": FRAME + 16}
STEP_BYTES = {"MortarMayhem-Grid-v0": 21305, "MortarMayhem-v0": 21309, "Endless-MortarMayhem-v0": 21821,
              "MysteryPath-v0": 21437, "Endless-SearingSpotlights-v0": 22205}
DEFAULT_ENVS = {"MortarMayhem-Grid-v0": 65536, "MortarMayhem-v0": 65536, "MortarMayhemB-Grid-v0": 65536, "MortarMayhemB-v0": 65536, "Endless-MortarMayhem-v0": 32768,
                "MysteryPath-v0": 32768, "MysteryPath-Grid-v0": 32768, "Endless-MysteryPath-v0": 32768, "SearingSpotlights-v0": 16384,
                "Endless-SearingSpotlights-v0": 16384}


def cpu_baseline(env_id, budget_s=15.0):
    """Time the CPU oracle (oracle/: a restatement of the reference's per-instance algorithm, incl. its software
    raster) on this host.  Bounded sample: `n` instances stepped with auto-reset for ~budget_s seconds."""
    import numpy as np

    import oracle_lib

    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    os.environ["OMP_NUM_THREADS"] = str(cores)
    n = 64 * cores
    b = oracle_lib.OracleBatch(env_id, n)
    disc = b.discrete
    b.reset(np.arange(n, dtype=np.int64))
    g = np.random.Generator(np.random.PCG64(0))
    acts = [(g.integers(0, 4, n) if disc else g.integers(0, 3, (n, 2))).astype(np.int32) for _ in range(16)]
    for a in acts[:2]:
        b.step(a, autoreset=True)
    t0 = time.perf_counter()
    steps = 0
    while time.perf_counter() - t0 < budget_s:
        b.step(acts[steps % 16], autoreset=True)
        steps += 1
    dt = time.perf_counter() - t0
    b.close()
    return {"value": n * steps / dt, "unit": "env steps/s", "cores": cores, "kind": "port",
            "sample": "%d instances x %d steps of %s, CPU oracle (C, OpenMP over instances, incl. software raster), %.1f s"
                      % (n, steps, env_id, dt)}


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


def secondary_workloads(primary, device_index, steps=200, warmup=30):
    """The other single-GPU BASELINE configs (C3, C4, the per-GPU shard of C5), measured the same way after the headline
    run so that one bench line carries them; informational (the contract's `value` is the headline workload's)."""
    import torch

    import memory_gym_amd

    out = []
    for env_id, label in (("MysteryPath-v0", "C3"), ("Endless-SearingSpotlights-v0", "C4"), ("Endless-MortarMayhem-v0", "C5 per-GPU shard")):
        if env_id == primary:
            continue
        n = DEFAULT_ENVS[env_id]
        env = memory_gym_amd.make(env_id, num_envs=n, device=device_index)
        env.reset(seed=0)
        g = torch.Generator(device="cuda").manual_seed(99)
        shape, hi = ((n,), 4) if env.action_dim == 1 else ((n, 2), 3)
        acts = [torch.randint(0, hi, shape, device="cuda", generator=g, dtype=torch.int32) for _ in range(32)]
        for k in range(warmup):
            env.step(acts[k % 32])
        env.set_profiling(8)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(steps):
            env.step(acts[k % 32])
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        rm, rn = env.get_profile(1)
        lm, ln = env.get_profile(0)
        env.close()
        out.append({"config": label, "workload": "%s, %d envs" % (env_id, n), "value": n * steps / dt, "unit": "env steps/s",
                    "ms_per_step": dt / steps * 1e3, "raster_avg_ms": rm / rn if rn else None, "logic_avg_ms": lm / ln if ln else None,
                    "raster_GBps": (FRAME + 16) * n / (rm / rn * 1e-3) / 1e9 if rn else None})
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=30)
    ap.add_argument("--env", default="MortarMayhem-Grid-v0")
    ap.add_argument("--envs-per-gpu", type=int, default=0)
    ap.add_argument("--obs-format", default="u8_xyc", choices=["u8_xyc", "f32_chw", "f16_chw", "bf16_chw"],
                    help="raster stream-out format; the BASELINE.json metric is quoted on the default (the reference's uint8 obs)")
    ap.add_argument("--gather", nargs="?", const="rccl", default=None, choices=["rccl", "peer"],
                    help="BASELINE config 5's observation gather to rank 0 every step: 'rccl' = torch.distributed.gather, "
                         "'peer' = the raster kernels store straight into rank 0's HBM (memory_gym_amd.dist.PeerObsBuffer)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the informational C3 / C4 / C5-shard measurements (N = 1 only)")
    ap.add_argument("--no-events", action="store_true", help="do not bracket kernels with HIP events")
    ap.add_argument("--event-stride", type=int, default=8, help="bracket every N-th step with HIP events (each bracketed step costs ~15 us)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    import memory_gym_amd

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    else:
        torch.cuda.set_device(0)
    if args.gpus != world and rank == 0 and world > 1:
        print("warning: --gpus %d but WORLD_SIZE %d" % (args.gpus, world), file=sys.stderr)
    dev = torch.device("cuda", local_rank)

    env_id = args.env
    n_local = args.envs_per_gpu or DEFAULT_ENVS[env_id]
    n_total = n_local * world
    peer = None
    if args.gather == "peer" and world > 1:
        from memory_gym_amd.dist import PeerObsBuffer
        code, dt, shape = memory_gym_amd.VecMemoryGym.OBS_FORMATS[args.obs_format]
        peer = PeerObsBuffer(n_total, frame_shape=shape, dtype=dt, device=dev)
    env = memory_gym_amd.make(env_id, num_envs=n_local, device=local_rank, obs_format=args.obs_format,
                              obs_buffer=peer.local if peer else None)
    obs_elem = {"u8_xyc": 1, "f32_chw": 4, "f16_chw": 2, "bf16_chw": 2}[args.obs_format]
    # instance i (global index) is seeded i whatever the world size -> results are world-size invariant
    from memory_gym_amd.dist import gather_to_rank0, shard_seeds
    seeds = shard_seeds(n_total, rank, world, base_seed=0, device=dev)
    env.reset(seed=seeds)

    K, W = args.steps, args.warmup
    g = torch.Generator(device=dev).manual_seed(1234 + rank)
    n_act_bufs = min(K + W, 64)
    if env.action_dim == 1:
        acts = [torch.randint(0, 4, (n_local,), device=dev, generator=g, dtype=torch.int32) for _ in range(n_act_bufs)]
    else:
        acts = [torch.randint(0, 3, (n_local, 2), device=dev, generator=g, dtype=torch.int32) for _ in range(n_act_bufs)]

    gather_bufs = [torch.empty_like(env.obs) for _ in range(world)] if (args.gather == "rccl" and world > 1 and rank == 0) else None

    def one_step(k):
        obs, rew, done, _, _ = env.step(acts[k % n_act_bufs])
        if peer is not None:  # the frames are already in rank 0's memory; one 4-byte all-reduce orders the streams
            peer.fence()
        elif args.gather and world > 1:  # equal shards: plain gather into preallocated buffers (no per-step allocation)
            dist.gather(obs, gather_bufs, dst=0)

    for k in range(W):
        one_step(k)
    if not args.no_events:
        env.set_profiling(max(1, args.event_stride))
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(K):
        one_step(W + k)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0

    raster_ms = raster_n = logic_ms = logic_n = 0
    if not args.no_events:
        raster_ms, raster_n = env.get_profile(1)
        logic_ms, logic_n = env.get_profile(0)
        env.set_profiling(False)

    t = torch.tensor([dt], device=dev, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt_max = float(t.item())

    if rank == 0:
        value = n_total * K / dt_max
        out = {
            "metric": "env steps/sec (aggregate)", "value": value, "unit": "env steps/s", "n_gpus": world, "steps": K,
            "warmup": W, "ms_per_step": dt_max / K * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u8", "data": "synthetic", "obs_format": args.obs_format,
            "config": {"workload": "%s, %d envs/GPU x %d GPU, 84x84x3 obs (%s), same-step auto-reset, uniform random "
                                   "actions generated on device%s" % (env_id, n_local, world, args.obs_format,
                                                                     (", peer-mapped obs stores into rank 0's HBM" if peer is not None else ", RCCL obs gather to rank 0") if args.gather and world > 1 else ""),
                       "env_id": env_id, "envs_per_gpu": n_local, "envs_total": n_total,
                       "parallelism": "env-sharded x%d, no data-path collective" % world if not args.gather else
                       "env-sharded x%d + %s(obs)->rank0" % (world, "peer-mapped stores" if peer is not None else "gather")},
        }
        if raster_n:
            avg_ms = raster_ms / raster_n
            rb = (FRAME * obs_elem + 16) * n_local
            achieved = rb / (avg_ms * 1e-3) / 1e9
            traffic = None
            pmc = os.path.join(ROOT, "profiles", "pmc_latest.json")
            if os.path.exists(pmc):
                try:
                    j = json.load(open(pmc))
                    if j.get("env_id") == env_id and j.get("envs_per_gpu") == n_local and obs_elem == 1:
                        traffic = j.get("hbm_bytes_per_launch")
                except Exception:
                    traffic = None
            out["roofline"] = {"bound": "hbm", "kernel": "raster", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                               "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic,
                               "bytes_per_launch": rb, "avg_launch_ms": avg_ms, "launches": raster_n, "event_stride": max(1, args.event_stride),
                               "logic_kernel_avg_ms": (logic_ms / logic_n) if logic_n else None,
                               "whole_step_GBps": (STEP_BYTES.get(env_id, FRAME) + FRAME * (obs_elem - 1)) * n_total / (dt_max / K) / 1e9}
        if getattr(env, "placement_probe_ms", None):  # raster time into each candidate allocation of the observation buffer
            out["obs_placement_probe_ms"] = [round(t, 4) for t in env.placement_probe_ms]
        if world == 1 and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(env_id)
            except Exception as e:  # the oracle is optional equipment on a box without gcc
                out["cpu_baseline"] = {"value": None, "unit": "env steps/s", "cores": os.cpu_count(), "kind": "port",
                                       "sample": "unavailable: %s" % e}
        if world == 1 and not args.no_cpu_baseline:
            out["pygame_baseline"] = pygame_baseline(env_id)
    env.close()
    if rank == 0:
        if world == 1 and not args.no_secondary and args.obs_format == "u8_xyc":
            try:
                out["secondary_workloads"] = secondary_workloads(env_id, local_rank)
            except Exception as e:
                out["secondary_workloads"] = "failed: %s" % e
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
