#!/usr/bin/env python3
"""Worker of tests/test_gpu_sparse_raster.py: for every case, a digest of everything a gymnasium-convention run and a run of masked resets
hand to their caller -- observations, terminal observations of the finished instances, rewards, dones, RNG words at the end.  Run once
with MEMGYM_SPARSE_RASTER=0 (the dense launches of rounds 1-5) and once without; the lab build reads the switch (csrc/mg_lab.hpp)."""
import hashlib
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "endless-memory-gym_amd"))
import memory_gym_amd  # noqa: E402

# (MysteryPath-v0 with max_steps = 14: every instance is truncated in the same step -- a "sparse" launch over ALL frames)
CASES = [("MortarMayhem-Grid-v0", 20001, 60, "u8_xyc", None), ("Endless-SearingSpotlights-v0", 16385, 25, "u8_xyc", None),
         ("MysteryPath-v0", 8191, 30, "u8_xyc", {"max_steps": 14}), ("Endless-MysteryPath-v0", 4097, 25, "u8_xyc", None),
         ("SearingSpotlights-v0", 3000, 25, "bf16_chw", None), ("MortarMayhemB-v0", 2049, 25, "u8_xyc", None)]


_W = {}


def upd(h, t):
    """position-sensitive checksums computed on the device (a row's words weighted by their place, the rows by theirs): hashing the bytes on
    the host would copy 0.4 GB per step"""
    if t.dtype == torch.bool:
        t = t.to(torch.uint8)
    t = t.contiguous()
    rows = t.shape[0] if t.dim() > 1 else 1
    b = t.view(torch.uint8).view(rows, -1)
    if b.shape[1] % 4 == 0:
        b = b.view(torch.int32)
    v = b.to(torch.int64)
    key = (v.shape[1], rows)
    if key not in _W:
        _W[key] = ((torch.arange(v.shape[1], device=v.device, dtype=torch.int64) * 2654435761 + 12345) % 1000003 + 1,
                   (torch.arange(rows, device=v.device, dtype=torch.int64) * 40503 + 7) % 999983 + 1)
    w1, w2 = _W[key]
    per_row = (v * w1).sum(1)
    h.update(np.array([int((per_row * w2).sum()), int(per_row.sum()), int(v.sum())], dtype=np.int64).tobytes())


if os.environ.get("MEMGYM_SPARSE_CASES") == "mortar":  # test_mortar_one_launch_keeps_terminal_observations: every mortar id, one of them at full size
    CASES = [("MortarMayhem-Grid-v0", 65536, 70, "u8_xyc", None), ("MortarMayhem-v0", 20001, 70, "u8_xyc", None),
             ("Endless-MortarMayhem-v0", 32768, 40, "u8_xyc", None), ("MortarMayhemB-Grid-v0", 12289, 40, "u8_xyc", None),
             ("MortarMayhemB-v0", 8193, 60, "u8_xyc", None)]
if os.environ.get("MEMGYM_SPARSE_CASES") == "spot":  # test_spot_fused_launch_keeps_terminal_observations
    CASES = [("SearingSpotlights-v0", 16385, 120, "u8_xyc", None), ("Endless-SearingSpotlights-v0", 16384, 150, "u8_xyc", None),
             ("Endless-SearingSpotlights-v0", 20001, 120, "u8_xyc", None), ("SearingSpotlights-v0", 3000, 150, "u8_xyc", {"black_background": True})]
if os.environ.get("MEMGYM_SPARSE_CASES") == "mystery":  # test_mystery_launches_keep_terminal_observations (short episodes: max_steps)
    CASES = [("MysteryPath-Grid-v0", 32768, 200, "u8_xyc", None), ("MysteryPath-v0", 20001, 60, "u8_xyc", {"max_steps": 24}),
             ("MysteryPath-Grid-v0", 4097, 120, "u8_xyc", {"max_steps": 16}), ("MysteryPath-v0", 32768, 120, "u8_xyc", {"max_steps": 48})]
if os.environ.get("MEMGYM_SPARSE_CASES") == "emp_big":  # test_emp_masked_resets_...: the arrangement of launches above ~20,000 instances
    CASES = [("Endless-MysteryPath-v0", 32768, 60, "u8_xyc", None)]

if os.environ.get("MEMGYM_SPARSE_CASES") == "emp":  # test_emp_launches_keep_terminal_observations: random agents and followers (a trailing
    # "follow": the agent follows info["ground_truth"], 10 % random moves; max_steps: many instances truncated in one step, some of them in the
    # very step in which they enter a new segment -- the one case in which the fused launch's SERVICE waves meet a terminal state), above and
    # below the ~20,000 instances at which the arrangement of launches changes
    # (a follower needs ~60 steps per segment: max_steps beyond the step in which the first segments are appended)
    CASES = [("Endless-MysteryPath-v0", 32768, 120, "u8_xyc", None), ("Endless-MysteryPath-v0", 32768, 330, "u8_xyc", {"max_steps": 150, "follow": 1}),
             ("Endless-MysteryPath-v0", 20001, 300, "u8_xyc", {"max_steps": 131, "follow": 1}), ("Endless-MysteryPath-v0", 4097, 300, "u8_xyc", {"max_steps": 140, "follow": 1})]

STEPS_X = int(os.environ.get("MEMGYM_SPARSE_STEPS_X", "1"))  # one-off long runs of the same comparison

for env_id, n, steps, fmt, options in CASES:
    steps *= STEPS_X
    h = hashlib.sha256()
    vis = (lambda o: o["visual_observation"] if isinstance(o, dict) else o)
    # (1) the gymnasium vector convention: terminal observations + same-call resets
    follow = bool(options and options.get("follow"))
    if options and "follow" in options:
        options = {k: v for k, v in options.items() if k != "follow"} or None
    envs = memory_gym_amd.GymnasiumVectorEnv(env_id, n, device=0, obs_format=fmt)
    adim = envs.env.action_dim
    n_act = 4 if adim == 1 else 3
    obs, _ = envs.reset(seed=7, options=options)
    upd(h, vis(obs))
    g = torch.Generator(device="cuda").manual_seed(3)
    finished = 0
    for t in range(steps):
        a = torch.randint(0, n_act, (n,) if adim == 1 else (n, adim), device="cuda", generator=g, dtype=torch.int32)
        if follow:  # (ground truth of the previous step: one-hot right / up / down = actions 1 / 2 / 3... the library's own action codes)
            gt = envs.env.gt
            a = torch.where(torch.rand(n, device="cuda", generator=g) < 0.1, a, gt.argmax(1).to(torch.int32) + 1)
        obs, rew, term, trunc, infos = envs.step(a)
        upd(h, vis(obs)); upd(h, rew); upd(h, term)
        d = infos["_final_observation"]
        finished += int(d.sum())
        if d.any():
            upd(h, vis(infos["final_observation"])[d])
    own = envs.env.debug_counter("emp_own_resets") if env_id == "Endless-MysteryPath-v0" else 0
    if env_id == "Endless-MysteryPath-v0" and os.environ.get("MEMGYM_SPARSE_CASES") == "emp":
        print("final_served %d" % envs.env.debug_counter("emp_final_served"), flush=True)
    for i in (0, n // 2, n - 1):
        h.update(np.asarray(envs.env.rng_words(i)).tobytes())
    envs.close()
    # (2) masked resets by the caller: a few instances, half of them, all but one
    env = memory_gym_amd.make(env_id, num_envs=n, device=0, obs_format=fmt)
    env.reset(seed=11)
    for frac in (0.01, 0.5, 0.999):
        mask = torch.rand(n, device="cuda", generator=g) < frac
        o, _ = env.reset(seed=1000, mask=mask)
        upd(h, vis(o))
        a = torch.randint(0, n_act, (n,) if adim == 1 else (n, adim), device="cuda", generator=g, dtype=torch.int32)
        upd(h, vis(env.step(a)[0]))
    env.check_errors()
    env.close()
    print("digest %s %d %s %s finished=%d" % (env_id, n, fmt, h.hexdigest(), finished), flush=True)
    print("own_resets %d" % own, flush=True)
print("ok: all cases")
