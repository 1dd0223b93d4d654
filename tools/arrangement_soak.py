#!/usr/bin/env python3
"""tools/arrangement_soak.py MODE [STEPS] -- one-off hunts (round 6) in the manner of tools/vector_soak.py: two handles of the same id that the
library runs through DIFFERENT launch arrangements, stepped with the same actions, every frame of every instance compared on the device after
every step.
  formats  a uint8 handle (the fused launches: one launch per step, resets / paths served inside the raster launch, lazy segments) against an
           f32_chw handle (the float formats take the plain arrangements: step kernel, queue server, plain raster): obs_f32 == obs_u8 / 255 in CHW
  sets     a default handle against one whose odd instances run under a second option set that differs in a reward only (per-instance option
           sets: the <PS> forms of the kernels, none of the fused launches): observations and dones equal, rewards equal on the even instances
  sizes    a handle of n instances against a handle of the first m = 4,099 of them (an instance's episode does not depend on how many others
           there are; the library chooses its launch arrangement by the size: queue entries or lane jobs, resets inside the raster launch or in
           the step kernel, small or large raster grids): obs[:m], rewards[:m], dones[:m] equal
  render   ONE handle: after every step mg_render (the plain raster over the descriptors in memory) into a second buffer must reproduce the
           observations the step's own launch drew (a frame drawn from a stale or half-written descriptor shows up here too)
  checkpoint  a handle against a second one that takes over its state_dict() every 61 steps and must then follow it frame for frame
           (Family::sync_state: owed segments generated, queues drained, at full size)
  streams  two handles of the same id and seeds stepped CONCURRENTLY on two streams (no synchronisation between the two launches of a
           step): both must draw the same frames -- nothing of a handle (queues, claim words, atlases, pooled buffers) may be shared
Sizes choose the large-launch arrangements of the uint8 / default handle."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "endless-memory-gym_amd"))
import memory_gym_amd  # noqa: E402

mode = sys.argv[1]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
CASES = [("MortarMayhem-Grid-v0", 65536), ("MortarMayhem-v0", 20001), ("Endless-MortarMayhem-v0", 32768), ("MysteryPath-v0", 32768),
         ("MysteryPath-Grid-v0", 24577), ("Endless-MysteryPath-v0", 32768), ("SearingSpotlights-v0", 16385), ("Endless-SearingSpotlights-v0", 20001),
         ("Endless-SearingSpotlights-v0", 16384), ("MortarMayhemB-Grid-v0", 12289), ("MortarMayhemB-v0", 8193)]
if os.environ.get("SOAK_CASES"):  # "id:n,id:n,...": sizes of one's own (the thresholds at which the library changes its arrangement)
    CASES = [(c.split(":")[0], int(c.split(":")[1])) for c in os.environ["SOAK_CASES"].split(",")]
if os.environ.get("SOAK_ONLY"):
    CASES = [c for c in CASES if c[0] in os.environ["SOAK_ONLY"].split(",")]
OTHER = {"MortarMayhem": {"reward_command_success": 0.25}, "MysteryPath": {"reward_fall_off": -0.5}, "SearingSpotlights": {"reward_inside_spotlight": -0.125}}
DIV = torch.tensor(255.0, device="cuda")
FMT = os.environ.get("SOAK_FORMAT", "f32_chw")  # formats mode: f32_chw / bf16_chw / f16_chw
vis = (lambda o: o["visual_observation"] if isinstance(o, dict) else o)
for env_id, n in CASES:
    a_env = memory_gym_amd.make(env_id, num_envs=n, device=0)
    if mode == "streams":
        b_env = memory_gym_amd.make(env_id, num_envs=n, device=0)
        s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
        seeds = torch.arange(n, dtype=torch.int64, device="cuda") + 3
        adim = a_env.action_dim
        n_act = 4 if adim == 1 else 3
        g = torch.Generator(device="cuda").manual_seed(29)
        with torch.cuda.stream(s1):
            oa, _ = a_env.reset(seed=seeds)
        with torch.cuda.stream(s2):
            ob, _ = b_env.reset(seed=seeds)
        torch.cuda.synchronize()
        assert torch.equal(vis(oa), vis(ob))
        for t in range(steps):
            a = torch.randint(0, n_act, (n,) if adim == 1 else (n, adim), device="cuda", generator=g, dtype=torch.int32)
            torch.cuda.synchronize()
            with torch.cuda.stream(s1):
                oa, ra, da, _, _ = a_env.step(a)
            with torch.cuda.stream(s2):
                ob, rb, db, _, _ = b_env.step(a)
            torch.cuda.synchronize()
            if not (torch.equal(vis(oa), vis(ob)) and torch.equal(ra, rb) and torch.equal(da, db)):
                print("MISMATCH %s (streams) step %d" % (env_id, t))
                sys.exit(1)
        a_env.check_errors()
        b_env.check_errors()
        print("ok %-30s %6d instances x %d steps (streams)" % (env_id, n, steps), flush=True)
        a_env.close()
        b_env.close()
        continue
    if mode in ("render", "checkpoint"):
        from memory_gym_amd import _native
        adim = a_env.action_dim
        n_act = 4 if adim == 1 else 3
        g = torch.Generator(device="cuda").manual_seed(23)
        obs, _ = a_env.reset(seed=torch.arange(n, dtype=torch.int64, device="cuda") + 3)
        b_env = memory_gym_amd.make(env_id, num_envs=n, device=0) if mode == "checkpoint" else None
        again = torch.empty_like(vis(obs))
        following = False
        for t in range(steps):
            a = torch.randint(0, n_act, (n,) if adim == 1 else (n, adim), device="cuda", generator=g, dtype=torch.int32)
            obs, r, d, _, _ = a_env.step(a)
            if mode == "render":
                again.fill_(7)
                _native.check(_native.LIB.mg_render(a_env._h, again.data_ptr(), a_env._stream()), "mg_render")
                if not torch.equal(again, vis(obs)):
                    bad = (again != vis(obs)).flatten(1).any(1).nonzero().flatten()[:4].tolist()
                    print("MISMATCH %s (render) step %d instances %s done %s" % (env_id, t, bad, d[bad].tolist()))
                    sys.exit(1)
            else:
                if following:
                    ob, rb, db, _, _ = b_env.step(a)
                    if not (torch.equal(vis(ob), vis(obs)) and torch.equal(rb, r) and torch.equal(db, d)):
                        print("MISMATCH %s (checkpoint) step %d" % (env_id, t))
                        sys.exit(1)
                if t % 61 == 60:
                    b_env.load_state_dict(a_env.state_dict())
                    following = True
        a_env.check_errors()
        print("ok %-30s %6d instances x %d steps (%s)" % (env_id, n, steps, mode), flush=True)
        a_env.close()
        if b_env is not None:
            b_env.close()
        continue
    m = 4099 if mode == "sizes" else n
    if mode == "formats":
        b_env = memory_gym_amd.make(env_id, num_envs=n, device=0, obs_format=FMT)
    else:
        b_env = memory_gym_amd.make(env_id, num_envs=m, device=0)
    seeds = torch.arange(n, dtype=torch.int64, device="cuda") + 3
    oa, _ = a_env.reset(seed=seeds)
    if mode == "sets":
        odd = (torch.arange(n, device="cuda") % 2) == 1
        opt = next(v for k, v in OTHER.items() if k in env_id)
        b_env.reset(seed=seeds, mask=~odd)
        ob, _ = b_env.reset(seed=seeds, options=opt, mask=odd)
    else:
        ob, _ = b_env.reset(seed=seeds[:m])

    def same(x, y):
        x, y = vis(x), vis(y)
        if mode == "formats":  # [N, 84 x, 84 y, 3] uint8 -> [N, 3, 84 y, 84 x] float32 = value / 255 (the correctly rounded quotient)
            # (a DEVICE divisor: with a Python scalar torch multiplies by the rounded reciprocal, which is not the quotient for 126 bytes)
            q = x.permute(0, 3, 2, 1).to(torch.float32) / DIV
            return torch.equal(q if FMT == "f32_chw" else q.to(y.dtype), y)  # (16-bit formats: the float32 quotient rounded to nearest even)
        return torch.equal(x[:m], y)
    assert same(oa, ob), env_id + ": reset frames"
    adim = a_env.action_dim
    n_act = 4 if adim == 1 else 3
    g = torch.Generator(device="cuda").manual_seed(17)
    follow = a_env.gt.clone() if (os.environ.get("SOAK_POLICY") == "follower" and a_env.gt_dim == 3) else None
    finished = 0
    for t in range(steps):
        a = torch.randint(0, n_act, (n,) if adim == 1 else (n, adim), device="cuda", generator=g, dtype=torch.int32)
        if follow is not None:  # SOAK_POLICY=follower: the way the ground truth names, a random action with probability 0.02 (deep episodes)
            a = torch.where(torch.rand(n, device="cuda", generator=g) < 0.02, a, follow.argmax(1).to(torch.int32) + 1)
        oa, ra, da, _, ia = a_env.step(a)
        if follow is not None:
            follow = ia["ground_truth"]
        ob, rb, db, _, _ = b_env.step(a[:m])
        ok = same(oa, ob) and torch.equal(da[:m], db) and (torch.equal(ra[:m], rb) if mode != "sets" else torch.equal(ra[::2], rb[::2]))
        if not ok:
            print("MISMATCH %s (%s) step %d: obs %s dones %s" % (env_id, mode, t, same(oa, ob), torch.equal(da[:m], db)))
            sys.exit(1)
        finished += int(da.sum())
    for i in (0, m // 3, m - 1):
        assert np.array_equal(a_env.rng_words(i), b_env.rng_words(i)), (env_id, i)
    a_env.check_errors()
    b_env.check_errors()
    print("ok %-30s %6d instances x %d steps (%s), %d episodes finished" % (env_id, n, steps, mode, finished), flush=True)
    a_env.close()
    b_env.close()
