#!/usr/bin/env python3
"""tools/vector_soak.py [STEPS] -- one-off hunt (round 6): every env id behind the gymnasium vector front end (terminal rows copied, masked
resets drawn by the mask, Endless-MysteryPath's masked resets from records ahead of time) in lock-step with a handle of the same id that
steps with the same-step auto-reset the parity suite pins to the oracle: observations, rewards and dones must be equal after EVERY step,
the generator's words of sample instances at the end; terminal observations must differ from the new episode's first frame only where an
episode really ended.  Sizes choose the large-launch arrangements."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "endless-memory-gym_amd"))
import memory_gym_amd  # noqa: E402
from memory_gym_amd.vector import GymnasiumVectorEnv  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
CASES = [("MortarMayhem-Grid-v0", 65536), ("MortarMayhem-v0", 20001), ("Endless-MortarMayhem-v0", 32768), ("MysteryPath-v0", 32768),
         ("MysteryPath-Grid-v0", 24577), ("Endless-MysteryPath-v0", 32768), ("SearingSpotlights-v0", 16385), ("Endless-SearingSpotlights-v0", 16384),
         ("MortarMayhemB-Grid-v0", 12289), ("MortarMayhemB-v0", 8193)]
if os.environ.get("VECTOR_SOAK_ONLY"):
    CASES = [c for c in CASES if c[0] in os.environ["VECTOR_SOAK_ONLY"].split(",")]
vis = (lambda o: o["visual_observation"] if isinstance(o, dict) else o)
for env_id, n in CASES:
    venv = GymnasiumVectorEnv(env_id, n, device=0)
    fused = memory_gym_amd.make(env_id, num_envs=n, device=0)
    adim = fused.action_dim
    n_act = 4 if adim == 1 else 3
    o1, _ = venv.reset(seed=5)
    o2, _ = fused.reset(seed=5)
    assert torch.equal(vis(o1), vis(o2))
    g = torch.Generator(device="cuda").manual_seed(9)
    follow = fused.gt.clone() if (os.environ.get("SOAK_POLICY") == "follower" and fused.gt_dim == 3) else None
    finished = 0
    for t in range(steps):
        a = torch.randint(0, n_act, (n,) if adim == 1 else (n, adim), device="cuda", generator=g, dtype=torch.int32)
        if follow is not None:  # SOAK_POLICY=follower: the way the ground truth names, a random action with probability 0.02 (deep episodes)
            a = torch.where(torch.rand(n, device="cuda", generator=g) < 0.02, a, follow.argmax(1).to(torch.int32) + 1)
        prev2 = vis(o2).clone() if os.environ.get("VECTOR_SOAK_DIAG") else None
        o1, r1, d1, tr, infos = venv.step(a)
        o2, r2, d2, _, i2 = fused.step(a)
        if follow is not None:
            follow = i2["ground_truth"]
        if not (torch.equal(vis(o1), vis(o2)) and torch.equal(r1, r2) and torch.equal(d1, d2)):
            bad = (vis(o1) != vis(o2)).flatten(1).any(1).nonzero().flatten()[:5].tolist()
            print("MISMATCH %s step %d instances %s (obs); rewards equal %s, dones equal %s; done of those %s; differing bytes %s" % (
                env_id, t, bad, torch.equal(r1, r2), torch.equal(d1, d2), d1[bad].tolist() if bad else None,
                [(int((vis(o1)[b] != vis(o2)[b]).sum())) for b in bad]))
            if bad:  # who is right?  replay the instance on the CPU oracle with the same actions
                sys.path.insert(0, os.path.join(ROOT, "tests"))
                import oracle_lib
                b = bad[0]
                ref = oracle_lib.OracleEnv(env_id)
                ro = ref.reset(5 + b)
                g2 = torch.Generator(device="cuda").manual_seed(9)
                for u in range(t + 1):
                    au = torch.randint(0, n_act, (n,) if adim == 1 else (n, adim), device="cuda", generator=g2, dtype=torch.int32)[b].cpu().numpy()
                    ro, _, rd = ref.step(au if adim == 2 else [int(au), 0])
                    term = ro
                    if rd:
                        ro = ref.reset(None)
                v, f = vis(o1)[b].cpu().numpy(), vis(o2)[b].cpu().numpy()
                print("  oracle: done %s; vector convention == oracle: %s (%d bytes off), auto-reset step == oracle: %s (%d bytes off); final_observation == oracle's terminal frame: %s; vector obs == terminal frame: %s" % (
                    rd, np.array_equal(v, ro), int((v != ro).sum()), np.array_equal(f, ro), int((f != ro).sum()),
                    np.array_equal(vis(infos["final_observation"])[b].cpu().numpy(), term), np.array_equal(v, term)))
                if prev2 is not None:
                    print("  auto-reset obs == its own previous frame (not drawn this step): %s; == oracle's terminal frame: %s; bytes off vs terminal %d, vs previous %d" % (
                        np.array_equal(f, prev2[b].cpu().numpy()), np.array_equal(f, term), int((f != term).sum()), int((f != prev2[b].cpu().numpy()).sum())))
                    dd = np.argwhere((f != ro).any(2))
                    print("  differing pixels bbox x %d..%d y %d..%d, %d pixels; channel sums auto %s oracle %s" % (dd[:, 0].min(), dd[:, 0].max(), dd[:, 1].min(), dd[:, 1].max(), len(dd), f.reshape(-1, 3).sum(0), ro.reshape(-1, 3).sum(0)))
                print("  rng words: vector %s\n             fused  %s\n             oracle %s" % (venv.env.rng_words(b), fused.rng_words(b), ref.rng_words()))
            sys.exit(1)
        if isinstance(o1, dict) and not torch.equal(o1["vector_observation"], o2["vector_observation"]):
            print("MISMATCH %s step %d vector observation" % (env_id, t))
            sys.exit(1)
        finished += int(d1.sum())
        if t % 97 == 0 and d1.any():  # the terminal rows are frames of the OLD episode: for some instance they differ from the new first frame
            assert (vis(infos["final_observation"])[d1] != vis(o1)[d1]).flatten(1).any(1).any(), (env_id, t)
    for i in (0, n // 3, n - 1):
        assert np.array_equal(venv.env.rng_words(i), fused.rng_words(i)), (env_id, i)
    venv.env.check_errors()
    fused.check_errors()
    print("ok %-30s %6d instances x %d steps, %d episodes finished: vector convention == auto-reset step" % (env_id, n, steps, finished), flush=True)
    venv.close()
    fused.close()
