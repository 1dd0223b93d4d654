"""GPU (-m gpu): every frame of every instance after EVERY step, for thousands of steps -- by letting two independent launch arrangements
of the library check each other on the device (a comparison costs 0.1 ms; against the CPU oracle the full-batch tests can afford all frames
only every 20-32 steps).  The gymnasium vector convention (step without auto-reset, terminal rows copied, masked reset, frames drawn by the
mask: tests/test_gpu_vector_api.py pins it to the oracle) must deliver the same observations, rewards and dones as the same-step auto-reset
arrangement bench.py measures (pinned to the oracle by tests/test_gpu_full_batch.py).

Round 6: this is how a race of the round-4 fused spotlight launch was found (tools/vector_soak.py) -- a reset frame drawn from the old or a
half-written descriptor, one frame in ~10^7, which frames compared every 20th step of a 200-step run meet with a probability of a few
per cent per run (csrc/mg_spot.hip spot_raster_serve_kernel: scalar loads behind vector stores of the same workgroup)."""
import pytest

pytestmark = pytest.mark.gpu

CASES = [("SearingSpotlights-v0", 16385, 3000),          # resets served inside the raster launch (the arrangement that raced)
         ("Endless-SearingSpotlights-v0", 20001, 1500),  # ... the endless variant takes that arrangement above 16,384 instances
         ("Endless-SearingSpotlights-v0", 16384, 600),   # C4's own: resets in the step kernel
         ("MortarMayhem-Grid-v0", 65536, 400),           # one launch per step (claim words)
         ("Endless-MysteryPath-v0", 32768, 400),         # lazy segments, records ahead of time
         ("MysteryPath-v0", 32768, 560)]                 # (episodes of 512 steps: every instance is truncated in the same step once)


@pytest.mark.parametrize("env_id,n,steps", CASES)
def test_two_arrangements_agree_after_every_step(env_id, n, steps):
    import memory_gym_amd
    import torch
    from memory_gym_amd.vector import GymnasiumVectorEnv

    venv = GymnasiumVectorEnv(env_id, n, device=0)
    fused = memory_gym_amd.make(env_id, num_envs=n, device=0)
    adim = fused.action_dim
    n_act = 4 if adim == 1 else 3
    o1, _ = venv.reset(seed=5)
    o2, _ = fused.reset(seed=5)
    assert torch.equal(o1, o2)
    g = torch.Generator(device="cuda").manual_seed(9)
    finished = 0
    for t in range(steps):
        a = torch.randint(0, n_act, (n,) if adim == 1 else (n, adim), device="cuda", generator=g, dtype=torch.int32)
        o1, r1, d1, _, _ = venv.step(a)
        o2, r2, d2, _, _ = fused.step(a)
        if not torch.equal(o1, o2):
            bad = (o1 != o2).flatten(1).any(1).nonzero().flatten()[:4].tolist()
            pytest.fail("%s: observations of instances %s differ after step %d (done: %s)" % (env_id, bad, t, d1[bad].tolist()))
        assert torch.equal(r1, r2) and torch.equal(d1, d2), "%s step %d" % (env_id, t)
        finished += int(d1.sum())
    assert finished > 0
    venv.env.check_errors()
    fused.check_errors()
    venv.close()
    fused.close()


FORMAT_CASES = [("MortarMayhem-Grid-v0", 65536, 250), ("Endless-MysteryPath-v0", 32768, 250), ("SearingSpotlights-v0", 16385, 400),
                ("Endless-SearingSpotlights-v0", 20001, 300), ("Endless-MortarMayhem-v0", 32768, 200)]


@pytest.mark.parametrize("env_id,n,steps", FORMAT_CASES)
def test_fused_uint8_launches_agree_with_the_plain_float_arrangement(env_id, n, steps):
    """The uint8 handle runs the fused launches (one launch per step, resets / paths served inside the raster launch, lazy segments); an
    f32_chw handle of the same id takes the plain arrangements (step kernel, queue server, plain raster) and the float stream-out.  After
    every step: obs_f32 == obs_u8 / 255 in CHW order, for every instance (tools/arrangement_soak.py formats)."""
    import memory_gym_amd
    import torch

    a_env = memory_gym_amd.make(env_id, num_envs=n, device=0)
    b_env = memory_gym_amd.make(env_id, num_envs=n, device=0, obs_format="f32_chw")
    div = torch.tensor(255.0, device="cuda")  # (a device divisor: with a Python scalar torch multiplies by the rounded reciprocal)

    def same(x, y):
        return torch.equal(x.permute(0, 3, 2, 1).to(torch.float32) / div, y)
    seeds = torch.arange(n, dtype=torch.int64, device="cuda") + 3
    oa, _ = a_env.reset(seed=seeds)
    ob, _ = b_env.reset(seed=seeds)
    assert same(oa, ob)
    adim = a_env.action_dim
    n_act = 4 if adim == 1 else 3
    g = torch.Generator(device="cuda").manual_seed(17)
    for t in range(steps):
        a = torch.randint(0, n_act, (n,) if adim == 1 else (n, adim), device="cuda", generator=g, dtype=torch.int32)
        oa, ra, da, _, _ = a_env.step(a)
        ob, rb, db, _, _ = b_env.step(a)
        assert same(oa, ob), "%s: frames differ after step %d" % (env_id, t)
        assert torch.equal(ra, rb) and torch.equal(da, db), "%s step %d" % (env_id, t)
    a_env.check_errors()
    b_env.check_errors()
    a_env.close()
    b_env.close()


@pytest.mark.parametrize("env_id,n,steps", [("SearingSpotlights-v0", 16385, 2000), ("Endless-SearingSpotlights-v0", 20001, 1000),
                                            ("MortarMayhem-Grid-v0", 65536, 300), ("Endless-MysteryPath-v0", 32768, 300)])
def test_render_reproduces_every_frame_of_every_step(env_id, n, steps):
    """mg_render draws the descriptors that lie in memory with the plain raster launch; the step's own launch drew its frames from them
    too -- fused with the step, beside the resets or the path service.  After every step the two must agree on every instance (the
    pre-fix spot_raster_serve_kernel fails this within a few hundred steps: tools/arrangement_soak.py render)."""
    import memory_gym_amd
    import torch
    from memory_gym_amd import _native

    env = memory_gym_amd.make(env_id, num_envs=n, device=0)
    obs, _ = env.reset(seed=torch.arange(n, dtype=torch.int64, device="cuda") + 3)
    adim = env.action_dim
    n_act = 4 if adim == 1 else 3
    g = torch.Generator(device="cuda").manual_seed(23)
    again = torch.empty_like(obs)
    for t in range(steps):
        a = torch.randint(0, n_act, (n,) if adim == 1 else (n, adim), device="cuda", generator=g, dtype=torch.int32)
        obs, _, d, _, _ = env.step(a)
        again.fill_(7)
        _native.check(_native.LIB.mg_render(env._h, again.data_ptr(), env._stream()), "mg_render")
        if not torch.equal(again, obs):
            bad = (again != obs).flatten(1).any(1).nonzero().flatten()[:4].tolist()
            pytest.fail("%s: mg_render and the step disagree on instances %s after step %d (done: %s)" % (env_id, bad, t, d[bad].tolist()))
    env.check_errors()
    env.close()
