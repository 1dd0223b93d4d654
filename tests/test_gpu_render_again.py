"""GPU (-m gpu): mg_render draws the current frames again (into any buffer, without stepping) and the placement probe of the
Python mirror leaves the first observation and the following trajectory untouched."""
import ctypes as C

import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("env_id,adim,n_act", [("MortarMayhem-Grid-v0", 1, 4), ("Endless-SearingSpotlights-v0", 2, 3), ("Endless-MysteryPath-v0", 1, 4)])
def test_render_reproduces_the_current_frames(env_id, adim, n_act):
    import memory_gym_amd
    import torch
    from memory_gym_amd import _native

    n = 300
    env = memory_gym_amd.make(env_id, num_envs=n, device=0)
    env.reset(seed=4)
    g = torch.Generator(device="cuda").manual_seed(1)
    for t in range(70):
        a = torch.randint(0, n_act, (n,) if adim == 1 else (n, adim), device="cuda", generator=g, dtype=torch.int32)
        obs = env.step(a)[0]
        if t % 10 == 9:
            words = [env.rng_words(i).copy() for i in (0, n - 1)]
            again = torch.full_like(obs, 7)
            _native.check(_native.LIB.mg_render(env._h, again.data_ptr(), env._stream()), "mg_render")
            assert torch.equal(again, obs), "step %d" % t
            assert all((env.rng_words(i) == w).all() for i, w in zip((0, n - 1), words)), "mg_render must not touch the state"
    env.close()


def test_balanced_obs_buffer_is_transparent():
    """mg_obs_alloc (pieces from two HBM zones mapped into one virtual range) holds the same frames as a plain tensor,
    for the uint8 and a float format; its memory outlives the handle while a tensor views it."""
    import memory_gym_amd
    import torch

    n = 24576  # 520 MB of uint8 observations: two 304-MiB pieces (smaller buffers are plain allocations)
    for fmt in ("u8_xyc", "bf16_chw"):
        a = memory_gym_amd.make("Endless-MortarMayhem-v0", num_envs=n, device=0, obs_placement="balanced", obs_format=fmt)
        b = memory_gym_amd.make("Endless-MortarMayhem-v0", num_envs=n, device=0, obs_placement="plain", obs_format=fmt)
        info = a.obs_placement_info
        assert info is not None and info["zones"] in (1, 2, 3)
        assert info["zones"] == 1 or info["pieces"] * info["piece_bytes"] >= a.obs.numel() * a.obs.element_size()  # 1: plain allocation
        assert b.obs_placement_info is None
        oa, _ = a.reset(seed=3)
        ob, _ = b.reset(seed=3)
        assert torch.equal(oa, ob)
        g = torch.Generator(device="cuda").manual_seed(1)
        for t in range(40):
            act = torch.randint(0, 3, (n, 2), device="cuda", generator=g, dtype=torch.int32)
            ra, rb = a.step(act), b.step(act)
            assert torch.equal(ra[0], rb[0]) and torch.equal(ra[1], rb[1]) and torch.equal(ra[2], rb[2]), "step %d" % t
        keep = a.obs
        want = keep.clone()
        a.close()
        b.close()
        del a
        torch.cuda.synchronize()
        assert torch.equal(keep, want)  # the buffer belongs to its tensors, not to the handle


def test_obs_alloc_small_and_no_search():
    import ctypes as C

    from memory_gym_amd import _native

    for nbytes, budget in ((1 << 20, _native.MG_OBS_SEARCH_DEFAULT), (300 << 20, 0)):
        p, info = C.c_void_p(), _native.ObsAllocInfo()
        _native.check(_native.LIB.mg_obs_alloc(0, nbytes, C.c_size_t(budget), C.byref(p), C.byref(info)), "mg_obs_alloc")
        assert p.value and info.zones == 0 and info.searched_bytes == 0
        assert _native.LIB.mg_obs_free(p) == 0
    assert _native.LIB.mg_obs_free(C.c_void_p(12345)) != 0


@pytest.mark.parametrize("env_id,adim,n_act", [("MortarMayhem-Grid-v0", 1, 4), ("SearingSpotlights-v0", 2, 3)])
def test_masked_reset_after_a_buffer_swap_leaves_no_stale_rows(env_id, adim, n_act):
    """use_obs_buffer() then reset(mask=...): the rows of the instances that are NOT reset must show their current frames in
    the new buffer, not whatever the buffer held (a gatherer that double-buffers would ship them)."""
    import memory_gym_amd
    import torch

    n = 512
    env = memory_gym_amd.make(env_id, num_envs=n, device=0)
    env.reset(seed=9)
    g = torch.Generator(device="cuda").manual_seed(2)
    for t in range(25):
        obs = env.step(torch.randint(0, n_act, (n,) if adim == 1 else (n, adim), device="cuda", generator=g, dtype=torch.int32))[0]
    before = obs.clone()
    other = torch.full_like(obs, 0xAB)
    env.use_obs_buffer(other)
    mask = torch.zeros(n, dtype=torch.bool, device="cuda")
    mask[::3] = True
    obs2, _ = env.reset(mask=mask)
    assert obs2.data_ptr() == other.data_ptr()
    assert torch.equal(obs2[~mask], before[~mask]), "rows of instances that were not reset are stale in the new buffer"
    assert not torch.equal(obs2[mask], before[mask])
    env.check_errors()
    env.close()
