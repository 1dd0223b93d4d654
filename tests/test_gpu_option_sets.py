"""GPU (-m gpu): per-instance reset options (include/memgym.h: mg_set_option_set / mg_bind_option_sets).

In the reference reset(seed, options) belongs to ONE environment instance (mortar_mayhem_grid.py:213-236): a pool of workers
runs different curricula side by side.  reset(options=..., mask=...) of the Python mirror means that: the options belong to
the instances that are being reset.  Checked: one handle whose halves (and, later, a third group) run under different
`command_count` / reward / `max_steps` / display options, each group bit-exact against its OWN oracle batch -- frames,
rewards, dones, RNG streams, through same-step auto-resets and a checkpoint; and reference sessions recorded under different
option dictionaries (tests/golden/fuzzd_*.npz) replayed two per handle."""
import glob
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))

CASES = [
    ("MortarMayhem-Grid-v0", dict(command_count=[3], reward_command_success=0.5, explosion_delay=[4]),
     dict(command_count=[5, 7], command_show_duration=[2], reward_command_failure=-0.25, visual_feedback=False)),
    ("MortarMayhem-v0", dict(command_count=[2], allowed_commands=5), dict(command_count=[4], reward_episode_success=1.0, explosion_duration=[4])),
    ("Endless-MortarMayhem-v0", dict(max_steps=40, reward_new_command_success=0.3), dict(max_steps=25, initial_command_count=3, command_show_delay=[2])),
    ("MortarMayhemB-Grid-v0", dict(command_count=[4]), dict(command_count=[9], reward_command_success=0.2)),
    ("Endless-SearingSpotlights-v0", dict(steps_per_coin=40, reward_coin=0.5, agent_health=4, coins_visible=True),
     dict(spawn_interval=20, initial_spawns=5, coin_show_duration=2, spot_damage=2.0, max_steps=70)),
    ("SearingSpotlights-v0", dict(num_coins=[2], reward_exit=2.0, sample_agent_position=False),
     dict(num_coins=[1, 3], agent_health=2, coins_visible=True, max_steps=60, black_background=True)),
    ("MysteryPath-Grid-v0", dict(max_steps=24, reward_goal=2.0, show_origin=True, cardinal_origin_choice=[0, 2]),
     dict(max_steps=40, reward_fall_off=-0.1, reward_step=-0.01, show_goal=True, visual_feedback=False)),
    ("MysteryPath-v0", dict(max_steps=30, reward_path_progress=0.2, show_goal=True), dict(max_steps=50, cardinal_origin_choice=[1, 3], show_origin=True)),
    ("Endless-MysteryPath-v0", dict(max_steps=40, stamina_level=10, show_stamina=True, reward_path_progress_dense=0.05),
     dict(max_steps=64, show_past_path=False, show_background=True, reward_fall_off=-0.2)),
]


def _full(env_id, opts):
    from memory_gym_amd.reset_params import process_reset_params
    return process_reset_params(env_id, opts)


@pytest.mark.parametrize("env_id,opt_a,opt_b", CASES, ids=[c[0] for c in CASES])
def test_two_halves_two_option_sets(env_id, opt_a, opt_b):
    import memory_gym_amd
    import oracle_lib
    import torch

    n = 256
    env = memory_gym_amd.make(env_id, num_envs=n, device=0)
    half = n // 2
    idx_a, idx_b = np.arange(0, half), np.arange(half, n)
    ref_a = oracle_lib.OracleBatch(env_id, half, options=_full(env_id, opt_a))
    ref_b = oracle_lib.OracleBatch(env_id, half, options=_full(env_id, opt_b))
    seeds = np.arange(n, dtype=np.int64) + 11
    mask_a = torch.zeros(n, dtype=torch.bool, device="cuda")
    mask_a[:half] = True
    env.reset(seed=seeds)  # every instance under the defaults first
    env.reset(seed=seeds, options=opt_a, mask=mask_a)
    obs, _ = env.reset(seed=seeds, options=opt_b, mask=~mask_a)
    vis = obs["visual_observation"] if isinstance(obs, dict) else obs
    assert np.array_equal(vis[:half].cpu().numpy(), ref_a.reset(seeds[:half])), "reset frames of the first half"
    assert np.array_equal(vis[half:].cpu().numpy(), ref_b.reset(seeds[half:])), "reset frames of the second half"
    disc = env.action_dim == 1
    g = np.random.Generator(np.random.PCG64(5))
    sd = None
    for t in range(120):
        a = (g.integers(0, 4, n) if disc else g.integers(0, 3, (n, 2))).astype(np.int32)
        obs, rew, done, _, _ = env.step(a)
        vis = obs["visual_observation"] if isinstance(obs, dict) else obs
        for ref, ix, name in ((ref_a, idx_a, "first"), (ref_b, idx_b, "second")):
            o2, r2, d2 = ref.step(a[ix], autoreset=True)
            assert np.array_equal(done[ix].cpu().numpy(), d2.astype(bool)), "%s half: done differs at step %d" % (name, t)
            assert np.array_equal(rew[ix].cpu().numpy(), r2.astype(np.float32)), "%s half: reward differs at step %d" % (name, t)
            assert np.array_equal(vis[ix].cpu().numpy(), o2), "%s half: frames differ at step %d" % (name, t)
        if t == 60:  # a checkpoint carries the sets and who runs under which
            sd = env.state_dict()
            env.close()
            env = memory_gym_amd.make(env_id, num_envs=n, device=0)
            env.load_state_dict(sd)
    for ref, ix in ((ref_a, idx_a), (ref_b, idx_b)):
        for j in (0, len(ix) - 1):
            assert np.array_equal(env.rng_words(int(ix[j])), ref.envs[j].rng_words())
    env.check_errors()
    env.close()


def test_geometry_options_are_refused_for_a_subset_instead_of_reparametrising_everybody():
    import memory_gym_amd
    import torch

    env = memory_gym_amd.make("Endless-MysteryPath-v0", num_envs=64, device=0)
    env.reset(seed=1)
    mask = torch.zeros(64, dtype=torch.bool, device="cuda")
    mask[:10] = True
    env.reset(mask=mask)  # options=None: the instances keep what they have
    env.reset(options=dict(), mask=mask)  # the defaults they run under anyway: nothing to refuse
    with pytest.raises(NotImplementedError):
        env.reset(options=dict(camera_offset_scale=3.0), mask=mask)  # the camera is the handle's
    env.close()
    env = memory_gym_amd.make("SearingSpotlights-v0", num_envs=64, device=0)  # a geometry option cannot differ between instances
    env.reset(seed=1)
    with pytest.raises(NotImplementedError):
        env.reset(options=dict(coin_scale=0.5), mask=mask)
    env.close()


@pytest.mark.parametrize("env_id,geo,other", [("SearingSpotlights-v0", dict(agent_scale=0.5), dict(num_coins=[2])),
                                              ("MortarMayhem-v0", dict(agent_scale=0.4), dict(command_count=[5])),
                                              ("MysteryPath-v0", dict(agent_scale=0.5), dict(max_steps=40)),
                                              ("Endless-MysteryPath-v0", dict(camera_offset_scale=3.0), dict(stamina_level=9))])
def test_a_subset_cannot_silently_run_under_the_handles_non_default_geometry(env_id, geo, other):
    """the handle's geometry is NOT the reference's default; a masked reset whose options hold the default geometry
    value (by not naming the key) asks for a geometry the handle does not have -- that must be refused, not run under the
    handle's.  With the handle's geometry named, the subset gets its set; a later FULL reset that changes the geometry forgets
    the sets (their geometry entries would be stale) and unbinds the per-instance index."""
    import memory_gym_amd
    import oracle_lib
    import torch

    n = 32
    env = memory_gym_amd.make(env_id, num_envs=n, device=0)
    seeds = np.arange(n, dtype=np.int64) + 5
    env.reset(seed=seeds, options=geo)
    mask = torch.zeros(n, dtype=torch.bool, device="cuda")
    mask[: n // 2] = True
    with pytest.raises(NotImplementedError):
        env.reset(seed=seeds, options=other, mask=mask)  # (other) + default geometry != the handle's geometry
    both = dict(geo, **other)
    obs, _ = env.reset(seed=seeds, options=both, mask=mask)
    ref = oracle_lib.OracleBatch(env_id, n // 2, options=_full(env_id, both))
    vis = obs["visual_observation"] if isinstance(obs, dict) else obs
    assert np.array_equal(vis[: n // 2].cpu().numpy(), ref.reset(seeds[: n // 2]))
    ref.close()
    assert env._set_of is not None
    # a full reset back to the reference's defaults: one set again, nothing bound
    obs, _ = env.reset(seed=seeds)
    assert env._set_of is None and len(env._set_params) == 1
    ref = oracle_lib.OracleBatch(env_id, n)
    vis = obs["visual_observation"] if isinstance(obs, dict) else obs
    assert np.array_equal(vis.cpu().numpy(), ref.reset(seeds))
    ref.close()
    # the set written above named the OLD geometry: asking for it again is a geometry mismatch now, not a match with a stale set
    with pytest.raises(NotImplementedError):
        env.reset(seed=seeds, options=both, mask=mask)
    obs, _ = env.reset(seed=seeds, options=other, mask=mask)  # (other) + default geometry == the handle's geometry now
    ref = oracle_lib.OracleBatch(env_id, n // 2, options=_full(env_id, other))
    vis = obs["visual_observation"] if isinstance(obs, dict) else obs
    assert np.array_equal(vis[: n // 2].cpu().numpy(), ref.reset(seeds[: n // 2]))
    ref.close()
    env.close()


def test_masked_reset_validates_before_it_touches_option_state():
    import memory_gym_amd
    import torch

    env = memory_gym_amd.make("MortarMayhem-Grid-v0", num_envs=8, device=0)
    mask = torch.ones(8, dtype=torch.bool, device="cuda")
    with pytest.raises(RuntimeError, match="earlier full reset"):
        env.reset(options=dict(command_count=[3]), mask=mask)
    assert env._set_of is None and len(env._set_params) == 1
    env.close()


@pytest.mark.parametrize("env_id", ["MortarMayhem-Grid-v0", "MortarMayhem-v0", "Endless-MortarMayhem-v0", "MortarMayhemB-Grid-v0", "MortarMayhemB-v0",
                                    "SearingSpotlights-v0", "Endless-SearingSpotlights-v0", "MysteryPath-v0", "MysteryPath-Grid-v0",
                                    "Endless-MysteryPath-v0"])
def test_reference_sessions_two_per_handle(env_id):
    """Recorded sessions of the UNMODIFIED reference under different random option dictionaries (tests/golden/fuzzd_*.npz),
    two at a time through ONE handle: instance 0 replays one session, instance 1 the next -- each reset is a masked reset with
    that session's options.  Rewards as the reference's Python floats, dones and the numpy PCG64 words after every call."""
    import memory_gym_amd
    import torch

    # (fuzzd: the fuzz generators without their *_scale keys -- geometry belongs to the handle, so two sessions can share one)
    z = np.load(os.path.join(HERE, "golden", "fuzzd_" + env_id.replace("-", "_") + ".npz"))
    metas = json.loads(str(z["meta"]))
    disc = None
    pairs = checked = refused = 0
    # options that fix the geometry the handle's instances share (atlases, templates) cannot differ between two instances of one
    # handle: sessions are paired with a partner of the same geometry, and the handle is brought to it by a full reset first
    geo = lambda o: tuple(repr(o.get(k)) for k in (  # noqa: E731
        "arena_size", "agent_scale", "agent_speed", "coin_scale", "show_last_action", "initial_spawn_interval", "spawn_interval_threshold", "exit_scale",
        "camera_offset_scale"))
    order = sorted(range(len(metas)), key=lambda j: geo(metas[j]["options"]))
    todo = [(order[j], order[j + 1]) for j in range(len(order) - 1) if geo(metas[order[j]]["options"]) == geo(metas[order[j + 1]]["options"])]
    # (make_golden.py --fuzz --default-geometry records one extra session per id with the geometry of its trial 0: no id is left without a pair)
    assert todo, "no two recorded sessions of %s share their geometry options" % env_id
    for sa, sb in todo:
        env = memory_gym_amd.make(env_id, num_envs=2, device=0)
        env.autoreset = False
        disc = env.action_dim == 1
        try:
            env.reset(seed=np.zeros(2, dtype=np.int64), options=metas[sa]["options"])
        except NotImplementedError:
            refused += 1
            env.close()
            continue
        sess = []
        for k, sj in enumerate((sa, sb)):
            p = "s%d_" % sj
            sess.append(dict(kind=z[p + "kind"], seed=z[p + "seed"], action=z[p + "action"], reward=z[p + "reward"], done=z[p + "done"],
                             rng=z[p + "rng"], options=metas[sj]["options"], row=0))
        masks = [torch.tensor([True, False], device="cuda"), torch.tensor([False, True], device="cuda")]
        try:
            while all(q["row"] < len(q["kind"]) for q in sess):
                progressed = False
                for k, q in enumerate(sess):  # resets first: they touch their own instance only
                    r = q["row"]
                    if q["kind"][r] == 0:
                        sd = q["seed"][r]
                        env.reset(seed=None if sd < 0 else np.full(2, int(sd), dtype=np.int64), options=q["options"], mask=masks[k])
                        assert np.array_equal(env.rng_words(k), q["rng"][r]), "%s sessions %d+%d: RNG after the reset of instance %d (row %d)" % (env_id, sa, sb, k, r)
                        q["row"] += 1
                        progressed = True
                if progressed:
                    continue
                a = np.zeros((2, 1 if disc else 2), dtype=np.int32)
                for k, q in enumerate(sess):
                    a[k, :] = q["action"][q["row"]][:a.shape[1]]
                env.step(a[:, 0] if disc else a)
                rw = env.reward64.cpu().numpy()
                dn = env.done_u8.cpu().numpy()
                for k, q in enumerate(sess):
                    r = q["row"]
                    ctx = "%s sessions %d+%d instance %d row %d" % (env_id, sa, sb, k, r)
                    assert rw[k] == q["reward"][r], ctx + ": reward %r, reference %r" % (rw[k], q["reward"][r])
                    assert bool(dn[k]) == bool(q["done"][r]), ctx + ": done"
                    assert np.array_equal(env.rng_words(k), q["rng"][r]), ctx + ": RNG state diverged from the reference's"
                    q["row"] += 1
                    checked += 1
            pairs += 1
        except NotImplementedError:  # an option this build refuses, or a geometry option that cannot differ between instances
            refused += 1
        env.close()
    assert pairs >= 1 and checked > 100, "pairs replayed: %d (refused %d), rows checked: %d" % (pairs, refused, checked)
