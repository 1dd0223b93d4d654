"""GPU parity (-m gpu) for MortarMayhemB-Grid-v0 / MortarMayhemB-v0: HIP path through the C ABI vs the
CPU oracle -- frames, rewards, dones, RNG words, end-of-episode info and the one-hot "vector_observation"."""
import numpy as np
import pytest

from gpu_parity import check_terminal_info, run_parity

pytestmark = pytest.mark.gpu


def _toward(d):
    return 0 if d == 0 else (1 if d < 0 else 2)


def grid_expert(e, prng):
    """Rotate towards / step onto the target tile, wait while the death tiles are on; occasional mistakes."""
    if prng.random() > 0.95:
        return [int(prng.integers(0, 4)), 0]
    if e.get("tiles_on"):
        return [0, 0]
    gx, gy, tx, ty = e.get("nx"), e.get("ny"), e.get("tx"), e.get("ty")
    if (gx, gy) == (tx, ty):
        return [0, 0]
    want = 270 if tx > gx else (90 if tx < gx else (0 if ty < gy else 180))
    rot = e.get("arot")
    if rot == want:
        return [3, 0]
    return [1 if (want - rot) % 360 in (90, 180) else 2, 0]


def free_expert(e, prng):
    if prng.random() > 0.97:
        return [int(prng.integers(0, 3)), int(prng.integers(0, 3))]
    if e.get("tiles_on"):
        return [0, 0]
    n = 5
    x0 = 42 - 14 * n / 2
    dx = x0 + 14 * e.get("tx") + 7 - e.get("ax")
    dy = x0 + 14 * e.get("ty") + 7 - e.get("ay")
    dx = 0 if abs(dx) < 3 else dx
    dy = 0 if abs(dy) < 3 else dy
    return [_toward(dx), _toward(dy)]


GRID_OPTS = [
    None,
    dict(arena_size=6, allowed_commands=9, command_count=[3, 5, 20], explosion_duration=[2, 3], explosion_delay=[4, 6, 8],
         reward_command_failure=-0.1, reward_episode_success=1.0),
    dict(arena_size=3, allowed_commands=4, command_count=[4], visual_feedback=False),
]
FREE_OPTS = [
    None,
    dict(command_count=[3, 6, 20], explosion_duration=[4, 6], explosion_delay=[12, 18], reward_command_failure=-0.5,
         reward_episode_success=2.0, allowed_commands=5),
]


@pytest.mark.parametrize("opt_idx", range(len(GRID_OPTS)))
def test_grid_parity(opt_idx):
    n_done = run_parity("MortarMayhemB-Grid-v0", GRID_OPTS[opt_idx], n=160, steps=260, policy=grid_expert if opt_idx == 0 else None,
                        n_policy=64)
    assert n_done > 0


@pytest.mark.parametrize("opt_idx", range(len(FREE_OPTS)))
def test_free_parity(opt_idx):
    n_done = run_parity("MortarMayhemB-v0", FREE_OPTS[opt_idx], n=160, steps=420, policy=free_expert if opt_idx == 0 else None,
                        n_policy=64)
    assert n_done > 0


def test_terminal_info():
    assert check_terminal_info("MortarMayhemB-Grid-v0", steps=120) > 0
    assert check_terminal_info("MortarMayhemB-v0", steps=200) > 0


def test_api_shape_and_assertion():
    import memory_gym_amd

    env = memory_gym_amd.make("MortarMayhemB-Grid-v0", num_envs=4, device=0)
    obs, info = env.reset(seed=0)
    assert set(obs) == {"visual_observation", "vector_observation"}
    assert obs["visual_observation"].shape == (4, 84, 84, 3) and obs["vector_observation"].shape == (4, 180)
    v = obs["vector_observation"].cpu().numpy()
    assert ((v == 0) | (v == 1)).all() and (v.reshape(4, 20, 9).sum(2)[:, :10] == 1).all() and v.reshape(4, 20, 9)[:, 10:].sum() == 0
    assert env.max_episode_steps == (6 + 2) * 10 - 2 + 1
    with pytest.raises(AssertionError, match="20 commands are allowed at maximum"):
        env.reset(seed=0, options=dict(command_count=[21]))
    env.close()
    single = memory_gym_amd.make("MortarMayhemB-v0")
    o, _ = single.reset(seed=3)
    assert o["visual_observation"].shape == (84, 84, 3) and o["vector_observation"].dtype == np.float32
    o, r, d, t, i = single.step([0, 0])
    assert isinstance(r, float) and t is False
    single.close()
