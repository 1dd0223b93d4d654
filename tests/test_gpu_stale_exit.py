"""GPU (-m gpu): `use_exit = False` on an instance that has had an exit before, replayed from the reference's own sessions
(tests/golden/stale_exit.npz, tests/golden/make_stale_exit_fixture.py) straight through the HIP path: reward as the reference's
Python float, done, the PCG64 words after every call; every frame against the oracle (which the CPU suite pins to the same
fixture).  And the refused case: an instance that never had an exit raises error bit 256 where the reference raises."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
ENV_ID = "SearingSpotlights-v0"


def test_reference_sessions_with_a_stale_exit():
    import memory_gym_amd
    import oracle_lib
    from memory_gym_amd.reset_params import process_reset_params

    z = np.load(os.path.join(HERE, "golden", "stale_exit.npz"))
    options = json.loads(str(z["options"]))
    env = memory_gym_amd.make(ENV_ID)
    ref = oracle_lib.OracleEnv(ENV_ID, scale=0.25)
    n_off = 0
    for k in range(len(z["kind"])):
        opts = options[int(z["phase"][k])]
        if z["kind"][k] == 0:
            seed = int(z["seed"][k])
            obs, _ = env.reset(seed=None if seed < 0 else seed, options=opts)
            want = ref.reset(None if seed < 0 else seed, options=process_reset_params(ENV_ID, opts))
        else:
            a = np.array([int(z["a0"][k]), int(z["a1"][k])])
            obs, r, d, _, info = env.step(a)
            want, _, _ = ref.step([int(a[0]), int(a[1])])
            assert r == z["reward"][k] and d == bool(z["done"][k]), "row %d: reward %r / done %r, reference %r / %r" % (k, r, d, z["reward"][k], z["done"][k])
            if d:
                assert info["success"] == z["success"][k] and info["reward"] == z["info_reward"][k], "row %d: terminal info" % k
        assert np.array_equal(env.vec.rng_words(0), z["rng"][k]), "row %d: the PCG64 stream diverged from the reference's" % k
        assert np.array_equal(obs, want), "row %d (phase %d): frame differs from the oracle's" % (k, int(z["phase"][k]))
        n_off += int(opts.get("use_exit", True) is False)
    assert n_off > 300
    env.vec.check_errors()
    env.close()
    ref.close()


def test_use_exit_false_without_an_earlier_exit_is_flagged():
    import memory_gym_amd

    env = memory_gym_amd.make(ENV_ID, num_envs=8, device=0)
    env.reset(seed=3, options=dict(use_exit=False))
    with pytest.raises(RuntimeError, match="0x100"):
        env.check_errors()
    env.close()
