"""GPU (-m gpu): randomised sweep over the reset-options space.  For every env id a seeded generator draws option
dictionaries over every key of the reference's reset-options dictionaries (tests/option_fuzz.py: lists for the "sample one
per episode" keys, rewards, durations, sizes, speeds and the *_scale geometry options), and the HIP path must stay bit-exact
with the oracle under random actions.  The same generators produce the reference-recorded tests/golden/fuzz_*.npz."""
import os

import numpy as np
import pytest

from gpu_parity import run_parity

pytestmark = pytest.mark.gpu


from option_fuzz import CASES  # noqa: E402


@pytest.mark.parametrize("env_id,gen", CASES, ids=[c[0] for c in CASES])
def test_random_option_sets(env_id, gen):
    import memory_gym_amd

    # MEMGYM_FUZZ_TRIALS / MEMGYM_FUZZ_SEED: one-off hunts with more and other draws
    trials = int(os.environ.get("MEMGYM_FUZZ_TRIALS", "10"))
    rng = np.random.Generator(np.random.PCG64(sum(map(ord, env_id)) + int(os.environ.get("MEMGYM_FUZZ_SEED", "0"))))
    tried = 0
    for trial in range(trials):
        options = gen(rng, env_id)
        try:  # ranges the build rejects raise from both sides alike; skip those draws (they are errors, not mismatches)
            memory_gym_amd.reset_params.process_reset_params(env_id, options)
            run_parity(env_id, options, n=64, steps=140, seed0=11 + trial)
        except NotImplementedError:
            continue
        tried += 1
    assert tried >= 3
