"""GPU (-m gpu): replay the REFERENCE's recorded sessions (tests/golden/{logic,fuzz,fuzzd,long}_*.npz: captured from the unmodified
reference by tests/golden/make_golden.py) straight through the HIP path -- the public single-instance API over the C ABI --
without the oracle in between: reward (the reference's Python float, bit for bit), done, the numpy PCG64 words after every
call, info["ground_truth"] and the terminal info dict must equal what the reference produced."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
ENV_IDS = ["MortarMayhem-Grid-v0", "MortarMayhem-v0", "Endless-MortarMayhem-v0", "MysteryPath-v0",
           "Endless-MysteryPath-v0", "SearingSpotlights-v0", "Endless-SearingSpotlights-v0", "MysteryPath-Grid-v0",
           "MortarMayhemB-Grid-v0", "MortarMayhemB-v0"]


def load(env_id, kind):
    return np.load(os.path.join(GOLDEN, kind + "_" + env_id.replace("-", "_") + ".npz"))


# "long": option lists of 9..41 entries (tests/golden/make_golden.py --long), incl. lists beyond the 32 entries the kernels
# keep in their arguments
CASES = [(e, k) for e in ENV_IDS for k in ("logic", "fuzz", "fuzzd")] + [
    (e, "long") for e in ENV_IDS if os.path.exists(os.path.join(GOLDEN, "long_" + e.replace("-", "_") + ".npz"))]


@pytest.mark.parametrize("env_id,kind", CASES, ids=["%s-%s" % c for c in CASES])
def test_reference_sessions_through_hip(env_id, kind):
    import memory_gym_amd

    z = load(env_id, kind)
    metas = json.loads(str(z["meta"]))
    fields = [str(f) for f in z["fields"]]
    info_cols = {f[5:]: i for i, f in enumerate(fields) if f.startswith("info_")}
    gt_cols = [i for i, f in enumerate(fields) if f.startswith("gt")]
    env = memory_gym_amd.make(env_id)
    disc = env.vec.action_dim == 1
    n_rows = n_term = n_refused = 0
    for si, meta in enumerate(metas):
        p = "s%d_" % si
        rows, seed, action = z[p + "kind"], z[p + "seed"], z[p + "action"]
        reward, done, rng, snap = z[p + "reward"], z[p + "done"], z[p + "rng"], z[p + "snap"]
        try:
            for r in range(len(rows)):
                ctx = "%s %s session %d row %d" % (env_id, kind, si, r)
                if rows[r] == 0:
                    _, info = env.reset(seed=None if seed[r] < 0 else int(seed[r]), options=meta["options"])
                else:
                    a = int(action[r][0]) if disc else action[r].astype(np.int64)
                    _, rw, dn, trunc, info = env.step(a)
                    assert rw == reward[r], ctx + ": reward %r, reference %r" % (rw, reward[r])
                    assert dn == bool(done[r]) and trunc is False, ctx + ": done"
                    if dn:
                        n_term += 1
                        for name, col in info_cols.items():
                            if np.isnan(snap[r, col]):
                                continue
                            assert name in info, ctx + ": terminal info lacks " + name
                            # "reward" (a double sum) and "length" are exact; the other entries travel as float32
                            same = info[name] == snap[r, col] if name in ("reward", "length") else np.float32(info[name]) == np.float32(snap[r, col])
                            assert same, ctx + ": info[%s] = %r, reference %r" % (name, info[name], snap[r, col])
                assert np.array_equal(env.vec.rng_words(0), rng[r]), ctx + ": RNG state diverged from the reference's"
                if gt_cols and not np.isnan(snap[r, gt_cols[0]]):
                    want = snap[r, gt_cols].astype(np.float32)
                    assert np.array_equal(info["ground_truth"].astype(np.float32)[:len(want)], want), ctx + ": ground truth"
                n_rows += 1
        except NotImplementedError:  # an option value this build refuses loudly; the session is skipped
            n_refused += 1
    env.close()
    assert n_refused * 2 <= len(metas), "most sessions were refused"
    assert n_rows > 100 and (n_term > 0 or "Endless" in env_id)
