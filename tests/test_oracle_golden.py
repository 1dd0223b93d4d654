"""Pin the oracle's LOGIC (state, reward, done, info, RNG consumption) to the reference:
replay tests/golden/logic_<env>.npz (captured from the unmodified reference under shims by
tests/golden/make_golden.py) through the CPU restatement and require exact agreement on every row."""
import json
import os

import numpy as np
import pytest

import oracle_lib

GOLDEN = os.environ.get("MEMGYM_GOLDEN_DIR", os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
ENV_IDS = ["MortarMayhem-Grid-v0", "MortarMayhem-v0", "Endless-MortarMayhem-v0", "MysteryPath-v0",
           "Endless-MysteryPath-v0", "SearingSpotlights-v0", "Endless-SearingSpotlights-v0", "MysteryPath-Grid-v0",
           "MortarMayhemB-Grid-v0", "MortarMayhemB-v0"]


def load(env_id, kind="logic"):
    return np.load(os.path.join(GOLDEN, kind + "_" + env_id.replace("-", "_") + ".npz"))


def sessions(env_id, kind="logic"):
    z = load(env_id, kind)
    meta = json.loads(str(z["meta"]))
    return [(env_id, i, kind) for i in range(len(meta))]


# "logic": hand-picked sessions; "fuzz": seeded random option dictionaries (tests/option_fuzz.py, make_golden.py --fuzz);
# "long": "sample one per episode" option lists of 9..41 entries (make_golden.py --long)
LONG_IDS = [e for e in ENV_IDS if os.path.exists(os.path.join(GOLDEN, "long_" + e.replace("-", "_") + ".npz"))]
# "fuzzd": the fuzz generators without their *_scale keys (make_golden.py --fuzz --default-geometry)
FUZZD_IDS = [e for e in ENV_IDS if os.path.exists(os.path.join(GOLDEN, "fuzzd_" + e.replace("-", "_") + ".npz"))]
ALL = ([s for e in ENV_IDS for s in sessions(e)] + [s for e in ENV_IDS for s in sessions(e, "fuzz")] + [s for e in LONG_IDS for s in sessions(e, "long")]
       + [s for e in FUZZD_IDS for s in sessions(e, "fuzzd")])


@pytest.mark.parametrize("env_id,si,kind", ALL, ids=["%s-%s%d" % (a[0], {"logic": "l", "fuzz": "f", "long": "lo", "fuzzd": "fd"}[a[2]], a[1]) for a in ALL])
def test_replay_matches_reference(env_id, si, kind):
    z = load(env_id, kind)
    meta = json.loads(str(z["meta"]))[si]
    fields = [str(f) for f in z["fields"]]
    p = "s%d_" % si
    kind, seed, action = z[p + "kind"], z[p + "seed"], z[p + "action"]
    reward, done, rng, snap = z[p + "reward"], z[p + "done"], z[p + "rng"], z[p + "snap"]
    lists = {k[len(p) + 2:]: z[k] for k in z.files if k.startswith(p + "L_")}
    try:
        env = oracle_lib.OracleEnv(env_id, 0.25)
    except ValueError:
        pytest.skip("oracle does not implement " + env_id + " yet")
    env.set_options(meta["options"])
    n_checked = 0
    for r in range(len(kind)):
        if kind[r] == 0:
            env.reset(None if seed[r] < 0 else int(seed[r]), want_obs=False)
            rw, dn = 0.0, False
        else:
            _, rw, dn = env.step(action[r], want_obs=False)
        ctx = "%s session %d row %d (kind %d action %s)" % (env_id, si, r, kind[r], action[r].tolist())
        assert rw == reward[r], ctx + " reward %r != %r" % (rw, reward[r])
        assert dn == bool(done[r]), ctx + " done"
        assert np.array_equal(env.rng_words(), rng[r]), ctx + " RNG state diverged"
        for fi, f in enumerate(fields):
            exp = snap[r, fi]
            if np.isnan(exp):
                continue
            got = env.get(f)
            assert got is not None, ctx + " oracle lacks field " + f
            assert got == exp, ctx + " field %s: oracle %r reference %r" % (f, got, exp)
            n_checked += 1
        for name, arr in lists.items():
            exp = arr[r]
            exp = exp[~np.isnan(exp)]
            got = env.get_list(name)
            assert got is not None, ctx + " oracle lacks list " + name
            assert len(got) == len(exp) and np.array_equal(got, exp), ctx + " list %s: oracle %s reference %s" % (name, got, exp)
    assert n_checked > 0
    env.close()
