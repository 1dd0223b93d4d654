"""Worker of tests/test_gpu_switches.py: one lock-step parity run (HIP vs oracle) in a fresh process, so that the
library reads the MEMGYM_* switches of the environment it was started with.  Usage: switch_worker.py ENV_ID N STEPS"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path[:0] = [HERE, os.path.join(ROOT, "endless-memory-gym_amd"), ROOT]

from gpu_parity import run_parity  # noqa: E402

if __name__ == "__main__":
    env_id, n, steps = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    want = [w for w in os.environ.get("MEMGYM_SWITCH_WORKER_WANT", "").split(",") if w]  # counters that must have moved
    done = run_parity(env_id, None, n=n, steps=steps, want_counters=want)
    print("ok: %s, %d instances x %d steps, %d episodes ended" % (env_id, n, steps, done))
