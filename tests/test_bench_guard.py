"""CPU: `bench.py --gpus N` cannot end without its headline line.  The control flow of an N-rank run --
rendezvous with a timeout, every rank's report through the store, the headline, the three config-5 legs with their agreement
before and after, the line guard's watchdog -- runs here with a stand-in for the measurement (MEMGYM_BENCH_FAKE=1: a few gloo
all-reduces per leg, no GPU), and MEMGYM_BENCH_TEST_FAULT makes one rank die, raise or hang inside one leg.  In every case rank 0's
ONE JSON line must appear, with the headline intact, well inside the limit -- under bench.py's own launcher and under
torch.distributed.run (the driver's)."""
import json
import os
import socket
import subprocess
import sys
import time

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run(fault=None, launcher="self", leg_limit=30.0, timeout=120):
    env = dict(os.environ, MEMGYM_BENCH_FAKE="1")
    env.pop("MEMGYM_BENCH_TEST_FAULT", None)
    if fault:
        env["MEMGYM_BENCH_TEST_FAULT"] = fault
    args = ["--gpus", "2", "--backend", "gloo", "--steps", "10", "--warmup", "2", "--leg-limit", str(leg_limit)]
    if launcher == "self":
        cmd = [sys.executable, BENCH] + args
    else:
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
               "--master-port", str(port), BENCH] + args
    t0 = time.monotonic()
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout)
    dt = time.monotonic() - t0
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, "exactly one JSON line from rank 0; stdout %r stderr %r" % (p.stdout[-400:], p.stderr[-1500:])
    return json.loads(lines[0]), p, dt


def test_healthy_run_has_ranks_and_three_legs():
    j, p, dt = _run()
    assert p.returncode == 0, p.stderr[-1500:]
    assert j["n_gpus"] == 2 and j["value"] > 0 and "ended_early" not in j
    assert j["ranks"]["world"] == 2 and j["ranks"]["backend"] == "gloo" and [d["rank"] for d in j["ranks"]["devices"]] == [0, 1]
    assert j["ranks"]["process_group_timeout_s"] == 120.0
    assert all(isinstance(j["config5"][k], dict) for k in ("no_gather", "gather_rccl", "gather_peer"))


@pytest.mark.parametrize("launcher", ["self", "torchrun"])
@pytest.mark.parametrize("fault", ["1:gather_rccl:exit", "1:no_gather:raise", "0:gather_rccl:raise", "1:gather_peer:exit"])
def test_a_rank_that_dies_or_raises_mid_leg_does_not_take_the_headline(fault, launcher):
    j, p, dt = _run(fault, launcher)
    assert dt < 60, "the line took %.0f s" % dt
    assert j["n_gpus"] == 2 and j["value"] > 0 and j["ranks"]["world"] == 2
    rank, leg, mode = fault.split(":")
    done_before = {"no_gather": [], "gather_rccl": ["no_gather"], "gather_peer": ["no_gather", "gather_rccl"]}[leg]
    for k in done_before:  # the legs that finished before the fault are in the line
        assert isinstance(j["config5"][k], dict) and j["config5"][k]["value"] > 0, j["config5"]
    said = json.dumps(j.get("config5", {}).get(leg, "")) + j.get("ended_early", "")
    assert "ended_early" in j or "failed" in said, j
    if launcher == "self":
        assert p.returncode == 0, "rank 0 printed its line: the launcher reports rank 0's word (stderr %r)" % p.stderr[-800:]


def test_a_rank_that_hangs_is_ended_by_the_watchdog():
    j, p, dt = _run("1:no_gather:hang", "self", leg_limit=6.0)
    assert dt < 45, "the line took %.0f s" % dt
    assert j["value"] > 0 and "watchdog" in j["ended_early"] and "no_gather" in j["ended_early"], j.get("ended_early")
