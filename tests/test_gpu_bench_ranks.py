"""GPU (-m gpu): the N > 1 path of bench.py end to end on ONE GPU -- `python bench.py --gpus 2` spawns its two ranks itself,
both ranks share GPU 0 (MEMGYM_BENCH_ONE_DEVICE=1) and rendezvous over gloo (RCCL refuses two ranks on one device), the
instances are sharded, rank 0 prints one JSON line with n_gpus = 2 and the config-5 variants.  What this cannot show is
RCCL / xGMI itself; no multi-GPU node was available to the build."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_self_launched_two_ranks():
    env = dict(os.environ, MEMGYM_BENCH_ONE_DEVICE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--steps", "12", "--warmup", "3",
                          "--settle", "30", "--envs-per-gpu", "4096", "--config5-envs", "4096", "--config5-steps", "24"], env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, "exactly one JSON line, from rank 0: %r" % out.stdout[-500:]
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["config"]["envs_total"] == 8192 and j["config"]["envs_per_gpu"] == 4096
    assert j["value"] > 5e5 and abs(j["per_gpu_value"] * 2 - j["value"]) < 1e-3 * j["value"] and j["scaling"] == "weak"
    assert "cpu_baseline" not in j  # rank 0, N = 1 only
    assert j["ranks"]["world"] == 2 and j["ranks"]["backend"] == "gloo" and [d["rank"] for d in j["ranks"]["devices"]] == [0, 1]
    assert j["value_2000"]["steps"] == 2000 and j["value_2000"]["value"] > 5e5 and "ended_early" not in j
    c5 = j["config5"]
    assert c5["workload"].startswith("Endless-MortarMayhem-v0, 4096 envs/GPU x 2")  # (--config5-envs: the default is BASELINE's 32,768)
    assert c5["no_gather"]["value"] > 5e5
    assert isinstance(c5["gather_peer"], dict) and c5["gather_peer"]["value"] > 1e5, c5["gather_peer"]  # peer-mapped stores (same device here)
    # gather_rccl needs the nccl backend (gloo has no CUDA gather): under this test it must fail softly, not take the line down
    assert isinstance(c5["gather_rccl"], (dict, str))


def test_killed_rank_mid_leg_keeps_the_headline():
    """The real measurement, two ranks on one GPU: rank 1 dies inside the second config-5 leg (MEMGYM_BENCH_TEST_FAULT); rank 0's ONE
    line still appears -- headline, `ranks`, the leg that finished -- and says what ended the run (tests/test_bench_guard.py has the
    other failure modes with a stand-in for the measurement)."""
    env = dict(os.environ, MEMGYM_BENCH_ONE_DEVICE="1", HSA_ENABLE_IPC_MODE_LEGACY="0", MEMGYM_BENCH_TEST_FAULT="1:gather_peer:exit")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--steps", "12", "--warmup", "3", "--long-window", "200",
                          "--settle", "30", "--envs-per-gpu", "4096", "--config5-envs", "4096", "--config5-steps", "24", "--leg-limit", "120"], env=env, capture_output=True, text=True, timeout=900)
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, "exactly one JSON line, from rank 0: %r / %r" % (out.stdout[-500:], out.stderr[-1500:])
    j = json.loads(lines[0])
    assert out.returncode == 0 and j["n_gpus"] == 2 and j["value"] > 5e5
    assert j["ranks"]["world"] == 2 and len(j["ranks"]["devices"]) == 2 and j["ranks"]["devices"][1]["name"]
    assert j["value_2000"]["steps"] == 200 and j["value_2000"]["value"] > 5e5
    assert j["config5"]["no_gather"]["value"] > 5e5
    assert "ended_early" in j or "failed" in json.dumps(j["config5"].get("gather_peer", ""))


def test_under_torch_distributed_run_like_the_driver():
    """The driver's own launch line for N > 1 -- `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1
    --master-port P bench.py --gpus N --steps K --warmup W` -- with both ranks on GPU 0 and gloo in RCCL's place: one JSON line on the
    launcher's stdout, from rank 0, with the N-rank headline, `ranks` and the config-5 legs."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, MEMGYM_BENCH_ONE_DEVICE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                          "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "12", "--warmup", "3", "--backend", "gloo",
                          "--settle", "30", "--envs-per-gpu", "4096", "--config5-envs", "4096", "--config5-steps", "24", "--long-window", "100"],
                         env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, "exactly one JSON line: %r" % out.stdout[-500:]
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["value"] > 5e5 and j["ranks"]["world"] == 2 and j["value_2000"]["steps"] == 100
    assert j["config5"]["no_gather"]["value"] > 5e5 and "ended_early" not in j
