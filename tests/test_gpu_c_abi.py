"""GPU (-m gpu): the drop-in boundary without Python in the loop.  tests/c_abi/c_abi_parity.cpp is a C++ host program
that links libmemgym_hip.so (include/memgym.h) and the oracle library, owns its hipMalloc'd buffers and stream, and
compares every frame / reward / done bit-exactly -- what a compiled trainer would do with the library."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "tests", "c_abi", "c_abi_parity")


def build_binary():
    import __graft_entry__
    return __graft_entry__.build_c_abi_test()


@pytest.mark.parametrize("env_id,steps", [("MortarMayhem-Grid-v0", 100), ("Endless-MysteryPath-v0", 100), ("SearingSpotlights-v0", 90),
                                          ("Endless-MortarMayhem-v0", 100)])
def test_c_host_program_matches_the_oracle(env_id, steps):
    exe = build_binary()
    env = dict(os.environ)
    env["LD_LIBRARY_PATH"] = os.pathsep.join([os.path.join(ROOT, "endless-memory-gym_amd", "lib"), os.path.join(ROOT, "oracle", "_build"),
                                              "/opt/rocm/lib", env.get("LD_LIBRARY_PATH", "")])
    out = subprocess.run([exe, env_id, "96", str(steps)], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    assert out.stdout.startswith("OK " + env_id)
