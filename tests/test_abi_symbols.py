"""CPU test of the boundary: libmemgym_hip.so loads without a GPU and exports every function include/memgym.h
declares (no compute calls are made here); the Python mirror keeps the reference's option keys and assertion text."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "endless-memory-gym_amd", "lib", "libmemgym_hip.so")


def declared_symbols():
    h = open(os.path.join(ROOT, "include", "memgym.h")).read()
    h = re.sub(r"/\*.*?\*/", "", h, flags=re.S)
    return sorted(set(re.findall(r"\b(mg_[a-z_0-9]+)\s*\(", h)))


def test_library_exports_every_declared_symbol():
    if not os.path.exists(LIB):
        import __graft_entry__
        __graft_entry__.build_hip()
    import torch  # noqa: F401  (same load order as the package: torch's HIP runtime first)
    lib = ctypes.CDLL(LIB)
    names = declared_symbols()
    assert len(names) >= 14
    for n in names:
        assert hasattr(lib, n), "libmemgym_hip.so does not export " + n


def test_unknown_env_id_is_rejected_without_touching_the_gpu():
    import torch  # noqa: F401
    lib = ctypes.CDLL(LIB)
    lib.mg_last_error.restype = ctypes.c_char_p
    h = ctypes.c_void_p()
    rc = lib.mg_create(b"NoSuchEnv-v0", 4, 0, ctypes.byref(h))
    assert rc != 0 and h.value is None


def test_reset_params_mirror_the_reference():
    import importlib.util
    spec = importlib.util.spec_from_file_location("rp", os.path.join(ROOT, "endless-memory-gym_amd", "memory_gym_amd", "reset_params.py"))
    rp = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(rp)
    assert set(rp.DEFAULTS) == {"MortarMayhem-Grid-v0", "MortarMayhem-v0", "Endless-MortarMayhem-v0", "MysteryPath-v0",
                                "Endless-MysteryPath-v0", "SearingSpotlights-v0", "Endless-SearingSpotlights-v0", "MysteryPath-Grid-v0",
                                "MortarMayhemB-Grid-v0", "MortarMayhemB-v0"}  # all ten ids of memory_gym/__init__.py:13-61
    with pytest.raises(AssertionError, match="20 commands are allowed at maximum"):
        rp.process_reset_params("MortarMayhemB-v0", {"command_count": [5, 21]})
    assert "command_show_duration" not in rp.DEFAULTS["MortarMayhemB-Grid-v0"]
    p = rp.process_reset_params("MortarMayhem-Grid-v0", {"arena_size": 6})
    assert p["arena_size"] == 6 and p["command_count"] == [10] and p["explosion_delay"] == [6]
    with pytest.raises(AssertionError, match=r"Provided reset parameter \(agent_speeed\) is not valid. Check spelling."):
        rp.process_reset_params("MortarMayhem-v0", {"agent_speeed": 1})
    with pytest.raises(AssertionError):
        rp.process_reset_params("MortarMayhem-Grid-v0", {"arena_size": 7})
    # calc_max_episode_steps with the reference's swapped arguments: 119 for MM-Grid, 275 for MM
    assert rp.calc_max_episode_steps(10, 3, 1, 6, 2) == 119 and rp.calc_max_episode_steps(10, 3, 1, 18, 6) == 275


def test_public_header_compiles_as_plain_c_and_cxx_without_hip(tmp_path):
    """include/memgym.h promises `void*` streams so that a trainer needs no HIP include (mg_store_probe had been declared
    with a hipStream_t): a C and a C++ translation unit that include nothing but the header must compile with the host compilers."""
    import shutil
    import subprocess

    inc = os.path.join(ROOT, "include")
    for cc, ext, flags in (("gcc", "c", ["-std=c99"]), ("g++", "cpp", ["-std=c++11"])):
        if not shutil.which(cc):
            pytest.skip(cc + " not installed")
        src = tmp_path / ("use_header." + ext)
        src.write_text('#include "memgym.h"\nint probe(void* b, void* s) { mg_single_io io; io.struct_size = sizeof io; (void)io; return mg_store_probe(b, 1, 0, s); }\n')
        p = subprocess.run([cc] + flags + ["-Wall", "-Wextra", "-Werror", "-pedantic", "-I", inc, "-c", str(src), "-o", str(tmp_path / ("o." + ext))], capture_output=True, text=True)
        assert p.returncode == 0, p.stderr
