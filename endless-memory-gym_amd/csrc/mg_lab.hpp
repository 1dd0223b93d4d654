// mg_lab.hpp -- measurement and test switches.
//
// The shipped library (lib/libmemgym_hip.so) reads NO tuning or test switch from the environment: lab_env() is a constant
// there and every branch behind it folds away.  The same sources built with -DMG_LAB (lib/lab/libmemgym_hip_lab.so,
// __graft_entry__.build_lab()) honour the MEMGYM_* switches named at their call sites; tools/ and the few tests that need a
// hook (tests/test_gpu_switches.py, test_gpu_error_bits.py, test_gpu_one_launch.py) load that build through MEMGYM_HIP_LIB.
// What a USER can set stays outside the library altogether: MEMGYM_OBS_PLACEMENT / MEMGYM_OBS_SEARCH_GB / MEMGYM_OBS_SEARCH_MS are
// read by the Python mirror (vec_env.py) and travel as arguments (mg_obs_alloc's budget, mg_obs_set_search_ms).
#pragma once
#include <stdlib.h>

namespace mg {
#ifdef MG_LAB
inline const char* lab_env(const char* name) { return getenv(name); }
constexpr bool LAB_BUILD = true;
#else
inline const char* lab_env(const char*) { return nullptr; }
constexpr bool LAB_BUILD = false;
#endif
inline int lab_int(const char* name, int dflt) {
    const char* e = lab_env(name);
    return e ? atoi(e) : dflt;
}
// MEMGYM_SPARSE_RASTER=0 (lab build): masked resets draw their frames with the dense persistent launch of rounds 1-5, and a
// gymnasium-convention step (mg_info_buffers.final_obs_dev) re-draws the terminal frames instead of copying them (A/B, bit-exactness tests)
inline bool sparse_masked_raster() {
    static const bool on = lab_int("MEMGYM_SPARSE_RASTER", 1) != 0;
    return on;
}
}  // namespace mg
