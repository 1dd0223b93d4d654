// mg_api.hip -- the C ABI of include/memgym.h on top of the per-family implementations.
#include <algorithm>
#include <chrono>
#include <cstddef>
#include <map>
#include <mutex>

#include "mg_family.hpp"
#include "mg_lab.hpp"
using mg::lab_env;

namespace mg {
static thread_local std::string g_last_error;
void set_error(const std::string& msg) { g_last_error = msg; }
}  // namespace mg

// One handle = one device, num_envs instances, one Family object (state, atlases, queues); everything runs on the caller's stream.
// (Rounds 3-5 could split a handle into "instance groups" on streams of their own -- mg_set_groups; measured x0.8 on ROCm 7.2,
// profiles/r03_groups.md -- removed in round 6: dead weight in the ABI, the checkpoint header and this file.)
struct mg_env {
    mg::Family* fam = nullptr;
    int device = 0;
    int num_envs = 0;
    int variant = 0, family = 0;  // what make_* was called with
    bool started = false;         // a reset has happened
    std::string id;
    int obs_format = MG_OBS_U8_XYC;
    float* vec_dev = nullptr;     // the caller's vector-observation binding (mg_bind_vector_obs), or the single-instance block's
    float* vec_dev_caller = nullptr;
    int prof_stride = 0;
    // mg_single_open: pinned, device-mapped host buffers of the single-instance fast path (host address, device address)
    struct Single {
        bool open = false;
        char *host = nullptr, *dev = nullptr;
        size_t bytes = 0;
        size_t o_action = 0, o_seed = 0, o_obs = 0, o_vec = 0, o_reward32 = 0, o_reward = 0, o_done = 0, o_gt32 = 0, o_gt = 0, o_ep_reward = 0,
               o_ep_length = 0, o_aux = 0, o_flag = 0;
        uint32_t ticket = 0;      // completion flag protocol of mg_single_step (see there)
    } single;
};

namespace {
// mg_info_buffers as the caller's header laid it out (include/memgym.h: struct_size)
constexpr size_t INFO_SIZE_MIN = offsetof(mg_info_buffers, final_obs_dev);  // the episode-record pointers every layout has
mg_info_buffers read_info(const mg_info_buffers* info) {
    mg_info_buffers ib;
    memset(&ib, 0, sizeof(ib));
    if (!info) return ib;
    const size_t sz = info->struct_size;
    // A caller built against the header of round 2 (no struct_size member) has a device POINTER in this place: 8-aligned and huge.
    // Nothing larger than this build's struct plus room for a few future members is a layout of include/memgym.h.
    constexpr size_t INFO_SIZE_MAX = sizeof(mg_info_buffers) + 16 * sizeof(void*);
    if (sz < INFO_SIZE_MIN || sz > INFO_SIZE_MAX || sz % sizeof(void*) != 0)
        throw std::runtime_error("mg_step: mg_info_buffers.struct_size = " + std::to_string(sz) + " is not a layout of include/memgym.h (" +
                                 std::to_string(sizeof(mg_info_buffers)) + " in this build); set it to sizeof(mg_info_buffers)");
    memcpy(&ib, info, sz < sizeof(ib) ? sz : sizeof(ib));  // a shorter (older) struct: the fields it lacks stay NULL
    ib.struct_size = sizeof(ib);
    return ib;
}

struct StateHeader {
    char magic[8];
    uint32_t version, num_envs;
    uint64_t payload, id_hash;
    uint8_t pad[32];
};
static_assert(sizeof(StateHeader) == 64, "state header is 64 bytes");
uint64_t fnv1a(const std::string& s) {
    uint64_t h = 1469598103934665603ull;
    for (unsigned char c : s) h = (h ^ c) * 1099511628211ull;
    return h;
}

struct DeviceGuard {
    int prev = -1;
    explicit DeviceGuard(int device) {
        MG_HIP(hipGetDevice(&prev));
        if (prev != device) MG_HIP(hipSetDevice(device));
        else prev = -1;
    }
    ~DeviceGuard() {
        if (prev >= 0) (void)hipSetDevice(prev);
    }
};

template <typename F>
int guarded(mg_env* env, F&& f) {
    try {
        if (!env || !env->fam) {
            mg::set_error("null handle");
            return -1;
        }
        DeviceGuard guard(env->device);  // the caller's current device is left as it was
        f();
        return 0;
    } catch (const mg::OptionError& e) {
        mg::set_error(e.msg);
        return e.code;
    } catch (const std::exception& e) {
        mg::set_error(e.what());
        return -1;
    }
}
}  // namespace

namespace {
mg::Family* make_family(int family, int variant, int n) {
    if (family == 0) return mg::make_mortar(variant, n);
    if (family == 1) return mg::make_spot(variant, n);
    return mg::make_mystery(variant, n);
}
void destroy_family(mg_env* e) {
    if (e->single.host) {  // (the family holds device views of it: mg_destroy synchronises first)
        if (e->vec_dev && e->vec_dev == (float*)(e->single.dev + e->single.o_vec)) e->vec_dev = e->vec_dev_caller;  // never leave a pointer into freed memory
        (void)hipHostFree(e->single.host);
        e->single = mg_env::Single();
    }
    delete e->fam;
    e->fam = nullptr;
}
size_t obs_bytes_of(int f) {
    const size_t elem = f == MG_OBS_F32_CYX ? 4 : ((f == MG_OBS_F16_CYX || f == MG_OBS_BF16_CYX) ? 2 : 1);
    return elem * 84 * 84 * 3;
}
}  // namespace

extern "C" {

const char* mg_last_error(void) { return mg::g_last_error.c_str(); }

int mg_create(const char* env_id, int32_t num_envs, int device, mg_env** out) {
    try {
        if (!env_id || !out || num_envs < 1) {
            mg::set_error("mg_create: bad arguments");
            return -1;
        }
        DeviceGuard guard(device);
        std::string id(env_id);
        int family = -1, variant = 0;
        if (id == "MortarMayhem-Grid-v0") { family = 0; variant = 0; }
        else if (id == "MortarMayhem-v0") { family = 0; variant = 1; }
        else if (id == "Endless-MortarMayhem-v0") { family = 0; variant = 2; }
        else if (id == "MortarMayhemB-Grid-v0") { family = 0; variant = 3; }
        else if (id == "MortarMayhemB-v0") { family = 0; variant = 4; }
        else if (id == "Endless-SearingSpotlights-v0") { family = 1; variant = 1; }
        else if (id == "SearingSpotlights-v0") { family = 1; variant = 0; }
        else if (id == "MysteryPath-v0") { family = 2; variant = 0; }
        else if (id == "Endless-MysteryPath-v0") { family = 2; variant = 1; }
        else if (id == "MysteryPath-Grid-v0") { family = 2; variant = 2; }
        else {
            mg::set_error("mg_create: environment id not available in this build: " + id);
            return -5;
        }
        mg_env* e = new mg_env();
        e->device = device;
        e->num_envs = num_envs;
        e->id = id;
        e->family = family;
        e->variant = variant;
        try {
            e->fam = make_family(family, variant, num_envs);
        } catch (...) {
            delete e;
            throw;
        }
        *out = e;
        return 0;
    } catch (const std::exception& e) {
        mg::set_error(e.what());
        return -1;
    }
}

void mg_destroy(mg_env* env) {
    if (!env) return;
    int prev = -1;
    (void)hipGetDevice(&prev);
    (void)hipSetDevice(env->device);
    (void)hipDeviceSynchronize();
    destroy_family(env);
    delete env;
    if (prev >= 0) (void)hipSetDevice(prev);
}

int32_t mg_num_envs(const mg_env* env) { return env ? env->num_envs : 0; }
int32_t mg_action_dim(const mg_env* env) { return env ? env->fam->action_dim() : 0; }
int32_t mg_gt_dim(const mg_env* env) { return env ? env->fam->gt_dim() : 0; }
int32_t mg_vec_dim(const mg_env* env) { return env ? env->fam->vec_dim() : 0; }
int mg_bind_vector_obs(mg_env* env, float* vec_dev) {
    return guarded(env, [&] {
        if (env->fam->vec_dim() == 0 && vec_dev) throw std::runtime_error("mg_bind_vector_obs: this env id has no vector observation");
        env->vec_dev = env->vec_dev_caller = vec_dev;
        env->fam->bind_vector_obs(vec_dev);
    });
}
const char* mg_info_name(const mg_env* env, int k) { return env ? env->fam->info_name(k) : nullptr; }

int mg_set_option(mg_env* env, const char* key, const double* values, int n) {
    return guarded(env, [&] {
        if (!key || !values || n < 1) throw mg::OptionError{-3, "mg_set_option: bad arguments"};
        env->fam->set_option(key, values, n);
    });
}

int mg_set_option_set(mg_env* env, int set_id, const char* key, const double* values, int n) {
    return guarded(env, [&] {
        if (!key || !values || n < 1) throw mg::OptionError{-3, "mg_set_option_set: bad arguments"};
        if (set_id < 0 || set_id >= MG_MAX_OPTION_SETS) throw mg::OptionError{-3, "mg_set_option_set: set index out of range"};
        env->fam->set_option_set(set_id, key, values, n);
    });
}

int mg_bind_option_sets(mg_env* env, const int32_t* set_of_dev) {
    return guarded(env, [&] {
        env->fam->bind_option_sets(set_of_dev);
    });
}

int mg_set_capacity(mg_env* env, const char* what, int64_t value) {
    return guarded(env, [&] {
        if (!what) throw mg::OptionError{-3, "mg_set_capacity: NULL"};
        if (env->started) throw std::runtime_error("mg_set_capacity: before the first mg_reset / mg_set_state");
        env->fam->set_capacity(what, value);
    });
}
int64_t mg_capacity(mg_env* env, const char* what) {
    int64_t v = -1;
    const int rc = guarded(env, [&] {
        if (!what) throw mg::OptionError{-3, "mg_capacity: NULL"};
        v = env->fam->capacity(what);
    });
    return rc == 0 ? v : -1;
}

int mg_set_obs_format(mg_env* env, int format) {
    return guarded(env, [&] {
        if (format != MG_OBS_U8_XYC && format != MG_OBS_F32_CYX && format != MG_OBS_F16_CYX && format != MG_OBS_BF16_CYX)
            throw mg::OptionError{-3, "mg_set_obs_format: unknown format"};
        env->obs_format = env->fam->obs_format = format;
    });
}

size_t mg_obs_bytes(const mg_env* env) {
    if (!env) return 0;
    return obs_bytes_of(env->obs_format);
}

int mg_reset(mg_env* env, const int64_t* seeds_dev, const uint8_t* mask_dev, void* obs_dev, float* gt_dev, void* stream) {
    return guarded(env, [&] {
        if (!obs_dev) throw std::runtime_error("mg_reset: obs_dev is NULL");
        env->fam->reset(seeds_dev, mask_dev, obs_dev, gt_dev, (hipStream_t)stream);
        env->started = true;
    });
}

int mg_render(mg_env* env, void* obs_dev, void* stream) {
    return guarded(env, [&] {
        if (!obs_dev) throw std::runtime_error("mg_render: obs_dev is NULL");
        env->fam->raster_only(obs_dev, nullptr, (hipStream_t)stream);
    });
}

// Rows of the observation buffer whose flag is set, copied to the same rows of another buffer (mg_step with mg_info_buffers.final_obs_dev:
// after a step without auto-reset the rows of the finished instances ARE the terminal observations).  A workgroup looks at 16 flags with one
// load and copies the rows a ballot names, six 16-byte vectors per lane in flight (a uint8 row is 1,323 vectors: one round); rows are whole
// vectors in every output format.
typedef uint32_t row_vec16 __attribute__((ext_vector_type(4)));
constexpr int COPY_CHUNK = 16;
__global__ __launch_bounds__(256) void copy_rows_kernel(const uint8_t* __restrict__ flags, const row_vec16* __restrict__ src, row_vec16* __restrict__ dst,
                                                      int n, int vec_per_row) {
    const int tid = threadIdx.x;
    for (int base = blockIdx.x * COPY_CHUNK; base < n; base += gridDim.x * COPY_CHUNK) {
        const int e = base + (tid & (COPY_CHUNK - 1));
        uint32_t m = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)__ballot(e < n && flags[e] != 0)) & ((1u << COPY_CHUNK) - 1u);
        while (m) {
            const size_t row = (size_t)(base + __builtin_ctz(m)) * (size_t)vec_per_row;
            m &= m - 1;
            for (int q0 = 0; q0 < vec_per_row; q0 += 6 * 256) {
                row_vec16 v[6];
#pragma unroll
                for (int k = 0; k < 6; ++k) {
                    const int q = q0 + tid + 256 * k;
                    if (q < vec_per_row) v[k] = src[row + q];
                }
#pragma unroll
                for (int k = 0; k < 6; ++k) {
                    const int q = q0 + tid + 256 * k;
                    if (q < vec_per_row) dst[row + q] = v[k];
                }
            }
        }
    }
}

namespace {
// pygame.transform.scale(surface, (336, 336)) of an 84x84 surface = every pixel four times per axis (transform.c stretch()),
// then fliplr(rot90(array3d, 3)) = image order: out[n][y][x][c] = frame[n][x / 4][y / 4][c]
__global__ __launch_bounds__(256) void debug_stretch_kernel(const uint8_t* __restrict__ frames, uint8_t* __restrict__ out, int n) {
    const size_t total = (size_t)n * 336 * 336;
    for (size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x; p < total; p += (size_t)gridDim.x * blockDim.x) {
        const int x = (int)(p % 336), y = (int)((p / 336) % 336);
        const size_t e = p / (336 * 336);
        const uint8_t* src = frames + e * MG_OBS_BYTES + ((size_t)(x >> 2) * 84 + (y >> 2)) * 3;
        uint8_t* dst = out + p * 3;
        dst[0] = src[0];
        dst[1] = src[1];
        dst[2] = src[2];
    }
}
}  // namespace

int mg_render_debug(mg_env* env, uint8_t* rgb_dev, void* stream) {
    return guarded(env, [&] {
        if (!rgb_dev) throw std::runtime_error("mg_render_debug: rgb_dev is NULL");
        hipStream_t st = (hipStream_t)stream;
        uint8_t* frames = nullptr;
        MG_HIP(hipMalloc((void**)&frames, (size_t)env->num_envs * MG_OBS_BYTES));
        try {
            env->fam->sync_state();
            env->fam->raster_debug(frames, st);
            const size_t total = (size_t)env->num_envs * 336 * 336;
            const unsigned grid = (unsigned)std::min<size_t>((total + 255) / 256, 65536);
            hipLaunchKernelGGL(debug_stretch_kernel, dim3(grid), dim3(256), 0, st, frames, rgb_dev, env->num_envs);
            MG_HIP(hipGetLastError());
            MG_HIP(hipStreamSynchronize(st));
        } catch (...) {
            (void)hipFree(frames);
            throw;
        }
        MG_HIP(hipFree(frames));
    });
}

int mg_step(mg_env* env, const int32_t* actions_dev, void* obs_dev, float* reward_dev, uint8_t* done_dev, float* gt_dev,
            const mg_info_buffers* info, int autoreset, void* stream) {
    return guarded(env, [&] {
        if (!actions_dev || !obs_dev || !reward_dev || !done_dev) throw std::runtime_error("mg_step: NULL buffer");
        hipStream_t st = (hipStream_t)stream;
        const mg_info_buffers ib = read_info(info);
        mg::Family* f = env->fam;
        if (autoreset && ib.final_obs_dev && !(mg::sparse_masked_raster() && f->keeps_final_obs(st))) {
            // terminal frames wanted: step without auto-reset (obs rows of finished instances = terminal frames), keep
            // a copy of exactly those rows, then reset the finished instances with seed=None -- the same RNG
            // consumption and frames as the fused path (tests/test_gpu_vector_api.py)
            f->step(actions_dev, obs_dev, reward_dev, done_dev, gt_dev, &ib, 0, st);
            if (mg::sparse_masked_raster()) {  // the rows are there: copied, not drawn again (round 6)
                const int n = env->num_envs, vec_per_row = (int)(mg_obs_bytes(env) / 16);
                hipLaunchKernelGGL(copy_rows_kernel, dim3(std::min((n + COPY_CHUNK - 1) / COPY_CHUNK, 16384)), dim3(256), 0, st, done_dev, (const row_vec16*)obs_dev,
                                   (row_vec16*)ib.final_obs_dev, n, vec_per_row);
                MG_HIP(hipGetLastError());
            } else f->raster_only(ib.final_obs_dev, done_dev, st);
            f->reset(nullptr, done_dev, obs_dev, gt_dev, st);
        } else {
            f->step(actions_dev, obs_dev, reward_dev, done_dev, gt_dev, &ib, autoreset, st);
        }
        if (ib.gt64_dev) f->ground_truth64(ib.gt64_dev, st);
    });
}

int mg_ground_truth64(mg_env* env, double* gt64_dev, void* stream) {
    return guarded(env, [&] {
        if (!env->fam->gt_dim()) return;
        if (!gt64_dev) throw std::runtime_error("mg_ground_truth64: gt64_dev is NULL");
        env->fam->sync_state();
        env->fam->ground_truth64(gt64_dev, (hipStream_t)stream);
    });
}

// ---- the single-instance fast path (include/memgym.h: mg_single_io) ----
} // extern "C"
namespace {
// Wait for everything enqueued on `st` WITHOUT hipStreamSynchronize (round 6): a stream memory operation writes the
// call's ticket into the pinned block behind the step's launches (the command processor executes it when they have completed -- and
// with them their stores into the same pinned block), and the host polls that word.  hipStreamSynchronize on this runtime costs
// more than the step's kernels once the work is this small.  A wait that takes longer than 50 ms (a fault, a debugger) falls back to the
// synchronising call, which also reports the stream's error; a runtime without stream memory operations synchronises as before.
void single_wait(mg_env::Single& S, hipStream_t st, bool armed = false) {
    static bool use_flag = true;
    if (use_flag) {
        // armed: the step's own last kernel stores the ticket (Family::arm_done_flag; S.ticket was advanced when it was armed)
        const uint32_t want = armed ? S.ticket : ++S.ticket;
        if (armed || hipStreamWriteValue32(st, S.dev + S.o_flag, want, 0) == hipSuccess) {
            volatile uint32_t* flag = (volatile uint32_t*)(S.host + S.o_flag);
            const auto t0 = std::chrono::steady_clock::now();
            for (uint32_t spins = 0;; ++spins) {
                if (__atomic_load_n(flag, __ATOMIC_ACQUIRE) == want) return;
                __builtin_ia32_pause();
                if ((spins & 1023u) == 1023u && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(50)) break;
            }
        } else {
            (void)hipGetLastError();
            use_flag = false;
        }
    }
    MG_HIP(hipStreamSynchronize(st));
}
}  // namespace
extern "C" {
int mg_single_open(mg_env* env, mg_single_io* io) {
    return guarded(env, [&] {
        if (!io || io->struct_size != sizeof(mg_single_io)) throw std::runtime_error("mg_single_open: set io->struct_size = sizeof(mg_single_io)");
        if (env->num_envs != 1) throw std::runtime_error("mg_single_open: the handle must hold exactly one instance");
        mg_env::Single& S = env->single;
        if (!S.open) {
            size_t o = 0;
            auto take = [&](size_t bytes) { const size_t at = o; o = (o + bytes + 255) & ~(size_t)255; return at; };
            S.o_action = take(8); S.o_seed = take(8);
            S.o_obs = take(84 * 84 * 3 * 4);  // room for every observation format
            S.o_vec = take(sizeof(float) * 256);
            S.o_reward32 = take(4); S.o_reward = take(8); S.o_done = take(1);
            S.o_gt32 = take(sizeof(float) * 8); S.o_gt = take(sizeof(double) * 8);
            S.o_ep_reward = take(8); S.o_ep_length = take(4); S.o_aux = take(sizeof(float) * MG_INFO_SLOTS * 64);  // (one 256-byte line per slot)
            S.o_flag = take(4);
            S.bytes = o;
            MG_HIP(hipHostMalloc((void**)&S.host, S.bytes, hipHostMallocMapped | hipHostMallocCoherent));
            memset(S.host, 0, S.bytes);
            MG_HIP(hipHostGetDevicePointer((void**)&S.dev, S.host, 0));
            S.open = true;
        }
        if (env->fam->vec_dim()) {  // the vector observation is written at every reset, into the mapped buffer from now on (io->vec;
            env->vec_dev = (float*)(S.dev + S.o_vec);  // a buffer bound earlier with mg_bind_vector_obs is no longer written)
            env->fam->bind_vector_obs(env->vec_dev);
        }
        io->obs = S.host + S.o_obs;
        io->vec = env->fam->vec_dim() ? (float*)(S.host + S.o_vec) : nullptr;
        io->reward = (double*)(S.host + S.o_reward);
        io->done = (uint8_t*)(S.host + S.o_done);
        io->gt = (double*)(S.host + S.o_gt);
        io->ep_reward = (double*)(S.host + S.o_ep_reward);
        io->ep_length = (int32_t*)(S.host + S.o_ep_length);
        for (int k = 0; k < MG_INFO_SLOTS; ++k) io->aux[k] = (float*)(S.host + S.o_aux + 256 * k);
    });
}

int mg_single_reset(mg_env* env, int64_t seed, int has_seed, void* stream) {
    return guarded(env, [&] {
        mg_env::Single& S = env->single;
        if (!S.open) throw std::runtime_error("mg_single_reset: mg_single_open first");
        hipStream_t st = (hipStream_t)stream;
        *(int64_t*)(S.host + S.o_seed) = seed;
        mg::Family* f = env->fam;
        f->reset(has_seed ? (const int64_t*)(S.dev + S.o_seed) : nullptr, nullptr, S.dev + S.o_obs, f->gt_dim() ? (float*)(S.dev + S.o_gt32) : nullptr, st);
        f->ground_truth64((double*)(S.dev + S.o_gt), st);
        env->started = true;
        single_wait(S, st);
    });
}

int mg_single_step(mg_env* env, int32_t a0, int32_t a1, void* stream) {
    int flags = 0;
    const int rc = guarded(env, [&] {
        mg_env::Single& S = env->single;
        if (!S.open) throw std::runtime_error("mg_single_step: mg_single_open first");
        hipStream_t st = (hipStream_t)stream;
        int32_t* act = (int32_t*)(S.host + S.o_action);
        act[0] = a0;
        act[1] = a1;
        mg_info_buffers ib;
        memset(&ib, 0, sizeof(ib));
        ib.struct_size = sizeof(ib);
        ib.ep_reward_dev = (double*)(S.dev + S.o_ep_reward);
        ib.ep_length_dev = (int32_t*)(S.dev + S.o_ep_length);
        for (int k = 0; k < MG_INFO_SLOTS; ++k) ib.aux_dev[k] = (float*)(S.dev + S.o_aux + 256 * k);
        ib.reward64_dev = (double*)(S.dev + S.o_reward);
        mg::Family* f = env->fam;
        const bool armed = f->arm_done_flag((uint32_t*)(S.dev + S.o_flag), S.ticket + 1);
        if (armed) ++S.ticket;
        f->step((const int32_t*)(S.dev + S.o_action), S.dev + S.o_obs, (float*)(S.dev + S.o_reward32), (uint8_t*)(S.dev + S.o_done),
                f->gt_dim() ? (float*)(S.dev + S.o_gt32) : nullptr, &ib, 0, st);
        f->ground_truth64((double*)(S.dev + S.o_gt), st);
        single_wait(S, st, armed);
        flags = f->peek_errors();  // (the word lives in pinned host memory: a plain read; saves the caller a second native call per step)
    });
    return rc != 0 ? rc : flags;
}

size_t mg_state_size(const mg_env* env) {
    if (!env) return 0;
    size_t t = sizeof(StateHeader);
    for (auto& b : env->fam->state_blobs()) t += b.second;
    return t;
}

int mg_get_state(mg_env* env, void* host_buf, size_t size) {
    return guarded(env, [&] {
        if (!host_buf || size < mg_state_size(env)) throw std::runtime_error("mg_get_state: buffer too small");
        MG_HIP(hipDeviceSynchronize());
        env->fam->sync_state();
        StateHeader h;
        memset(&h, 0, sizeof(h));
        memcpy(h.magic, "MGSTATE1", 8);
        h.version = MG_STATE_VERSION;
        h.num_envs = (uint32_t)env->num_envs;
        h.payload = mg_state_size(env) - sizeof(StateHeader);
        h.id_hash = fnv1a(env->id);
        memcpy(host_buf, &h, sizeof(h));
        char* p = (char*)host_buf + sizeof(StateHeader);
        for (auto& b : env->fam->state_blobs()) {
            MG_HIP(hipMemcpy(p, b.first, b.second, hipMemcpyDeviceToHost));
            p += b.second;
        }
    });
}

int mg_set_state(mg_env* env, const void* host_buf, size_t size) {
    return guarded(env, [&] {
        if (!host_buf || size < sizeof(StateHeader)) throw std::runtime_error("mg_set_state: buffer too small for a state header");
        StateHeader h;
        memcpy(&h, host_buf, sizeof(h));
        if (memcmp(h.magic, "MGSTATE1", 8) != 0) throw std::runtime_error("mg_set_state: not a memgym state blob (bad magic)");
        if (h.version != MG_STATE_VERSION)
            throw std::runtime_error("mg_set_state: state version " + std::to_string(h.version) + ", this library reads version " +
                                     std::to_string(MG_STATE_VERSION));
        if (h.id_hash != fnv1a(env->id)) throw std::runtime_error("mg_set_state: the blob belongs to another env id than " + env->id);
        if (h.num_envs != (uint32_t)env->num_envs)
            throw std::runtime_error("mg_set_state: the blob holds " + std::to_string(h.num_envs) + " instances, the handle " +
                                     std::to_string(env->num_envs));
        if (h.payload != mg_state_size(env) - sizeof(StateHeader) || size < mg_state_size(env))
            throw std::runtime_error("mg_set_state: payload size differs from this handle's state");
        MG_HIP(hipDeviceSynchronize());
        const char* p = (const char*)host_buf + sizeof(StateHeader);
        for (auto& b : env->fam->state_blobs()) {
            MG_HIP(hipMemcpy(b.first, p, b.second, hipMemcpyHostToDevice));
            p += b.second;
        }
        env->fam->on_state_loaded();  // reset(seed=None) / auto-reset are legal on a restored handle
        env->started = true;
    });
}

int mg_set_profiling(mg_env* env, int on) {
    return guarded(env, [&] {
        env->prof_stride = on < 0 ? 0 : on;
        env->fam->prof.stride = env->prof_stride;
        env->fam->prof.count[0] = env->fam->prof.count[1] = 0;
    });
}

int mg_get_profile(mg_env* env, int kind, double* total_ms, int64_t* launches) {
    return guarded(env, [&] {
        if (kind < 0 || kind > 1 || !total_ms || !launches) throw std::runtime_error("mg_get_profile: bad arguments");
        env->fam->prof.collect(kind, total_ms, launches);
    });
}

int mg_poll_errors(mg_env* env, int* flags) {
    return guarded(env, [&] {
        if (!flags) throw std::runtime_error("mg_poll_errors: NULL");
        MG_HIP(hipDeviceSynchronize());
        *flags = env->fam->poll_errors();
    });
}

int mg_peek_errors(mg_env* env, int* flags) {
    return guarded(env, [&] {
        if (!flags) throw std::runtime_error("mg_peek_errors: NULL");
        *flags = env->fam->peek_errors();
    });
}

int mg_enable_peer_access(int device, int peer_device) {
    try {
        int prev = 0;
        MG_HIP(hipGetDevice(&prev));
        struct Restore {
            int d;
            ~Restore() { (void)hipSetDevice(d); }
        } restore{prev};
        if (device == peer_device) return 0;
        int can = 0;
        MG_HIP(hipDeviceCanAccessPeer(&can, device, peer_device));
        if (!can) {
            mg::set_error("device " + std::to_string(device) + " has no peer access to device " + std::to_string(peer_device));
            return -1;
        }
        MG_HIP(hipSetDevice(device));
        hipError_t e = hipDeviceEnablePeerAccess(peer_device, 0);
        if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) {
            (void)hipGetLastError();
            mg::set_error(std::string("hipDeviceEnablePeerAccess: ") + hipGetErrorString(e));
            return -1;
        }
        (void)hipGetLastError();
        return 0;
    } catch (const std::exception& e) {
        mg::set_error(e.what());
        return -1;
    }
}

int mg_debug_counter(mg_env* env, const char* name, int64_t* value) {
    return guarded(env, [&] {
        if (!name || !value) throw std::runtime_error("mg_debug_counter: NULL");
        MG_HIP(hipDeviceSynchronize());
        if (!env->fam->debug_counter(name, value)) throw std::runtime_error(std::string("mg_debug_counter: no counter named ") + name + " for " + env->id);
    });
}

int mg_debug_rng(mg_env* env, int32_t i, uint64_t* out) {
    return guarded(env, [&] {
        if (i < 0 || i >= env->num_envs) throw std::runtime_error("mg_debug_rng: index out of range");
        MG_HIP(hipDeviceSynchronize());
        env->fam->sync_state();
        env->fam->debug_rng(i, out);
    });
}

}  // extern "C"
