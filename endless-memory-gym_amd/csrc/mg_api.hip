// mg_api.hip -- the C ABI of include/memgym.h on top of the per-family implementations.
#include <algorithm>
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

constexpr int MG_MAX_GROUPS = 8;

// One handle = one device, num_envs instances in `groups` contiguous blocks (include/memgym.h: mg_set_groups), each block a
// Family object of its own (state, atlases, queues).  groups == 1: everything runs on the caller's stream as it always did.
struct mg_env {
    std::vector<mg::Family*> fams;
    std::vector<int> base;  // first instance of block g; base[groups] = num_envs
    mg::Family* fam = nullptr;  // fams[0]: static properties
    int device = 0;
    int num_envs = 0;
    int variant = 0, family = 0;  // what make_* was called with
    bool started = false;         // a reset has happened: the grouping is fixed
    std::string id;
    struct Opt {
        int set;
        std::string key;
        std::vector<double> values;
    };
    std::vector<Opt> options;  // every mg_set_option / mg_set_option_set so far (replayed by mg_set_groups)
    const int32_t* set_of_dev = nullptr;
    int obs_format = MG_OBS_U8_XYC;
    float* vec_dev = nullptr;
    int prof_stride = 0;
    // mg_single_open: pinned, device-mapped host buffers of the single-instance fast path (host address, device address)
    struct Single {
        bool open = false;
        char *host = nullptr, *dev = nullptr;
        size_t bytes = 0;
        size_t o_action = 0, o_seed = 0, o_obs = 0, o_vec = 0, o_reward32 = 0, o_reward = 0, o_done = 0, o_gt32 = 0, o_gt = 0, o_ep_reward = 0,
               o_ep_length = 0, o_aux = 0;
    } single;
    hipStream_t gs[MG_MAX_GROUPS] = {};
    hipEvent_t ev_in = nullptr, ev_logic[MG_MAX_GROUPS] = {}, ev_done[MG_MAX_GROUPS] = {};
    int groups() const { return (int)fams.size(); }
    int count(int g) const { return base[g + 1] - base[g]; }
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
void destroy_families(mg_env* e) {
    if (e->single.host) {  // (the families hold device views of it: mg_destroy synchronises first)
        (void)hipHostFree(e->single.host);
        e->single = mg_env::Single();
    }
    for (auto* f : e->fams) delete f;
    e->fams.clear();
    e->fam = nullptr;
    for (int g = 0; g < MG_MAX_GROUPS; ++g) {
        if (e->gs[g]) (void)hipStreamDestroy(e->gs[g]);
        if (e->ev_logic[g]) (void)hipEventDestroy(e->ev_logic[g]);
        if (e->ev_done[g]) (void)hipEventDestroy(e->ev_done[g]);
        e->gs[g] = nullptr;
        e->ev_logic[g] = e->ev_done[g] = nullptr;
    }
    if (e->ev_in) (void)hipEventDestroy(e->ev_in);
    e->ev_in = nullptr;
}
// (re)build the blocks of a handle; options, observation format, vector-observation binding and profiling are carried over
void build_groups(mg_env* e, int groups) {
    destroy_families(e);
    e->base.assign(groups + 1, 0);
    for (int g = 0; g <= groups; ++g) e->base[g] = (int)((int64_t)e->num_envs * g / groups);
    for (int g = 0; g < groups; ++g) {
        mg::Family* f = make_family(e->family, e->variant, e->count(g));
        e->fams.push_back(f);
        f->obs_format = e->obs_format;
        f->prof.stride = e->prof_stride;
        for (auto& o : e->options) f->set_option_set(o.set, o.key, o.values.data(), (int)o.values.size());
        if (e->set_of_dev) f->bind_option_sets(e->set_of_dev + e->base[g]);
        if (e->vec_dev && f->vec_dim()) f->bind_vector_obs(e->vec_dev + (size_t)e->base[g] * f->vec_dim());
    }
    e->fam = e->fams[0];
    if (groups > 1) {
        MG_HIP(hipEventCreateWithFlags(&e->ev_in, hipEventDisableTiming));
        for (int g = 0; g < groups; ++g) {
            MG_HIP(hipStreamCreateWithFlags(&e->gs[g], hipStreamNonBlocking));
            MG_HIP(hipEventCreateWithFlags(&e->ev_logic[g], hipEventDisableTiming));
            MG_HIP(hipEventCreateWithFlags(&e->ev_done[g], hipEventDisableTiming));
            e->fams[g]->logic_event = e->ev_logic[g];
        }
    }
}
size_t obs_bytes_of(int f) {
    const size_t elem = f == MG_OBS_F32_CYX ? 4 : ((f == MG_OBS_F16_CYX || f == MG_OBS_BF16_CYX) ? 2 : 1);
    return elem * 84 * 84 * 3;
}
// Run `body(g, stream)` for every block: on the caller's stream for one block; otherwise on the blocks' own streams, which
// first wait for what the caller's stream has enqueued so far, and the caller's stream waits for all of them afterwards.
// stagger: block g's stream additionally waits for block g - 1's logic kernel (Family::logic_event).
template <typename F>
void for_groups(mg_env* e, hipStream_t s, bool stagger, F&& body) {
    const int G = e->groups();
    if (G == 1) {
        body(0, s);
        return;
    }
    // measurement switches (tools/groups_probe.py): MEMGYM_GROUPS_LAB bit 0 = no fork from / join into the caller's stream
    // (results are then NOT ordered with it), bit 1 = no stagger
    static const int lab = [] {
        const char* v = lab_env("MEMGYM_GROUPS_LAB");
        return v ? atoi(v) : 0;
    }();
    if (!(lab & 1)) MG_HIP(hipEventRecord(e->ev_in, s));
    for (int g = 0; g < G; ++g) {
        if (!(lab & 1)) MG_HIP(hipStreamWaitEvent(e->gs[g], e->ev_in, 0));
        if (stagger && g > 0 && !(lab & 2)) MG_HIP(hipStreamWaitEvent(e->gs[g], e->ev_logic[g - 1], 0));
        body(g, e->gs[g]);
        if (!(lab & 1)) MG_HIP(hipEventRecord(e->ev_done[g], e->gs[g]));
    }
    if (!(lab & 1))
        for (int g = 0; g < G; ++g) MG_HIP(hipStreamWaitEvent(s, e->ev_done[g], 0));
}
template <typename T>
T* off(T* p, size_t n) { return p ? p + n : nullptr; }
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
            build_groups(e, 1);
        } catch (...) {
            destroy_families(e);
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
    (void)hipDeviceSynchronize();  // the blocks' streams may still run
    destroy_families(env);
    delete env;
    if (prev >= 0) (void)hipSetDevice(prev);
}

int mg_set_groups(mg_env* env, int groups) {
    return guarded(env, [&] {
        if (groups != 1 && groups != 2 && groups != 4 && groups != 8) throw std::runtime_error("mg_set_groups: 1, 2, 4 or 8 groups");
        if (env->num_envs % groups != 0 || env->num_envs / groups < 1) throw std::runtime_error("mg_set_groups: num_envs must be divisible by the number of groups");
        if (env->started) throw std::runtime_error("mg_set_groups: the grouping is fixed by the first mg_reset");
        if (env->single.open) throw std::runtime_error("mg_set_groups: the handle is open for the single-instance fast path");
        if (groups != env->groups()) {
            const int before = env->groups();
            try {
                build_groups(env, groups);
            } catch (...) {  // e.g. out of memory for the second set of state arrays: the handle keeps working as it was
                try {
                    build_groups(env, before);
                } catch (...) {
                    destroy_families(env);
                }
                throw;
            }
        }
    });
}
int32_t mg_groups(const mg_env* env) { return env ? env->groups() : 0; }

int32_t mg_num_envs(const mg_env* env) { return env ? env->num_envs : 0; }
int32_t mg_action_dim(const mg_env* env) { return env ? env->fam->action_dim() : 0; }
int32_t mg_gt_dim(const mg_env* env) { return env ? env->fam->gt_dim() : 0; }
int32_t mg_vec_dim(const mg_env* env) { return env ? env->fam->vec_dim() : 0; }
int mg_bind_vector_obs(mg_env* env, float* vec_dev) {
    return guarded(env, [&] {
        if (env->fam->vec_dim() == 0 && vec_dev) throw std::runtime_error("mg_bind_vector_obs: this env id has no vector observation");
        env->vec_dev = vec_dev;
        for (int g = 0; g < env->groups(); ++g) env->fams[g]->bind_vector_obs(off(vec_dev, (size_t)env->base[g] * env->fam->vec_dim()));
    });
}
const char* mg_info_name(const mg_env* env, int k) { return env ? env->fam->info_name(k) : nullptr; }

int mg_set_option(mg_env* env, const char* key, const double* values, int n) {
    return guarded(env, [&] {
        if (!key || !values || n < 1) throw mg::OptionError{-3, "mg_set_option: bad arguments"};
        for (auto* f : env->fams) f->set_option(key, values, n);
        env->options.push_back({0, std::string(key), std::vector<double>(values, values + n)});
    });
}

int mg_set_option_set(mg_env* env, int set_id, const char* key, const double* values, int n) {
    return guarded(env, [&] {
        if (!key || !values || n < 1) throw mg::OptionError{-3, "mg_set_option_set: bad arguments"};
        if (set_id < 0 || set_id >= MG_MAX_OPTION_SETS) throw mg::OptionError{-3, "mg_set_option_set: set index out of range"};
        for (auto* f : env->fams) f->set_option_set(set_id, key, values, n);
        env->options.push_back({set_id, std::string(key), std::vector<double>(values, values + n)});
    });
}

int mg_bind_option_sets(mg_env* env, const int32_t* set_of_dev) {
    return guarded(env, [&] {
        for (int g = 0; g < env->groups(); ++g) env->fams[g]->bind_option_sets(off(set_of_dev, (size_t)env->base[g]));
        env->set_of_dev = set_of_dev;
    });
}

int mg_set_obs_format(mg_env* env, int format) {
    return guarded(env, [&] {
        if (format != MG_OBS_U8_XYC && format != MG_OBS_F32_CYX && format != MG_OBS_F16_CYX && format != MG_OBS_BF16_CYX)
            throw mg::OptionError{-3, "mg_set_obs_format: unknown format"};
        env->obs_format = format;
        for (auto* f : env->fams) f->obs_format = format;
    });
}

size_t mg_obs_bytes(const mg_env* env) {
    if (!env) return 0;
    return obs_bytes_of(env->obs_format);
}

int mg_reset(mg_env* env, const int64_t* seeds_dev, const uint8_t* mask_dev, void* obs_dev, float* gt_dev, void* stream) {
    return guarded(env, [&] {
        if (!obs_dev) throw std::runtime_error("mg_reset: obs_dev is NULL");
        const size_t ob = obs_bytes_of(env->obs_format);
        const int gd = env->fam->gt_dim();
        for_groups(env, (hipStream_t)stream, false, [&](int g, hipStream_t st) {
            const size_t b = (size_t)env->base[g];
            env->fams[g]->reset(off(seeds_dev, b), off(mask_dev, b), (char*)obs_dev + b * ob, off(gt_dev, b * gd), st);
        });
        env->started = true;
    });
}

int mg_render(mg_env* env, void* obs_dev, void* stream) {
    return guarded(env, [&] {
        if (!obs_dev) throw std::runtime_error("mg_render: obs_dev is NULL");
        const size_t ob = obs_bytes_of(env->obs_format);
        for_groups(env, (hipStream_t)stream, false, [&](int g, hipStream_t st) {
            env->fams[g]->raster_only((char*)obs_dev + (size_t)env->base[g] * ob, nullptr, st);
        });
    });
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
            for (int g = 0; g < env->groups(); ++g) {
                env->fams[g]->sync_state();
                env->fams[g]->raster_debug(frames + (size_t)env->base[g] * MG_OBS_BYTES, st);
            }
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
        const size_t ob = obs_bytes_of(env->obs_format);
        const int ad = env->fam->action_dim(), gd = env->fam->gt_dim();
        for_groups(env, st, true, [&](int g, hipStream_t gst) {
            const size_t b = (size_t)env->base[g];
            mg_info_buffers gi = ib;  // this block's rows of every array
            gi.ep_reward_dev = off(ib.ep_reward_dev, b);
            gi.ep_length_dev = off(ib.ep_length_dev, b);
            for (int k = 0; k < MG_INFO_SLOTS; ++k) gi.aux_dev[k] = off(ib.aux_dev[k], b);
            gi.final_obs_dev = ib.final_obs_dev ? (char*)ib.final_obs_dev + b * ob : nullptr;
            gi.reward64_dev = off(ib.reward64_dev, b);
            gi.gt64_dev = off(ib.gt64_dev, b * gd);
            mg::Family* f = env->fams[g];
            void* obs_g = (char*)obs_dev + b * ob;
            if (autoreset && gi.final_obs_dev) {
                // terminal frames wanted: step without auto-reset (obs rows of finished instances = terminal frames), keep
                // a copy of exactly those rows, then reset the finished instances with seed=None -- the same RNG
                // consumption and frames as the fused path (tests/test_gpu_vector_api.py)
                f->step(actions_dev + b * ad, obs_g, reward_dev + b, done_dev + b, off(gt_dev, b * gd), &gi, 0, gst);
                f->raster_only(gi.final_obs_dev, done_dev + b, gst);
                f->reset(nullptr, done_dev + b, obs_g, off(gt_dev, b * gd), gst);
            } else {
                f->step(actions_dev + b * ad, obs_g, reward_dev + b, done_dev + b, off(gt_dev, b * gd), &gi, autoreset, gst);
            }
            if (gi.gt64_dev) f->ground_truth64(gi.gt64_dev, gst);
        });
    });
}

int mg_ground_truth64(mg_env* env, double* gt64_dev, void* stream) {
    return guarded(env, [&] {
        const int gd = env->fam->gt_dim();
        if (!gd) return;
        if (!gt64_dev) throw std::runtime_error("mg_ground_truth64: gt64_dev is NULL");
        for_groups(env, (hipStream_t)stream, false, [&](int g, hipStream_t st) {
            env->fams[g]->sync_state();
            env->fams[g]->ground_truth64(gt64_dev + (size_t)env->base[g] * gd, st);
        });
    });
}

// ---- the single-instance fast path (include/memgym.h: mg_single_io) ----
int mg_single_open(mg_env* env, mg_single_io* io) {
    return guarded(env, [&] {
        if (!io || io->struct_size != sizeof(mg_single_io)) throw std::runtime_error("mg_single_open: set io->struct_size = sizeof(mg_single_io)");
        if (env->num_envs != 1 || env->groups() != 1) throw std::runtime_error("mg_single_open: the handle must hold exactly one instance in one group");
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
            S.bytes = o;
            MG_HIP(hipHostMalloc((void**)&S.host, S.bytes, hipHostMallocMapped | hipHostMallocCoherent));
            memset(S.host, 0, S.bytes);
            MG_HIP(hipHostGetDevicePointer((void**)&S.dev, S.host, 0));
            S.open = true;
        }
        if (env->fam->vec_dim()) {  // the vector observation is written at every reset, into the mapped buffer from now on
            env->vec_dev = (float*)(S.dev + S.o_vec);
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
        MG_HIP(hipStreamSynchronize(st));
    });
}

int mg_single_step(mg_env* env, int32_t a0, int32_t a1, void* stream) {
    return guarded(env, [&] {
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
        f->step((const int32_t*)(S.dev + S.o_action), S.dev + S.o_obs, (float*)(S.dev + S.o_reward32), (uint8_t*)(S.dev + S.o_done),
                f->gt_dim() ? (float*)(S.dev + S.o_gt32) : nullptr, &ib, 0, st);
        f->ground_truth64((double*)(S.dev + S.o_gt), st);
        MG_HIP(hipStreamSynchronize(st));
    });
}

size_t mg_state_size(const mg_env* env) {
    if (!env) return 0;
    size_t t = sizeof(StateHeader);
    for (auto* f : env->fams)
        for (auto& b : f->state_blobs()) t += b.second;
    return t;
}

int mg_get_state(mg_env* env, void* host_buf, size_t size) {
    return guarded(env, [&] {
        if (!host_buf || size < mg_state_size(env)) throw std::runtime_error("mg_get_state: buffer too small");
        MG_HIP(hipDeviceSynchronize());
        for (auto* f : env->fams) f->sync_state();
        StateHeader h;
        memset(&h, 0, sizeof(h));
        memcpy(h.magic, "MGSTATE1", 8);
        h.version = MG_STATE_VERSION;
        h.num_envs = (uint32_t)env->num_envs;
        h.payload = mg_state_size(env) - sizeof(StateHeader);
        h.id_hash = fnv1a(env->id);
        h.pad[0] = (uint8_t)env->groups();
        memcpy(host_buf, &h, sizeof(h));
        char* p = (char*)host_buf + sizeof(StateHeader);
        for (auto* f : env->fams)
            for (auto& b : f->state_blobs()) {
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
        if ((int)h.pad[0] != env->groups())
            throw std::runtime_error("mg_set_state: the blob was taken with " + std::to_string((int)h.pad[0]) + " instance group(s), the handle has " +
                                     std::to_string(env->groups()) + " (mg_set_groups before the first reset)");
        if (h.payload != mg_state_size(env) - sizeof(StateHeader) || size < mg_state_size(env))
            throw std::runtime_error("mg_set_state: payload size differs from this handle's state");
        MG_HIP(hipDeviceSynchronize());
        const char* p = (const char*)host_buf + sizeof(StateHeader);
        for (auto* f : env->fams) {
            for (auto& b : f->state_blobs()) {
                MG_HIP(hipMemcpy(b.first, p, b.second, hipMemcpyHostToDevice));
                p += b.second;
            }
            f->on_state_loaded();  // reset(seed=None) / auto-reset are legal on a restored handle
        }
        env->started = true;
    });
}

int mg_set_profiling(mg_env* env, int on) {
    return guarded(env, [&] {
        env->prof_stride = on < 0 ? 0 : on;
        for (auto* f : env->fams) {
            f->prof.stride = env->prof_stride;
            f->prof.count[0] = f->prof.count[1] = 0;
        }
    });
}

int mg_get_profile(mg_env* env, int kind, double* total_ms, int64_t* launches) {
    return guarded(env, [&] {
        if (kind < 0 || kind > 1 || !total_ms || !launches) throw std::runtime_error("mg_get_profile: bad arguments");
        *total_ms = 0;
        *launches = 0;
        for (auto* f : env->fams) {  // with several blocks: the sum over the blocks' (concurrent) launches
            double ms = 0;
            int64_t n = 0;
            f->prof.collect(kind, &ms, &n);
            *total_ms += ms;
            *launches += n;
        }
    });
}

int mg_poll_errors(mg_env* env, int* flags) {
    return guarded(env, [&] {
        if (!flags) throw std::runtime_error("mg_poll_errors: NULL");
        MG_HIP(hipDeviceSynchronize());
        *flags = 0;
        for (auto* f : env->fams) *flags |= f->poll_errors();
    });
}

int mg_peek_errors(mg_env* env, int* flags) {
    return guarded(env, [&] {
        if (!flags) throw std::runtime_error("mg_peek_errors: NULL");
        *flags = 0;
        for (auto* f : env->fams) *flags |= f->peek_errors();
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
        int64_t total = 0;
        for (auto* f : env->fams) {
            int64_t v = 0;
            if (!f->debug_counter(name, &v)) throw std::runtime_error(std::string("mg_debug_counter: no counter named ") + name + " for " + env->id);
            total += v;
        }
        *value = total;
    });
}

int mg_debug_rng(mg_env* env, int32_t i, uint64_t* out) {
    return guarded(env, [&] {
        if (i < 0 || i >= env->num_envs) throw std::runtime_error("mg_debug_rng: index out of range");
        MG_HIP(hipDeviceSynchronize());
        int g = 0;
        while (g + 1 < env->groups() && i >= env->base[g + 1]) ++g;
        env->fams[g]->sync_state();
        env->fams[g]->debug_rng(i - env->base[g], out);
    });
}

}  // extern "C"
