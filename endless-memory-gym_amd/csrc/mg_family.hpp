// mg_family.hpp -- host-side interface every environment family implements behind the C ABI (include/memgym.h).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/memgym.h"
#include "mg_lab.hpp"
#include "mg_device.hpp"

namespace mg {

void set_error(const std::string& msg);
static_assert(MG_MAX_OPTION_SETS == 8, "set_index() (mg_device.hpp) masks an instance's set index with 7");

#define MG_HIP(expr)                                                                                 \
    do {                                                                                             \
        hipError_t e_ = (expr);                                                                      \
        if (e_ != hipSuccess)                                                                        \
            throw std::runtime_error(std::string(#expr) + " failed: " + hipGetErrorString(e_));      \
    } while (0)

// RAII device array
// One error word in pinned, coherent host memory mapped into the device's address space: kernels raise bits with a
// system-scope atomic OR (rare path), the host reads them without touching the stream (mg_peek_errors).
struct ErrorWord {
    int* host = nullptr;
    int* dev = nullptr;
    ErrorWord() {}
    ErrorWord(const ErrorWord&) = delete;
    ErrorWord& operator=(const ErrorWord&) = delete;
    ~ErrorWord() {
        if (host) (void)hipHostFree(host);
    }
    void alloc() {
        MG_HIP(hipHostMalloc((void**)&host, 64, hipHostMallocMapped | hipHostMallocCoherent));
        *host = 0;
        MG_HIP(hipHostGetDevicePointer((void**)&dev, host, 0));
    }
    int peek() const { return __atomic_load_n(host, __ATOMIC_ACQUIRE); }
    int take() { return __atomic_exchange_n(host, 0, __ATOMIC_ACQ_REL); }
};

#ifdef __HIPCC__
__device__ __forceinline__ void raise_error(int* err, int bit) {
    __hip_atomic_fetch_or(err, bit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
// Deferred-reset queues (`cap` entries + a counter that the serving launch zeroes when it has drained them): a push never
// writes past the end.  The counter can only exceed `cap` if an earlier serving launch failed after its step kernel had
// queued entries; that is flagged (bit 64) instead of becoming an out-of-bounds store, and the servers clamp the count.
constexpr int ERR_QUEUE_OVERFLOW = 64;
__device__ __forceinline__ void queue_push(int* queue, int* counter, int cap, int value, int* err) {
    const int slot = atomicAdd(counter, 1);
    if (__builtin_expect(slot < cap, 1)) queue[slot] = value;
    else raise_error(err, ERR_QUEUE_OVERFLOW);
}
__device__ __forceinline__ int queue_count(const int* counter, int cap) {
    const int c = *counter;
    return c < cap ? c : cap;
}
#endif

template <typename T>
struct DevArray {
    T* p = nullptr;
    size_t n = 0;
    DevArray() {}
    DevArray(const DevArray&) = delete;
    DevArray& operator=(const DevArray&) = delete;
    ~DevArray() { release(); }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        n = 0;
    }
    void alloc(size_t count, bool zero = true) {
        release();
        n = count;
        MG_HIP(hipMalloc((void**)&p, sizeof(T) * (count ? count : 1)));
        if (zero) MG_HIP(hipMemset(p, 0, sizeof(T) * (count ? count : 1)));
    }
    void upload(const std::vector<T>& v) {
        alloc(v.size(), false);
        MG_HIP(hipMemcpy(p, v.data(), sizeof(T) * v.size(), hipMemcpyHostToDevice));
    }
    size_t bytes() const { return sizeof(T) * n; }
};

// Per-instance RNG streams (mg::RngSoA, mg_device.hpp) as five device arrays; part of every family's checkpoint
struct RngStore {
    DevArray<uint64_t> s_hi, s_lo, i_hi, i_lo, buf;
    void alloc(size_t n) { s_hi.alloc(n); s_lo.alloc(n); i_hi.alloc(n); i_lo.alloc(n); buf.alloc(n); }
    RngSoA view() { return RngSoA{s_hi.p, s_lo.p, i_hi.p, i_lo.p, buf.p}; }
    void blobs(std::vector<std::pair<void*, size_t>>& v) {
        v.push_back({s_hi.p, s_hi.bytes()}); v.push_back({s_lo.p, s_lo.bytes()}); v.push_back({i_hi.p, i_hi.bytes()});
        v.push_back({i_lo.p, i_lo.bytes()}); v.push_back({buf.p, buf.bytes()});
    }
    void debug(int i, uint64_t out[6]) {
        uint64_t b;
        MG_HIP(hipMemcpy(&out[0], s_hi.p + i, 8, hipMemcpyDeviceToHost));
        MG_HIP(hipMemcpy(&out[1], s_lo.p + i, 8, hipMemcpyDeviceToHost));
        MG_HIP(hipMemcpy(&out[2], i_hi.p + i, 8, hipMemcpyDeviceToHost));
        MG_HIP(hipMemcpy(&out[3], i_lo.p + i, 8, hipMemcpyDeviceToHost));
        MG_HIP(hipMemcpy(&b, buf.p + i, 8, hipMemcpyDeviceToHost));
        out[4] = (b >> 32) & 1;
        out[5] = b & 0xFFFFFFFFull;
    }
};

// Host side of an OptList (mg_device.hpp): packs the entries, owns the device copy of lists that do not fit the inline form.
struct OptListStore {
    DevArray<int32_t> ext;
    std::vector<int> host;  // the entries as set (capacity checks at reset time)
    // largest entry of the list `l` this store backs.  A store that was never set() (an option set > 0 whose parameter block was copied
    // from the defaults) has no host mirror: the defaults are short byte lists held in l's inline words -- read those (the
    // capacity check at reset time saw 0 for such a set).
    int max(const OptList& l) const {
        int m = 0;
        if (!host.empty() || l.ext) {
            for (int x : host) m = x > m ? x : m;
            return m;
        }
        for (int k = 0; k < l.n && k < OPT_INLINE; ++k) {
            const int x = (int)((l.w[k >> 2] >> (8 * (k & 3))) & 0xFFu);
            m = x > m ? x : m;
        }
        return m;
    }
    void set(OptList& l, const std::vector<int>& v) {
        host = v;
        l.n = (int)v.size();
        for (int j = 0; j < OPT_INLINE / 4; ++j) l.w[j] = 0u;
        l.ext = nullptr;
        bool bytes = true;
        for (int x : v) bytes = bytes && x >= 0 && x <= 255;
        for (int k = 0; bytes && k < l.n && k < OPT_INLINE; ++k) l.w[k >> 2] |= (uint32_t)(v[k] & 0xFF) << (8 * (k & 3));
        if (l.n > OPT_INLINE || !bytes) {
            std::vector<int32_t> b(v.begin(), v.end());
            // the previous array may still be read by kernels in flight on some stream: synchronise before it goes
            MG_HIP(hipDeviceSynchronize());
            ext.upload(b);
            l.ext = ext.p;
        }
    }
};

struct OptionError {
    int code;  // -2 unknown key, -3 unsupported value
    std::string msg;
};

// Optional per-kernel timing with HIP events recorded on the launch stream (bench.py's roofline leg).
struct KernelProfile {
    int stride = 0;          // 0 = off; N = bracket every N-th step (events cost ~7 us per bracketed launch)
    uint64_t count[2] = {0, 0};
    bool live[2] = {false, false};
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev[2];  // 0 = logic kernel, 1 = raster kernel
    void begin(int kind, hipStream_t s) {
        live[kind] = stride > 0 && (count[kind]++ % (uint64_t)stride) == 0;
        if (!live[kind]) return;
        hipEvent_t a, b;
        MG_HIP(hipEventCreate(&a));
        MG_HIP(hipEventCreate(&b));
        MG_HIP(hipEventRecord(a, s));
        ev[kind].push_back({a, b});
    }
    void end(int kind, hipStream_t s) {
        if (!live[kind]) return;
        MG_HIP(hipEventRecord(ev[kind].back().second, s));
    }
    // sum of elapsed ms and launch count for `kind`; synchronises and clears
    void collect(int kind, double* ms, int64_t* launches) {
        double t = 0;
        for (auto& p : ev[kind]) {
            MG_HIP(hipEventSynchronize(p.second));
            float f = 0;
            MG_HIP(hipEventElapsedTime(&f, p.first, p.second));
            t += f;
            (void)hipEventDestroy(p.first);
            (void)hipEventDestroy(p.second);
        }
        *ms = t;
        *launches = (int64_t)ev[kind].size();
        ev[kind].clear();
    }
};

class Family {
   public:
    KernelProfile prof;
    void end_logic(hipStream_t s) { prof.end(0, s); }
    int obs_format = MG_OBS_U8_XYC;  // stream-out format of the raster kernel (include/memgym.h)
    virtual ~Family() {}
    virtual int action_dim() const = 0;
    virtual int gt_dim() const = 0;
    // mg_single_step (mg_api.hip): can the family's next step() store `ticket` to *flag_dev itself when its results are out?  false: the
    // caller waits through a stream memory operation behind the step's launches
    virtual bool arm_done_flag(uint32_t* /*flag_dev*/, uint32_t /*ticket*/) { return false; }
    // capacities of the per-instance lists the reference grows without limit (include/memgym.h: mg_set_capacity / mg_capacity)
    virtual void set_capacity(const std::string& what, int64_t) { throw OptionError{-2, "this env id has no capacity named " + what}; }
    virtual int64_t capacity(const std::string& what) const { throw OptionError{-2, "this env id has no capacity named " + what}; }
    virtual int vec_dim() const { return 0; }          // size of obs["vector_observation"] (MortarMayhemB*), else 0
    virtual void bind_vector_obs(float* /*dev*/) {}    // caller buffer [num_envs][vec_dim], written at every reset
    virtual const char* info_name(int k) const = 0;
    virtual void set_option(const std::string& key, const double* v, int n) = 0;  // throws OptionError
    // per-instance option sets (include/memgym.h: mg_set_option_set / mg_bind_option_sets); families without them refuse sets > 0
    virtual void set_option_set(int set, const std::string& key, const double* v, int n) {
        if (set != 0) throw OptionError{-3, "this env id has no per-instance option sets in this build (use one handle per option set)"};
        set_option(key, v, n);
    }
    virtual void bind_option_sets(const int32_t* set_of_dev) {
        if (set_of_dev) throw OptionError{-3, "this env id has no per-instance option sets in this build (use one handle per option set)"};
    }
    virtual void reset(const int64_t* seeds, const uint8_t* mask, void* obs, float* gt, hipStream_t s) = 0;
    virtual void step(const int32_t* actions, void* obs, float* reward, uint8_t* done, float* gt,
                      const mg_info_buffers* info, int autoreset, hipStream_t s) = 0;
    // rasterise the CURRENT frame descriptors of the instances with only[i] != 0 into `obs` (others untouched)
    virtual void raster_only(void* obs, const uint8_t* only, hipStream_t s) = 0;
    // true: step(..., autoreset = 1) with info->final_obs_dev set keeps the terminal observations ITSELF (the frame workgroup of a finishing
    // instance draws the terminal frame into final_obs_dev, then the reset frame into obs); false: mg_step takes the generic path (a step
    // without auto-reset, the terminal rows copied, a masked reset).  Asked once per call, on the stream of the call.
    virtual bool keeps_final_obs(hipStream_t /*s*/) { return false; }
    // render("debug_rgb_array") before its final x4 stretch: the debug surface of every instance (the reference's
    // _build_debug_surface, e.g. mortar_mayhem_grid.py:104-135) as uint8 [num_envs][84 x][84 y][3], the observation layout
    virtual void raster_debug(void* frames, hipStream_t s) = 0;
    // checkpoint: list of (device pointer, bytes) making up the state
    virtual std::vector<std::pair<void*, size_t>> state_blobs() = 0;
    // called by mg_set_state after the blobs were restored: the instances now carry seeded RNG streams
    virtual void on_state_loaded() {}
    // called before anything looks at the state from outside (mg_get_state, mg_debug_rng, mg_render_debug): work a family has
    // put off without changing any result (Endless Mystery Path: owed path segments) is done now.  Synchronous.
    virtual void sync_state() {}
    virtual void debug_rng(int i, uint64_t out[6]) = 0;
    // info["ground_truth"] as the reference returns it -- float64 (e.g. endless_mortar_mayhem.py:259,358) -- of every instance,
    // [num_envs][gt_dim], computed from the CURRENT state (the float32 gt_dev of mg_step / mg_reset is its rounding); a small
    // launch of its own, only when a caller asks (mg_info_buffers.gt64_dev, mg_ground_truth64).  No-op for gt_dim() == 0.
    virtual void ground_truth64(double* /*gt64_dev*/, hipStream_t /*s*/) {}
    // device-side error bits accumulated since the last call (0 = none); synchronises
    virtual int poll_errors() { return 0; }
    // the same bits as seen right now, without synchronising or clearing
    virtual int peek_errors() { return 0; }
    // test / telemetry counters by name (mg_debug_counter); false = this family has no such counter.  Synchronous.
    virtual bool debug_counter(const std::string& /*name*/, int64_t* /*out*/) { return false; }
};

Family* make_mortar(int variant, int num_envs);
Family* make_spot(int endless, int num_envs);
Family* make_mystery(int variant, int num_envs);

// Workgroup size of the one-lane-per-instance logic kernels (all are written for any multiple of 64 up to 256).
// MEMGYM_STEP_BLOCK overrides `shipped` per process (measurements: profiles/r03_step_blocks.md).
inline int step_block(int shipped) {
    static const int forced = [] {
        const char* e = lab_env("MEMGYM_STEP_BLOCK");
        const int v = e ? atoi(e) : 0;
        return (v == 64 || v == 128 || v == 256) ? v : 0;
    }();
    return forced ? forced : shipped;
}

inline int to_int_checked(double v, const char* key) {
    int i = (int)v;
    if ((double)i != v) throw OptionError{-3, std::string("option ") + key + " must be integral"};
    return i;
}

}  // namespace mg
