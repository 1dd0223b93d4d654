// mg_placement.hip -- mg_obs_alloc / mg_obs_free: an observation buffer whose physical pages come from TWO of the
// MI355X's three HBM zones, alternating every 64 MiB.
//
// Why (profiles/r02_zones.md): the 288 GB of a MI355X fall into three zones of ~96 GB (presumably the three ranks of the
// 12-high HBM3E stacks).  The raster kernel's store stream -- ~1,800 persistent workgroups writing 21 KB frames in
// several 300-MB-apart windows at once, the evictions of the 256-MB Infinity Cache behind them -- runs at 5.1-5.5 TB/s
// when everything it writes lies in ONE zone and at 6.2-6.4 TB/s when the traffic is split over two (a linear fill does
// not care).  A process' allocations come out of the driver's VRAM manager in address order, so a buffer of ordinary
// size lies in one zone unless it happens to straddle a boundary (round 1's "fast and slow allocations").
//
// What: the buffer is assembled with the HIP virtual-memory API from physical pieces of 304 MiB (one window of the
// probe below).  Every piece is classified against the first one with a two-window store probe (the raster's store shape
// over both pieces at once: ~5.1 TB/s when they share a zone, ~6.3 TB/s when they do not).  Pieces are requested until
// half of the buffer can be taken from the first piece's zone and half from elsewhere; pieces that are not needed and
// 8-GiB spacer allocations (never mapped or written) keep the driver's allocator moving and are released before the
// function returns.  The chosen pieces are mapped alternately into one contiguous virtual range.  The driver serves
// requests of different sizes from different free lists, which is why the PIECES THEMSELVES are probed, not their
// neighbours.  If no second zone turns up within the budget the buffer still works, from one zone (info.zones == 1).
#include <algorithm>
#include <chrono>
#include <cmath>
#include <map>
#include <mutex>
#include <vector>

#include "mg_family.hpp"
#include "mg_lab.hpp"
using mg::lab_env;

namespace {

constexpr size_t MiB = 1ull << 20, GiB = 1ull << 30;
constexpr size_t PIECE = 304 * MiB;         // = one window of the probe: 14,336 frames of 21,168 B (64-MiB pieces measured the
                                            // same, 16 MiB +6 %, 2 MiB +12 %: profiles/r02_zones.md)
constexpr int FILL_STEP = 32;               // filler handles per step of the walk
constexpr int PROBE_GRID = 14336;
// Classification of a probed pair of pieces.  Pairs in one zone measured 4.9-5.5 TB/s, pairs across zones 6.1-6.5
// (profiles/r02_zones.md) -- on THESE boxes.  Those absolute figures are only the PRIOR (round 2 classified by them alone):
// every probe of two distinct pieces feeds the per-device calibration (Calib), and once the samples show two clusters the
// thresholds become relative: the geometric mean across the gap between the clusters -/+ 1.5 %.  A
// search whose earlier decisions the calibrated thresholds would change starts over once.  (A piece paired with ITSELF is no
// calibration: both windows then hit the same cache lines, 11.6 TB/s.)  MEMGYM_OBS_SAME_TBPS / MEMGYM_OBS_CROSS_TBPS set the prior.
static const double PRIOR_SAME_TBPS = lab_env("MEMGYM_OBS_SAME_TBPS") ? atof(lab_env("MEMGYM_OBS_SAME_TBPS")) : 5.35;
static const double PRIOR_CROSS_TBPS = lab_env("MEMGYM_OBS_CROSS_TBPS") ? atof(lab_env("MEMGYM_OBS_CROSS_TBPS")) : 5.85;
struct Calib {
    // Round 4: the thresholds turn relative only on evidence of TWO clusters -- at least two probes on either side of the
    // widest gap between neighbouring samples, that gap >= 7 % and the whole spread >= 15 %.  The first version switched as soon as
    // the fastest probe was 15 % above the slowest: pairs of ONE zone differ by up to 12 % among themselves and a single noisy
    // probe did the rest, after which same-zone pairs at 5.5 TB/s were classed "other zone" for the rest of the process.
    std::vector<double> seen;  // every probe of two distinct pieces so far (bounded)
    bool relative = false;
    double same_t = PRIOR_SAME_TBPS, cross_t = PRIOR_CROSS_TBPS;
    void observe(double t) {
        if (seen.size() < 256) seen.push_back(t);
        std::vector<double> v(seen);
        std::sort(v.begin(), v.end());
        relative = false;
        same_t = PRIOR_SAME_TBPS;
        cross_t = PRIOR_CROSS_TBPS;
        if (v.size() < 4 || v.back() < 1.15 * v.front()) return;
        size_t cut = 0;
        double gap = 0;
        for (size_t k = 2; k + 2 <= v.size(); ++k)  // two samples at least on either side
            if (v[k] / v[k - 1] > gap) {
                gap = v[k] / v[k - 1];
                cut = k;
            }
        if (cut == 0 || gap < 1.07) return;
        const double mid = std::sqrt(v[cut - 1] * v[cut]);
        relative = true;
        same_t = std::max(v[cut - 1] * 1.005, 0.985 * mid);  // (never below the fastest sample of the slow cluster)
        cross_t = std::min(v[cut] * 0.995, 1.015 * mid);
    }
    int classify(double t) const { return t < same_t ? 0 : (t <= cross_t ? 1 : 2); }  // 0 same zone, 1 unclear, 2 other zone
};

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// the raster's store shape without the compose work: workgroup b writes frame b of window 0, then frame b of window 1
__global__ __launch_bounds__(256) void zone_probe_kernel(u32x4* w0, u32x4* w1) {
    extern __shared__ unsigned char occupancy_pad[];  // 22 KiB requested: seven workgroups per CU, like the raster
    const int tid = threadIdx.x;
#pragma unroll 1
    for (int k = 0; k < 2; ++k) {
        u32x4* dst = (k ? w1 : w0) + (size_t)blockIdx.x * 1323;
#pragma unroll
        for (int j = 0; j < 5; ++j) dst[tid + 256 * j] = (u32x4)(0u);
        if (tid < 43) dst[tid + 1280] = (u32x4)(0u);
    }
}

// mg_store_probe: what THIS box, THIS buffer placement and THIS launch size allow a pure store stream (bench.py's per-box
// control next to the headline).  Pattern 0: linear fill, one 16-byte store per thread in short-lived workgroups -- the
// memory system's ceiling for stores.  Pattern 1: the raster's store SHAPE without any compose work -- persistent 256-lane
// workgroups (the raster's grid and LDS request, seven per CU), each writing whole 21,168-byte frames b, b + grid, ... --
// the ceiling of a frame-shaped stream.
__global__ __launch_bounds__(256) void store_probe_linear_kernel(u32x4* out, size_t nvec) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < nvec) out[i] = (u32x4)(0u);
}
__global__ __launch_bounds__(256) void store_probe_frames_kernel(u32x4* out, int n) {
    extern __shared__ unsigned char occupancy_pad[];  // 22 KiB requested: seven workgroups per CU, like the raster
    const int tid = threadIdx.x;
    for (int f = blockIdx.x; f < n; f += gridDim.x) {
        u32x4* dst = out + (size_t)f * 1323;
#pragma unroll
        for (int j = 0; j < 5; ++j) dst[tid + 256 * j] = (u32x4)(0u);
        if (tid < 43) dst[tid + 1280] = (u32x4)(0u);
    }
}

// Pattern 4 (round 6): the frame walk with LINE-ALIGNED ownership -- a frame is 21,168 B = 165.375 cache lines of 128 B, so
// every frame boundary but each eighth splits a line between two workgroups; here workgroup f writes [up128(21,168 f), up128(21,168 (f + 1)))
// instead: the same bytes per launch, every 128-byte line written by exactly one workgroup (what a raster would do that takes the stray
// 48 .. 80 bytes of its neighbour's frame through an exchange).
__global__ __launch_bounds__(256) void store_probe_lines_kernel(u32x4* out, int n) {
    extern __shared__ unsigned char occupancy_pad[];
    const int tid = threadIdx.x;
    const size_t end = (size_t)n * 1323;
    for (int f = blockIdx.x; f < n; f += gridDim.x) {
        size_t v0 = ((size_t)f * 1323 + 7) & ~(size_t)7, v1 = ((size_t)(f + 1) * 1323 + 7) & ~(size_t)7;  // in 16-byte vectors: 8 per line
        if (f == 0) v0 = 0;
        if (v1 > end) v1 = end;
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            const size_t v = v0 + tid + 256 * j;
            if (v < v1) out[v] = (u32x4)(0u);
        }
    }
}
// Pattern 6: the raster's frame walk with an XCD-AWARE order of frames: workgroups are dealt to the eight XCDs round-robin (workgroup b ->
// XCD b % 8, each with an L2 of its own), so neighbouring frames -- whose shared 64-byte block at the boundary is written half by one
// workgroup and half by the next -- lie in DIFFERENT L2s and leave as two partial writes.  Here ordinal v = 64 q + 8 r + x draws frame
// 64 q + 8 x + r: the eight workgroups of one XCD within a block of 64 own eight CONSECUTIVE frames (= 1,323 whole lines), so every split
// block is written by two workgroups of the same XCD and its L2 can merge the halves before the line leaves.  Same bytes, same stores.
__device__ __forceinline__ int xcd_grouped_frame(int v) { return (v & ~63) | ((v & 7) << 3) | ((v >> 3) & 7); }
__global__ __launch_bounds__(256) void store_probe_frames_xcd_kernel(u32x4* out, int n) {
    extern __shared__ unsigned char occupancy_pad[];
    const int tid = threadIdx.x;
    for (int v = blockIdx.x; v < n; v += gridDim.x) {
        const int f = (v | 63) < n ? xcd_grouped_frame(v) : v;  // (a last, partial block of 64 keeps the plain order)
        u32x4* dst = out + (size_t)f * 1323;
#pragma unroll
        for (int j = 0; j < 5; ++j) dst[tid + 256 * j] = (u32x4)(0u);
        if (tid < 43) dst[tid + 1280] = (u32x4)(0u);
    }
}
// Pattern 7: a workgroup writes EIGHT CONSECUTIVE frames one after another, each as the 64-byte-aligned span [up64(start), up64(next start))
// (what a raster would do that holds a frame's last partial block back and writes it together with the head of its next frame): no block is
// ever shared between two workgroups; the concurrently written window is eight times as wide.
__global__ __launch_bounds__(256) void store_probe_octets_kernel(u32x4* out, int n) {
    extern __shared__ unsigned char occupancy_pad[];
    const int tid = threadIdx.x;
    const size_t end = (size_t)n * 1323;
    for (int g = blockIdx.x; g * 8 < n; g += gridDim.x) {
#pragma unroll 1
        for (int k = 0; k < 8 && g * 8 + k < n; ++k) {
            const size_t f = (size_t)g * 8 + k;
            size_t v0 = (f * 1323 + 3) & ~(size_t)3, v1 = ((f + 1) * 1323 + 3) & ~(size_t)3;  // in 16-byte vectors: 4 per 64-byte block
            if (v1 > end) v1 = end;
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                const size_t v = v0 + tid + 256 * j;
                if (v < v1) out[v] = (u32x4)(0u);
            }
        }
    }
}
// Pattern 5: one frame per workgroup, no LDS request, no persistent loop (grid = frames): the frame's shape alone, dispatched like a fill.
__global__ __launch_bounds__(256) void store_probe_frame_per_wg_kernel(u32x4* out) {
    const int tid = threadIdx.x;
    u32x4* dst = out + (size_t)blockIdx.x * 1323;
#pragma unroll
    for (int j = 0; j < 5; ++j) dst[tid + 256 * j] = (u32x4)(0u);
    if (tid < 43) dst[tid + 1280] = (u32x4)(0u);
}

// Pattern 2 (lab: tools/store_shapes.py): the frame walk with each lane writing PAIRS of adjacent vectors (32 contiguous bytes per lane:
// a linear sweep of that kind is as fast as one vector per thread, profiles/r01g_store_patterns.md) -- three store rounds per frame
// instead of six.  Pattern 3: pairs, and the workgroup's four waves write contiguous quarters of the frame.
__global__ __launch_bounds__(256) void store_probe_frames_pairs_kernel(u32x4* out, int n, int wave_quarters) {
    extern __shared__ unsigned char occupancy_pad[];
    const int tid = threadIdx.x;
    for (int f = blockIdx.x; f < n; f += gridDim.x) {
        u32x4* dst = out + (size_t)f * 1323;
        if (!wave_quarters) {
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const int v = 2 * tid + 512 * j;
                if (v < 1323) dst[v] = (u32x4)(0u);
                if (v + 1 < 1323) dst[v + 1] = (u32x4)(0u);
            }
        } else {
            const int w = tid >> 6, lane = tid & 63, lo = w * 332, hi = w == 3 ? 1323 : lo + 332;  // 332 vectors per wave (even)
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const int v = lo + 2 * lane + 128 * j;
                if (v < hi) dst[v] = (u32x4)(0u);
                if (v + 1 < hi) dst[v + 1] = (u32x4)(0u);
            }
        }
    }
}

// Virtual ranges are NEVER handed back to the runtime.  Measured on ROCm 7.2 / MI355X (tools/vmm_stress.py, profiles/
// r02_zones.md): when a range freed with hipMemAddressFree is reserved again and mapped onto other physical memory, stores
// through it can land in the OLD physical pages (5 of 80 allocate-fill-verify-free cycles read back wrong, up to 95 % of a
// 1.4-GB buffer; 0 of 80 when no range is ever reused) -- stale translations, which would also let the probe below
// scribble over memory that belongs to somebody else by now.  A reservation costs address space only (a few GiB of the
// 128-TiB space per call), so unmapped ranges simply stay reserved.  MEMGYM_OBS_REUSE_VA=1 restores the frees (experiments).
inline void va_free(void* va, size_t bytes) {
    static const bool reuse = lab_env("MEMGYM_OBS_REUSE_VA") && atoi(lab_env("MEMGYM_OBS_REUSE_VA")) != 0;
    if (reuse) (void)hipMemAddressFree(va, bytes);
}

// write-then-read check of a freshly assembled range: every 16-byte vector gets a value derived from its index
__global__ void verify_fill_kernel(u32x4* p, size_t nvec, unsigned salt) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += (size_t)gridDim.x * blockDim.x) {
        const unsigned h = (unsigned)i * 2654435761u + salt;
        p[i] = (u32x4){h, h ^ 0x9E3779B9u, (unsigned)(i >> 32) + salt, ~h};
    }
}
__global__ void verify_check_kernel(const u32x4* p, size_t nvec, unsigned salt, unsigned long long* bad) {
    unsigned long long mine = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += (size_t)gridDim.x * blockDim.x) {
        const unsigned h = (unsigned)i * 2654435761u + salt;
        const u32x4 v = p[i];
        mine += (v.x != h) | (v.y != (h ^ 0x9E3779B9u)) | (v.z != (unsigned)(i >> 32) + salt) | (v.w != ~h);
    }
    if (mine) atomicAdd(bad, mine);
}

struct Piece {
    hipMemGenericAllocationHandle_t h = nullptr;
    size_t bytes = 0;
};

struct Mapping {
    size_t va_bytes = 0;
    void* base = nullptr;                          // start of the virtual range (the caller's pointer lies lead bytes into it)
    std::vector<std::pair<size_t, Piece>> pieces;  // (offset from base, piece)
    std::vector<int> piece_class;                  // group id of every piece (pieces of one id are slow together)
    bool plain = false;                            // hipMalloc fallback
    int device = 0;
};

std::mutex g_mu;
std::map<void*, Mapping> g_live;

hipMemAllocationProp prop_for(int device) {
    hipMemAllocationProp p = {};
    p.type = hipMemAllocationTypePinned;
    p.location.type = hipMemLocationTypeDevice;
    p.location.id = device;
    return p;
}

// `exportable`: handle type POSIX file descriptor.  Such pieces come out of a different place of the driver's VRAM manager
// than ordinary ones (often the other end of the memory, i.e. another zone, without any walking); same speed otherwise.
size_t g_va_reserved = 0;  // bytes of virtual address space reserved so far (never handed back: see va_free)
hipError_t va_reserve(void** va, size_t bytes) {
    const hipError_t e = hipMemAddressReserve(va, bytes, 2 * MiB, nullptr, 0);
    if (e == hipSuccess) g_va_reserved += bytes;
    return e;
}

bool create_piece(int device, size_t bytes, Piece* out, bool exportable = false) {
    static const bool no_vmm = lab_env("MEMGYM_OBS_NO_VMM") && atoi(lab_env("MEMGYM_OBS_NO_VMM")) != 0;  // tests: a runtime without hipMemCreate
    if (no_vmm) return false;
    hipMemAllocationProp p = prop_for(device);
    if (exportable) p.requestedHandleType = hipMemHandleTypePosixFileDescriptor;
    hipMemGenericAllocationHandle_t h;
    if (hipMemCreate(&h, bytes, &p, 0) != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    out->h = h;
    out->bytes = bytes;
    return true;
}

void release_piece(Piece& p) {
    if (p.h) (void)hipMemRelease(p.h);
    p.h = nullptr;
}

// Access for the owning device and -- so that the buffer can be read by (or gathered to) the other GPUs of the node, like an
// ordinary allocation after hipDeviceEnablePeerAccess -- for every device that reports peer access to it.  Peers that refuse
// are skipped: the owner's access is what the library itself needs.
void map_at(void* va, const Piece& p, int device) {
    MG_HIP(hipMemMap(va, p.bytes, 0, p.h, 0));
    static const std::vector<int> all = [] {
        int n = 0;
        if (hipGetDeviceCount(&n) != hipSuccess) n = 1;
        std::vector<int> v(n);
        for (int i = 0; i < n; ++i) v[i] = i;
        return v;
    }();
    hipMemAccessDesc acc = {};
    acc.location.type = hipMemLocationTypeDevice;
    acc.location.id = device;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    MG_HIP(hipMemSetAccess(va, p.bytes, &acc, 1));
    for (int d : all) {
        if (d == device) continue;
        int can = 0;
        if (hipDeviceCanAccessPeer(&can, d, device) != hipSuccess || !can) {
            (void)hipGetLastError();
            continue;
        }
        acc.location.id = d;
        if (hipMemSetAccess(va, p.bytes, &acc, 1) != hipSuccess) (void)hipGetLastError();
    }
}

// a candidate piece, mapped on its own for the probe
struct Cand {
    Piece piece;
    void* va = nullptr;
    double tbps = 0;  // two-window probe against the reference piece
    bool make(int device, bool exportable = false) {
        if (!create_piece(device, PIECE, &piece, exportable)) return false;
        if (va_reserve(&va, PIECE) != hipSuccess) {
            release_piece(piece);
            return false;
        }
        map_at(va, piece, device);
        return true;
    }
    void unmap() {
        if (va) {
            (void)hipMemUnmap(va, PIECE);
            va_free(va, PIECE);
            va = nullptr;
        }
    }
    void drop() {
        unmap();
        release_piece(piece);
    }
};

// number of 16-byte vectors of [p, p + bytes) that do not read back what a previous kernel wrote there
unsigned long long verify_range(void* p, size_t bytes, unsigned salt) {
    unsigned long long* bad = nullptr;
    MG_HIP(hipMalloc((void**)&bad, sizeof *bad));
    MG_HIP(hipMemset(bad, 0, sizeof *bad));
    const size_t nvec = bytes / 16;
    hipLaunchKernelGGL(verify_fill_kernel, dim3(4096), dim3(256), 0, 0, (u32x4*)p, nvec, salt);
    MG_HIP(hipDeviceSynchronize());
    hipLaunchKernelGGL(verify_check_kernel, dim3(4096), dim3(256), 0, 0, (const u32x4*)p, nvec, salt, bad);
    unsigned long long h = 0;
    MG_HIP(hipMemcpy(&h, bad, sizeof h, hipMemcpyDeviceToHost));
    (void)hipFree(bad);
    return h;
}

// TB/s of the two-window store probe over (a, b); best of three launches after one warm-up
double probe_tbps(void* a, void* b) {
    hipEvent_t e0, e1;
    MG_HIP(hipEventCreate(&e0));
    MG_HIP(hipEventCreate(&e1));
    float best = 1e30f;
    for (int r = 0; r < 4; ++r) {
        MG_HIP(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(zone_probe_kernel, dim3(PROBE_GRID), dim3(256), 22528, 0, (u32x4*)a, (u32x4*)b);
        MG_HIP(hipEventRecord(e1, 0));
        MG_HIP(hipEventSynchronize(e1));
        float ms = 0;
        MG_HIP(hipEventElapsedTime(&ms, e0, e1));
        if (r > 0) best = std::min(best, ms);
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    return 2.0 * PROBE_GRID * 21168.0 / (best * 1e-3) / 1e12;
}

}  // namespace

// Per device, for the life of the process: a bounded pool of spare pieces in GROUPS of pieces that are slow together (one
// zone each) -- pieces a search classified but did not need, and the pieces of buffers that were freed.  The next buffer
// starts from the pool: often no new piece, let alone a walk, is needed.
struct ZoneCache {
    std::map<int, std::vector<Piece>> pool;  // group id -> spare pieces
    size_t pooled = 0;
    int next_id = 0;
    Calib calib;  // what the probes of this process have shown so far
};
constexpr size_t POOL_CAP = 10;  // pieces (3 GiB) kept at most
std::map<int, ZoneCache> g_zones;

void pool_put(ZoneCache& Z, int group, Piece p) {
    if (group >= 0 && Z.pooled < POOL_CAP) {
        Z.pool[group].push_back(p);
        Z.pooled++;
    } else {
        release_piece(p);
    }
}

struct Group {
    int id = -1;
    std::vector<Cand> pcs;  // pcs[0] is the group's reference during a search (mapped on its own while the search runs)
};

namespace {
double g_search_ms = 1500.0;  // mg_obs_set_search_ms
}

extern "C" {

// Time bound of mg_obs_alloc's search (milliseconds; the call as a whole stays within 1.5 x the bound plus the final assembly of
// the buffer).  The library reads no environment variable for it: the Python mirror forwards MEMGYM_OBS_SEARCH_MS.
int mg_obs_set_search_ms(double ms) {
    if (!(ms >= 0)) {
        mg::set_error("mg_obs_set_search_ms: need a bound >= 0");
        return -1;
    }
    std::lock_guard<std::mutex> lk(g_mu);
    g_search_ms = ms;
    return 0;
}

// bench.py's per-box control (see store_probe_*_kernel): one launch of `pattern` over n_frames x 21,168 bytes at `buf`, on `stream`.
int mg_store_probe(void* buf, size_t n_frames, int pattern, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!buf || n_frames == 0 || n_frames > (1u << 30) || pattern < 0 || pattern > 7) {
        mg::set_error("mg_store_probe: bad arguments");
        return -1;
    }
    if (pattern == 0) {
        const size_t nvec = n_frames * 1323;
        hipLaunchKernelGGL(store_probe_linear_kernel, dim3((unsigned)((nvec + 255) / 256)), dim3(256), 0, stream, (u32x4*)buf, nvec);
    } else if (pattern == 1) {
        const int grid = (int)std::min<size_t>(n_frames, PROBE_GRID);
        hipLaunchKernelGGL(store_probe_frames_kernel, dim3(grid), dim3(256), 22528, stream, (u32x4*)buf, (int)n_frames);
    } else if (pattern == 4) {
        const int grid = (int)std::min<size_t>(n_frames, PROBE_GRID);
        hipLaunchKernelGGL(store_probe_lines_kernel, dim3(grid), dim3(256), 22528, stream, (u32x4*)buf, (int)n_frames);
    } else if (pattern == 6) {
        const int grid = (int)std::min<size_t>(n_frames, PROBE_GRID);
        hipLaunchKernelGGL(store_probe_frames_xcd_kernel, dim3(grid), dim3(256), 22528, stream, (u32x4*)buf, (int)n_frames);
    } else if (pattern == 7) {
        const int grid = (int)std::min<size_t>((n_frames + 7) / 8, PROBE_GRID);
        hipLaunchKernelGGL(store_probe_octets_kernel, dim3(grid), dim3(256), 22528, stream, (u32x4*)buf, (int)n_frames);
    } else if (pattern == 5) {
        hipLaunchKernelGGL(store_probe_frame_per_wg_kernel, dim3((unsigned)n_frames), dim3(256), 0, stream, (u32x4*)buf);
    } else {
        const int grid = (int)std::min<size_t>(n_frames, PROBE_GRID);
        hipLaunchKernelGGL(store_probe_frames_pairs_kernel, dim3(grid), dim3(256), 22528, stream, (u32x4*)buf, (int)n_frames, pattern == 3 ? 1 : 0);
    }
    if (hipGetLastError() != hipSuccess) {
        mg::set_error("mg_store_probe: launch failed");
        return -1;
    }
    return 0;
}

// Which zone (0 .. G - 1) each of the k pieces of an assembled range comes from: piece p lies in window w(p) of the frame walk (by its
// centre; the buffer starts `lead` bytes into the range, a window = the bytes between two concurrently written fronts) as that window's
// j-th piece and comes from zone (w + j) mod 2; with three zones, piece p from zone p mod 3 (mg_obs_alloc_for below).
static void plan_zones(size_t k, size_t lead, size_t window, size_t G, std::vector<size_t>& want, std::vector<size_t>& count) {
    want.assign(k, 0);
    count.assign(G, 0);
    size_t prev_w = (size_t)-1, j = 0;
    for (size_t p = 0; p < k; ++p) {
        const size_t centre = p * PIECE + PIECE / 2, w = centre > lead ? (centre - lead) / window : 0;
        j = w == prev_w ? j + 1 : 0;
        prev_w = w;
        // (three zones: one after the other -- 3 has no common factor with the 1, 2 or 4 pieces of a window, the fronts of every format
        // fall into different zones as they are; tests/test_obs_plan.py)
        want[p] = G == 2 ? (w + j) % 2 : p % G;
        ++count[want[p]];
    }
}
static size_t range_lead(size_t k, size_t bytes) { return ((k * PIECE - bytes) / 2) & ~(size_t)(2 * MiB - 1); }

int mg_obs_plan(size_t bytes, size_t frame_bytes, int zones, size_t* piece_bytes, size_t* lead_bytes, int* zone_of_piece, int max_pieces) {
    if (bytes == 0 || zones < 2 || zones > 3) {
        mg::set_error("mg_obs_plan: bad arguments");
        return -1;
    }
    const size_t k = (bytes + PIECE - 1) / PIECE;
    std::vector<size_t> want, count;
    plan_zones(k, range_lead(k, bytes), (frame_bytes ? frame_bytes : (size_t)mg::FRAME_BYTES) * (size_t)PROBE_GRID, (size_t)zones, want, count);
    if (piece_bytes) *piece_bytes = PIECE;
    if (lead_bytes) *lead_bytes = range_lead(k, bytes);
    for (size_t p = 0; p < k && (int)p < max_pieces; ++p)
        if (zone_of_piece) zone_of_piece[p] = (int)want[p];
    return (int)k;
}

int mg_obs_alloc(int device, size_t bytes, size_t search_budget_bytes, void** out, mg_obs_alloc_info* info) {
    return mg_obs_alloc_for(device, bytes, (size_t)mg::FRAME_BYTES, search_budget_bytes, out, info);
}

int mg_obs_alloc_for(int device, size_t bytes, size_t frame_bytes, size_t search_budget_bytes, void** out, mg_obs_alloc_info* info) {
    mg_obs_alloc_info I = {};
    try {
        if (!out || bytes == 0) {
            mg::set_error("mg_obs_alloc: bad arguments");
            return -1;
        }
        int prev = 0;
        MG_HIP(hipGetDevice(&prev));
        MG_HIP(hipSetDevice(device));
        struct Restore {
            int d;
            ~Restore() { (void)hipSetDevice(d); }
        } restore{prev};
        const auto t0 = std::chrono::steady_clock::now();
        const bool debug = lab_env("MEMGYM_OBS_DEBUG") != nullptr;
        *out = nullptr;
        size_t free_b = 0, total_b = 0;
        MG_HIP(hipMemGetInfo(&free_b, &total_b));
        // Default budget of the transient walk: half of the free memory, at most 128 GiB (round 2: 55 % / 160 GiB), and
        // the time bound (mg_obs_set_search_ms, 1.5 s) on top.  Usually nothing is walked at all (the first few pieces already differ), but a
        // pristine VRAM can hand out 100-130 GiB of ONE zone in a row (0.3 s of walking there); on memory earlier processes
        // have dirtied the driver wipes what it hands out (~27 ms per GiB) and the time bound ends the walk after ~55 GiB.
        // 32 and 64 GiB were tried this round: one in three processes of a busy box ended on a one-zone buffer (-15 % on the
        // raster).  The fillers are never mapped or touched, but they ARE memory other processes on the same GPU cannot have
        // while the search lasts: several processes per GPU should set MEMGYM_OBS_SEARCH_GB (or obs_placement="plain").
        if (search_budget_bytes == MG_OBS_SEARCH_DEFAULT) search_budget_bytes = std::min<size_t>(free_b / 2, 128 * GiB);
        const size_t k = (bytes + PIECE - 1) / PIECE;
        auto plain = [&](int zones) {  // (called without the lock)
            void* p = nullptr;
            MG_HIP(hipMalloc(&p, bytes));
            Mapping m;
            m.plain = true;
            m.device = device;
            std::lock_guard<std::mutex> lk(g_mu);
            g_live[p] = m;
            *out = p;
            I.zones = zones;
            I.search_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
            if (info) *info = I;
        };
        std::unique_lock<std::mutex> lock(g_mu);
        ZoneCache& Z = g_zones[device];
        // buffers of one piece or less (the 256-MB Infinity Cache absorbs most of those) and boxes without room: plain
        if (k < 2 || (Z.pooled < k && (search_budget_bytes < PIECE || free_b < (k + 4) * PIECE))) {
            lock.unlock();
            plain(0);
            return 0;
        }
        // No group may contribute more than half of the pieces (rounded up); once a walk would be needed, "at least a
        // quarter of the pieces from other groups" is good enough (one in five measured within 2 % of an even split).
        const size_t half = (k + 1) / 2, loose_cap = k - std::max<size_t>(1, k / 4);
        std::vector<Group> groups;
        for (auto& kv : Z.pool) {  // start from the pool
            if (kv.second.empty()) continue;
            Group g;
            g.id = kv.first;
            for (auto& p : kv.second) {
                Cand c;
                c.piece = p;
                g.pcs.push_back(c);
            }
            groups.push_back(g);
        }
        Z.pool.clear();
        Z.pooled = 0;
        auto usable = [&](size_t cap) {
            size_t u = 0;
            for (auto& g : groups) u += std::min(g.pcs.size(), cap);
            return u;
        };
        auto ref_va = [&](Group& g) {  // the group's reference, mapped on its own
            Cand& r = g.pcs[0];
            if (!r.va) {
                if (va_reserve(&r.va, PIECE) != hipSuccess) throw std::runtime_error("mg_obs_alloc: hipMemAddressReserve failed");
                map_at(r.va, r.piece, device);
            }
            return r.va;
        };
        std::vector<Cand> unclear;
        std::vector<Piece> spacers;
        size_t walked = 0;
        std::vector<std::pair<double, int>> decided;  // (probe, class) of this search: rechecked when the calibration turns relative
        bool restarted = false;
        int tries_after_good = 0;
        bool exportable = false;
        auto give_back = [&] {  // everything that is not part of the buffer
            for (auto& g : groups)
                for (auto& c : g.pcs) {
                    c.unmap();
                    pool_put(Z, g.id, c.piece);
                }
            groups.clear();
            for (auto& c : unclear) c.drop();
            unclear.clear();
            for (auto& sp : spacers) release_piece(sp);
            spacers.clear();
        };
        try {
            // Also bounded in time (mg_obs_set_search_ms, default 1,500 ms): on memory a previous process dirtied the driver wipes
            // what it hands out (~27 ms per GiB).  The bound holds for the WHOLE call up to a factor 1.5 (round 5; VERDICT r4 #8:
            // the clock used to be read at the head of this loop only, a driver call under rocprofv3 --pmc took 5 s): the walk
            // ends when the time spent, plus the longest single step seen so far, plus what handing everything back is projected
            // to cost (every handle held x the cost of one hipMemRelease, measured on the first filler) would pass the bound; the
            // filler loop looks at the same clock after every handle.
            const double max_ms = g_search_ms;
            auto elapsed_ms = [&] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); };
            double step_max_ms = 0, release_ms_each = 0;
            auto held = [&] {
                size_t h = spacers.size() + unclear.size();
                for (auto& g : groups) h += g.pcs.size();
                return h;
            };
            auto out_of_time = [&] { return elapsed_ms() + step_max_ms + (double)held() * release_ms_each >= max_ms; };
            while ((usable(half) < k || groups.size() < 2) && walked <= search_budget_bytes && !out_of_time()) {
                const double step_t0 = elapsed_ms();
                struct StepClock {
                    double t0, *mx;
                    std::chrono::steady_clock::time_point base;
                    ~StepClock() {
                        const double now = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - base).count();
                        *mx = std::max(*mx, now - t0);
                    }
                } step_clock{step_t0, &step_max_ms, t0};
                Cand c;
                if (!c.make(device, exportable)) break;
                int home = -1;
                bool odd = false;
                for (size_t j = 0; j < groups.size() && home < 0; ++j) {
                    c.tbps = probe_tbps(ref_va(groups[j]), c.va);
                    Z.calib.observe(c.tbps);
                    const int cls = Z.calib.classify(c.tbps);
                    decided.push_back({c.tbps, cls});
                    if (cls == 0) {
                        home = (int)j;
                        I.probe_same_tbps = std::max(I.probe_same_tbps, c.tbps);
                    } else if (cls == 1) {
                        odd = true;
                        break;
                    } else {
                        I.probe_cross_tbps = I.probe_cross_tbps == 0 ? c.tbps : std::min(I.probe_cross_tbps, c.tbps);
                    }
                }
                if (!restarted && Z.calib.relative) {  // would the thresholds as they stand now have decided differently before?
                    bool differs = false;
                    for (auto& d : decided) differs = differs || Z.calib.classify(d.first) != d.second;
                    if (differs) {
                        if (debug) fprintf(stderr, "mg_obs_alloc: thresholds now %.2f / %.2f TB/s (relative): starting over\n", Z.calib.same_t, Z.calib.cross_t);
                        restarted = true;
                        c.drop();
                        for (auto& g : groups)
                            for (auto& q : g.pcs) q.drop();
                        groups.clear();
                        for (auto& q : unclear) q.drop();
                        unclear.clear();
                        decided.clear();
                        I.probe_same_tbps = I.probe_cross_tbps = 0;
                        continue;
                    }
                }
                if (debug)
                    fprintf(stderr, "mg_obs_alloc: %5.1f GiB walked, %s piece: last probe %.2f TB/s -> %s\n", walked / (double)GiB,
                            exportable ? "exportable" : "ordinary", c.tbps, odd ? "unclear" : home >= 0 ? "known group" : "new group");
                bool useful = false;
                if (odd) {
                    unclear.push_back(c);
                } else if (home < 0) {  // fast with every group so far: a piece of another zone
                    Group g;
                    g.id = Z.next_id++;
                    g.pcs.push_back(c);
                    groups.push_back(g);
                    useful = true;
                } else {
                    useful = groups[home].pcs.size() < half;
                    c.unmap();  // only references stay mapped
                    groups[home].pcs.push_back(c);
                }
                if (useful) continue;
                walked += PIECE;
                // good enough (a quarter from another zone)?  Six more pieces of alternating kind first, without fillers: a
                // second piece of the minority zone is often one request away (3 : 2 measures 0.785-0.79, 4 : 1 0.767-0.77)
                // (round 6: ... and while less than a quarter of the time bound is spent, the walk goes on WITH fillers until the split is
                // even: 4 : 1 was the outcome of two or three processes in eight on some boxes, 282 against 289-291 M env-steps/s)
                const bool good_enough = groups.size() >= 2 && usable(loose_cap) >= k;
                const bool early = elapsed_ms() < 0.25 * max_ms;
                if (good_enough && ++tries_after_good > 6 && !early) break;
                exportable = !exportable;
                if (!exportable && (!good_enough || early)) {  // every second time: step the ordinary allocator further -- or give up when that is not allowed.
                    // With piece-sized handles of the ordinary kind, 32 at a time (9.5 GiB): the driver serves requests of
                    // different sizes from different lists (big spacers -- 96 GiB, then 16 GiB at a time, the first version --
                    // moved the pieces' list on some boxes and not at all on others: 158 GiB walked, every piece from one zone),
                    // and a pristine VRAM hands out 100-130 GiB of one zone in a row.
                    const size_t before = spacers.size();
                    for (int f = 0; f < FILL_STEP && walked + PIECE <= search_budget_bytes && !out_of_time(); ++f) {
                        Piece sp;
                        const double c0 = elapsed_ms();
                        if (!create_piece(device, PIECE, &sp)) break;
                        if (release_ms_each == 0) {  // what one hipMemRelease costs here (0.01 ms plain, milliseconds under a profiler)
                            const double r0 = elapsed_ms();
                            release_piece(sp);
                            release_ms_each = std::max(1e-3, elapsed_ms() - r0);
                            if (!create_piece(device, PIECE, &sp)) break;
                        }
                        step_max_ms = std::max(step_max_ms, elapsed_ms() - c0);
                        spacers.push_back(sp);
                        walked += PIECE;
                    }
                    if (spacers.size() == before) break;
                }
            }
        } catch (...) {
            give_back();
            throw;
        }
        I.searched_bytes = walked;
        I.zones = (int)std::min<size_t>(groups.size(), 3);
        if (groups.size() < 2 || usable(k) < k) {  // one zone only (or out of memory half-way): an assembled buffer has nothing
            const bool one = groups.size() < 2;       // over an ordinary allocation (measured: slower)
            give_back();
            lock.unlock();
            if (debug) fprintf(stderr, "mg_obs_alloc: %s after %.1f GiB: plain allocation\n", one ? "one zone only" : "not enough pieces", walked / (double)GiB);
            plain(1);
            return 0;
        }
        // choose, under the strictest cap that still yields k pieces.  WHICH zone a piece of the range comes from follows the frame walk
        // (round 6): a raster launch writes at FRONTS one window = RASTER_GRID frames apart (its persistent workgroups stride over the
        // frames), and what is fast is a split of the concurrently written fronts over the zones (profiles/r02_zones.md).  A uint8 window is
        // 0.95 pieces, so alternating pieces alternate the fronts -- rounds 2-5 mapped the groups round-robin.  A window of the 16-bit
        // formats is 1.9 pieces: round-robin puts all five fronts into pieces of the same parity, i.e. into ONE zone at any time
        // (bfloat16 at 65,536 instances: 0.72 of peak on a balanced buffer against 0.83 on a lucky plain one).  Piece p lies in window
        // w(p) (by its centre) as that window's j-th piece and comes from zone (w + j) mod 2: neighbouring windows start in different
        // zones and a window's pieces alternate; for a window of about one piece that is the round-robin order.  (Three groups, when the
        // two largest cannot supply that: round-robin, which splits the fronts of every format as it is.)
        const size_t cap = usable(half) >= k ? half : (usable(loose_cap) >= k ? loose_cap : k);
        std::sort(groups.begin(), groups.end(), [](const Group& x, const Group& y) { return x.pcs.size() > y.pcs.size(); });
        const size_t lead = range_lead(k, bytes);  // the buffer sits in the MIDDLE of the range (below)
        const size_t window = (frame_bytes ? frame_bytes : (size_t)mg::FRAME_BYTES) * (size_t)PROBE_GRID;  // (= the rasters' persistent grid, RASTER_GRID)
        std::vector<size_t> want, count;
        plan_zones(k, lead, window, 2, want, count);
        bool alternating = true;
        for (size_t p = 0; p < k; ++p) alternating = alternating && want[p] == p % 2;
        std::vector<std::pair<Piece, int>> order;
        if (alternating) {
            // windows of about one piece (the uint8 frames every headline number is measured on): exactly the order of rounds 2-5 --
            // round-robin over ALL groups, largest first
            for (size_t round = 0; round < cap && order.size() < k; ++round)
                for (auto& g : groups)
                    if (round < g.pcs.size() && order.size() < k) {
                        g.pcs[round].unmap();
                        order.push_back({g.pcs[round].piece, g.id});
                        g.pcs[round].piece.h = nullptr;  // taken
                    }
        } else {
            // Two zones are enough (measured: a 3 : 2 split of the fronts is as fast as 2 : 2 : 1); a third group joins the pattern only
            // when the two largest cannot supply it (a search that found three groups of a third of the pieces each).
            size_t zones_used = 0;
            for (size_t G = 2; G <= std::min<size_t>(groups.size(), 3); ++G) {
                plan_zones(k, lead, window, G, want, count);
                zones_used = G;
                std::vector<size_t> need(count);
                std::sort(need.begin(), need.end(), [](size_t a, size_t b) { return a > b; });
                bool fits = true;
                for (size_t r = 0; r < G; ++r) fits = fits && need[r] <= std::min(groups[r].pcs.size(), cap);
                if (fits) break;  // (else: one more zone, or -- behind the last try -- the fullest other group stands in below)
            }
            // the zone the pattern asks for most often = the largest group (groups are sorted), and so on
            std::vector<size_t> by_count(zones_used), group_of(zones_used);
            for (size_t z = 0; z < zones_used; ++z) by_count[z] = z;
            std::stable_sort(by_count.begin(), by_count.end(), [&](size_t a, size_t b) { return count[a] > count[b]; });
            for (size_t r = 0; r < zones_used; ++r) group_of[by_count[r]] = r;
            std::vector<size_t> taken(groups.size(), 0);
            for (size_t p = 0; p < k; ++p) {
                size_t gi = group_of[want[p]];
                if (taken[gi] >= std::min(groups[gi].pcs.size(), cap)) {  // that group has no piece left (an uneven search result): the fullest other one
                    size_t best = groups.size(), room = 0;
                    for (size_t g2 = 0; g2 < groups.size(); ++g2) {
                        const size_t lim = std::min(groups[g2].pcs.size(), cap), left = lim - std::min(taken[g2], lim);
                        if (left > room) {
                            room = left;
                            best = g2;
                        }
                    }
                    if (best == groups.size()) throw std::runtime_error("mg_obs_alloc: internal error (pieces ran out while assembling)");
                    gi = best;
                }
                Cand& c = groups[gi].pcs[taken[gi]++];
                c.unmap();
                order.push_back({c.piece, groups[gi].id});
                c.piece.h = nullptr;  // taken
            }
        }
        for (auto& g : groups) {
            std::vector<Cand> rest;
            for (auto& c : g.pcs)
                if (c.piece.h) rest.push_back(c);
            g.pcs.swap(rest);
        }
        give_back();
        // one contiguous virtual range
        void* va = nullptr;
        if (va_reserve(&va, k * PIECE) != hipSuccess) {
            for (auto& o : order) pool_put(Z, o.second, o.first);
            throw std::runtime_error("mg_obs_alloc: hipMemAddressReserve failed");
        }
        Mapping m;
        m.va_bytes = k * PIECE;
        m.device = device;
        for (size_t i = 0; i < k; ++i) {
            map_at((char*)va + i * PIECE, order[i].first, device);
            m.pieces.push_back({i * PIECE, order[i].first});
            m.piece_class.push_back(order[i].second);
        }
        // belt and braces: what one kernel writes through the new range, the next one must read
        if (const unsigned long long bad = verify_range(va, k * PIECE, 0x5EEDu)) {
            for (auto& op : m.pieces) {
                (void)hipMemUnmap((char*)va + op.first, op.second.bytes);
                (void)hipMemRelease(op.second.h);
            }
            throw std::runtime_error("mg_obs_alloc: " + std::to_string(bad) + " of " + std::to_string(k * PIECE / 16) +
                                     " vectors of the assembled range did not read back (stale translations?)");
        }
        // The buffer sits in the MIDDLE of the range: what the pieces hold beyond `bytes` is split between the first and
        // the last piece, so that a buffer of 1.1 pieces is half in one zone and half in the other, not 10 : 1.
        m.base = va;
        void* user = (char*)va + lead;
        g_live[user] = m;
        *out = user;
        I.pieces = (int)k;
        I.piece_bytes = PIECE;
        I.search_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        if (debug) {
            std::string lay;
            for (auto& o : order) lay += (char)('A' + (o.second % 26));
            fprintf(stderr, "mg_obs_alloc: %zu pieces %s, %.1f GiB walked, %.0f ms, %zu spare pieces pooled\n", k, lay.c_str(),
                    walked / (double)GiB, I.search_ms, Z.pooled);
        }
        if (info) *info = I;
        return 0;
    } catch (const std::exception& e) {
        mg::set_error(e.what());
        return -1;
    }
}

// Test hook (tests/test_gpu_obs_alloc.py): what the allocator holds right now -- live buffers, pooled spare pieces (all devices),
// bytes of virtual address space reserved since the process started (never handed back, see va_free).
int mg_obs_debug_stats(size_t* live_buffers, size_t* pooled_pieces, size_t* reserved_va_bytes) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (live_buffers) *live_buffers = g_live.size();
    size_t pooled = 0;
    for (auto& z : g_zones) pooled += z.second.pooled;
    if (pooled_pieces) *pooled_pieces = pooled;
    if (reserved_va_bytes) *reserved_va_bytes = g_va_reserved;
    return 0;
}

int mg_obs_free(void* p) {
    if (!p) return 0;
    Mapping m;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        auto it = g_live.find(p);
        if (it == g_live.end()) {
            mg::set_error("mg_obs_free: not a pointer returned by mg_obs_alloc");
            return -1;
        }
        m = it->second;
        g_live.erase(it);
    }
    int prev = 0;
    (void)hipGetDevice(&prev);
    (void)hipSetDevice(m.device);
    struct Restore {
        int d;
        ~Restore() { (void)hipSetDevice(d); }
    } restore{prev};
    if (m.plain) return hipFree(p) == hipSuccess ? 0 : -1;
    (void)hipDeviceSynchronize();
    std::lock_guard<std::mutex> lk(g_mu);
    ZoneCache& Z = g_zones[m.device];
    for (size_t i = 0; i < m.pieces.size(); ++i) {
        (void)hipMemUnmap((char*)m.base + m.pieces[i].first, m.pieces[i].second.bytes);
        pool_put(Z, i < m.piece_class.size() ? m.piece_class[i] : -1, m.pieces[i].second);  // spare pieces of a known zone
    }
    va_free(m.base, m.va_bytes);
    return 0;
}

}  // extern "C"
