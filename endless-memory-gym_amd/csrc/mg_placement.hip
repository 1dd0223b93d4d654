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
#include <map>
#include <mutex>
#include <vector>

#include "mg_family.hpp"

namespace {

constexpr size_t MiB = 1ull << 20, GiB = 1ull << 30;
constexpr size_t PIECE = 304 * MiB;         // = one window of the probe: 14,336 frames of 21,168 B (64-MiB pieces measured the
                                            // same, 16 MiB +6 %, 2 MiB +12 %: profiles/r02_zones.md)
constexpr size_t SPACER = 16 * GiB;
constexpr size_t ZONE = 96 * GiB;            // a third of the 288 GB
constexpr int PROBE_GRID = 14336;
static const double CROSS_ZONE_TBPS = getenv("MEMGYM_OBS_CROSS_TBPS") ? atof(getenv("MEMGYM_OBS_CROSS_TBPS")) : 5.85;  // different zones 6.1-6.5 ...
constexpr double SAME_ZONE_TBPS = 5.35;     // ... same zone 4.9-5.3 (profiles/r02_zones.md); in between: a piece that straddles

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

// Virtual ranges are NEVER handed back to the runtime.  Measured on ROCm 7.2 / MI355X (tools/vmm_stress.py, profiles/
// r02_zones.md): when a range freed with hipMemAddressFree is reserved again and mapped onto other physical memory, stores
// through it can land in the OLD physical pages (5 of 80 allocate-fill-verify-free cycles read back wrong, up to 95 % of a
// 1.4-GB buffer; 0 of 80 when no range is ever reused) -- stale translations, which would also let the probe below
// scribble over memory that belongs to somebody else by now.  A reservation costs address space only (a few GiB of the
// 128-TiB space per call), so unmapped ranges simply stay reserved.  MEMGYM_OBS_REUSE_VA=1 restores the frees (experiments).
inline void va_free(void* va, size_t bytes) {
    static const bool reuse = getenv("MEMGYM_OBS_REUSE_VA") && atoi(getenv("MEMGYM_OBS_REUSE_VA")) != 0;
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
    std::vector<std::pair<size_t, Piece>> pieces;  // (offset, piece)
    std::vector<int> piece_class;                  // zone class of every piece (index into the device's references)
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
bool create_piece(int device, size_t bytes, Piece* out, bool exportable = false) {
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

void map_at(void* va, const Piece& p, int device) {
    MG_HIP(hipMemMap(va, p.bytes, 0, p.h, 0));
    hipMemAccessDesc acc = {};
    acc.location.type = hipMemLocationTypeDevice;
    acc.location.id = device;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    MG_HIP(hipMemSetAccess(va, p.bytes, &acc, 1));
}

// a candidate piece, mapped on its own for the probe
struct Cand {
    Piece piece;
    void* va = nullptr;
    double tbps = 0;  // two-window probe against the reference piece
    bool make(int device, bool exportable = false) {
        if (!create_piece(device, PIECE, &piece, exportable)) return false;
        if (hipMemAddressReserve(&va, PIECE, 2 * MiB, nullptr, 0) != hipSuccess) {
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

// Per device, for the life of the process: one dedicated reference piece per zone class found so far (mapped, never part
// of a buffer: the probe writes into it) and a bounded pool of spare pieces whose class is known -- pieces a search
// classified but did not need, and the pieces of buffers that were freed.  The second buffer of a process is assembled
// from the pool or with a handful of probes instead of another walk.
struct ZoneCache {
    std::vector<Cand> refs;                 // class j = "slow together with refs[j]"
    std::vector<std::vector<Piece>> pool;   // spare pieces by class
    size_t pooled = 0;
};
constexpr size_t POOL_CAP = 10;             // pieces (3 GiB) kept at most
std::map<int, ZoneCache> g_zones;

void pool_put(ZoneCache& Z, int cls, Piece p) {
    if (cls >= 0 && cls < (int)Z.refs.size() && Z.pooled < POOL_CAP) {
        if (Z.pool.size() < Z.refs.size()) Z.pool.resize(Z.refs.size());
        Z.pool[cls].push_back(p);
        Z.pooled++;
    } else {
        release_piece(p);
    }
}

extern "C" {

int mg_obs_alloc(int device, size_t bytes, size_t search_budget_bytes, void** out, mg_obs_alloc_info* info) {
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
        const bool debug = getenv("MEMGYM_OBS_DEBUG") != nullptr;
        *out = nullptr;
        size_t free_b = 0, total_b = 0;
        MG_HIP(hipMemGetInfo(&free_b, &total_b));
        if (search_budget_bytes == MG_OBS_SEARCH_DEFAULT) search_budget_bytes = std::min<size_t>(free_b / 20 * 11, 160 * GiB);
        const size_t k = (bytes + PIECE - 1) / PIECE;
        auto plain = [&](int zones) {
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
        // chosen[j] = pieces of class j for this buffer; no class may contribute more than half (rounded up) -- relaxed to
        // "at least a quarter from other zones" once a walk would be needed (one in five measured within 2 % of an even split)
        const size_t half = (k + 1) / 2, loose_cap = k - std::max<size_t>(1, k / 4);
        std::vector<std::vector<Piece>> chosen(3);
        auto usable = [&](size_t cap) {
            size_t u = 0;
            for (auto& v : chosen) u += std::min(v.size(), cap);
            return u;
        };
        auto classes_used = [&] {
            int c = 0;
            for (auto& v : chosen) c += !v.empty();
            return c;
        };
        // 1. from the pool
        for (size_t j = 0; j < Z.pool.size() && j < 3; ++j)
            while (!Z.pool[j].empty() && chosen[j].size() < half && usable(half) < k) {
                chosen[j].push_back(Z.pool[j].back());
                Z.pool[j].pop_back();
                Z.pooled--;
            }
        // 2. new pieces, classified against the references (a piece that is fast with all of them founds a new class and
        //    becomes its dedicated reference); pieces of no use and spacers keep the driver's allocator moving
        std::vector<Cand> unclear;
        std::vector<std::pair<int, Cand>> surplus;
        std::vector<Piece> spacers;
        size_t walked = 0;
        bool exportable = false;
        while ((usable(half) < k || classes_used() < 2) && walked <= search_budget_bytes) {
            Cand c;
            if (!c.make(device, exportable)) break;
            int home = -1;
            bool odd = false;
            for (size_t j = 0; j < Z.refs.size() && home < 0; ++j) {
                c.tbps = probe_tbps(Z.refs[j].va, c.va);
                if (c.tbps < SAME_ZONE_TBPS) {
                    home = (int)j;
                    I.probe_same_tbps = std::max(I.probe_same_tbps, c.tbps);
                } else if (c.tbps <= CROSS_ZONE_TBPS) {
                    odd = true;
                    break;
                } else {
                    I.probe_cross_tbps = I.probe_cross_tbps == 0 ? c.tbps : std::min(I.probe_cross_tbps, c.tbps);
                }
            }
            if (debug)
                fprintf(stderr, "mg_obs_alloc: %5.1f GiB walked, %s piece: last probe %.2f TB/s -> %s\n", walked / (double)GiB,
                        exportable ? "exportable" : "ordinary", c.tbps,
                        odd ? "unclear" : home >= 0 ? "known class" : Z.refs.size() < 3 ? "new class" : "fast with every reference");
            if (!odd && home < 0 && Z.refs.size() < 3) {  // a new zone: this piece stays mapped as its reference
                Z.refs.push_back(c);
                Z.pool.resize(Z.refs.size());
                continue;
            }
            if (!odd && home < 0) {  // fast with all three references: as good a partner as any -- counted with the emptiest class
                home = 0;
                for (int j = 1; j < 3; ++j)
                    if (chosen[j].size() < chosen[home].size()) home = j;
            }
            if (!odd && home >= 0 && chosen[home].size() < half) {
                c.unmap();
                chosen[home].push_back(c.piece);
                continue;
            }
            // of no use right now (unclear, or its class is full): keep it out of the way and move on
            if (odd || home < 0) unclear.push_back(c);
            else surplus.push_back({home, c});
            walked += PIECE;
            if (classes_used() >= 2 && usable(loose_cap) >= k) break;
            exportable = !exportable;
            if (!exportable) {  // every second time: step the ordinary allocator further -- or give up when that is not allowed.
                // The first step is one zone long (a pristine VRAM hands out ~100 GiB of one zone in a row); then 16 GiB.
                Piece sp;
                size_t step = spacers.empty() ? ZONE : SPACER;
                if (walked + step > search_budget_bytes) step = SPACER;
                if (walked + step > search_budget_bytes || !create_piece(device, step, &sp)) break;
                spacers.push_back(sp);
                walked += step;
            }
        }
        I.searched_bytes = walked;
        for (auto& sp : spacers) release_piece(sp);
        for (auto& c : unclear) c.drop();
        auto surplus_to_pool = [&] {
            for (auto& sc : surplus) {
                sc.second.unmap();
                pool_put(Z, sc.first, sc.second.piece);
            }
            surplus.clear();
        };
        I.zones = classes_used();
        if (I.zones < 2) {  // one zone only: an assembled buffer has nothing over an ordinary allocation (measured: slower)
            for (size_t j = 0; j < chosen.size(); ++j)
                for (auto& p : chosen[j]) pool_put(Z, (int)j, p);
            surplus_to_pool();
            lock.unlock();
            if (debug) fprintf(stderr, "mg_obs_alloc: one zone only after %.1f GiB: plain allocation\n", walked / (double)GiB);
            plain(1);
            return 0;
        }
        // the rest (if the strict balance was not reached): surplus pieces of any class, then fresh ones of unknown class
        std::vector<Piece> unknown;
        while (usable(k) + unknown.size() < k) {
            if (!surplus.empty()) {
                surplus.back().second.unmap();
                chosen[surplus.back().first].push_back(surplus.back().second.piece);
                surplus.pop_back();
                continue;
            }
            Piece p;
            if (!create_piece(device, PIECE, &p)) {
                for (size_t j = 0; j < chosen.size(); ++j)
                    for (auto& q : chosen[j]) pool_put(Z, (int)j, q);
                for (auto& q : unknown) release_piece(q);
                throw std::runtime_error("mg_obs_alloc: out of device memory");
            }
            unknown.push_back(p);
        }
        surplus_to_pool();
        // order: round-robin over the classes, largest first
        std::vector<int> by_size = {0, 1, 2};
        std::sort(by_size.begin(), by_size.end(), [&](int x, int y) { return chosen[x].size() > chosen[y].size(); });
        std::vector<std::pair<Piece, int>> order;
        for (size_t round = 0; order.size() + unknown.size() < k; ++round) {
            bool any = false;
            for (int j : by_size)
                if (round < chosen[j].size() && order.size() + unknown.size() < k) {
                    order.push_back({chosen[j][round], j});
                    any = true;
                }
            if (!any) break;
        }
        for (auto& q : unknown) order.push_back({q, -1});
        for (int j : by_size)  // pieces beyond k (cannot happen with the caps above; kept for safety)
            for (size_t r = 0; r < chosen[j].size(); ++r) {
                bool used = false;
                for (auto& o : order) used = used || o.first.h == chosen[j][r].h;
                if (!used) pool_put(Z, j, chosen[j][r]);
            }
        // one contiguous virtual range
        void* va = nullptr;
        if (hipMemAddressReserve(&va, k * PIECE, 2 * MiB, nullptr, 0) != hipSuccess) {
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
        g_live[va] = m;
        *out = va;
        I.pieces = (int)k;
        I.piece_bytes = PIECE;
        I.search_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        if (debug)
            fprintf(stderr, "mg_obs_alloc: %zu pieces, zones %d, %.1f GiB walked, %.0f ms, %zu spare pieces pooled\n", k, I.zones,
                    walked / (double)GiB, I.search_ms, Z.pooled);
        if (info) *info = I;
        return 0;
    } catch (const std::exception& e) {
        mg::set_error(e.what());
        return -1;
    }
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
        (void)hipMemUnmap((char*)p + m.pieces[i].first, m.pieces[i].second.bytes);
        pool_put(Z, i < m.piece_class.size() ? m.piece_class[i] : -1, m.pieces[i].second);  // spare pieces of a known zone
    }
    va_free(p, m.va_bytes);
    return 0;
}

}  // extern "C"
