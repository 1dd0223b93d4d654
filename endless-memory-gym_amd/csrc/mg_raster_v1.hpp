// mg_raster_v1.hpp -- raster skeleton, generation 1 (namespace mg::v1): used by the template-dominated families
// (mortar, mystery).  The spotlight family uses generation 2 (mg_raster.hpp: constant-address-space descriptors,
// register-prefetched layers, darkening fused into the template pass).  Measured side by side on the same MI355X
// (profiles/r01c_raster_generations.md): generation 2 is 25-30 % faster for the spotlight frames and 8-11 % SLOWER
// for the mortar frames, whose kernel time is set by how the HBM write stream is paced, not by instruction count.
//
//
// raster_kernel<Composer>: PERSISTENT workgroups (256 lanes = 4 waves); each walks frames
// env = blockIdx.x, blockIdx.x + gridDim.x, ...  For one frame it
//   1. reads the family's small per-instance frame descriptor (workgroup-uniform -> scalar loads),
//   2. composes the 84x84x3 observation in 21,168 B of LDS with the helpers below, in the reference's blit order
//      (its _draw_surfaces(), e.g. memory_gym/mortar_mayhem_grid.py:92-102,367-370,
//      endless_searing_spotlights.py:464-479, endless_mystery_path.py:134-160),
//   3. streams it to HBM as 1,323 x 16-byte stores, lane-contiguous (1 KiB per wave instruction).
// The stores are fire-and-forget, so the workgroup composes its next frame while they drain; a
// one-frame-per-workgroup launch keeps the LDS hostage until the stores are acknowledged (measured 322 us vs 264 us
// per 65,536 frames; an interpreter over a generic display list measured 390-415 us: tools/microbench/raster_bench.hip).
// Roofline: HBM write bandwidth; algorithmic traffic per instance-step = 21,168 B written + sizeof(Desc) read.
//
// Helpers (all lanes of the workgroup call them together; callers place __syncthreads() between overlapping layers):
//   fill_template  copy a pre-rendered full frame (mortar arena variants, chessboards) from the L2-resident atlas
//   fill_clear     black frame
//   stamp          colour-keyed blit of a palette-indexed stamp (agent sprites, glyphs, cross, coin, exit), clipped
//   rect           filled rectangle with optional 1-px inset border (tiles, bars), clipped
// (The spotlight family's layers live in generation 2, mg_raster.hpp; the hole-mask words stay part of this
// skeleton's LDS request because its occupancy and pacing were tuned with them: profiles/r01c_raster_generations.md.)
#pragma once
#include <algorithm>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "../../include/memgym.h"
#include "mg_lab.hpp"
#include "mg_device.hpp"
#include "mg_stream_out.hpp"

namespace mg {
namespace v1 {

constexpr int MAX_STAMPS = 48;
constexpr int PALETTE_SIZE = 32;
constexpr int MASK_WORDS = 3;                 // 84 bits per column
constexpr int RASTER_GRID = 256 * 7 * 8;      // persistent workgroups (bench sweep at seven per CU: best of 2..37 rounds)
constexpr int RASTER_LDS = FRAME_BYTES + SCREEN * MASK_WORDS * 4;

struct StampInfo {
    uint32_t off;  // pixel offset into the stamp data, pixels stored [x][y] (column-major like the frame)
    uint16_t w, h;
};

struct AtlasTables {
    StampInfo stamps[MAX_STAMPS];
    uint32_t palette[PALETTE_SIZE];   // r | g<<8 | b<<16
    uint8_t border_of[PALETTE_SIZE];  // palette id of the 1-px border drawn around a bordered rect of this fill colour
};

// Everything the raster kernel samples (device pointers; small enough to sit in the scalar/L1/L2 caches).
struct RasterAtlas {
    const uint8_t* templates;   // [n_templates][84][84][3]
    const uint32_t* stamp_data; // r | g<<8 | b<<16 | 0xFF<<24 per opaque pixel, 0 = transparent (colour key)
    const AtlasTables* tables;
};

// Palette ids shared by all families
enum : uint8_t {
    C_KEY = 0, C_BODY = 1, C_HAND = 2, C_OUTLINE = 3, C_WHITE = 4, C_RED = 5, C_GREEN = 6, C_BLUE = 7, C_YELLOW = 8,
    C_ORANGE = 9, C_GREY50 = 10, C_GREY120 = 11, C_PURPLE = 12, C_ACT_ORANGE = 13, C_GREY210 = 14, C_BLACK = 15,
    C_EXIT_OPEN = 16, C_EXIT_CLOSED = 17, C_ICY = 18
};

struct RasterCtx {
    uint8_t* frame;         // LDS, [x][y][c]
    uint32_t* mask;         // LDS, [84][MASK_WORDS] hole mask scratch
    const AtlasTables* T;   // palette / stamp infos (global; indices are workgroup-uniform -> scalar loads)
    RasterAtlas A;
    int tid;
};

__device__ __forceinline__ void put_rgb(uint8_t* frame, int x, int y, uint32_t rgb) {
    uint8_t* p = frame + (x * SCREEN + y) * 3;
    p[0] = (uint8_t)rgb;
    p[1] = (uint8_t)(rgb >> 8);
    p[2] = (uint8_t)(rgb >> 16);
}

// all six 16-byte loads are issued before the first LDS write (one L2 round trip, not six)
__device__ __forceinline__ void fill_template(const RasterCtx& R, int t) {
    uint4* lds16 = reinterpret_cast<uint4*>(R.frame);
    const uint4* src = reinterpret_cast<const uint4*>(R.A.templates + (size_t)t * FRAME_BYTES);
    const int tid = R.tid;
    uint4 v0 = src[tid], v1 = src[tid + 256], v2 = src[tid + 512], v3 = src[tid + 768], v4 = src[tid + 1024];
    uint4 v5 = make_uint4(0, 0, 0, 0);
    if (tid < TAIL) v5 = src[tid + 1280];
    lds16[tid] = v0; lds16[tid + 256] = v1; lds16[tid + 512] = v2; lds16[tid + 768] = v3; lds16[tid + 1024] = v4;
    if (tid < TAIL) lds16[tid + 1280] = v5;
}

__device__ __forceinline__ void fill_clear(const RasterCtx& R) {
    uint4* lds16 = reinterpret_cast<uint4*>(R.frame);
    const uint4 z = make_uint4(0, 0, 0, 0);
    const int tid = R.tid;
    lds16[tid] = z; lds16[tid + 256] = z; lds16[tid + 512] = z; lds16[tid + 768] = z; lds16[tid + 1024] = z;
    if (tid < TAIL) lds16[tid + 1280] = z;
}

__device__ __forceinline__ void stamp(const RasterCtx& R, int id, int x, int y) {
    const StampInfo si = R.T->stamps[id];
    const uint32_t* sp = R.A.stamp_data + si.off;
    const int h = si.h, npx = si.w * h;
    for (int p = R.tid; p < npx; p += 256) {
        int px = p / h, py = p - px * h;
        uint32_t c = sp[p];
        int X = x + px, Y = y + py;
        if ((c >> 24) && (unsigned)X < (unsigned)SCREEN && (unsigned)Y < (unsigned)SCREEN) put_rgb(R.frame, X, Y, c);
    }
}

// The same blit split in two so that the stamp's pixels are requested from global memory EARLY (together with the
// template loads) and applied later: one memory round trip per frame instead of one per layer.  K*256 >= w*h.
template <int K>
struct StampRegs {
    uint32_t px[K];
    int h;
};
template <int K>
__device__ __forceinline__ StampRegs<K> stamp_fetch(const RasterCtx& R, int id) {
    StampRegs<K> s;
    const StampInfo si = R.T->stamps[id];
    const uint32_t* sp = R.A.stamp_data + si.off;
    s.h = si.h;
    const int npx = si.w * si.h;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        int p = R.tid + k * 256;
        s.px[k] = p < npx ? sp[p] : 0u;
    }
    return s;
}
template <int K>
__device__ __forceinline__ void stamp_apply(const RasterCtx& R, const StampRegs<K>& s, int x, int y) {
    const int h = s.h;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        int p = R.tid + k * 256;
        int px = p / h, py = p - px * h;
        int X = x + px, Y = y + py;
        uint32_t c = s.px[k];
        if ((c >> 24) && (unsigned)X < (unsigned)SCREEN && (unsigned)Y < (unsigned)SCREEN) put_rgb(R.frame, X, Y, c);
    }
}

__device__ __forceinline__ void rect(const RasterCtx& R, int x, int y, int w, int h, int fill, bool bordered) {
    const uint32_t cf = R.T->palette[fill], ce = R.T->palette[R.T->border_of[fill]];
    for (int p = R.tid; p < w * h; p += 256) {
        int px = p / h, py = p - px * h;
        int X = x + px, Y = y + py;
        bool on_edge = bordered && (px == 0 || py == 0 || px == w - 1 || py == h - 1);
        if ((unsigned)X < (unsigned)SCREEN && (unsigned)Y < (unsigned)SCREEN) put_rgb(R.frame, X, Y, on_edge ? ce : cf);
    }
}

#ifdef MG_LAB  // measurement builds only (tools/placement_lab.py): window k of the frame walk is displaced by g_lab_win_off[k] bytes
static __device__ long long g_lab_win_off[16];
#endif

template <class Composer, int FMT>
__global__ __launch_bounds__(256, 7) void raster_kernel(const typename Composer::Desc* __restrict__ descs, RasterAtlas A,
                                                     void* __restrict__ obs, int n, const uint8_t* __restrict__ only) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    RasterCtx R;
    R.frame = smem;
    R.mask = reinterpret_cast<uint32_t*>(smem + FRAME_BYTES);
    R.A = A;
    R.T = A.tables;
    R.tid = threadIdx.x;
    const int tid = threadIdx.x;
    for (int v = blockIdx.x; v < n; v += gridDim.x) {
        const int env = xcd_grouped_frame(v, n);
        const typename Composer::Desc* d = descs + env;  // workgroup-uniform
        if (Composer::skip(d) || (only && !only[env])) continue;  // `only`: per-frame filter (final observations)
        Composer::compose(d, R);
        __syncthreads();
#ifdef MG_LAB
        store_frame<FMT, false>(smem, static_cast<uint8_t*>(obs) + g_lab_win_off[(env / (int)gridDim.x) & 15], env, tid);
#else
        store_frame<FMT, false>(smem, obs, env, tid);  // plain stores (mg_stream_out.hpp)
#endif
        __syncthreads();  // the LDS frame is reused by the next iteration
    }
}

// The same for a launch that draws FEW of the n frames (a masked reset; round 6): the persistent loop above has every workgroup look at
// its n / grid descriptors one after the other -- ~0.7 us each, 28 us for the ~1,400 frames a gymnasium-convention step of 65,536 instances
// resets.  Here a workgroup owns SPARSE_CHUNK consecutive instances, reads their descriptors (and the caller's mask) with one vector load
// and draws the ones a ballot names.  (A kernel of its own: the dense launches are the measured ones and stay as they are.)
constexpr int SPARSE_CHUNK = 32;
template <class Composer, int FMT>
__global__ __launch_bounds__(256) void raster_sparse_kernel(const typename Composer::Desc* __restrict__ descs, RasterAtlas A,
                                                          void* __restrict__ obs, int n, const uint8_t* __restrict__ only) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    RasterCtx R;
    R.frame = smem;
    R.mask = reinterpret_cast<uint32_t*>(smem + FRAME_BYTES);
    R.A = A;
    R.T = A.tables;
    R.tid = threadIdx.x;
    const int tid = threadIdx.x;
    for (int base = blockIdx.x * SPARSE_CHUNK; base < n; base += gridDim.x * SPARSE_CHUNK) {
        const int e = base + (tid & (SPARSE_CHUNK - 1));  // (every wave looks at the same SPARSE_CHUNK instances: the same list in all four)
        const bool want = e < n && !Composer::skip(descs + e) && (!only || only[e]);
        uint32_t m = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)__ballot(want));
        while (m) {
            const int env = base + __builtin_ctz(m);
            m &= m - 1;
            Composer::compose(descs + env, R);
            __syncthreads();
            store_frame<FMT, false>(smem, obs, env, tid);
            __syncthreads();  // the LDS frame is reused by the next iteration
        }
    }
}

// Workgroups of a raster launch over n frames (MEMGYM_RASTER_GRID overrides: tuning experiments)
inline int raster_grid(int n) {
    static const int forced = [] {
        const char* e = lab_env("MEMGYM_RASTER_GRID");
        return e ? atoi(e) : 0;
    }();
    (void)n;
    return forced > 0 ? forced : RASTER_GRID;
}

template <class Composer>
inline void launch_raster(const typename Composer::Desc* descs, const RasterAtlas& atlas, void* obs, int fmt, int n, hipStream_t s,
                          const uint8_t* only = nullptr) {
    const int tuned = raster_grid(n);
    const int grid = n < tuned ? n : tuned;
    // MEMGYM_RASTER_LDS inflates the LDS request = fewer resident workgroups per CU (tuning only).  Seven per CU (what
    // fits) is the optimum once the observation buffer sits in a fast allocation (profiles/r01l_placement.md):
    // MortarMayhem-Grid 223.4-224.5 us at 7, 229.4 at 6, 243-244 at 5; in a slow allocation 6 was 1-3 % ahead of 7.
    static const int lds = [] {
        const char* e = lab_env("MEMGYM_RASTER_LDS");
        return e && atoi(e) >= RASTER_LDS ? atoi(e) : RASTER_LDS;
    }();
    if (fmt == MG_OBS_F32_CYX)
        hipLaunchKernelGGL((raster_kernel<Composer, MG_OBS_F32_CYX>), dim3(grid), dim3(256), lds, s, descs, atlas, obs, n, only);
    else if (fmt == MG_OBS_BF16_CYX)
        hipLaunchKernelGGL((raster_kernel<Composer, MG_OBS_BF16_CYX>), dim3(grid), dim3(256), lds, s, descs, atlas, obs, n, only);
    else if (fmt == MG_OBS_F16_CYX)
        hipLaunchKernelGGL((raster_kernel<Composer, MG_OBS_F16_CYX>), dim3(grid), dim3(256), lds, s, descs, atlas, obs, n, only);
    else
        hipLaunchKernelGGL((raster_kernel<Composer, MG_OBS_U8_XYC>), dim3(grid), dim3(256), lds, s, descs, atlas, obs, n, only);
}

template <class Composer>
inline void launch_raster_sparse(const typename Composer::Desc* descs, const RasterAtlas& atlas, void* obs, int fmt, int n, hipStream_t s,
                                 const uint8_t* only) {
    const int grid = std::min((n + SPARSE_CHUNK - 1) / SPARSE_CHUNK, 8192);
    if (fmt == MG_OBS_F32_CYX)
        hipLaunchKernelGGL((raster_sparse_kernel<Composer, MG_OBS_F32_CYX>), dim3(grid), dim3(256), RASTER_LDS, s, descs, atlas, obs, n, only);
    else if (fmt == MG_OBS_BF16_CYX)
        hipLaunchKernelGGL((raster_sparse_kernel<Composer, MG_OBS_BF16_CYX>), dim3(grid), dim3(256), RASTER_LDS, s, descs, atlas, obs, n, only);
    else if (fmt == MG_OBS_F16_CYX)
        hipLaunchKernelGGL((raster_sparse_kernel<Composer, MG_OBS_F16_CYX>), dim3(grid), dim3(256), RASTER_LDS, s, descs, atlas, obs, n, only);
    else
        hipLaunchKernelGGL((raster_sparse_kernel<Composer, MG_OBS_U8_XYC>), dim3(grid), dim3(256), RASTER_LDS, s, descs, atlas, obs, n, only);
}

}  // namespace v1
}  // namespace mg
