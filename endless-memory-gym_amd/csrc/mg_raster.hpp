// mg_raster.hpp -- raster skeleton, generation 2 (gfx950): the layered composer used by the spotlight family.
// (The template-dominated families -- mortar, mystery -- stay on generation 1, mg_raster_v1.hpp; see the note there
// and profiles/r01c_raster_generations.md for the side-by-side measurements behind that split.)
//
// raster_kernel<Composer, Format>: PERSISTENT workgroups (256 lanes = 4 waves); each walks frames
// env = blockIdx.x, blockIdx.x + gridDim.x, ...  For one frame it
//   1. reads the family's per-instance frame descriptor and the atlas tables through the CONSTANT address space
//      (workgroup-uniform -> scalar loads that do not queue behind the observation stores),
//   2. prefetch(): issues every global read of the frame at once (template, stamp pixels, disc spans) into registers,
//   3. compose(): builds the 84x84x3 observation in 21,168 B of LDS in the reference's blit order
//      (endless_searing_spotlights.py:464-479, searing_spotlights.py:524-545) touching LDS only; the spotlight layer
//      is not a pass of its own: the hole mask is built first and every layer below it is darkened while written,
//   4. streams the frame to HBM as 1,323 x 16-byte stores, lane-contiguous (1 KiB per wave instruction); plain or non-temporal
//      by launch size (raster_nt()).
// Roofline: HBM write bandwidth; algorithmic traffic per instance-step = 21,168 B written + sizeof(Desc) read.
//
// Helpers (all lanes of the workgroup call them together; callers place __syncthreads() between overlapping layers):
//   templ_fetch / templ_apply_dark   background template, darkened outside the holes on the way into LDS
//   stamp_fetch / stamp_apply_lit    colour-keyed RGBA stamp (agent sprites, coin, exit), clipped, darkened per pixel
//   zero_mask / hole_fetch8 / hole_apply8 / hole_mask   the lit discs as an 84x84 bit mask
//   darken2 / darken4                SDL's surface-alpha rule d - floor(d*alpha/255), two bytes per multiply
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

// Pointers to data that no kernel of this library writes while the raster kernel runs (frame descriptors written by
// the preceding logic kernel, atlas tables uploaded at creation) are viewed through the CONSTANT address space: the
// compiler then reads workgroup-uniform fields with scalar loads (s_load, counted by lgkmcnt).  As plain global loads
// they would be vector loads counted by vmcnt together with the observation stores, and every descriptor access
// would wait for the previous frame's stores to drain.
#define MG_CONST_AS __attribute__((address_space(4)))
template <class T>
using cptr = const T MG_CONST_AS*;
template <class T>
__device__ __forceinline__ cptr<T> as_const(const T* p) {
    return (cptr<T>)p;  // NOLINT: address-space cast
}

constexpr int DISC_RMAX = 64;
constexpr int MAX_STAMPS = 48;
constexpr int PALETTE_SIZE = 32;
constexpr int MASK_WORDS = 3;                 // 84 bits per column
constexpr int RASTER_GRID = 256 * 7 * 8;      // workgroups of a launch over many frames (65,536 and more)
constexpr int RASTER_GRID_SMALL = 256 * 38;   // ... over 24,576 frames or fewer: see raster_grid()
constexpr int RASTER_PLAIN_MAX = 16384;       // launches up to this many frames use plain stores, larger ones non-temporal: raster_nt()
constexpr int RASTER_LDS = FRAME_BYTES + SCREEN * MASK_WORDS * 4;
constexpr int RASTER_LDS_REQUEST = 25 * 1024;  // non-temporal uint8 stream: six workgroups per CU, see launch_raster

struct StampInfo {
    uint32_t off;  // pixel offset into the stamp data, pixels stored [x][y] (column-major like the frame)
    // 32-bit bit-fields / arrays only in anything the raster kernel reads with scalar loads (no sub-dword s_load)
    uint32_t w : 16, h : 16;
    uint32_t sh : 16, pad : 16;  // column stride = 1 << sh >= h (padding pixels are transparent); mono stamps: colour
};

struct AtlasTables {
    StampInfo stamps[MAX_STAMPS];
    uint32_t palette[PALETTE_SIZE];   // r | g<<8 | b<<16
    uint32_t border_of[PALETTE_SIZE];  // palette id of the 1-px border drawn around a bordered rect of this fill colour
};

// Everything the raster kernel samples (device pointers; small enough to sit in the scalar/L1/L2 caches).
struct RasterAtlas {
    const uint8_t* templates;   // [n_templates][84][84][3]
    const uint32_t* stamp_data; // r | g<<8 | b<<16 | 0xFF<<24 per opaque pixel, 0 = transparent (colour key)
    const int8_t* disc_span;    // [DISC_RMAX+1][2*DISC_RMAX][2]: per column i of a radius-r disc, (lo, hi) y offsets; lo > hi = empty
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
    cptr<AtlasTables> T;    // palette / stamp infos (indices are workgroup-uniform -> scalar loads)
    RasterAtlas A;
    int tid;
};

__device__ __forceinline__ void put_rgb(uint8_t* frame, int x, int y, uint32_t rgb) {
    uint8_t* p = frame + (x * SCREEN + y) * 3;
    p[0] = (uint8_t)rgb;
    p[1] = (uint8_t)(rgb >> 8);
    p[2] = (uint8_t)(rgb >> 16);
}

// d - floor(d * a / 255) for the four bytes of a dword (SDL ALPHA_BLEND_RGB towards black), two bytes at a time in
// 16-bit lanes: t = d*a <= 65025 and t + 1 + (t >> 8) <= 65280 never carry into the neighbouring lane, and
// (t + 1 + (t >> 8)) >> 8 == t / 255 for every t <= 65535.
__device__ __forceinline__ uint32_t darken2(uint32_t x, uint32_t a) {  // x = 0x00dd00dd
    uint32_t t = x * a;
    uint32_t q = ((t + 0x00010001u + ((t >> 8) & 0x00FF00FFu)) >> 8) & 0x00FF00FFu;
    return x - q;
}
__device__ __forceinline__ uint32_t darken4(uint32_t v, uint32_t a) {
    return darken2(v & 0x00FF00FFu, a) | (darken2((v >> 8) & 0x00FF00FFu, a) << 8);
}

// The same blit split in two so that the stamp's pixels are requested from global memory EARLY (together with the
// template loads) and applied later: one memory round trip per frame instead of one per layer.  K*256 >= w*h.
template <int K>
struct StampRegs {
    uint32_t px[K];
    const uint32_t* src;  // pixels beyond K*256 (option-scaled stamps) are read at apply time
    int sh, npx;
};
template <int K>
__device__ __forceinline__ void stamp_fetch(const RasterCtx& R, int id, StampRegs<K>& s) {
    s.src = R.A.stamp_data + R.T->stamps[id].off;
    s.sh = R.T->stamps[id].sh;
    s.npx = (int)R.T->stamps[id].w << s.sh;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        int p = R.tid + k * 256;
        s.px[k] = p < s.npx ? s.src[p] : 0u;
    }
}
template <int K>
__device__ __forceinline__ void stamp_none(StampRegs<K>& s) {
#pragma unroll
    for (int k = 0; k < K; ++k) s.px[k] = 0u;
    s.src = nullptr;
    s.sh = 0;
    s.npx = 0;
}
__device__ __forceinline__ bool never_skip(int, int) { return false; }
// alpha != 0: the pixel is blended towards black unless its hole-mask bit is set (the stamp is below the spotlight
// layer); skip(X, Y): pixels owned by a later layer that is written in the same phase.
template <int K, class Skip>
__device__ __forceinline__ void stamp_apply_lit(const RasterCtx& R, const StampRegs<K>& s, int x, int y, uint32_t alpha, Skip skip) {
    const int sh = s.sh, ym = (1 << sh) - 1;
    auto one = [&](int p, uint32_t c) {
        int px = p >> sh, py = p & ym;
        int X = x + px, Y = y + py;
        if ((c >> 24) && (unsigned)X < (unsigned)SCREEN && (unsigned)Y < (unsigned)SCREEN && !skip(X, Y)) {
            if (alpha) {
                uint32_t lit = (R.mask[X * MASK_WORDS + (Y >> 5)] >> (Y & 31)) & 1u;
                if (!lit) c = alpha < 255u ? darken4(c, alpha) : 0u;
            }
            put_rgb(R.frame, X, Y, c);
        }
    };
#pragma unroll
    for (int k = 0; k < K; ++k) one(R.tid + k * 256, s.px[k]);
    for (int p = R.tid + K * 256; p < s.npx; p += 256) one(p, s.src[p]);
}

__device__ __forceinline__ uint32_t pack_hole(int x, int y, int r) {
    return (uint32_t)(x + 128) | ((uint32_t)(y + 128) << 9) | ((uint32_t)r << 18);
}
__device__ __forceinline__ int hole_radius(uint32_t hv) { return (int)((hv >> 18) & 0x1FFFu); }  // bit 31: the disc has a white border

// holes[i] = pack_hole(x, y, r): filled discs (pygame's even-diameter midpoint disc) that stay lit.
// hole_mask: union of the discs as an 84x84 bit mask in LDS; tasks = (hole, column), 4 holes x 64 columns per round,
// so the span-table loads of all holes are in flight together.  The mask must have been zeroed (and synchronised).
__device__ __forceinline__ void zero_mask(const RasterCtx& R) {
    if (R.tid < SCREEN * MASK_WORDS) R.mask[R.tid] = 0u;
}
// (hole_at(h): the h-th packed hole of the frame's descriptor, h workgroup- or wave-uniform; the descriptor may sit behind scalar
// loads or in a register: mg_spot.hip SpotView)
template <class HoleAt>
__device__ __forceinline__ void hole_mask(const RasterCtx& R, HoleAt hole_at, int nholes) {
    const int sub = __builtin_amdgcn_readfirstlane(R.tid >> 6), col0 = R.tid & 63;
    for (int base = 0; base < nholes; base += 4) {
        int hI = base + sub;
        if (hI >= nholes) continue;
        const uint32_t hv = hole_at(hI);
        const int hx = (int)(hv & 511u) - 128, hy = (int)((hv >> 9) & 511u) - 128, r = hole_radius(hv);
        for (int col = col0; col < 2 * r; col += 64) {
            int X = hx - r + col;
            int lo = R.A.disc_span[(r * 2 * DISC_RMAX + col) * 2], hi = R.A.disc_span[(r * 2 * DISC_RMAX + col) * 2 + 1];
            int y0 = hy + lo, y1 = hy + hi;
            y0 = y0 < 0 ? 0 : y0;
            y1 = y1 > SCREEN - 1 ? SCREEN - 1 : y1;
            if ((unsigned)X < (unsigned)SCREEN && y0 <= y1) {
                for (int wI = 0; wI < MASK_WORDS; ++wI) {
                    int a0 = y0 - 32 * wI, a1 = y1 - 32 * wI;
                    a0 = a0 < 0 ? 0 : a0;
                    a1 = a1 > 31 ? 31 : a1;
                    if (a0 <= a1) {
                        uint32_t bits = (a1 - a0 == 31) ? 0xFFFFFFFFu : (((1u << (a1 - a0 + 1)) - 1u) << a0);
                        atomicOr(&R.mask[X * MASK_WORDS + wI], bits);
                    }
                }
            }
        }
    }
}
// ---- fused form of the spotlight layer (<= 16 holes of radius <= 16, the reference's 7..13) --------------------
// 8 holes x 32 columns per round, two rounds; a column's lit span is at most 32 rows -> at most two mask words.
struct HoleRegs8 {
    uint32_t hole[2];  // the packed hole of this lane's (round, column) task, 0 = no task
    uint32_t span[2];  // its span-table entry: lo | hi << 8 (int8 y offsets); decoded in hole_apply8, so that hole_fetch8
                       // only ISSUES loads and the frame's whole prefetch is one memory round trip
};
template <class HoleAt>
__device__ __forceinline__ bool holes_small(HoleAt hole_at, int nholes) {
    bool ok = nholes <= 16;
    for (int h = 0; h < nholes; ++h) ok = ok && hole_radius(hole_at(h)) <= 16;
    return ok;
}
// A wave handles two holes per round (its lower and upper 32 lanes): their packed words are read with WAVE-UNIFORM indices --
// scalar loads like the rest of the descriptor -- and picked per lane, so that the span-table read is the only vector-memory
// hop of the spotlight layer and leaves together with the frame's other loads (as a lane-indexed load the word cost a
// round trip of its own in front of the span read).
template <class HoleAt>
__device__ __forceinline__ void hole_fetch8(const RasterCtx& R, HoleAt hole_at, int nholes, HoleRegs8& H) {
    const int sub = R.tid >> 5, col = R.tid & 31;
    const int wave2 = __builtin_amdgcn_readfirstlane(R.tid >> 6) * 2;
    const uint16_t* spans = reinterpret_cast<const uint16_t*>(R.A.disc_span);
#pragma unroll
    for (int rnd = 0; rnd < 2; ++rnd) {
        const int hI = rnd * 8 + sub, h0 = rnd * 8 + wave2;
        const uint32_t lo_word = h0 < nholes ? hole_at(h0) : 0u, hi_word = h0 + 1 < nholes ? hole_at(h0 + 1) : 0u;
        const uint32_t hv = (R.tid & 32) ? hi_word : lo_word;
        const int r = hole_radius(hv);
        const bool task = hI < nholes && col < 2 * r;
        H.hole[rnd] = task ? hv : 0u;  // a task's hole has r >= 1, so its packed word is never 0
        H.span[rnd] = task ? (uint32_t)spans[r * 2 * DISC_RMAX + col] : 0u;
    }
}
__device__ __forceinline__ void hole_apply8(const RasterCtx& R, const HoleRegs8& H) {
    const int col = R.tid & 31;
#pragma unroll
    for (int rnd = 0; rnd < 2; ++rnd) {
        const uint32_t hv = H.hole[rnd];
        if (!hv) continue;
        const int hx = (int)(hv & 511u) - 128, hy = (int)((hv >> 9) & 511u) - 128, r = hole_radius(hv);
        const int lo = (int8_t)(H.span[rnd] & 0xFFu), hi = (int8_t)(H.span[rnd] >> 8);
        const int X = hx - r + col;
        int y0 = hy + lo, y1 = hy + hi;
        y0 = y0 < 0 ? 0 : y0;
        y1 = y1 > SCREEN - 1 ? SCREEN - 1 : y1;
        if ((unsigned)X >= (unsigned)SCREEN || y0 > y1) continue;
        const int w0 = y0 >> 5, w1 = y1 >> 5;
        const uint32_t lo_bits = 0xFFFFFFFFu << (y0 & 31), hi_bits = 0xFFFFFFFFu >> (31 - (y1 & 31));
        if (w0 == w1) {
            atomicOr(&R.mask[X * MASK_WORDS + w0], lo_bits & hi_bits);
        } else {
            atomicOr(&R.mask[X * MASK_WORDS + w0], lo_bits);
            atomicOr(&R.mask[X * MASK_WORDS + w1], hi_bits);
        }
    }
}

// Background template with the darkening applied on the way into LDS.  One task per lane: 28 pixels of one column
// (a column is 252 B = 3 x 84 B, so lane t owns bytes [84t, 84t + 84) of the frame; 252 of the 256 lanes work).  The
// 28 mask bits of the task come from at most two mask words; pixels are handled in groups of 4 = 3 dwords.
struct TemplRegs {
    uint32_t v[21];
};
__device__ __forceinline__ void templ_fetch(const RasterCtx& R, int t, TemplRegs& T) {
    const int lane = R.tid < 252 ? R.tid : 251;
    const uint32_t* src = reinterpret_cast<const uint32_t*>(R.A.templates + (size_t)t * FRAME_BYTES) + 21 * lane;
#pragma unroll
    for (int i = 0; i < 21; ++i) T.v[i] = src[i];
}
__device__ __forceinline__ void templ_apply_dark(const RasterCtx& R, const TemplRegs& T, uint32_t alpha) {
    if (R.tid >= 252) return;
    uint32_t* dst = reinterpret_cast<uint32_t*>(R.frame) + 21 * R.tid;
    uint32_t lit = 0x0FFFFFFFu;
    if (alpha) {
        const int X = R.tid / 3, y0 = (R.tid - 3 * X) * 28;
        const uint32_t* mw = R.mask + X * MASK_WORDS;
        const int w = y0 >> 5;
        const uint64_t two = (uint64_t)mw[w] | ((uint64_t)(w + 1 < MASK_WORDS ? mw[w + 1] : 0u) << 32);
        lit = (uint32_t)(two >> (y0 & 31)) & 0x0FFFFFFFu;
    }
#pragma unroll
    for (int g = 0; g < 7; ++g) {
        uint32_t v0 = T.v[3 * g], v1 = T.v[3 * g + 1], v2 = T.v[3 * g + 2];
        if (alpha) {
            // per-pixel 0 / ~0 from the four mask bits, then byte masks of the three dwords (RGBR GBRG BRGB)
            const uint32_t s0 = 0u - ((lit >> (4 * g)) & 1u), s1 = 0u - ((lit >> (4 * g + 1)) & 1u);
            const uint32_t s2 = 0u - ((lit >> (4 * g + 2)) & 1u), s3 = 0u - ((lit >> (4 * g + 3)) & 1u);
            const uint32_t m0 = (s0 & 0x00FFFFFFu) | (s1 & 0xFF000000u);
            const uint32_t m1 = (s1 & 0x0000FFFFu) | (s2 & 0xFFFF0000u);
            const uint32_t m2 = (s2 & 0x000000FFu) | (s3 & 0xFFFFFF00u);
            uint32_t d0 = 0u, d1 = 0u, d2 = 0u;
            if (alpha < 255u) {  // workgroup-uniform: only the dim ramp at the start of an episode
                d0 = darken4(v0, alpha);
                d1 = darken4(v1, alpha);
                d2 = darken4(v2, alpha);
            }
            v0 = (v0 & m0) | (d0 & ~m0);
            v1 = (v1 & m1) | (d1 & ~m1);
            v2 = (v2 & m2) | (d2 & ~m2);
        }
        dst[3 * g] = v0;
        dst[3 * g + 1] = v1;
        dst[3 * g + 2] = v2;
    }
}

// Composer concept:
//   struct Desc;                                   trivially copyable, sizeof % 16 == 0
//   struct Pre;                                    every global operand of one frame, in registers
//   static __device__ bool skip(cptr<Desc>);       true: leave the frame untouched (masked reset)
//   static __device__ void prefetch(cptr<Desc>, const RasterCtx&, Pre&);   issues the loads, no LDS access
//   static __device__ void compose(cptr<Desc>, const Pre&, const RasterCtx&);   LDS only; leaves the frame complete
//   static __device__ void recycle(const RasterCtx&);   re-initialise per-frame LDS scratch (not the frame) after compose()
// The descriptor is read through its (workgroup-uniform, constant address space) pointer, so every field access --
// also array elements with a run-time index -- is a scalar load; a by-value copy would push indexed arrays to scratch.
// compose() does not touch global memory (gfx9 counts loads and stores in one in-order counter, vmcnt: a load issued
// inside compose() could only be consumed after the previous frame's stores had drained).  A software-pipelined loop
// (frame i+1's prefetch issued before frame i's stores) was built and measured: no gain over this simple loop for the
// spotlight frames, a loss for the mortar frames (profiles/r01c_raster_generations.md).
// Stream-out: mg_stream_out.hpp, buffer stores; NT = non-temporal (uint8 format only, chosen per launch: raster_nt()).
template <class Composer, int FMT, bool NT>
__global__ __launch_bounds__(256, 7) void raster_kernel(const typename Composer::Desc* __restrict__ descs, RasterAtlas A,
                                                     void* __restrict__ obs, int n, const uint8_t* __restrict__ only) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    RasterCtx R;
    R.frame = smem;
    R.mask = reinterpret_cast<uint32_t*>(smem + FRAME_BYTES);
    R.A = A;
    R.T = as_const(A.tables);
    R.tid = threadIdx.x;
    const int tid = threadIdx.x, stride = gridDim.x;
    const cptr<typename Composer::Desc> cdescs = as_const(descs);
    Composer::recycle(R);  // scratch state compose() expects (e.g. a zeroed hole mask); published by the barriers below
    __syncthreads();
    for (int env = blockIdx.x; env < n; env += stride) {
        if (Composer::skip(cdescs + env) || (only && !only[env])) continue;  // `only`: per-frame filter (final observations)
        typename Composer::Pre P;
        Composer::prefetch(cdescs + env, R, P);
        Composer::compose(cdescs + env, P, R);
        __syncthreads();
        Composer::recycle(R);  // overlaps the stream-out, saves a barrier at the start of the next compose()
        store_frame<FMT, NT, true>(smem, obs, env, tid);
        __syncthreads();  // the LDS frame is reused by the next iteration
    }
}

// A launch that draws FEW of the n frames (a masked reset; round 6, see mg_raster_v1.hpp: raster_sparse_kernel): a workgroup owns
// SPARSE_CHUNK consecutive instances, reads their descriptors (and the caller's mask) with one vector load and draws the ones a ballot names.
constexpr int SPARSE_CHUNK = 32;
template <class Composer, int FMT>
__global__ __launch_bounds__(256) void raster_sparse_kernel(const typename Composer::Desc* __restrict__ descs, RasterAtlas A,
                                                          void* __restrict__ obs, int n, const uint8_t* __restrict__ only) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    RasterCtx R;
    R.frame = smem;
    R.mask = reinterpret_cast<uint32_t*>(smem + FRAME_BYTES);
    R.A = A;
    R.T = as_const(A.tables);
    R.tid = threadIdx.x;
    const int tid = threadIdx.x;
    const cptr<typename Composer::Desc> cdescs = as_const(descs);
    Composer::recycle(R);  // scratch state compose() expects (e.g. a zeroed hole mask); published by the barriers below
    __syncthreads();
    for (int base = blockIdx.x * SPARSE_CHUNK; base < n; base += gridDim.x * SPARSE_CHUNK) {
        const int e = base + (tid & (SPARSE_CHUNK - 1));  // (every wave looks at the same SPARSE_CHUNK instances: the same list in all four)
        const bool want = e < n && !Composer::skip(cdescs + e) && (!only || only[e]);
        uint32_t m = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)__ballot(want));
        while (m) {
            const int env = base + __builtin_ctz(m);
            m &= m - 1;
            typename Composer::Pre P;
            Composer::prefetch(cdescs + env, R, P);
            Composer::compose(cdescs + env, P, R);
            __syncthreads();
            Composer::recycle(R);
            store_frame<FMT, false, true>(smem, obs, env, tid);
            __syncthreads();  // the LDS frame is reused by the next iteration
        }
    }
}

// Workgroups of a raster launch over n frames (MEMGYM_RASTER_GRID overrides: tuning experiments).  A workgroup that draws
// several frames has each next frame's loads queued behind its own stores (gfx9 counts both in vmcnt), a launch of one
// workgroup per frame pays ~14,000 wave launches: the optimum lies in between and moves with the launch size.  Round 3,
// Endless-SearingSpotlights at 16,384 frames, same call (profiles/r03_spot_grid.md): 3,584 / 5,120 / 7,168 / 8,960 / 9,728 /
// 10,752 / 14,336 / 16,384 workgroups at six per CU -> 78.0 / 73.8 / 69.0 / 65.7 / 65.5 / 67.0 / 72.5 / 77.4 us; the same
// optimum with plain stores at seven per CU (profiles/r03_spot_store_lab.md: 7,168 / 9,728 / 14,336 -> 60.7 / 59.7 / 64.5 us).
// At 32,768 frames and beyond 14,336 workgroups win (32,768: 121.4 against 124.1-127.1 us for 7,168-12,288).
inline int raster_grid(int n) {
    static const int forced = [] {
        const char* e = lab_env("MEMGYM_RASTER_GRID");
        return e ? atoi(e) : 0;
    }();
    return forced > 0 ? forced : (n <= 24576 ? RASTER_GRID_SMALL : RASTER_GRID);
}

// Store flavour of the uint8 stream for a launch over n frames.  Plain stores are the faster stream (16,384 frames: 59.7 us
// = 5.8 TB/s against 63.3 us non-temporal) but pass through the caches, and the logic kernel then finds less of its state
// there: +2-3 us at 16,384 instances (212 vs 208 M env-steps/s, plain wins), +10 us at 65,536 (223 vs 230 M, non-temporal
// wins).  MEMGYM_RASTER_NT = 0 / 1 forces one (tuning only).  profiles/r03_spot_store_lab.md.
inline bool raster_nt(int n) {
    static const int forced = [] {
        const char* e = lab_env("MEMGYM_RASTER_NT");
        return e ? (atoi(e) != 0 ? 1 : 0) : -1;
    }();
    return forced >= 0 ? forced != 0 : n > RASTER_PLAIN_MAX;
}

template <class Composer>
inline void launch_raster(const typename Composer::Desc* descs, const RasterAtlas& atlas, void* obs, int fmt, int n, hipStream_t s,
                          const uint8_t* only = nullptr) {
    const int tuned = raster_grid(n);
    // The kernel needs RASTER_LDS (22,176 B: 7 workgroups per CU).  A NON-TEMPORAL stream is faster with fewer concurrent
    // writers than fit and asks for 25 KiB = SIX per CU (profiles/r03_spot_grid.md, Endless-SearingSpotlights 16,384 frames: six
    // per CU x 8,960-10,240 workgroups 65.4-66.1 us, five x 14,336 (round 2) 71.4, seven x 8,960 68.4, four 80-84); plain
    // stores -- small launches and the float formats -- want all seven (7: 59.7, 6: 62.3 us; profiles/r03_spot_store_lab.md).
    // MEMGYM_RASTER_LDS overrides (tuning only).
    static const int forced_lds = [] {
        const char* e = lab_env("MEMGYM_RASTER_LDS");
        return e && atoi(e) >= RASTER_LDS ? atoi(e) : 0;
    }();
    const bool nt = fmt == MG_OBS_U8_XYC && raster_nt(n);
    const int lds = forced_lds ? forced_lds : (nt ? RASTER_LDS_REQUEST : RASTER_LDS);
    const int grid = n < tuned ? n : tuned;
    if (fmt == MG_OBS_F32_CYX)
        hipLaunchKernelGGL((raster_kernel<Composer, MG_OBS_F32_CYX, false>), dim3(grid), dim3(256), lds, s, descs, atlas, obs, n, only);
    else if (fmt == MG_OBS_BF16_CYX)
        hipLaunchKernelGGL((raster_kernel<Composer, MG_OBS_BF16_CYX, false>), dim3(grid), dim3(256), lds, s, descs, atlas, obs, n, only);
    else if (fmt == MG_OBS_F16_CYX)
        hipLaunchKernelGGL((raster_kernel<Composer, MG_OBS_F16_CYX, false>), dim3(grid), dim3(256), lds, s, descs, atlas, obs, n, only);
    else if (nt)
        hipLaunchKernelGGL((raster_kernel<Composer, MG_OBS_U8_XYC, true>), dim3(grid), dim3(256), lds, s, descs, atlas, obs, n, only);
    else
        hipLaunchKernelGGL((raster_kernel<Composer, MG_OBS_U8_XYC, false>), dim3(grid), dim3(256), lds, s, descs, atlas, obs, n, only);
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

}  // namespace mg
