// mg_stream_out.hpp -- the ONE frame stream-out both raster generations use (mg_raster_v1.hpp, mg_raster.hpp).
//
// A composed 84x84x3 frame sits in LDS in the reference's observation order ([x][y][c] uint8, what
// pygame.surfarray.array3d returns, e.g. memory_gym/mortar_mayhem_grid.py:277,372); store_frame writes it to the
// caller's observation buffer in the format chosen with mg_set_obs_format:
//   MG_OBS_U8_XYC    1,323 x 16-byte stores, lane-contiguous (1 KiB per wave instruction), the headline format;
//   MG_OBS_F32_CYX / MG_OBS_F16_CYX / MG_OBS_BF16_CYX   value / 255 in image order [c][y][x]: the
//                    transpose is done LDS-side (byte gathers, stride 252 B), the global stores stay 16-byte vectors.
// NT (uint8 format only): non-temporal stores.  A measured choice, not a taste: a non-temporal stream does not displace the
// logic kernel's state from the caches (spotlight family at 65,536 instances: logic kernel 38 vs 48 us) but is itself slower
// than a plain one (16,384 frames: 63 vs 60 us); the mortar frames (generation 1) lose 40 % with it
// (profiles/r01c_raster_generations.md, profiles/r03_spot_store_lab.md).  Generation 2 picks per launch size (mg_raster.hpp).
// BUF (uint8 format only): the same six stores as raw BUFFER stores on a per-frame resource -- the sixth needs no lane
// predicate (the range check drops what lies beyond the frame); generation 2 uses them (-2 us per 16,384 frames).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/memgym.h"
#include "mg_device.hpp"

namespace mg {

constexpr int TAIL = FRAME_VEC16 - 5 * 256;  // 43 lanes of a 256-lane workgroup carry a sixth 16-byte chunk

// Measurement builds (tools/build_all_variant.sh, profiles/r06_store_counters.md): MG_LAB_OBS_STRIDE = bytes between the uint8 frames of
// neighbouring instances (21,248 = 166 whole cache lines: no 64-byte block is shared by two frames); MG_LAB_XCD_GROUP: the frame a
// workgroup's ordinal draws is permuted so that neighbouring frames are composed on the same XCD (see xcd_grouped_frame).
#ifdef MG_LAB_OBS_STRIDE
constexpr size_t OBS_STRIDE_U8 = MG_LAB_OBS_STRIDE;
#else
constexpr size_t OBS_STRIDE_U8 = FRAME_BYTES;
#endif
// Workgroups are dealt to the eight XCDs round-robin (workgroup b -> XCD b % 8, each with an L2 of its own).  Ordinal v = 64 q + 8 r + x
// draws frame 64 q + 8 x + r: the eight workgroups of one XCD within a block of 64 own eight CONSECUTIVE frames (= 1,323 whole lines).
__device__ __forceinline__ int xcd_grouped_frame(int v, int n) {
#ifdef MG_LAB_FRAMES_DOWN  // measurement builds: the launch walks the buffer from its end to its start
    return n - 1 - v;
#elif defined(MG_LAB_XCD_GROUP)
    return (v | 63) < n ? ((v & ~63) | ((v & 7) << 3) | ((v >> 3) & 7)) : v;  // (a last, partial block of 64 keeps the plain order)
#else
    (void)n;
    return v;
#endif
}

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));  // native vector: stays in registers inside a struct (HIP's uint4 class members end up in scratch)

// b / 255 as the correctly rounded float32 quotient, without the ~10 instructions of an IEEE division: the rounded
// reciprocal, one residual, one correction (Markstein); equal to the division for all 256 bytes (tests/test_unit_division.py)
__device__ __forceinline__ float byte_to_unit(uint8_t b) {
    const float v = (float)b, r = 1.0f / 255.0f;
    const float q0 = v * r;
    return __fmaf_rn(__fmaf_rn(-q0, 255.0f, v), r, q0);
}

template <int FMT, bool NT, bool BUF = false>
__device__ __forceinline__ void store_frame(const uint8_t* __restrict__ frame, void* __restrict__ obs, int env, int tid) {
    if constexpr (FMT == MG_OBS_U8_XYC && BUF) {
        const u32x4* lds16 = reinterpret_cast<const u32x4*>(frame);
        // raw buffer over this frame only: stride 0, FRAME_BYTES records, dword 3 = 32-bit untyped data (gfx9 encoding)
        __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(static_cast<uint8_t*>(obs) + (size_t)env * OBS_STRIDE_U8, 0, FRAME_BYTES, 0x00020000);
        u32x4 v0 = lds16[tid], v1 = lds16[tid + 256], v2 = lds16[tid + 512], v3 = lds16[tid + 768], v4 = lds16[tid + 1024];
        u32x4 v5 = (u32x4)(0u);
        if (tid < TAIL) v5 = lds16[tid + 1280];
        constexpr int AUX = NT ? 2 : 0;  // cache policy operand: bit 1 = nt
        // (upwards: the plain path's downward order measured 1-5 % WORSE here, tail-first and alternating orders and a sleep between the
        // stores nothing -- profiles/r06_store_counters.md, addendum 2)
        __builtin_amdgcn_raw_buffer_store_b128(v0, rs, tid * 16, 0, AUX);
        __builtin_amdgcn_raw_buffer_store_b128(v1, rs, (tid + 256) * 16, 0, AUX);
        __builtin_amdgcn_raw_buffer_store_b128(v2, rs, (tid + 512) * 16, 0, AUX);
        __builtin_amdgcn_raw_buffer_store_b128(v3, rs, (tid + 768) * 16, 0, AUX);
        __builtin_amdgcn_raw_buffer_store_b128(v4, rs, (tid + 1024) * 16, 0, AUX);
        __builtin_amdgcn_raw_buffer_store_b128(v5, rs, (tid + 1280) * 16, 0, AUX);  // lanes >= TAIL: out of range, dropped
    } else if constexpr (FMT == MG_OBS_U8_XYC) {
        const u32x4* lds16 = reinterpret_cast<const u32x4*>(frame);
        u32x4* dst = reinterpret_cast<u32x4*>(static_cast<uint8_t*>(obs) + (size_t)env * OBS_STRIDE_U8);
        u32x4 v0 = lds16[tid], v1 = lds16[tid + 256], v2 = lds16[tid + 512], v3 = lds16[tid + 768], v4 = lds16[tid + 1024];
        u32x4 v5 = (u32x4)(0u);
        if (tid < TAIL) v5 = lds16[tid + 1280];
        if constexpr (NT) {
            __builtin_nontemporal_store(v0, &dst[tid]); __builtin_nontemporal_store(v1, &dst[tid + 256]);
            __builtin_nontemporal_store(v2, &dst[tid + 512]); __builtin_nontemporal_store(v3, &dst[tid + 768]);
            __builtin_nontemporal_store(v4, &dst[tid + 1024]);
            if (tid < TAIL) __builtin_nontemporal_store(v5, &dst[tid + 1280]);
        } else {
            // DOWNWARDS, the partial tail first (round 6): the same six stores in the opposite order measure 2.2-2.6 % faster on the mortar
            // family's one-launch step at 65,536 instances (289.6-289.9 -> 297.2-297.4 M env-steps/s, A/B/A/B on one box), 1-2 % at 32,768,
            // nothing at 16,384 and on the Mystery Path launches (profiles/r06_store_counters.md, "pacing"); upwards was rounds 1-5.
#if defined(MG_LAB_STORE_ORDER) && MG_LAB_STORE_ORDER == 0   // measurement builds: upwards
            dst[tid] = v0; dst[tid + 256] = v1; dst[tid + 512] = v2; dst[tid + 768] = v3; dst[tid + 1024] = v4;
            if (tid < TAIL) dst[tid + 1280] = v5;
#else
            if (tid < TAIL) dst[tid + 1280] = v5;
            dst[tid + 1024] = v4; dst[tid + 768] = v3; dst[tid + 512] = v2; dst[tid + 256] = v1; dst[tid] = v0;
#endif
        }
    } else if constexpr (FMT == MG_OBS_F32_CYX) {
        float4* dst = reinterpret_cast<float4*>(static_cast<float*>(obs) + (size_t)env * FRAME_BYTES);
        constexpr int PER_ROW = SCREEN / 4, TOTAL = 3 * SCREEN * PER_ROW;  // 21 float4 per (c, y) row, 5,292 per frame
        for (int q = tid; q < TOTAL; q += 256) {
            const int row = q / PER_ROW, x0 = (q - row * PER_ROW) * 4;
            const int c = row / SCREEN, y = row - c * SCREEN;
            const uint8_t* src = frame + x0 * COL_BYTES + y * 3 + c;
            float4 v;
            v.x = byte_to_unit(src[0]);
            v.y = byte_to_unit(src[COL_BYTES]);
            v.z = byte_to_unit(src[2 * COL_BYTES]);
            v.w = byte_to_unit(src[3 * COL_BYTES]);
            dst[q] = v;
        }
    } else {
        // The 16-bit formats move half the bytes of float32 behind the same gather and were bound by the vector unit, not by the memory
        // (round 6, profiles/r06_float_formats.md).  Per element now: one byte read, one conversion, ONE multiply -- byte * (1/255)f rounds to
        // the same bfloat16 / half as the correctly rounded float32 quotient for all 256 bytes (tests/test_unit_division.py), so the
        // two-instruction correction of byte_to_unit is the float32 format's alone -- the 16-bit rounding, half a pack; the (row, column
        // group) of a lane's stores advances by additions (512 groups = 24 rows + 8 groups per round) instead of two divisions per group.
        u32x4* dst = reinterpret_cast<u32x4*>(static_cast<uint16_t*>(obs) + (size_t)env * FRAME_BYTES);
        constexpr int PER_ROW = SCREEN / 4, TOTAL = 3 * SCREEN * PER_ROW / 2;  // 21 four-x groups per (c, y) row; 8 halves (two groups) per 16-B store
        constexpr int ROWS_PER_ROUND = (2 * 256) / PER_ROW, GROUPS_PER_ROUND = (2 * 256) % PER_ROW;  // 24, 8
        int xg[2], y[2], at[2];  // column group, row, and the LDS offset of the group's first byte: x0 * 252 + y * 3 + c
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            const int qq = 2 * tid + g, row = qq / PER_ROW;
            xg[g] = qq - row * PER_ROW;
            const int c0 = row / SCREEN;
            y[g] = row - c0 * SCREEN;
            at[g] = xg[g] * (4 * COL_BYTES) + y[g] * 3 + c0;
        }
        for (int q = tid; q < TOTAL; q += 256) {
            uint32_t w[4];
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                const uint8_t* src = frame + at[g];
                float f[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) f[k] = (float)src[k * COL_BYTES] * (1.0f / 255.0f);
                // two elements per conversion (gfx950: v_cvt_pk_bf16_f32 / v_cvt_pk_f16_f32, round to nearest even)
                typedef float f32x2 __attribute__((ext_vector_type(2)));
                const f32x2 lo = {f[0], f[1]}, hi = {f[2], f[3]};
                if constexpr (FMT == MG_OBS_BF16_CYX) {
                    typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
                    union { bf16x2 v; uint32_t u; } a, b;
                    a.v = __builtin_convertvector(lo, bf16x2);
                    b.v = __builtin_convertvector(hi, bf16x2);
                    w[2 * g] = a.u;
                    w[2 * g + 1] = b.u;
                } else {
                    typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
                    union { f16x2 v; uint32_t u; } a, b;
                    a.v = __builtin_convertvector(lo, f16x2);
                    b.v = __builtin_convertvector(hi, f16x2);
                    w[2 * g] = a.u;
                    w[2 * g + 1] = b.u;
                }
                // the lane's next store, 512 groups on: 24 rows down and 8 groups to the right, by additions only
                xg[g] += GROUPS_PER_ROUND;
                const bool carry = xg[g] >= PER_ROW;  // past the row's end: the next row's start
                xg[g] -= carry ? PER_ROW : 0;
                y[g] += ROWS_PER_ROUND + (carry ? 1 : 0);
                const bool wrap = y[g] >= SCREEN;     // past the channel's last row: the next channel
                y[g] -= wrap ? SCREEN : 0;
                at[g] += GROUPS_PER_ROUND * (4 * COL_BYTES) + ROWS_PER_ROUND * 3 + (carry ? 3 - PER_ROW * (4 * COL_BYTES) : 0) + (wrap ? 1 - SCREEN * 3 : 0);
            }
            u32x4 v;
            v.x = w[0]; v.y = w[1]; v.z = w[2]; v.w = w[3];
            dst[q] = v;
        }
    }
}

}  // namespace mg
