// mg_device.hpp -- device-side building blocks shared by the per-family logic/raster kernels (gfx950).
//
//  * Pcg: numpy-compatible Generator(PCG64(SeedSequence(seed))) stream per environment instance
//    (what gymnasium's Env.reset(seed) hands the reference: every draw site listed in SURVEY.md 8(a),
//    e.g. mortar_mayhem_grid.py:181,186,244).  State lives in HBM as five SoA arrays.
//  * frame streaming helpers for the 84x84x3 observation ([x][y][c], 21,168 B = 1,323 x 16 B).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mg {

constexpr int SCREEN = 84;
constexpr int FRAME_BYTES = SCREEN * SCREEN * 3;  // 21168
constexpr int FRAME_VEC16 = FRAME_BYTES / 16;     // 1323
constexpr int COL_BYTES = SCREEN * 3;             // 252 bytes per x column

typedef unsigned __int128 u128;

// SoA view of the per-instance RNG streams.
struct RngSoA {
    uint64_t* s_hi;
    uint64_t* s_lo;
    uint64_t* inc_hi;
    uint64_t* inc_lo;
    uint64_t* buf;  // bit 32 = has_uint32, bits 0..31 = buffered high half
};

struct Pcg {
    u128 state, inc;
    uint32_t buf;
    bool has;

    __device__ __forceinline__ void load(const RngSoA& r, int i) {
        state = ((u128)r.s_hi[i] << 64) | r.s_lo[i];
        inc = ((u128)r.inc_hi[i] << 64) | r.inc_lo[i];
        uint64_t b = r.buf[i];
        buf = (uint32_t)b;
        has = (b >> 32) & 1;
    }
    // A use the compiler cannot move: placed behind other requests, it keeps a speculative load() up front among them
    // instead of where the (rare) first draw is -- which would be a memory round trip of its own in that path.
    __device__ __forceinline__ void pin() {
        uint64_t a = (uint64_t)state, b = (uint64_t)(state >> 64), c = (uint64_t)inc, d = (uint64_t)(inc >> 64);
        asm volatile("" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(buf));
        state = ((u128)b << 64) | a;
        inc = ((u128)d << 64) | c;
    }
    __device__ __forceinline__ void store(const RngSoA& r, int i) const {
        r.s_hi[i] = (uint64_t)(state >> 64);
        r.s_lo[i] = (uint64_t)state;
        r.inc_hi[i] = (uint64_t)(inc >> 64);
        r.inc_lo[i] = (uint64_t)inc;
        r.buf[i] = (uint64_t)buf | ((uint64_t)(has ? 1 : 0) << 32);
    }
    __device__ __forceinline__ void advance() {
        const u128 mult = (((u128)0x2360ED051FC65DA4ull) << 64) | (u128)0x4385DF649FCCF645ull;
        state = state * mult + inc;
    }
    // np.random.SeedSequence(seed).generate_state(4, uint64) -> PCG64 seeding
    __device__ void seed(uint64_t s) {
        const uint32_t INIT_A = 0x43b0d7e5u, MULT_A = 0x931e8875u, INIT_B = 0x8b51f9ddu, MULT_B = 0x58f38dedu;
        const uint32_t MIX_L = 0xca01f9ddu, MIX_R = 0x4973f715u;
        uint32_t ent[4] = {(uint32_t)s, (uint32_t)(s >> 32), 0u, 0u};
        uint32_t pool[4], hc = INIT_A;
        auto hashmix = [&](uint32_t v) {
            v ^= hc;
            hc *= MULT_A;
            v *= hc;
            v ^= v >> 16;
            return v;
        };
        auto mix = [&](uint32_t x, uint32_t y) {
            uint32_t r = MIX_L * x - MIX_R * y;
            r ^= r >> 16;
            return r;
        };
        for (int i = 0; i < 4; ++i) pool[i] = hashmix(ent[i]);
        for (int a = 0; a < 4; ++a)
            for (int d = 0; d < 4; ++d)
                if (a != d) pool[d] = mix(pool[d], hashmix(pool[a]));
        uint32_t w[8];
        hc = INIT_B;
        for (int i = 0; i < 8; ++i) {
            uint32_t d = pool[i & 3] ^ hc;
            hc *= MULT_B;
            d *= hc;
            d ^= d >> 16;
            w[i] = d;
        }
        u128 initstate = ((u128)(((uint64_t)w[1] << 32) | w[0]) << 64) | (((uint64_t)w[3] << 32) | w[2]);
        u128 initseq = ((u128)(((uint64_t)w[5] << 32) | w[4]) << 64) | (((uint64_t)w[7] << 32) | w[6]);
        inc = (initseq << 1) | 1u;
        state = 0;
        advance();
        state += initstate;
        advance();
        has = false;
        buf = 0;
    }
    __device__ __forceinline__ uint64_t next64() {
        advance();
        uint64_t hi = (uint64_t)(state >> 64), lo = (uint64_t)state, x = hi ^ lo;
        unsigned rot = (unsigned)(hi >> 58);
        return (x >> rot) | (x << ((64 - rot) & 63));
    }
    __device__ __forceinline__ uint32_t next32() {  // low half first, high half buffered
        if (has) {
            has = false;
            return buf;
        }
        uint64_t v = next64();
        has = true;
        buf = (uint32_t)(v >> 32);
        return (uint32_t)v;
    }
    __device__ __forceinline__ double next_double() { return (double)(next64() >> 11) * (1.0 / 9007199254740992.0); }
    // Generator.integers(lo, hi): Lemire bounded draw on the 32-bit path; span 1 consumes nothing.
    __device__ __forceinline__ int integers(int lo, int hi) {
        uint32_t rng = (uint32_t)(hi - 1 - lo);
        if (rng == 0) return lo;
        uint32_t n = rng + 1u;
        uint64_t m = (uint64_t)next32() * n;
        uint32_t left = (uint32_t)m;
        if (left < n) {
            uint32_t thr = (0xFFFFFFFFu - rng) % n;
            while (left < thr) {
                m = (uint64_t)next32() * n;
                left = (uint32_t)m;
            }
        }
        return lo + (int)(m >> 32);
    }
    // Generator.uniform(a, b) = a + (b - a) * next_double()   (no fused multiply-add: built with -ffp-contract=off)
    __device__ __forceinline__ double uniform(double a, double b) { return a + (b - a) * next_double(); }
};

// CharacterController.step (character_controller.py:89-146): 8-way move with `int()`-truncated velocity
// (v_axis = int(speed), v_diag = int(speed / sqrt(2))), rotation from the velocity signs, optional clamp of the
// centre to [lo_x, hi_x] x [lo_y, hi_y].  rot8 = rotation / 45.
__device__ __forceinline__ void free_move(int a0, int a1, int v_axis, int v_diag, int& ax, int& ay, uint8_t& rot8, bool clamp,
                                          int lo_x, int hi_x, int lo_y, int hi_y) {
    int dxs = a0 == 1 ? -1 : (a0 == 2 ? 1 : 0), dys = a1 == 1 ? -1 : (a1 == 2 ? 1 : 0);
    int rot = rot8 * 45;
    if (a0 == 1) rot = 90;
    if (a0 == 2) rot = 270;
    if (a1 == 1) rot = 0;
    if (a1 == 2) rot = 180;
    if (dxs < 0 && dys < 0) rot = 45;
    if (dxs < 0 && dys > 0) rot = 135;
    if (dxs > 0 && dys < 0) rot = 315;
    if (dxs > 0 && dys > 0) rot = 225;
    rot8 = (uint8_t)(rot / 45);
    int v = (dxs != 0 && dys != 0) ? v_diag : v_axis;
    ax += dxs * v;
    ay += dys * v;
    if (clamp) {
        ax = ax > hi_x ? hi_x : ax;
        ax = ax < lo_x ? lo_x : ax;
        ay = ay > hi_y ? hi_y : ay;
        ay = ay < lo_y ? lo_y : ay;
    }
}

// Index of the option set an instance runs under (include/memgym.h: mg_bind_option_sets), from the caller's int32 array: masked
// to the MG_MAX_OPTION_SETS = 8 parameter blocks every handle uploads, so that a stray entry reads SOME block of the handle
// (never-written sets hold the reference's defaults) instead of memory beyond them.
__device__ __forceinline__ int set_index(const int32_t* set_of, int i) { return set_of[i] & 7; }

// "sample one per episode" option list (np_random.choice(list) == list[integers(0, len)], e.g.
// mortar_mayhem_grid.py:181,253-254,268-269; mystery_path.py:154; searing_spotlights.py:408).  The reference accepts lists
// of any length with any ints.  A list of up to OPT_INLINE entries that all fit a byte (every default, every list a curriculum
// plausibly uses) travels in the kernel arguments, four entries per dword: the word is picked with a select chain over eight
// uniform (scalar) operands and the byte with a shift -- indexing the list with a lane's draw would be a load from the
// kernel-argument segment, one more dependent memory round trip in the reset's serial code.  Longer lists and lists with
// larger entries (round 5: 32-bit) live in a device array owned by the family (OptListStore, mg_family.hpp) and cost that load.
constexpr int OPT_INLINE = 32;
struct OptList {
    int n;
    uint32_t w[OPT_INLINE / 4];
    const int32_t* ext;  // non-NULL: all n entries, device memory (n > OPT_INLINE or an entry beyond 255)
};
__device__ __forceinline__ int choice(Pcg& g, const OptList& l) {
    const int k = g.integers(0, l.n);
    if (__builtin_expect(l.ext != nullptr, 0)) return l.ext[k];  // uniform branch
    const int wi = k >> 2;
    uint32_t w = l.w[0];
#pragma unroll
    for (int j = 1; j < OPT_INLINE / 4; ++j) w = wi == j ? l.w[j] : w;
    return (int)((w >> (8 * (k & 3))) & 0xFFu);
}

}  // namespace mg
