/* oracle/mgo_rng.h -- TEST INFRASTRUCTURE (CPU oracle), not product code.
 *
 * Restatement of the numpy random stream the reference consumes through
 * gymnasium.Env.reset(seed) -> np.random.Generator(np.random.PCG64(np.random.SeedSequence(seed))).
 * The algorithm lives in third-party numpy (unpinned by the reference; validated here against
 * numpy 2.2.6, see tests/test_oracle_rng.py) -- call sites in the reference:
 *   memory_gym/mortar_mayhem_grid.py:181,186,244,253,254,268,269 and the list in SURVEY.md 8(a).
 * Published algorithms restated: SeedSequence (M.E. O'Neill seed_seq_fe128 variant used by numpy),
 * PCG64 XSL-RR 128/64, Lemire's nearly-divisionless bounded integers (numpy _bounded_integers),
 * next_double = (u64 >> 11) * 2^-53, 32-bit draws buffered LOW half first.
 */
#ifndef MGO_RNG_H
#define MGO_RNG_H
#include <stdint.h>

typedef unsigned __int128 mgo_u128;

typedef struct {
    mgo_u128 state, inc;
    int has_u32;
    uint32_t buf;
} mgo_rng;

static inline void mgo_seedseq_state(uint64_t seed, uint64_t out64[4]) {
    const uint32_t INIT_A = 0x43b0d7e5u, MULT_A = 0x931e8875u, INIT_B = 0x8b51f9ddu, MULT_B = 0x58f38dedu;
    const uint32_t MIX_L = 0xca01f9ddu, MIX_R = 0x4973f715u;
    uint32_t ent[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
    int n_ent = ent[1] ? 2 : 1;
    uint32_t pool[4], hc = INIT_A;
#define MGO_HASHMIX(v_in, res) do { uint32_t v_ = (v_in); v_ ^= hc; hc *= MULT_A; v_ *= hc; v_ ^= v_ >> 16; (res) = v_; } while (0)
#define MGO_MIX(x, y, res) do { uint32_t r_ = MIX_L * (x) - MIX_R * (y); r_ ^= r_ >> 16; (res) = r_; } while (0)
    for (int i = 0; i < 4; i++) {
        uint32_t e = i < n_ent ? ent[i] : 0u;
        MGO_HASHMIX(e, pool[i]);
    }
    for (int s = 0; s < 4; s++)
        for (int d = 0; d < 4; d++)
            if (s != d) {
                uint32_t h;
                MGO_HASHMIX(pool[s], h);
                MGO_MIX(pool[d], h, pool[d]);
            }
    /* entropy longer than the pool (never the case for 64-bit seeds: n_ent <= 2) would be mixed here */
    uint32_t w32[8];
    hc = INIT_B;
    for (int i = 0; i < 8; i++) {
        uint32_t d = pool[i % 4] ^ hc;
        hc *= MULT_B;
        d *= hc;
        d ^= d >> 16;
        w32[i] = d;
    }
    for (int k = 0; k < 4; k++) out64[k] = (uint64_t)w32[2 * k] | ((uint64_t)w32[2 * k + 1] << 32);
#undef MGO_HASHMIX
#undef MGO_MIX
}

#define MGO_PCG_MULT ((((mgo_u128)0x2360ED051FC65DA4ull) << 64) | (mgo_u128)0x4385DF649FCCF645ull)

static inline void mgo_rng_seed(mgo_rng* r, uint64_t seed) {
    uint64_t w[4];
    mgo_seedseq_state(seed, w);
    mgo_u128 initstate = ((mgo_u128)w[0] << 64) | w[1];
    mgo_u128 initseq = ((mgo_u128)w[2] << 64) | w[3];
    r->inc = (initseq << 1) | 1u;
    r->state = 0;
    r->state = r->state * MGO_PCG_MULT + r->inc;
    r->state += initstate;
    r->state = r->state * MGO_PCG_MULT + r->inc;
    r->has_u32 = 0;
    r->buf = 0;
}

static inline uint64_t mgo_next_u64(mgo_rng* r) {
    r->state = r->state * MGO_PCG_MULT + r->inc;
    uint64_t hi = (uint64_t)(r->state >> 64), lo = (uint64_t)r->state;
    uint64_t x = hi ^ lo;
    unsigned rot = (unsigned)(hi >> 58);
    return (x >> rot) | (x << ((64 - rot) & 63));
}

static inline uint32_t mgo_next_u32(mgo_rng* r) {
    if (r->has_u32) {
        r->has_u32 = 0;
        return r->buf;
    }
    uint64_t v = mgo_next_u64(r);
    r->has_u32 = 1;
    r->buf = (uint32_t)(v >> 32);
    return (uint32_t)v;
}

static inline double mgo_next_double(mgo_rng* r) { return (double)(mgo_next_u64(r) >> 11) * (1.0 / 9007199254740992.0); }

/* Generator.integers(lo, hi) with int64 dtype, endpoint=False; lo/hi already truncated toward zero.
 * Only ranges < 2^32 occur on the hot path. */
static inline int64_t mgo_integers(mgo_rng* r, int64_t lo, int64_t hi) {
    uint64_t rng = (uint64_t)(hi - 1 - lo);
    if (rng == 0) return lo;
    uint32_t rng_excl = (uint32_t)rng + 1u;
    uint64_t m = (uint64_t)mgo_next_u32(r) * rng_excl;
    uint32_t left = (uint32_t)m;
    if (left < rng_excl) {
        uint32_t thr = (0xFFFFFFFFu - (uint32_t)rng) % rng_excl;
        while (left < thr) {
            m = (uint64_t)mgo_next_u32(r) * rng_excl;
            left = (uint32_t)m;
        }
    }
    return lo + (int64_t)(m >> 32);
}

/* Generator.choice(list) == list[integers(0, len)] (a 1-element list consumes nothing) */
static inline int mgo_choice_index(mgo_rng* r, int n) { return (int)mgo_integers(r, 0, n); }

static inline double mgo_uniform(mgo_rng* r, double a, double b) { return a + (b - a) * mgo_next_double(r); }

#endif
