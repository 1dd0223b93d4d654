// mg_spot.hip -- Searing Spotlights family on gfx950: Endless-SearingSpotlights-v0 and SearingSpotlights-v0.
//
// Reference behaviour reproduced (bit-exact observations, rewards, dones, RNG consumption):
//   memory_gym/endless_searing_spotlights.py  reset :294-407  step :409-506  _step_spotlight_task :179-231
//                                             _spawn_coin :256-269  _step_coin_task :271-292
//   memory_gym/searing_spotlights.py          reset :332-450  step :452-562  helpers :137-330
//   memory_gym/pygame_assets.py               GridPositionSampler :7-59  Spotlight :61-131  Coin :133-167  Exit :169-220
//   memory_gym/character_controller.py        CharacterController.step :89-146
//
//   spot_step_kernel : SIXTEEN LANES per instance (4 instances per wave).  Lane s owns spotlight slot s: float64
//                      trajectory (lerp of lerp, un-fused multiply-add: the library is built with
//                      -ffp-contract=off) and hit test; wave ballots collect the 16 done / hit flags of an instance.
//                      The instance-level logic (agent, coin/exit, RNG, list bookkeeping) is executed redundantly by
//                      the 16 lanes -- free under SIMD -- and stored by lane 0; the grid sampler splits its 84 rows
//                      over them (sample_cell).  Kernels are templated on ENDLESS.  Slot arrays are
//                      [N][16] so the 16 lanes of an instance read one contiguous 128-byte row per field; the
//                      Python list semantics (append / remove-while-iterating) live in a 16-nibble order word.
//   raster_kernel<SpotComposer> (generation 2, mg_raster.hpp): hole mask -> chessboard template, coin(s)/exit, agent, each
//                      darkened outside the holes while written -> coin(s) shown above the dark layer -> top bar.
#include <memory>

#include "mg_atlas.hpp"
#include "mg_lab.hpp"
#include "mg_device.hpp"
#include "mg_family.hpp"
#include "mg_raster.hpp"
#include "mg_stamps.hpp"

namespace mg {

constexpr int SLOTS = 16;
constexpr int MAX_COINS = 8;
constexpr int MAX_HOLES = 16;  // = SLOTS: a frame shows at most one disc per slot
constexpr int LAYER_COIN_ABOVE = 1, LAYER_EXIT_ABOVE = 2, LAYER_AGENT_TOP = 4;  // SpotDesc::coin_above: what is drawn over the dark layer

struct SpotParams {
    int endless, n;
    int max_steps, steps_per_coin, initial_spawns, spawn_interval, interval0, num_spawns;
    int visual_feedback, dim_duration, dim_step, light_threshold;
    int black_background, hide_chessboard;  // repaint the instance's background surfaces for good (see BG_MODE_SHIFT)
    int ordered_holes;              // a spotlight with a border has been possible: hole words in LIST order, border composer
    int layer_flags;                // LAYER_EXIT_ABOVE (exit_visible) | LAYER_AGENT_TOP (agent_visible), OR-ed into SpotDesc::coin_above
    int coin_enabled, coin_show_duration, coins_visible, sample_agent_position, show_last_action, show_last_positive_reward;
    int use_exit;                   // finite variant; 0: no exit is spawned, the instance's EARLIER exit stays in the frame (spot_reset)
    int r_lo, r_hi;                 // radius = integers(r_lo, r_hi)
    int agent_radius, sprite_half, coin_radius;
    int v_axis_i, v_diag_i;
    int spawn_clamp;                // _process_spawn_pos offset
    int bar_x, bar_w, quarter, bar_h;
    // finite variant: the Exit stamps the handle holds, one pair (closed, open) per GENERATION = distinct exit_scale still on
    // some instance's screen (a stale exit keeps the size it was made with: exit_gen_of); exit_gen = the one new exits get,
    // exit_halves = (int(20 * exit_scale) >> 1) of each generation, a byte each
    int exit_gen;
    uint64_t exit_halves;
    double speed_lo, speed_hi, damage, agent_health, exit_radius, half_diag;
    double r_inside, r_outside, r_death, r_coin, r_exit;
    OptList num_coins;
    const double* cos_tab;          // [360] integer degrees, exact for multiples of 90
    const double* sin_tab;
    const uint4* jump;              // [16][2] PCG64 jump constants {A^(k+1), S_(k+1)} (new_spots_at_reset)
    int lab_fallback;               // lab build (MEMGYM_SPOT_RESET_FALLBACK=k): every k-th instance takes new_spots_at_reset's fall-back, as after a rejected draw (tests)
};

struct __attribute__((aligned(16))) SpotCore {
    int16_t ax, ay;
    uint8_t rot8, alpha, la0, la1;
    uint8_t red_w, n_spots, exit_open, n_coins;
    uint8_t n_intervals, last_pos, bg_red, has_coin;
    int32_t spawn_timer, t, coin_t, coins_collected;
    int16_t coin_x, coin_y, exit_x, exit_y;
    int32_t num_coins, ep_len;
    double health, ep_sum;
    uint64_t order;  // spotlight list: nibble k = slot of the k-th element
    // pad: debug view, bit 31 = a sprite has been shown, 18..16 sprite, 15..8 y + 128, 7..0 x + 128;
    //      bits 21..20 / 23..22 = what the blue / red background surface of this instance looks like (BG_CHESS / WHITE / BLACK)
    uint32_t free_mask, pad;
};
static_assert(sizeof(SpotCore) == 80, "SpotCore must be 80 bytes");

// The frame descriptor: 32 dwords = ONE 128-byte line (round 4; it was 160 bytes over two or three lines).
//   w0  valid | bg << 8 | sprite << 16 | alpha << 24          w1  sx | sy << 16 (int16 each)
//   w2  n_holes | n_coins << 8 | coin_above << 16 | red_w << 24
//   w3  c_base | c_act0 << 8 | c_act1 << 16 | c_bar << 24      w4  bar_x | bar_w << 8 | quarter << 16 | exit_stamp << 24
//   w5  exit_x | exit_y << 16                                  w6, w7  unused
//   w8 .. w15  coins: (x + 128) | (y + 128) << 16, top-left of the coin stamp        w16 .. w31 holes
// The composers address it by WORD through scalar loads (SpotView over a pointer in the constant address space).
struct __attribute__((aligned(128))) SpotDesc {
    uint32_t valid : 8, bg : 8, sprite : 8, alpha : 8;
    int32_t sx : 16, sy : 16;
    uint32_t n_holes : 8, n_coins : 8, coin_above : 8, red_w : 8;
    uint32_t c_base : 8, c_act0 : 8, c_act1 : 8, c_bar : 8;
    uint32_t bar_x : 8, bar_w : 8, quarter : 8, exit_stamp : 8;
    int32_t exit_x : 16, exit_y : 16;
    uint32_t pad[2];
    uint32_t coins[MAX_COINS];
    uint32_t holes[MAX_HOLES];
};
static_assert(sizeof(SpotDesc) == 128 && MAX_HOLES == 16 && MAX_COINS == 8, "SpotDesc is one 128-byte line");
constexpr int DW_COINS = 8, DW_HOLES = 16;

struct DescWordsMem {  // the descriptor in memory, written by an EARLIER launch: scalar loads
    cptr<uint32_t> p;
    __device__ __forceinline__ uint32_t w(int k) const { return p[k]; }
};
template <class W>
struct SpotView {
    W s;
    __device__ __forceinline__ uint32_t valid() const { return s.w(0) & 0xFFu; }
    __device__ __forceinline__ uint32_t bg() const { return (s.w(0) >> 8) & 0xFFu; }
    __device__ __forceinline__ uint32_t sprite() const { return (s.w(0) >> 16) & 0xFFu; }
    __device__ __forceinline__ uint32_t alpha() const { return s.w(0) >> 24; }
    __device__ __forceinline__ int sx() const { return (int)(int16_t)(s.w(1) & 0xFFFFu); }
    __device__ __forceinline__ int sy() const { return (int)s.w(1) >> 16; }
    __device__ __forceinline__ int n_holes() const { return (int)(s.w(2) & 0xFFu); }
    __device__ __forceinline__ int n_coins() const { return (int)((s.w(2) >> 8) & 0xFFu); }
    __device__ __forceinline__ uint32_t coin_above() const { return (s.w(2) >> 16) & 0xFFu; }
    __device__ __forceinline__ int red_w() const { return (int)(s.w(2) >> 24); }
    __device__ __forceinline__ uint32_t c_base() const { return s.w(3) & 0xFFu; }
    __device__ __forceinline__ uint32_t c_act0() const { return (s.w(3) >> 8) & 0xFFu; }
    __device__ __forceinline__ uint32_t c_act1() const { return (s.w(3) >> 16) & 0xFFu; }
    __device__ __forceinline__ uint32_t c_bar() const { return s.w(3) >> 24; }
    __device__ __forceinline__ int bar_x() const { return (int)(s.w(4) & 0xFFu); }
    __device__ __forceinline__ int bar_w() const { return (int)((s.w(4) >> 8) & 0xFFu); }
    __device__ __forceinline__ int quarter() const { return (int)((s.w(4) >> 16) & 0xFFu); }
    __device__ __forceinline__ uint32_t exit_stamp() const { return s.w(4) >> 24; }
    __device__ __forceinline__ int exit_x() const { return (int)(int16_t)(s.w(5) & 0xFFFFu); }
    __device__ __forceinline__ int exit_y() const { return (int)s.w(5) >> 16; }
    __device__ __forceinline__ int coin_x(int k) const { return (int)(s.w(DW_COINS + k) & 0xFFFFu) - 128; }
    __device__ __forceinline__ int coin_y(int k) const { return (int)(s.w(DW_COINS + k) >> 16) - 128; }
    __device__ __forceinline__ uint32_t hole(int h) const { return s.w(DW_HOLES + h); }
};
typedef SpotView<DescWordsMem> SpotViewMem;
__device__ __forceinline__ SpotViewMem view_of(cptr<SpotDesc> dp) { return SpotViewMem{DescWordsMem{(cptr<uint32_t>)dp}}; }

constexpr int ST_COIN = 8, ST_EXIT0 = 9;  // exit of generation g: closed ST_EXIT0 + 2 g, open ST_EXIT0 + 2 g + 1
constexpr int EXIT_GENS = 8;
// hide_chessboard / black_background paint over the two background surfaces an environment object keeps for its lifetime
// (searing_spotlights.py:349-351, 234-235, 420-421; endless :313-315, 223-224, 376-377): per instance, sticky across
// episodes and option changes.  Templates: 0 blue board, 1 red board, 2 white, 3 black.
constexpr uint32_t BG_CHESS = 0, BG_WHITE = 1, BG_BLACK = 2, BG_MODE_SHIFT = 20, BG_MODE_MASK = 0xFu << BG_MODE_SHIFT;
// SpotCore::pad bit 24: the instance has had an exit (searing_spotlights.py: self.exit exists); sticky like the board modes
// bits 27..25: the generation of that exit (SpotParams::exit_gen when it was spawned) -- self.exit is an object of its own in the
// reference: with use_exit = False it stays on screen as it was made, also once exit_scale has changed (searing_spotlights.py:431-435)
constexpr uint32_t PAD_HAS_EXIT = 1u << 24, PAD_EXIT_GEN_SHIFT = 25, PAD_EXIT_GEN_MASK = (uint32_t)(EXIT_GENS - 1) << PAD_EXIT_GEN_SHIFT;
constexpr uint32_t PAD_STICKY = BG_MODE_MASK | PAD_HAS_EXIT | PAD_EXIT_GEN_MASK;
__device__ __forceinline__ int exit_gen_of(uint32_t pad) { return (int)((pad & PAD_EXIT_GEN_MASK) >> PAD_EXIT_GEN_SHIFT); }
constexpr int ERR_NO_EXIT = 256;  // include/memgym.h: use_exit = False for an instance that never had an exit
__device__ __forceinline__ uint32_t bg_mode(uint32_t pad, int red) { return (pad >> (BG_MODE_SHIFT + 2 * red)) & 3u; }
__device__ __forceinline__ uint32_t bg_set(uint32_t pad, int red, uint32_t m) {
    return (pad & ~(3u << (BG_MODE_SHIFT + 2 * red))) | (m << (BG_MODE_SHIFT + 2 * red));
}
__device__ __forceinline__ uint8_t bg_template(uint32_t pad, int red) {
    const uint32_t m = bg_mode(pad, red);
    return (uint8_t)(m == BG_CHESS ? (uint32_t)red : 1u + m);
}
constexpr int BAR_H = 4;  // top bar height: int(16 * SCALE)

// ---- spotlights with a border (Spotlight.draw: filled disc, then pygame's 1-px circle in white, pygame_assets.py:110-113) ----
// The spotlight surface ends up with three kinds of pixels: black (the dark layer), the colour key (a hole) and white (a
// border pixel, blended over what lies below with the layer's alpha).  Spotlights are drawn in list order, so a pixel shows
// a border iff the LAST disc covering it has one and the pixel lies on it.  One lane per column walks the hole words in
// order: a disc clears the ring bits of its column span, a border sets its own (draw_circle_bresenham_thin: the end points
// of the spans draw_circle_filled walks, for every x step).  Only the border composer does this.
template <class HoleAt>
__device__ __forceinline__ void ring_mask(const RasterCtx& R, HoleAt hole_at, int nholes, uint32_t* ring) {
    if (R.tid >= SCREEN) return;
    const int X = R.tid;
    uint32_t rg[MASK_WORDS] = {0u, 0u, 0u};
    for (int h = 0; h < nholes; ++h) {
        const uint32_t hv = hole_at(h);
        const int hx = (int)(hv & 511u) - 128, hy = (int)((hv >> 9) & 511u) - 128, r = hole_radius(hv);
        const int col = X - (hx - r);
        if (col < 0 || col >= 2 * r) continue;
        const int lo = R.A.disc_span[(r * 2 * DISC_RMAX + col) * 2], hi = R.A.disc_span[(r * 2 * DISC_RMAX + col) * 2 + 1];
        int y0 = hy + lo, y1 = hy + hi;
        y0 = y0 < 0 ? 0 : y0;
        y1 = y1 > SCREEN - 1 ? SCREEN - 1 : y1;
#pragma unroll
        for (int w = 0; w < MASK_WORDS; ++w) {
            int a0 = y0 - 32 * w, a1 = y1 - 32 * w;
            a0 = a0 < 0 ? 0 : a0;
            a1 = a1 > 31 ? 31 : a1;
            if (a0 <= a1) rg[w] &= ~((a1 - a0 == 31) ? 0xFFFFFFFFu : (((1u << (a1 - a0 + 1)) - 1u) << a0));
        }
        if (!(hv >> 31)) continue;
        auto put = [&](int px, int py) {
            if (px == X && (unsigned)py < (unsigned)SCREEN) {
                const uint32_t bit = 1u << (py & 31);
#pragma unroll
                for (int w = 0; w < MASK_WORDS; ++w) rg[w] |= (py >> 5) == w ? bit : 0u;
            }
        };
        int f = 1 - r, ddx = 0, ddy = -2 * r, x = 0, y = r;
        while (x < y) {
            if (f >= 0) {
                --y;
                ddy += 2;
                f += ddy;
            }
            ++x;
            ddx += 2;
            f += ddx + 1;
            put(hx + x - 1, hy + y - 1);
            put(hx - x, hy + y - 1);
            put(hx + x - 1, hy - y);
            put(hx - x, hy - y);
            put(hx + y - 1, hy + x - 1);
            put(hx + y - 1, hy - x);
            put(hx - y, hy + x - 1);
            put(hx - y, hy - x);
        }
    }
#pragma unroll
    for (int w = 0; w < MASK_WORDS; ++w) ring[X * MASK_WORDS + w] = rg[w];
}
// the border pixels over everything drawn so far: d += (255 - d) * A / 255 (SDL ALPHA_BLEND_RGB, source white)
__device__ __forceinline__ void ring_apply(const RasterCtx& R, const uint32_t* ring, uint32_t alpha) {
    if (R.tid >= SCREEN * MASK_WORDS) return;
    const int X = R.tid / MASK_WORDS, w = R.tid - X * MASK_WORDS;
    uint32_t bits = ring[R.tid];
    while (bits) {
        const int b = __ffs(bits) - 1;
        bits &= bits - 1;
        uint8_t* p = R.frame + (X * SCREEN + 32 * w + b) * 3;
#pragma unroll
        for (int c = 0; c < 3; ++c) p[c] = (uint8_t)(p[c] + ((255u - p[c]) * alpha) / 255u);
    }
}

template <bool BORDER>
__device__ __forceinline__ uint32_t* ring_words() {  // LDS of the border composers only
    if constexpr (BORDER) {
        __shared__ uint32_t ring[SCREEN * MASK_WORDS];
        return ring;
    } else {
        return nullptr;
    }
}

template <bool BORDER>
struct SpotComposerT {
    typedef SpotDesc Desc;
    static __device__ __forceinline__ bool skip(cptr<Desc> dp) { return view_of(dp).valid() == 0; }
    // top bar (rows y < BAR_H of every column); priority reward bar > action rects > red > green > base.
    // Returns false where no bar element covers column x (the scene shows through).
    template <class V>
    static __device__ __forceinline__ bool bar_colour(const V& d, cptr<AtlasTables> T, int x, uint32_t* c) {
        bool has = d.c_base() != 0xFF;
        uint32_t id = d.c_base();
        if (x < 2 * d.quarter()) { id = x < d.red_w() ? (uint32_t)C_RED : (uint32_t)C_GREEN; has = true; }
        else if (d.c_act0() != 0xFF) { id = x < 3 * d.quarter() ? d.c_act0() : d.c_act1(); has = true; }
        if (d.c_bar() != 0xFF && x >= d.bar_x() && x < d.bar_x() + d.bar_w()) { id = d.c_bar(); has = true; }
        if (has) *c = T->palette[id];
        return has;
    }
    template <class V>
    static __device__ __forceinline__ bool bar_covers(const V& d, int x) {
        return d.c_base() != 0xFF || x < 2 * d.quarter() || d.c_act0() != 0xFF || (d.c_bar() != 0xFF && x >= d.bar_x() && x < d.bar_x() + d.bar_w());
    }
    template <class V>
    static __device__ __forceinline__ void bar_columns(const V& d, cptr<AtlasTables> T, const RasterCtx& R) {
        static_assert(BAR_H == 4, "one bar column = 4 pixels = 3 dwords");
        if (R.tid < SCREEN) {
            uint32_t c = 0u;
            if (bar_colour(d, T, R.tid, &c)) {
                const uint32_t r = c & 0xFFu, g = (c >> 8) & 0xFFu, b = (c >> 16) & 0xFFu;
                uint32_t* p = reinterpret_cast<uint32_t*>(R.frame) + R.tid * (COL_BYTES / 4);
                p[0] = r | (g << 8) | (b << 16) | (r << 24);
                p[1] = g | (b << 8) | (r << 16) | (g << 24);
                p[2] = b | (r << 8) | (g << 16) | (b << 24);
            }
        }
    }
    // Order of the reference's _draw_surfaces (endless_searing_spotlights.py:464-479, searing_spotlights.py:524-545):
    // board, coins (unless drawn above), exit, agent, spotlight layer, coins above, top bar.  The spotlight layer is
    // not a pass of its own: the hole mask is built first and every layer below it is darkened as it is written.
    struct Pre {
        TemplRegs bg;
        StampRegs<1> agent, coin, exitp;
        HoleRegs8 holes;
    };
    // every global read of the frame (template, sprite / coin / exit pixels, disc spans)
    template <class V>
    static __device__ __forceinline__ void prefetch_v(const V& d, const RasterCtx& R, Pre& P) {
        templ_fetch(R, d.bg(), P.bg);
        stamp_fetch<1>(R, d.sprite(), P.agent);
        if (d.n_coins()) stamp_fetch<1>(R, ST_COIN, P.coin);
        else stamp_none<1>(P.coin);
        if (d.exit_stamp() != 0xFF) stamp_fetch<1>(R, d.exit_stamp(), P.exitp);
        else stamp_none<1>(P.exitp);
        P.holes.hole[0] = P.holes.hole[1] = P.holes.span[0] = P.holes.span[1] = 0u;
        auto hole_at = [&](int h) { return d.hole(h); };
        if (d.alpha() && holes_small(hole_at, d.n_holes())) hole_fetch8(R, hole_at, d.n_holes(), P.holes);
    }
    static __device__ __forceinline__ void prefetch(cptr<Desc> dp, const RasterCtx& R, Pre& P) { prefetch_v(view_of(dp), R, P); }
    static __device__ __forceinline__ void recycle(const RasterCtx& R) { zero_mask(R); }
    static __device__ __forceinline__ void compose(cptr<Desc> dp, const Pre& P, const RasterCtx& R) { compose_v(view_of(dp), P, R); }
    template <class V>
    static __device__ __forceinline__ void compose_v(const V& d, const Pre& P, const RasterCtx& R) {
        const cptr<AtlasTables> T = R.T;
        const uint32_t alpha = d.alpha();
        const StampRegs<1>&agent = P.agent, &coin = P.coin, &exitp = P.exitp;
        uint32_t* const ring = ring_words<BORDER>();
        auto hole_at = [&](int h) { return d.hole(h); };
        const int n_holes = d.n_holes(), n_coins = d.n_coins();
        if (alpha) {  // the hole mask is zero on entry (recycle())
            if (holes_small(hole_at, n_holes)) hole_apply8(R, P.holes);
            else hole_mask(R, hole_at, n_holes);  // radii beyond the reference's range: span table read in place
            if constexpr (BORDER) ring_mask(R, hole_at, n_holes, ring);
            __syncthreads();
        }
        templ_apply_dark(R, P.bg, alpha);
        __syncthreads();
        auto under_bar = [&](int X, int Y) { return Y < BAR_H && bar_covers(d, X); };
        const uint32_t lf = d.coin_above();
        const bool exit_here = d.exit_stamp() != 0xFF;
        const int sx = d.sx(), sy = d.sy(), exit_x = d.exit_x(), exit_y = d.exit_y();
        if constexpr (BORDER) {  // the same layers, with the border pixels blended in where the spotlight layer sits
            if (!(lf & LAYER_COIN_ABOVE))
                for (int k = 0; k < n_coins; ++k) stamp_apply_lit<1>(R, coin, d.coin_x(k), d.coin_y(k), alpha, never_skip);
            if (exit_here && !(lf & LAYER_EXIT_ABOVE)) stamp_apply_lit<1>(R, exitp, exit_x, exit_y, alpha, never_skip);
            __syncthreads();
            if (!(lf & LAYER_AGENT_TOP)) stamp_apply_lit<1>(R, agent, sx, sy, alpha, never_skip);
            __syncthreads();
            if (alpha) {
                ring_apply(R, ring, alpha);
                __syncthreads();
            }
            if (lf & LAYER_COIN_ABOVE)
                for (int k = 0; k < n_coins; ++k) stamp_apply_lit<1>(R, coin, d.coin_x(k), d.coin_y(k), 0u, under_bar);
            if (exit_here && (lf & LAYER_EXIT_ABOVE)) stamp_apply_lit<1>(R, exitp, exit_x, exit_y, 0u, under_bar);
            bar_columns(d, T, R);
            if (lf & LAYER_AGENT_TOP) {
                __syncthreads();
                stamp_apply_lit<1>(R, agent, sx, sy, 0u, never_skip);
            }
            return;
        }
        // coins keep their distance from each other and from the exit (sampler block radius): no overlap among them.
        // coins_visible / exit_visible / agent_visible move a layer from below the dark layer to above it (the agent:
        // to the very top, over the bar -- the reference's list.insert index is past the end of its surface list).
        if (!(lf & LAYER_COIN_ABOVE))
            for (int k = 0; k < n_coins; ++k) stamp_apply_lit<1>(R, coin, d.coin_x(k), d.coin_y(k), alpha, never_skip);
        if (exit_here && !(lf & LAYER_EXIT_ABOVE)) stamp_apply_lit<1>(R, exitp, exit_x, exit_y, alpha, never_skip);
        __syncthreads();
        if (!(lf & (LAYER_COIN_ABOVE | LAYER_EXIT_ABOVE))) {  // the bar follows without a barrier: leave its pixels alone
            if (!(lf & LAYER_AGENT_TOP)) stamp_apply_lit<1>(R, agent, sx, sy, alpha, under_bar);
        } else {
            if (!(lf & LAYER_AGENT_TOP)) stamp_apply_lit<1>(R, agent, sx, sy, alpha, never_skip);
            __syncthreads();
            if (lf & LAYER_COIN_ABOVE)
                for (int k = 0; k < n_coins; ++k) stamp_apply_lit<1>(R, coin, d.coin_x(k), d.coin_y(k), 0u, under_bar);
            if (exit_here && (lf & LAYER_EXIT_ABOVE)) stamp_apply_lit<1>(R, exitp, exit_x, exit_y, 0u, under_bar);
        }
        bar_columns(d, T, R);
        if (lf & LAYER_AGENT_TOP) {
            __syncthreads();
            stamp_apply_lit<1>(R, agent, sx, sy, 0u, never_skip);
        }
    }
};
typedef SpotComposerT<false> SpotComposer;
typedef SpotComposerT<true> SpotBorderComposer;  // black_background has been on: spotlights may have a border

// _build_debug_surface (searing_spotlights.py:157-185, endless_searing_spotlights.py:150-177): board, spotlight layer, then
// exit, coins and agent OVER it (undarkened), top bar last.  Same descriptor, prefetch and hole mask as the observation.
template <bool BORDER>
struct SpotDebugComposerT {
    typedef SpotDesc Desc;
    typedef SpotComposerT<false> Obs;
    typedef Obs::Pre Pre;
    static __device__ __forceinline__ bool skip(cptr<Desc>) { return false; }
    static __device__ __forceinline__ void prefetch(cptr<Desc> dp, const RasterCtx& R, Pre& P) { Obs::prefetch(dp, R, P); }
    static __device__ __forceinline__ void recycle(const RasterCtx& R) { zero_mask(R); }
    static __device__ __forceinline__ void compose(cptr<Desc> dp, const Pre& P, const RasterCtx& R) {
        const SpotViewMem d = view_of(dp);
        const uint32_t alpha = d.alpha();
        uint32_t* const ring = ring_words<BORDER>();
        auto hole_at = [&](int h) { return d.hole(h); };
        if (alpha) {
            if (holes_small(hole_at, d.n_holes())) hole_apply8(R, P.holes);
            else hole_mask(R, hole_at, d.n_holes());
            if constexpr (BORDER) ring_mask(R, hole_at, d.n_holes(), ring);
            __syncthreads();
        }
        templ_apply_dark(R, P.bg, alpha);
        __syncthreads();
        if constexpr (BORDER) {
            if (alpha) {
                ring_apply(R, ring, alpha);
                __syncthreads();
            }
        }
        auto under_bar = [&](int X, int Y) { return Y < BAR_H && Obs::bar_covers(d, X); };
        if (d.exit_stamp() != 0xFF) stamp_apply_lit<1>(R, P.exitp, d.exit_x(), d.exit_y(), 0u, under_bar);
        for (int k = 0; k < d.n_coins(); ++k) stamp_apply_lit<1>(R, P.coin, d.coin_x(k), d.coin_y(k), 0u, under_bar);
        __syncthreads();
        stamp_apply_lit<1>(R, P.agent, d.sx(), d.sy(), 0u, under_bar);
        Obs::bar_columns(d, R.T, R);
    }
};
typedef SpotDebugComposerT<false> SpotDebugComposer;
typedef SpotDebugComposerT<true> SpotBorderDebugComposer;

struct SpotIO {
    SpotCore* core;
    // A spotlight's slot record, [N][16] each: where it is on its way (t, f64), how fast (f64), its three angles in degrees (u32:
    // start | target << 9 | offset << 18, each already % 360) and its radius (u8, bit 7: has_border).  The six end points of
    // Spotlight.__init__ are c + cos/sin(angle) * (half_diag + radius): the same expression gives the same doubles at every step, so
    // they are recomputed from the (cache-resident) trig tables instead of stored -- round 3 kept them (six f64 arrays): 1,312 B of
    // slot state per instance, read by a step kernel that is a burst of cold loads (profiles/r04_spot_step.md); now 336 B.
    // `done` is t == 1.0 (the step clamps t to exactly 1.0 when it raises it, and nothing else writes either).
    double *sp_t, *sp_speed;
    uint32_t* sp_ang;
    uint8_t* sp_r;
    uint32_t* coins;  // [N][MAX_COINS] (x | y<<16), finite variant
    RngSoA rng;
    SpotDesc* desc;
    int* err;
    // resets put off by the step kernel and served inside the raster launch (spot_raster_serve_kernel)
    int* queue;  // [N] instances
    int* qctr;   // SQ_COUNT entries, SQ_LEFT service workgroups that have finished (the last one clears both)
    // per-instance option sets (mg_set_option_set / mg_bind_option_sets): instance i runs under sets[set_of[i]]; both NULL while
    // the handle has ONE set -- the kernels then take the parameters from their arguments
    const SpotParams* sets;
    const int32_t* set_of;
};
constexpr int SQ_COUNT = 0, SQ_LEFT = 32, SQ_WORDS = 64;  // one 128-byte line each
// SpotDesc::valid: 0 = leave the frame alone (masked reset), 1 = draw, 2 = a reset is queued, 3 = reset and drawn by a service
// workgroup.  The frame workgroups of the fused launch draw 1 only, everything else (raster_only, debug view) draws != 0.
constexpr uint32_t DESC_QUEUED = 2, DESC_SERVED = 3;

// floor(sqrt(v)), v < 2^24: single-precision estimate (a double-precision square root is ~20 dependent f64 instructions on this
// chip), made exact by the two integer corrections.
__device__ __forceinline__ int isqrt_floor(int v) {
    int r = (int)__fsqrt_rn((float)v);
    while (r * r > v) --r;
    while ((r + 1) * (r + 1) <= v) ++r;
    return r;
}
// sqrt(dx^2 + dy^2) <= R for integers (Coin / agent distance tests of the reference, computed there in doubles): the square root is
// correctly rounded and monotonic and sqrt(R^2) == R exactly, so the test is d2 <= R^2 -- without the f64 square root.
__device__ __forceinline__ bool within(int dx, int dy, int R) { return dx * dx + dy * dy <= R * R; }

// GridPositionSampler.sample: k-th un-blocked cell (row-major) of the 84x84 grid; discs: (x, y, r) with strict <.
// Each row's blocked set is a union of intervals [cx - hw, cx + hw] with hw = isqrt(r^2 - dy^2 - 1); rows are
// 84-bit masks built with shifts (no per-cell loops), free cells counted with popcounts.
typedef unsigned __int128 u128m;
// The blocked discs (agent, coins, exit; at most 1 + MAX_COINS + 1) of the instance a 16-lane group is resetting live in
// LDS (x[], y[], r[] of MAX_DISCS ints each in the group's DISC_INTS-int slot; all 16 lanes write the same values, each
// reads after its own write).  As register arrays -- compile-time indices under predicates -- they pushed the finite
// variant's fused raster / reset kernel 109 dwords past its 96 VGPRs: 436 B of scratch per lane, which a kernel pays for at
// EVERY wave launch (profiles/r02_spot_resets.md), and unrolled its loops over the discs.
constexpr int MAX_DISCS = 1 + MAX_COINS + 1;
constexpr int DISC_INTS = 3 * MAX_DISCS + MAX_COINS + 2;  // + the coins placed by a finite reset (spot_reset), 16-byte multiple
static_assert(DISC_INTS % 4 == 0, "group slots stay 16-byte aligned");
struct Discs {
    int* p;  // LDS slot of this lane's group
    int n;
    __device__ __forceinline__ void push(int X, int Y, int R) {
        p[n] = X;
        p[MAX_DISCS + n] = Y;
        p[2 * MAX_DISCS + n] = R;
        ++n;
    }
};
// the slot of the calling lane's group inside an array of (workgroup size / 16) * DISC_INTS ints
__device__ __forceinline__ int* disc_slot(int* lds, int grp) { return lds + grp * DISC_INTS; }
// Where a lane sits: its slot id, the group (= instance) of 16 lanes it belongs to within the workgroup, and that group's bit
// position in a wave ballot.
struct LaneCtx {
    int ls, grp, gshift;
};
__device__ __forceinline__ LaneCtx lane_ctx(int tix) { return LaneCtx{tix & 15, tix >> 4, ((tix >> 4) & 3) * 16}; }
__device__ __forceinline__ u128m row_mask(const Discs& D, int y) {
    u128m m = 0;
    for (int d = 0; d < D.n; ++d) {
        const int dx = D.p[d], dr = D.p[2 * MAX_DISCS + d];
        int ddy = y - D.p[MAX_DISCS + d], rem = dr * dr - ddy * ddy - 1;
        if (rem < 0) continue;
        int hw = isqrt_floor(rem);
        int a = dx - hw, b = dx + hw;
        a = a < 0 ? 0 : a;
        b = b > SCREEN - 1 ? SCREEN - 1 : b;
        if (a > b) continue;
        m |= (((u128m)1 << (b + 1)) - 1) ^ (((u128m)1 << a) - 1);
    }
    return m;
}
__device__ __forceinline__ int popc128(u128m m) { return __popcll((unsigned long long)m) + __popcll((unsigned long long)(m >> 64)); }

// Cooperative form: the 16 lanes of an instance call this together (same arguments, same RNG state in every lane).
// Lane ls owns the six rows [6 ls, 6 ls + 6) (lanes 14, 15 idle); free-cell counts are reduced / scanned across the
// group with shuffles, every lane performs the identical draw, and the lane whose rows contain the k-th free cell
// locates it.  A single lane walking all 84 rows twice took ~40 us (the tail of the whole step kernel whenever any
// instance re-spawned its coin).
constexpr int ROWS_PER_LANE = 6;
static_assert(ROWS_PER_LANE * 14 == SCREEN, "14 lanes x 6 rows cover the sampler grid");
// One row of the grid under one or two discs: the blocked cells as two sorted, disjoint intervals [a1, a1 + l1), [a2, a2 + l2)
// (a length of 0 = none; two overlapping or touching spans are returned as one).
__device__ __forceinline__ void row_spans(const Discs& D, int y, int& a1, int& l1, int& a2, int& l2) {
    int lo[2] = {0, 0}, hi[2] = {-1, -1};
#pragma unroll
    for (int d = 0; d < 2; ++d) {
        if (d >= D.n) continue;
        const int dx = D.p[d], dr = D.p[2 * MAX_DISCS + d], ddy = y - D.p[MAX_DISCS + d], rem = dr * dr - ddy * ddy - 1;
        if (rem < 0) continue;
        const int hw = isqrt_floor(rem);
        lo[d] = dx - hw < 0 ? 0 : dx - hw;
        hi[d] = dx + hw > SCREEN - 1 ? SCREEN - 1 : dx + hw;
    }
    const bool e0 = hi[0] >= lo[0], e1 = hi[1] >= lo[1];
    if (e0 && e1 && lo[0] <= hi[1] + 1 && lo[1] <= hi[0] + 1) {  // one span
        a1 = lo[0] < lo[1] ? lo[0] : lo[1];
        l1 = (hi[0] > hi[1] ? hi[0] : hi[1]) - a1 + 1;
        a2 = SCREEN;
        l2 = 0;
        return;
    }
    const bool first0 = e0 && (!e1 || lo[0] < lo[1]);  // which span comes first (an empty one goes last)
    const int fa = first0 ? lo[0] : lo[1], fb = first0 ? hi[0] : hi[1], sa = first0 ? lo[1] : lo[0], sb = first0 ? hi[1] : hi[0];
    const bool fe = first0 ? e0 : e1, se = first0 ? e1 : e0;
    a1 = fe ? fa : SCREEN;
    l1 = fe ? fb - fa + 1 : 0;
    a2 = se ? sa : SCREEN;
    l2 = se ? sb - sa + 1 : 0;
}

__device__ __forceinline__ int sample_cell(Pcg& g, const Discs& D, const LaneCtx& L, int* ox, int* oy) {
    const int ls = L.ls;
    if (D.n == 0) {  // empty mask: cell k itself
        int k = g.integers(0, SCREEN * SCREEN);
        *oy = k / SCREEN;
        *ox = k - *oy * SCREEN;
        return SCREEN * SCREEN;
    }
    const int y0 = ls * ROWS_PER_LANE;
    // One or two discs (the endless variant's coin re-sampling: the collected coin; the finite variant's first coin and, with one
    // coin, its exit: agent, agent + coin): a row's blocked cells are at most two spans -- no 128-bit masks, the k-th free cell by
    // comparisons (round 4; SearingSpotlights-v0 at 4,096 instances, where the launch is as long as one reset: 126 -> 135 M
    // env-steps/s with the one-disc form alone).
    const bool few = D.n <= 2;
    int local_free = 0;
    if (ls < 14) {
        if (few) {
            for (int j = 0; j < ROWS_PER_LANE; ++j) {
                int a1, l1, a2, l2;
                row_spans(D, y0 + j, a1, l1, a2, l2);
                local_free += SCREEN - l1 - l2;
            }
        } else {
            for (int j = 0; j < ROWS_PER_LANE; ++j) local_free += SCREEN - popc128(row_mask(D, y0 + j));
        }
    }
    // inclusive scan over the 16 lanes of the group (width-16 shuffles stay inside the instance's lanes)
    int incl = local_free;
    for (int off = 1; off < 16; off <<= 1) {
        int v = __shfl_up(incl, off, 16);
        if (ls >= off) incl += v;
    }
    const int free_total = __shfl(incl, 15, 16);
    int k = g.integers(0, free_total);  // identical in all 16 lanes
    const int excl = incl - local_free;
    int fx = -1, fy = -1;
    if (few && k >= excl && k < incl) {  // exactly one lane
        int kk = k - excl;
        for (int j = 0; j < ROWS_PER_LANE; ++j) {
            int a1, l1, a2, l2;
            row_spans(D, y0 + j, a1, l1, a2, l2);
            const int fr = SCREEN - l1 - l2;
            if (kk < fr) {
                int x = kk;
                if (x >= a1) x += l1;
                if (x >= a2) x += l2;
                fx = x;
                fy = y0 + j;
                break;
            }
            kk -= fr;
        }
    } else if (k >= excl && k < incl) {  // exactly one lane
        int kk = k - excl;
        for (int j = 0; j < ROWS_PER_LANE; ++j) {
            u128m m = row_mask(D, y0 + j);
            int fr = SCREEN - popc128(m);
            if (kk < fr) {
                // kk-th free cell of this row: skip whole bytes, then single bits
                int x = 0;
                for (;; x += 8) {
                    int zb = 8 - __popc((unsigned)(m >> x) & 0xFFu);
                    if (x + 8 > SCREEN) zb -= x + 8 - SCREEN;  // bits beyond the grid are not cells
                    if (kk < zb) break;
                    kk -= zb;
                }
                for (;; ++x) {
                    if (!((m >> x) & 1)) {
                        if (kk == 0) break;
                        --kk;
                    }
                }
                fx = x;
                fy = y0 + j;
                break;
            }
            kk -= fr;
        }
    }
    const int owner = __ffs((unsigned)(__ballot(fx >= 0) >> L.gshift) & 0xFFFFu) - 1;
    *ox = __shfl(fx, owner, 16);
    *oy = __shfl(fy, owner, 16);
    return free_total;
}

__device__ __forceinline__ void clamp_spawn(const SpotParams& P, int& x, int& y) {
    int off = P.spawn_clamp;
    if (x < off) x = off; else if (x > SCREEN - off) x = SCREEN - off;
    if (y < off) y = off; else if (y > SCREEN - off) y = SCREEN - off;
}

// Spotlight.__init__: 5 draws (radius, speed, start angle, target delta, offset delta)
// `ls` = this lane's slot id: every lane of the instance draws the same numbers, the owner of the chosen slot stores them
// The record of a spotlight as the lane owning its slot holds it in registers.
struct SlotRec {
    double t, speed;
    uint32_t ang;  // start | target << 9 | offset << 18 (degrees, % 360)
    int r;         // bit 7: has_border
};
__device__ __forceinline__ uint32_t pack_angles(int start, int target, int offset) {
    return (uint32_t)(start % 360) | ((uint32_t)(target % 360) << 9) | ((uint32_t)(offset % 360) << 18);
}
// cos / sin of integer degrees (host-built tables in global memory, see SpotFamily)
struct Trig {
    const double* c;
    const double* s;
};
// Spotlight.__init__: 5 draws (radius, speed, start angle, target delta, offset delta).  `ls` = this lane's slot id: every lane of
// the instance draws the same numbers; the lane owning the slot the free mask hands out stores the record AND gets it back in
// `rec` (reading it back from memory is two round trips in every wave in which any instance spawned, i.e. in every launch).
// Returns false when all SLOTS slots are taken (the reference's list is unbounded, endless_searing_spotlights.py:191): the draws are
// consumed, no spotlight is added, error bit 1 is raised -- and the step that called ends the episode (mg_info_buffers.capacity_dev).
__device__ __forceinline__ bool new_spot(const SpotParams& P, const SpotIO& io, int i, int ls, SpotCore& s, Pcg& g, SlotRec* rec = nullptr) {
    int radius = g.integers(P.r_lo, P.r_hi);
    double speed = g.uniform(P.speed_lo, P.speed_hi);
    int start = g.integers(0, 360);
    int target = start + 180 + g.integers(-45, 45);
    int offset = target + g.integers(-135, 135);
    if (s.n_spots >= SLOTS || s.free_mask == 0) {
        raise_error(io.err, 1);
        return false;
    }
    int slot = __ffs(s.free_mask) - 1;
    s.free_mask &= ~(1u << slot);
    s.order |= (uint64_t)slot << (4 * s.n_spots);
    s.n_spots++;
    if (slot != ls) return true;
    size_t k = (size_t)i * SLOTS + slot;
    SlotRec n;
    n.r = radius | (P.black_background ? 0x80 : 0);  // bit 7: Spotlight.has_border
    n.t = 0.0;
    n.speed = speed;
    n.ang = pack_angles(start, target, offset);
    io.sp_r[k] = (uint8_t)n.r;
    io.sp_t[k] = n.t;
    io.sp_speed[k] = n.speed;
    io.sp_ang[k] = n.ang;
    if (rec) *rec = n;
    return true;
}

// The spotlights a reset starts with (free_mask == 0xFFFF: the q-th takes slot q, which lane q owns): 5 draws each, one after another
// in the generator's stream -- 20 of the ~27 draws of a reset, ~3 of the ~4 us by which a resetting instance's wave outlasts the
// others (profiles/r04_spot_step.md).  The stream's NEXT 16 outputs do not have to be produced one after another: PCG64's state
// after k steps is A^k s + S_k inc (S_k = 1 + A + ... + A^(k-1)), so lane j of the instance's 16 computes output j + 1 directly
// (two 128-bit multiplications with its pair of constants, P.jump) and lane q < count picks the three outputs spotlight q
// consumes.  Which halves feed which draw depends on whether the stream arrives with a buffered half (numpy's next_uint32 hands
// out the low half of a fresh 64-bit output and keeps the high half; Generator.uniform takes a fresh output and leaves the
// buffer alone):   buffered:  radius <- the buffered half (spotlight 0: the stream's own; q > 0: the high half of output 3q),
//                             speed <- output 3q+1, start <- low(3q+2), target <- high(3q+2), offset <- low(3q+3);
//                  otherwise: radius <- low(3q+1), speed <- output 3q+2, start <- high(3q+1), target <- low(3q+3), offset <- high(3q+3).
// Either way a spotlight consumes three outputs, the stream ends with the same "buffered" flag it came with, and the buffer word holds
// the high half of output 3 count (numpy keeps a used half in place).  This holds as long as no bounded draw is rejected (Lemire:
// possible only when the low word of the product is below the range, ~1e-7 per draw): a lane that sees such a low word makes the
// whole group fall back to the one-after-another form below from the untouched stream -- that form is the definition.
// Returns false if it did nothing (the caller then runs the loop over new_spot()).
__device__ __forceinline__ bool new_spots_at_reset(const SpotParams& P, const SpotIO& io, int i, const LaneCtx& L, SpotCore& s, Pcg& g, int count,
                                                   const uint4 jm, const uint4 jq) {
    const uint32_t n_r = (uint32_t)(P.r_hi - P.r_lo);
    if (count < 1 || count > 5 || n_r < 2u) return false;  // (16 outputs = 5 spotlights; a one-value radius range draws nothing)
    const int ls = L.ls;
    const u128 M = ((u128)jm.w << 96) | ((u128)jm.z << 64) | ((u128)jm.y << 32) | jm.x;
    const u128 S = ((u128)jq.w << 96) | ((u128)jq.z << 64) | ((u128)jq.y << 32) | jq.x;
    const u128 st = M * g.state + S * g.inc;  // the state after ls + 1 steps
    uint32_t lo, hi;
    {
        const uint64_t h = (uint64_t)(st >> 64), l = (uint64_t)st, x = h ^ l;
        const unsigned rot = (unsigned)(h >> 58);
        const uint64_t o = (x >> rot) | (x << ((64 - rot) & 63));
        lo = (uint32_t)o;
        hi = (uint32_t)(o >> 32);
    }
    const int q = ls < count ? ls : 0;  // this lane's spotlight (lanes >= count follow spotlight 0 and store nothing)
    const uint32_t lo1 = __shfl(lo, 3 * q, 16), hi1 = __shfl(hi, 3 * q, 16);
    const uint32_t lo2 = __shfl(lo, 3 * q + 1, 16), hi2 = __shfl(hi, 3 * q + 1, 16);
    const uint32_t lo3 = __shfl(lo, 3 * q + 2, 16), hi3 = __shfl(hi, 3 * q + 2, 16);
    const uint32_t hi0 = __shfl(hi, q > 0 ? 3 * q - 1 : 0, 16);
    const bool buffered = g.has;
    const uint32_t x_radius = buffered ? (q > 0 ? hi0 : g.buf) : lo1;
    const uint64_t x_speed = buffered ? (((uint64_t)hi1 << 32) | lo1) : (((uint64_t)hi2 << 32) | lo2);
    const uint32_t x_start = buffered ? lo2 : hi1;
    const uint32_t x_target = buffered ? hi2 : lo3;
    const uint32_t x_offset = buffered ? lo3 : hi3;
    const uint64_t m_radius = (uint64_t)x_radius * n_r, m_start = (uint64_t)x_start * 360u, m_target = (uint64_t)x_target * 90u,
                   m_offset = (uint64_t)x_offset * 270u;
    const bool maybe_rejected = (uint32_t)m_radius < n_r || (uint32_t)m_start < 360u || (uint32_t)m_target < 90u || (uint32_t)m_offset < 270u ||
                                (P.lab_fallback > 0 && i % P.lab_fallback == 0);
    if (((uint32_t)(__ballot(maybe_rejected) >> L.gshift) & 0xFFFFu) != 0u) return false;
    // the stream after 3 count outputs (all 16 lanes hold the same copy)
    {
        const int last = 3 * count - 1;
        const uint32_t a = __shfl((uint32_t)st, last, 16), b = __shfl((uint32_t)(st >> 32), last, 16);
        const uint32_t c = __shfl((uint32_t)(st >> 64), last, 16), d = __shfl((uint32_t)(st >> 96), last, 16);
        g.state = ((u128)d << 96) | ((u128)c << 64) | ((u128)b << 32) | a;
        g.buf = __shfl(hi, last, 16);
    }
    s.n_spots = (uint8_t)count;
    s.free_mask = 0xFFFFu & ~((1u << count) - 1u);
    s.order = 0x43210ull & ((1ull << (4 * count)) - 1ull);
    if (ls < count) {  // Spotlight.__init__ of spotlight ls, as in new_spot()
        const int radius = P.r_lo + (int)(m_radius >> 32);
        const double speed = P.speed_lo + (P.speed_hi - P.speed_lo) * ((double)(x_speed >> 11) * (1.0 / 9007199254740992.0));
        const int start = (int)(m_start >> 32);
        const int target = start + 180 + (-45 + (int)(m_target >> 32));
        const int offset = target + (-135 + (int)(m_offset >> 32));
        const size_t k = (size_t)i * SLOTS + ls;
        io.sp_r[k] = (uint8_t)(radius | (P.black_background ? 0x80 : 0));
        io.sp_t[k] = 0.0;
        io.sp_speed[k] = speed;
        io.sp_ang[k] = pack_angles(start, target, offset);
    }
    return true;
}

template <bool EN>
__device__ __forceinline__ void fill_topbar(const SpotParams& P, const SpotCore& s, SpotDesc& d, bool reset_frame, int a0, int a1) {
    d.c_base = EN ? C_BLACK : C_GREY50;
    d.red_w = s.red_w;
    d.quarter = (uint8_t)P.quarter;
    d.c_act0 = d.c_act1 = 0xFF;
    if (P.show_last_action) {  // 0 -> grey, 1 -> purple, 2 -> orange
        d.c_act0 = a0 == 0 ? C_GREY120 : (a0 == 1 ? C_PURPLE : C_ACT_ORANGE);
        d.c_act1 = a1 == 0 ? C_GREY120 : (a1 == 1 ? C_PURPLE : C_ACT_ORANGE);
    }
    d.c_bar = 0xFF;
    d.bar_x = (uint8_t)P.bar_x;
    d.bar_w = (uint8_t)P.bar_w;
    if (!reset_frame && P.show_last_positive_reward) d.c_bar = s.last_pos ? C_YELLOW : C_GREY50;
}

// ENDLESS is a compile-time flag: the endless instantiation has no run-time indexed local arrays (coin lists), so the
// descriptor and the state stay in registers -- with both variants in one kernel they lived in 176 B of scratch per lane.
// stale_holes: the spotlight surface is NOT repainted by reset() (searing_spotlights.py:394-397 only set its alpha), so
// the first frame of an episode shows the holes of the last frame drawn before it; they only show when the alpha is not
// 0 at reset, i.e. with light_dim_off_duration == 0.  The hole words themselves are still in the descriptor.
template <bool EN>
__device__ __forceinline__ void spot_reset(const SpotParams& P, const SpotIO& io, int i, const LaneCtx& L, SpotCore& s, Pcg& g, SpotDesc& d, float* gt,
                                           int stale_holes, int* slot) {  // slot: disc_slot() of the calling kernel's LDS array
    const int ls = L.ls;
    const uint4 jm = P.jump[2 * ls], jq = P.jump[2 * ls + 1];  // (new_spots_at_reset: requested here, used after the first draws)
    s.t = 0;
    s.coin_t = 0;
    s.ep_sum = 0.0;
    s.ep_len = 0;
    s.la0 = s.la1 = 0;
    s.rot8 = (uint8_t)g.integers(0, 8);  // choice([0, 45, ..., 315])
    Discs D;
    D.p = slot;
    D.n = 0;
    int ax, ay;
    if (P.sample_agent_position) {
        int k = g.integers(0, SCREEN * SCREEN);  // sampler with an empty mask: cell k itself
        int cy = k / SCREEN, cx = k - cy * SCREEN;
        D.push(cx, cy, 28);
        ax = cx + g.integers(2, 4);
        ay = cy + g.integers(2, 4);
    } else {
        ax = SCREEN / 2;
        ay = SCREEN / 2;
        D.push(ax, ay, 21);
    }
    s.ax = (int16_t)ax;
    s.ay = (int16_t)ay;
    s.health = P.agent_health;
    s.red_w = 0;
    s.last_pos = 0;
    s.alpha = (uint8_t)(P.dim_duration > 0 ? 0 : (P.light_threshold > 255 ? 255 : (P.light_threshold < 0 ? 0 : P.light_threshold)));
    s.n_spots = 0;
    s.order = 0;
    s.free_mask = 0xFFFFu;
    s.spawn_timer = 0;
    s.n_intervals = (uint8_t)P.num_spawns;
    if (!new_spots_at_reset(P, io, i, L, s, g, P.initial_spawns, jm, jq))
        for (int k = 0; k < P.initial_spawns; ++k) new_spot(P, io, i, ls, s, g);
    s.coins_collected = 0;
    s.n_coins = 0;
    s.has_coin = 0;
    uint32_t* coins = io.coins + (size_t)i * MAX_COINS;
    int* const coin_pos = slot + 3 * MAX_DISCS;  // the coins as placed (finite variant), next to the disc list
    if constexpr (EN) {
        if (P.coin_enabled) {  // _spawn_coin: the sampler is reset first, self.coin is None -> nothing blocked
            int k = g.integers(0, SCREEN * SCREEN);
            int cy = k / SCREEN, cx = k - cy * SCREEN;
            cx += g.integers(2, 4);
            cy += g.integers(2, 4);
            clamp_spawn(P, cx, cy);
            s.coin_x = (int16_t)cx;
            s.coin_y = (int16_t)cy;
            s.has_coin = 1;
            s.n_coins = 1;
        }
    } else {
        int nc = P.num_coins.n > 0 ? choice(g, P.num_coins) : 0;
        s.num_coins = nc;
        for (int k = 0; k < nc && k < MAX_COINS; ++k) {  // deliberately not unrolled (code size)
            int cx, cy;
            sample_cell(g, D, L, &cx, &cy);
            D.push(cx, cy, 21);
            cx += g.integers(2, 4);
            cy += g.integers(2, 4);
            clamp_spawn(P, cx, cy);
            const uint32_t w = (uint32_t)(cx & 0xFFFF) | ((uint32_t)cy << 16);
            if (ls == 0) coins[k] = w;
            coin_pos[k] = (int)w;
            s.n_coins++;
        }
        if (P.use_exit) {  // _spawn_exit (searing_spotlights.py:280-286)
            int ex, ey;
            sample_cell(g, D, L, &ex, &ey);
            ex += g.integers(2, 4);
            ey += g.integers(2, 4);
            clamp_spawn(P, ex, ey);
            s.exit_x = (int16_t)ex;
            s.exit_y = (int16_t)ey;
            s.exit_open = 0;
            s.pad = (s.pad & ~PAD_EXIT_GEN_MASK) | ((uint32_t)P.exit_gen << PAD_EXIT_GEN_SHIFT) | PAD_HAS_EXIT;
        } else if (!(s.pad & PAD_HAS_EXIT)) {
            // use_exit == False: nothing is spawned, sampled or drawn (:413-416) and the frame keeps blitting self.exit -- the Exit
            // of an earlier episode, where it was and as it was last drawn (open / closed).  Without one the reference raises
            // AttributeError at this reset (:431-435); here the bit is raised and no exit is drawn.
            raise_error(io.err, ERR_NO_EXIT);
        }
    }
    s.bg_red = 0;
    if (P.hide_chessboard) s.pad = bg_set(bg_set(s.pad, 0, BG_WHITE), 1, BG_WHITE);  // (the reference does this first thing: no draw depends on it)
    if (P.black_background) s.pad = bg_set(s.pad, 0, BG_BLACK);

    // reset frame: blue board, sprite index 0 (not the sampled rotation), dark layer at the reset alpha with the hole
    // pattern the previous frame left, coin(s) shown above the dark layer while coin_t < coin_show_duration
    memset(&d, 0, sizeof(d));
    d.valid = 1;
    d.bg = bg_template(s.pad, 0);
    d.sprite = 0;
    d.sx = (int16_t)(ax - P.sprite_half);
    d.sy = (int16_t)(ay - P.sprite_half);
    d.alpha = s.alpha;
    d.n_holes = (uint8_t)stale_holes;
    d.exit_stamp = 0xFF;
    if constexpr (EN) {
        d.n_coins = s.n_coins;
        d.coin_above = (uint8_t)(((P.coins_visible || s.coin_t < P.coin_show_duration) ? LAYER_COIN_ABOVE : 0) | P.layer_flags);
        d.coins[0] = (uint32_t)(s.coin_x - P.coin_radius + 128) | ((uint32_t)(s.coin_y - P.coin_radius + 128) << 16);
    } else {
        d.n_coins = s.n_coins;
        d.coin_above = (uint8_t)((P.coins_visible ? LAYER_COIN_ABOVE : 0) | P.layer_flags);
#pragma unroll
        for (int k = 0; k < MAX_COINS; ++k) {
            if (k < s.n_coins) {
                const uint32_t w = (uint32_t)coin_pos[k];
                int cx = (int)(int16_t)(w & 0xFFFF), cy = (int)(w >> 16);
                d.coins[k] = (uint32_t)(cx - P.coin_radius + 128) | ((uint32_t)(cy - P.coin_radius + 128) << 16);
            }
        }
        {
            const int eg = exit_gen_of(s.pad), half = (int)((P.exit_halves >> (8 * eg)) & 0xFFu);
            d.exit_stamp = (s.pad & PAD_HAS_EXIT) ? (uint8_t)(ST_EXIT0 + 2 * eg + (s.exit_open ? 1 : 0)) : 0xFF;
            d.exit_x = (int16_t)(s.exit_x - half);
            d.exit_y = (int16_t)(s.exit_y - half);
        }
    }
    fill_topbar<EN>(P, s, d, true, 0, 0);
    if (gt) {
        gt[0] = (float)((double)ax / SCREEN);
        gt[1] = (float)((double)ay / SCREEN);
        gt[2] = (float)(P.coin_enabled ? (double)s.coin_x / SCREEN : 0.0);
        gt[3] = (float)(P.coin_enabled ? (double)s.coin_y / SCREEN : 0.0);
    }
}

// info["ground_truth"] in float64: agent and coin position / screen size (endless_searing_spotlights.py:407,496)
__global__ __launch_bounds__(256) void spot_gt64_kernel(SpotParams P0, SpotIO io, double* out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P0.n) return;
    const SpotParams& P = io.set_of ? io.sets[set_index(io.set_of, i)] : P0;
    const SpotCore s = io.core[i];
    out[4 * i + 0] = (double)s.ax / SCREEN;
    out[4 * i + 1] = (double)s.ay / SCREEN;
    out[4 * i + 2] = P.coin_enabled ? (double)s.coin_x / SCREEN : 0.0;
    out[4 * i + 3] = P.coin_enabled ? (double)s.coin_y / SCREEN : 0.0;
}

__global__ __launch_bounds__(256) void spot_init_kernel(int n, SpotCore* core) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    SpotCore s;
    memset(&s, 0, sizeof(s));
    core[i] = s;
}

// the leader stores the descriptor's header + coin positions (words 0..5 and 8..15); hole words are written by the slot
// lanes.  Packed field by field (the layout of the bit-fields above) so that `d` never has to exist in memory.
__device__ __forceinline__ void store_desc_head(SpotDesc* dst, const SpotDesc& d) {
    uint4* out = reinterpret_cast<uint4*>(dst);
    const uint32_t w0 = (uint32_t)d.valid | ((uint32_t)d.bg << 8) | ((uint32_t)d.sprite << 16) | ((uint32_t)d.alpha << 24);
    const uint32_t w1 = ((uint32_t)d.sx & 0xFFFFu) | ((uint32_t)d.sy << 16);
    const uint32_t w2 = (uint32_t)d.n_holes | ((uint32_t)d.n_coins << 8) | ((uint32_t)d.coin_above << 16) | ((uint32_t)d.red_w << 24);
    const uint32_t w3 = (uint32_t)d.c_base | ((uint32_t)d.c_act0 << 8) | ((uint32_t)d.c_act1 << 16) | ((uint32_t)d.c_bar << 24);
    const uint32_t w4 = (uint32_t)d.bar_x | ((uint32_t)d.bar_w << 8) | ((uint32_t)d.quarter << 16) | ((uint32_t)d.exit_stamp << 24);
    const uint32_t w5 = ((uint32_t)d.exit_x & 0xFFFFu) | ((uint32_t)d.exit_y << 16);
    out[0] = make_uint4(w0, w1, w2, w3);
    reinterpret_cast<uint2*>(dst)[2] = make_uint2(w4, w5);
    out[2] = make_uint4(d.coins[0], d.coins[1], d.coins[2], d.coins[3]);
    out[3] = make_uint4(d.coins[4], d.coins[5], d.coins[6], d.coins[7]);
}
static_assert(MAX_COINS == 8, "store_desc_head packs eight coin words");

// PS: per-instance option sets -- the parameters come from memory, io.sets[set_index(io.set_of, i)], instead of from the kernel arguments
template <bool EN, bool PS>
__global__ __launch_bounds__(256) void spot_reset_kernel(SpotParams P0, SpotIO io, const int64_t* seeds, const uint8_t* mask,
                                                         float* gt) {
    __shared__ int disc_lds[(256 / 16) * DISC_INTS];
    int gid = blockIdx.x * blockDim.x + threadIdx.x;
    int i = gid >> 4, ls = gid & 15;
    if (i >= P0.n) return;
    const SpotParams& P = PS ? io.sets[set_index(io.set_of, i)] : P0;
    if (mask && !mask[i]) {
        if (ls == 0) io.desc[i].valid = 0;
        return;
    }
    Pcg g;
    if (seeds) g.seed((uint64_t)seeds[i]);
    else g.load(io.rng, i);
    SpotCore s = io.core[i];
    SpotDesc d;
    const int stale_holes = (int)(reinterpret_cast<const uint32_t*>(&io.desc[i])[2] & 0xFFu);  // n_holes of the frame drawn last
    const LaneCtx L = lane_ctx((int)threadIdx.x);
    spot_reset<EN>(P, io, i, L, s, g, d, (gt && EN && ls == 0) ? gt + 4 * i : nullptr, stale_holes, disc_slot(disc_lds, L.grp));
    if (ls == 0) {
        io.core[i] = s;
        g.store(io.rng, i);
        store_desc_head(&io.desc[i], d);
    }
}

// What a step needs besides the instance: ONE struct, the head of the kernel-argument segment of both step kernels.
struct SpotStepArgs {
    SpotParams P;
    SpotIO io;
    const int32_t* actions;
    float* reward_out;
    uint8_t* done_out;
    float* gt;
    mg_info_buffers info;
    int autoreset, defer;
};

// The step of instance i as its 16 lanes execute it (lane ls owns spotlight slot ls).  The launch lasts as long as its slowest wave
// (all waves of a 16,384-instance launch are resident at once): every load the step can need is requested up front -- core record,
// generator stream, the lane's 17-byte slot record -- and the rare paths (spawn, coin re-sampling, reset) are kept short
// (profiles/r04_spot_step.md: 20.5 -> 14 us).  The core record lives in LDS (round 4: 123 -> 69-88 VGPRs).
#ifdef MG_LAB_SPOT_CLOCK  // measurement builds only (tools/spot_step_timeline.py): eight stamps + flags per wave of the step kernel
static __device__ unsigned long long g_lab_spot_clock[10 * 65536];
#define SPOT_CLOCK(slot) do { clk[slot] = (unsigned long long)clock64(); } while (0)
#else
#define SPOT_CLOCK(slot) do { } while (0)
#endif
template <bool EN, bool PS>
__device__ __forceinline__ void spot_step_body(int i, const LaneCtx& L, const SpotStepArgs& a, int* disc_lds, SpotCore* core_lds, const Trig& T) {
    const int ls = L.ls;
#ifdef MG_LAB_SPOT_CLOCK
    unsigned long long clk[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const unsigned long long wall0 = wall_clock64();
    bool f_spawn = false, f_coin = false;
#endif
    SPOT_CLOCK(0);
    const SpotIO& io = a.io;
    const SpotParams& P = PS ? io.sets[set_index(io.set_of, i)] : a.P;  // (PS: per-instance option sets)
    const int32_t* const actions = a.actions;
    float* const reward_out = a.reward_out;
    uint8_t* const done_out = a.done_out;
    float* const gt = a.gt;
    const mg_info_buffers& info = a.info;
    const int autoreset = a.autoreset, defer = a.defer;
    const int group_shift = L.gshift;  // bit position of this instance's 16 lanes in a wave ballot
    const bool leader = ls == 0;
    // The instance's core record lives in LDS for the length of the step (its 16 lanes write the same values to the same words):
    // twenty registers less at every point of a kernel that has to share its register budget with the raster (round 4:
    // 89 -> 65-73 VGPRs for this body), for a handful of LDS round trips on its critical path.
    SpotCore& s = core_lds[L.grp];
    s = io.core[i];
    // Everything the step may need is requested HERE, together: the generator's stream (spawns, coin re-sampling and resets are
    // rare, but each used to start with a memory round trip of its own, and a wave with two of them is what the launch waits
    // for) and this lane's slot record (16 bytes + a byte now).  Timeline per wave: profiles/r04_spot_step.md.
    Pcg g;
    g.load(io.rng, i);
    bool rng_used = false;  // (all 16 lanes of the instance take the same branches: they hold the same state)
    uint32_t* coins = io.coins + (size_t)i * MAX_COINS;
    const size_t k = (size_t)i * SLOTS + ls;  // lane ls looks after slot ls
    SlotRec mine;
    mine.t = io.sp_t[k];
    mine.speed = io.sp_speed[k];
    mine.ang = io.sp_ang[k];
    mine.r = io.sp_r[k];
    g.pin();

    // CharacterController.step(action, walkable_rect = (0, 4, 84, 80))
    int a0 = actions[2 * i], a1 = actions[2 * i + 1];
    int ax = s.ax, ay = s.ay;
    free_move(a0, a1, P.v_axis_i, P.v_diag_i, ax, ay, s.rot8, true, P.agent_radius, SCREEN - P.agent_radius, P.bar_h + P.agent_radius,
              SCREEN - P.agent_radius);
    s.ax = (int16_t)ax;
    s.ay = (int16_t)ay;
#ifdef MG_LAB_SPOT_CLOCK
    asm volatile("" ::"v"(ax), "v"(ay));
#endif
    SPOT_CLOCK(1);
    // the top bar shows the PREVIOUS action
    int shown0 = s.la0, shown1 = s.la1;
    if (EN || P.show_last_action) {
        s.la0 = (uint8_t)a0;
        s.la1 = (uint8_t)a1;
    }
    // dim the light until off
    if ((int)s.alpha <= P.light_threshold) {
        int a = P.dim_duration > 0 ? (int)s.alpha + P.dim_step : P.light_threshold;
        s.alpha = (uint8_t)(a > 255 ? 255 : (a < 0 ? 0 : a));  // Surface.set_alpha clamps
    }

    SpotDesc d;
    memset(&d, 0, sizeof(d));
    d.valid = 1;

    // ---- spotlight task ----
    double reward = 0.0, r = 0.0;
    bool spot_done = false, cap = false;  // cap: a spotlight was due and all slots are taken -- this step ends the episode (new_spot)
    s.spawn_timer++;
    if constexpr (EN) {
        if (__builtin_expect(s.spawn_timer >= P.spawn_interval, 0)) {
#ifdef MG_LAB_SPOT_CLOCK
            f_spawn = true;
#endif
            rng_used = true;
            cap = !new_spot(P, io, i, ls, s, g, &mine);
            s.spawn_timer = 0;
        }
    } else if (s.n_intervals > 0) {
        if (__builtin_expect(s.spawn_timer >= P.interval0, 0)) {
            rng_used = true;
            cap = !new_spot(P, io, i, ls, s, g, &mine);
            s.n_intervals--;
            s.spawn_timer = 0;
        }
    }
    SPOT_CLOCK(2);
    const int p_r = mine.r;  // bit 7: has_border
    const bool p_done = mine.t >= 1.0;
    const bool used = !((s.free_mask >> ls) & 1u);
    const bool my_done = used && p_done;
    const uint32_t done_mask = (uint32_t)(__ballot(my_done) >> group_shift) & 0xFFFFu;
    // `for spot in self.spotlights: if spot.done: self.spotlights.remove(spot) else: draw + hit test`:
    // removing while iterating skips the element that follows a removed one (it stays in the list untouched)
    uint32_t processed = 0;
    {
        uint64_t new_order = 0;
        int n_new = 0, n_old = s.n_spots;
        for (int pos = 0; pos < n_old;) {
            int slot = (int)((s.order >> (4 * pos)) & 15u);
            if ((done_mask >> slot) & 1u) {
                s.free_mask |= 1u << slot;
                if (pos + 1 < n_old) {
                    int nxt = (int)((s.order >> (4 * (pos + 1))) & 15u);
                    new_order |= (uint64_t)nxt << (4 * n_new++);
                }
                pos += 2;
            } else {
                processed |= 1u << slot;
                new_order |= (uint64_t)slot << (4 * n_new++);
                pos += 1;
            }
        }
        s.order = new_order;
        s.n_spots = (uint8_t)n_new;
    }
    bool my_hit = false;
    if ((processed >> ls) & 1u) {
        const int radius0 = p_r & 127;
        const double R = P.half_diag + (double)radius0, c = SCREEN / 2;  // Spotlight.__init__'s end points (see SpotIO)
        const int a_s = (int)(mine.ang & 511u), a_t = (int)((mine.ang >> 9) & 511u), a_o = (int)(mine.ang >> 18);
        const double p_sx = c + T.c[a_s] * R, p_sy = c + T.s[a_s] * R, p_tx = c + T.c[a_t] * R, p_ty = c + T.s[a_t] * R;
        const double p_ox = c + T.c[a_o] * R, p_oy = c + T.s[a_o] * R;
        double t = mine.t;
        double lx = p_tx * (1 - t) + p_ox * t, ly = p_ty * (1 - t) + p_oy * t;
        double cx = p_sx * (1 - t) + lx * t, cy = p_sy * (1 - t) + ly * t;
        const int radius = p_r & 127;
        int rank = __popc(processed & ((1u << ls) - 1u));
        if (P.ordered_holes) {  // a border is drawn over the discs before it and under the discs after it: list order
            rank = 0;
            for (int pos = 0; pos < (int)s.n_spots; ++pos) {
                const int slot = (int)((s.order >> (4 * pos)) & 15u);
                if (slot == ls) break;
                rank += (int)((processed >> slot) & 1u);
            }
        }
        io.desc[i].holes[rank] = pack_hole((int)cx, (int)cy, radius) | ((uint32_t)(p_r >> 7) << 31);
        t += mine.speed;
        if (t >= 1.0) t = 1.0;  // = done: removed from the list by the next step
        io.sp_t[k] = t;
        double ddx = (double)ax - cx, ddy = (double)ay - cy;
        my_hit = sqrt(ddx * ddx + ddy * ddy) <= (double)(radius + P.agent_radius);
    }
    const int hit = __popc((uint32_t)(__ballot(my_hit) >> group_shift) & 0xFFFFu);
    const int nh = __popc(processed);
#ifdef MG_LAB_SPOT_CLOCK
    asm volatile("" ::"v"(hit));
#endif
    SPOT_CLOCK(3);
    if (hit > 0) {
        s.health -= P.damage;
        r += P.r_inside;
        s.red_w = (uint8_t)(int)((SCREEN / 2) * (1 - s.health / P.agent_health));
        s.bg_red = P.visual_feedback ? 1 : 0;
    } else {
        s.bg_red = 0;
        r += P.r_outside;
    }
    if (P.black_background) s.pad = bg_set(s.pad, s.bg_red, BG_BLACK);  // bg.fill(0): that surface stays black
    if (s.health <= 0) {
        spot_done = true;
        r += P.r_death;
    }
    reward += r;

    // ---- coin / exit tasks ----
    bool done = false;
    int success = 0;
    uint32_t coin_pos[MAX_COINS] = {0, 0, 0, 0, 0, 0, 0, 0};
    if constexpr (EN) {
        if (P.coin_enabled) {
            double cr = 0.0;
            if (__builtin_expect(within(ax - (int)s.coin_x, ay - (int)s.coin_y, P.coin_radius + P.agent_radius), 0)) {
                cr += P.r_coin;
                s.coins_collected++;
                s.coin_t = 0;
                // _spawn_coin: sampler reset, previous coin blocked with r = 28
#ifdef MG_LAB_SPOT_CLOCK
                f_coin = true;
#endif
                rng_used = true;
                Discs D;
                D.p = disc_slot(disc_lds, L.grp);
                D.n = 0;
                D.push(s.coin_x, s.coin_y, 28);
                int cx, cy;
                sample_cell(g, D, L, &cx, &cy);
                cx += g.integers(2, 4);
                cy += g.integers(2, 4);
                clamp_spawn(P, cx, cy);
                s.coin_x = (int16_t)cx;
                s.coin_y = (int16_t)cy;
            }
            reward += cr;
        }
        if (spot_done) done = true;
        s.t++;
        s.coin_t++;
        if (s.coin_t == P.steps_per_coin && P.coin_enabled) done = true;
        if (s.t == P.max_steps) done = true;
    } else {
        bool coins_done;
        {  // the instance's coin list as two 16-byte loads (eight predicated dword loads were issued one after another)
            const uint4 c0 = reinterpret_cast<const uint4*>(coins)[0], c1 = reinterpret_cast<const uint4*>(coins)[1];
            const uint32_t cw[MAX_COINS] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
#pragma unroll
            for (int q = 0; q < MAX_COINS; ++q) coin_pos[q] = q < s.n_coins ? cw[q] : 0u;
        }
        if (s.num_coins > 0) {
            double cr = 0.0;
#pragma unroll
            for (int q = 0; q < MAX_COINS; ++q) {  // remove-while-iterating: the coin after a collected one is skipped
                if (q >= s.n_coins) break;
                int cx = (int)(int16_t)(coin_pos[q] & 0xFFFF), cy = (int)(coin_pos[q] >> 16);
                if (within(ax - cx, ay - cy, P.coin_radius + P.agent_radius)) {
#pragma unroll
                    for (int j = q; j < MAX_COINS - 1; ++j)
                        if (j < s.n_coins - 1) coin_pos[j] = coin_pos[j + 1];
                    s.n_coins--;
                    cr += P.r_coin;
                    s.coins_collected++;
                }
            }
            coins_done = s.n_coins == 0;
            reward += cr;
        } else {
            coins_done = true;
        }
        bool exit_done = false;
        double er = 0.0;
        if (coins_done && P.use_exit) {  // _step_exit_task (:313-330)
            s.exit_open = 1;
            double ddx = (double)ax - (double)s.exit_x, ddy = (double)ay - (double)s.exit_y;
            if (sqrt(ddx * ddx + ddy * ddy) <= P.exit_radius + (double)P.agent_radius) {
                exit_done = true;
                er = P.r_exit;
            }
        }
        reward += er;
        if (spot_done) done = true;
        else if (coins_done && (P.use_exit ? exit_done : s.num_coins > 0)) { done = true; success = 1; }  // (:499-511)
        s.t++;
        if (s.t == P.max_steps) done = true;
    }
    done = done || cap;
#ifdef MG_LAB_SPOT_CLOCK
    asm volatile("" ::"v"(done));
#endif
    SPOT_CLOCK(4);
    bool shown_last_pos = s.last_pos;
    if (P.show_last_positive_reward) s.last_pos = reward > 0 ? 1 : 0;
    s.ep_sum += reward;
    s.ep_len++;

    if (done && leader) {
        if (info.ep_reward_dev) info.ep_reward_dev[i] = s.ep_sum;
        if (info.ep_length_dev) info.ep_length_dev[i] = s.ep_len;
        if (info.aux_dev[0]) info.aux_dev[0][i] = (float)(s.health / P.agent_health);
        if constexpr (EN) {
            if (info.aux_dev[1]) info.aux_dev[1][i] = (float)s.coins_collected;
        } else {
            if (info.aux_dev[1]) info.aux_dev[1][i] = (float)((double)s.coins_collected / (double)s.num_coins);
            if (info.aux_dev[2]) info.aux_dev[2][i] = (float)success;
        }
    }
    if (leader) {
        reward_out[i] = (float)reward;
        if (info.reward64_dev) info.reward64_dev[i] = reward;  // the reference's Python float, unrounded
        done_out[i] = done ? 1 : 0;
        if (info.capacity_dev) info.capacity_dev[i] = cap ? 1 : 0;
    }

    // debug view only: the (rotated_agent_surface, rotated_agent_rect) pair of this step -- a reset leaves it alone, and the
    // reference's debug render shows that stale pair until the first step of the next episode
    s.pad = (s.pad & PAD_STICKY) | 0x80000000u | ((uint32_t)s.rot8 << 16) | (uint32_t)((ax + 128) & 0xFF) | ((uint32_t)((ay + 128) & 0xFF) << 8);
    // defer: the reset (position sampling on 84x84 masks: ~30 us for the 16 lanes of the instance, the tail of this launch
    // whenever any instance finishes) is queued and done by a service workgroup of the raster launch, which also draws the
    // frame; state, stream and the descriptor head (its n_holes are the reset frame's stale holes) are stored as after
    // any other step, exactly what a masked mg_reset(seed = None) would find.
    const bool reset_me = done && autoreset;
    SPOT_CLOCK(5);
    if (defer && reset_me && leader) queue_push(io.queue, &io.qctr[SQ_COUNT], P.n, i, io.err);
    if (__builtin_expect(reset_me && !defer, 0)) {  // cold: keep the reset code out of the hot instruction stream
        rng_used = true;
        spot_reset<EN>(P, io, i, L, s, g, d, (gt && EN && leader) ? gt + 4 * i : nullptr, nh, disc_slot(disc_lds, L.grp));
    } else {
        d.bg = bg_template(s.pad, s.bg_red);
        d.sprite = s.rot8;
        d.sx = (int16_t)(ax - P.sprite_half);
        d.sy = (int16_t)(ay - P.sprite_half);
        d.alpha = s.alpha;
        d.n_holes = (uint8_t)nh;
        d.exit_stamp = 0xFF;
        if constexpr (EN) {
            d.n_coins = P.coin_enabled ? 1 : 0;
            d.coin_above = (uint8_t)(((P.coins_visible || s.coin_t < P.coin_show_duration) ? LAYER_COIN_ABOVE : 0) | P.layer_flags);
            d.coins[0] = (uint32_t)(s.coin_x - P.coin_radius + 128) | ((uint32_t)(s.coin_y - P.coin_radius + 128) << 16);
        } else {
            d.n_coins = s.n_coins;
            d.coin_above = (uint8_t)((P.coins_visible ? LAYER_COIN_ABOVE : 0) | P.layer_flags);
#pragma unroll
            for (int q = 0; q < MAX_COINS; ++q) {
                if (q < s.n_coins) {
                    int cx = (int)(int16_t)(coin_pos[q] & 0xFFFF), cy = (int)(coin_pos[q] >> 16);
                    d.coins[q] = (uint32_t)(cx - P.coin_radius + 128) | ((uint32_t)(cy - P.coin_radius + 128) << 16);
                }
            }
            if (leader) {  // entries beyond n_coins are dead; written as two 16-byte stores
                reinterpret_cast<uint4*>(coins)[0] = make_uint4(coin_pos[0], coin_pos[1], coin_pos[2], coin_pos[3]);
                reinterpret_cast<uint4*>(coins)[1] = make_uint4(coin_pos[4], coin_pos[5], coin_pos[6], coin_pos[7]);
            }
            const int eg = exit_gen_of(s.pad), half = (int)((P.exit_halves >> (8 * eg)) & 0xFFu);
            d.exit_stamp = (s.pad & PAD_HAS_EXIT) ? (uint8_t)(ST_EXIT0 + 2 * eg + (s.exit_open ? 1 : 0)) : 0xFF;
            d.exit_x = (int16_t)(s.exit_x - half);
            d.exit_y = (int16_t)(s.exit_y - half);
        }
        if (reset_me) d.valid = DESC_QUEUED;
        SpotCore tb = s;
        tb.last_pos = shown_last_pos;  // the bar shows whether the PREVIOUS reward was positive
        fill_topbar<EN>(P, tb, d, false, shown0, shown1);
        if (gt && EN && leader) {
            gt[4 * i + 0] = (float)((double)ax / SCREEN);
            gt[4 * i + 1] = (float)((double)ay / SCREEN);
            gt[4 * i + 2] = (float)(P.coin_enabled ? (double)s.coin_x / SCREEN : 0.0);
            gt[4 * i + 3] = (float)(P.coin_enabled ? (double)s.coin_y / SCREEN : 0.0);
        }
    }
    SPOT_CLOCK(6);
    if (leader) {
        if (rng_used) g.store(io.rng, i);
        io.core[i] = s;
        store_desc_head(&io.desc[i], d);
    }
#ifdef MG_LAB_SPOT_CLOCK
    __builtin_amdgcn_s_waitcnt(0);
    SPOT_CLOCK(7);
    {
        const unsigned long long any_spawn = __ballot(f_spawn) != 0, any_coin = __ballot(f_coin) != 0, any_reset = __ballot(reset_me && !defer) != 0;
        const int wave = i >> 2;
        if ((threadIdx.x & 63) == 0 && wave < 65536) {
            unsigned long long* o = g_lab_spot_clock + 10 * (size_t)wave;
            for (int q = 0; q < 8; ++q) o[q] = clk[q];
            o[8] = any_spawn | (any_coin << 1) | (any_reset << 2);
            o[9] = (wall0 & 0xFFFFFFFFull) | (wall_clock64() << 32);
        }
    }
#endif
}

#ifdef MG_LAB
// Measurement (lab build, MEMGYM_SPOT_WARM=1; profiles/r06_spot.md): a launch behind the raster that reads exactly what the NEXT step
// kernel's lanes will read first -- core record, generator stream, slot record -- with the same block -> instance mapping, so that the
// step finds its state in its XCD's L2.  Only the step kernel's time is of interest (would a warm state be worth building into the
// raster launch's tail?); this launch's own cost is not hidden.
__global__ __launch_bounds__(256) void spot_warm_kernel(int n, SpotIO io) {
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    const int i = gid >> 4, ls = gid & 15;
    if (i >= n) return;
    const size_t k = (size_t)i * SLOTS + ls;
    const uint32_t* c = reinterpret_cast<const uint32_t*>(&io.core[i]);
    uint64_t acc = c[ls] ^ c[(ls + 16) % 20];
    acc ^= io.rng.s_hi[i] ^ io.rng.s_lo[i] ^ io.rng.inc_hi[i] ^ io.rng.inc_lo[i] ^ io.rng.buf[i];
    acc ^= (uint64_t)io.sp_ang[k] ^ (uint64_t)io.sp_r[k] ^ (uint64_t)__double_as_longlong(io.sp_t[k]) ^ (uint64_t)__double_as_longlong(io.sp_speed[k]);
    if (acc == 0x1234567812345678ull) io.err[0] |= 0;  // (never: keeps the loads alive)
}
#endif

template <bool EN, bool PS>
__global__ __launch_bounds__(256) void spot_step_kernel(SpotStepArgs a) {
    __shared__ int disc_lds[(256 / 16) * DISC_INTS];  // step_block() launches 256 lanes at most
    __shared__ SpotCore core_lds[256 / 16];
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    const int i = gid >> 4;
    // (a copy of the trig tables in LDS -- 5.8 KB per workgroup, one barrier -- measured: step kernel 19.9 -> 21.7 us, nothing gained)
    if (i < a.P.n) spot_step_body<EN, PS>(i, lane_ctx((int)threadIdx.x), a, disc_lds, core_lds, Trig{a.P.cos_tab, a.P.sin_tab});
}

// The step's raster launch with the put-off resets served inside it: the first workgroups take the queue entries, eight
// each (the 16 lanes of a quarter wave reset one instance, like spot_reset_kernel; waves 2 and 3 wait), then draw those
// eight frames; all other workgroups walk the frames of the instances that were not queued (SpotDesc::valid == 1).  The
// descriptors a service workgroup has just written are read back through the scalar cache like every descriptor: release,
// barrier, s_dcache_inv first.  Service workgroups without an entry leave at once and issue no atomic (thousands of them on
// one address: 22 ns each, in series).  What bounds the launch is a reset's latency next to the raster's waves (~45-80 us)
// plus the frames that follow it in the same workgroup; variants measured: profiles/r02_spot_resets.md.
#ifndef MG_SPOT_SVC_BATCH
#define MG_SPOT_SVC_BATCH 8
#endif
#ifndef MG_SPOT_SVC_WGS
#define MG_SPOT_SVC_WGS 512
#endif
constexpr int SPOT_SVC_WGS = MG_SPOT_SVC_WGS, SPOT_SVC_BATCH = MG_SPOT_SVC_BATCH;
static_assert(SPOT_SVC_BATCH * (DISC_INTS * 4 + (int)sizeof(SpotCore)) <= FRAME_BYTES, "a service batch's disc lists and core records fit into the frame area");
// Workgroups per CU of the fused launch (round 4, profiles/r04_spot_serve.md): SIX, non-temporal stores.  Rounds 2-3 ran it at five (96
// VGPRs and 104-124 B of scratch, 28 KiB of LDS): with the core record of a reset in LDS and the arguments of the service loop
// read where they are used, the endless variant needs 80 VGPRs and no scratch, the finite one 80 + 88-100 B.
#ifndef MG_SPOT_SERVE_OCC
#define MG_SPOT_SERVE_OCC 6
#endif
#define MG_KERNARG_AS __attribute__((address_space(4)))
// all arguments in one struct = the kernel-argument segment: the service workgroups read theirs through a pointer the compiler cannot
// see through, where they are used (held in scalar registers for the length of the service loop they spilled into vector lanes)
struct SpotServeArgs {
    const SpotDesc* descs;
    RasterAtlas A;
    void* obs;
    int n;
    SpotParams P;
    SpotIO io;
    float* gt;
    // Resets a serving workgroup takes per round: as few as serve every queued instance in ONE round (the launch is as long as a
    // reset plus the frames its workgroup draws behind it: 4,096 instances 74 -> 117 M env-steps/s with one instead of eight),
    // within [batch_min, batch_max] (host: 1 .. 8 up to 12,288 instances -- a step in which every instance is truncated at once
    // still takes few rounds -- and 8 beyond, where eight measured 1-2 % ahead of the adaptive choice).  profiles/r04_spot_step.md section 4.
    int batch_min, batch_max;
    void* final_obs;  // FINAL form (terminal observations kept, mg_info_buffers.final_obs_dev), else NULL
};
// FINAL (round 6): a call in the gymnasium vector convention.  The step kernel has stored a finishing instance's state and frame descriptor
// "as after any other step" (valid = DESC_QUEUED): that descriptor IS the terminal frame's -- the service workgroup draws it into final_obs
// before it resets the instance and draws the new episode's first frame into obs.  A kernel of its own; the measured ones are as they were.
template <bool EN, bool BORDER, bool NT, bool FINAL = false>
__global__ __launch_bounds__(256, MG_SPOT_SERVE_OCC) void spot_raster_serve_kernel(SpotServeArgs a) {
    typedef SpotComposerT<BORDER> Composer;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    RasterCtx R;
    R.frame = smem;
    R.mask = reinterpret_cast<uint32_t*>(smem + FRAME_BYTES);
    R.A = a.A;
    R.T = as_const(a.A.tables);
    R.tid = threadIdx.x;
    const int tid = threadIdx.x;
    const int n = a.n;
    void* const obs = a.obs;
    const cptr<SpotDesc> cdescs = as_const(a.descs);
    const bool service = (int)blockIdx.x < SPOT_SVC_WGS;
    const int count = service ? queue_count(&a.io.qctr[SQ_COUNT], n) : 0;
    int batch = (count + SPOT_SVC_WGS - 1) / SPOT_SVC_WGS;
    batch = batch < a.batch_min ? a.batch_min : (batch > a.batch_max ? a.batch_max : batch);
    if (service && (int)blockIdx.x * batch >= count) return;
    Composer::recycle(R);
    __syncthreads();
    auto draw = [&](cptr<SpotDesc> from, int env) {
        typename Composer::Pre Pq;
        Composer::prefetch(from + env, R, Pq);
        Composer::compose(from + env, Pq, R);
        __syncthreads();
        Composer::recycle(R);
        store_frame<MG_OBS_U8_XYC, NT, true>(smem, obs, env, tid);
        __syncthreads();
    };
    if (service) {
        for (int base = blockIdx.x * batch; base < count; base += SPOT_SVC_WGS * batch) {
            const SpotServeArgs MG_KERNARG_AS* ka = (const SpotServeArgs MG_KERNARG_AS*)__builtin_amdgcn_kernarg_segment_ptr();
            asm volatile("" : "+s"(ka));
            const SpotParams& P = *(const SpotParams*)&ka->P;
            const SpotIO& io = *(const SpotIO*)&ka->io;
            float* const gt = ka->gt;
            if constexpr (FINAL) {  // the terminal frames of this round's instances, from the descriptors the step kernel left
                // (the draw lambda's body once more with another target: as one lambda with a target argument, or one lambda calling the
                // other, every variant of the kernel took 96-112 B of scratch and the frame loop ran three times as long)
                void* const fin = ka->final_obs;
                for (int k = 0; k < batch && base + k < count; ++k) {
                    const int env = io.queue[base + k];
                    typename Composer::Pre Pq;
                    Composer::prefetch(cdescs + env, R, Pq);
                    Composer::compose(cdescs + env, Pq, R);
                    __syncthreads();
                    Composer::recycle(R);
                    store_frame<MG_OBS_U8_XYC, NT, true>(smem, fin, env, tid);
                    __syncthreads();
                }
            }
            const int e = base + (tid >> 4), ls = tid & 15;
            if (tid < 16 * batch && e < count) {
                const int i = io.queue[e];
                Pcg g;
                g.load(io.rng, i);
                // disc lists and core records of the batch: in the FRAME area -- nothing of this workgroup is being composed while it
                // resets (the barriers around draw() separate the two uses) -- so the launch asks for no more LDS than the raster alone;
                // the core record in LDS instead of registers is what lets this kernel run at the raster's occupancy (round 4)
                const LaneCtx L = lane_ctx(tid);
                SpotCore& s = reinterpret_cast<SpotCore*>(smem + SPOT_SVC_BATCH * DISC_INTS * 4)[L.grp];
                s = io.core[i];
                SpotDesc d;
                const int stale_holes = (int)(reinterpret_cast<const uint32_t*>(&io.desc[i])[2] & 0xFFu);
                spot_reset<EN>(P, io, i, L, s, g, d, (gt && EN && ls == 0) ? gt + 4 * i : nullptr, stale_holes,
                               disc_slot(reinterpret_cast<int*>(smem), L.grp));
                d.valid = DESC_SERVED;
                if (ls == 0) {
                    io.core[i] = s;
                    g.store(io.rng, i);
                    store_desc_head(&io.desc[i], d);
                }
            }
            // The descriptors just stored are read back by THIS workgroup's composers through the scalar cache: the stores have to
            // have reached the L2 (s_waitcnt vmcnt(0); the vector L1 writes through) and the scalar cache
            // must not answer from an older copy (s_dcache_inv).  NOT __threadfence(): at agent scope that is buffer_wbl2 +
            // buffer_inv -- a write-back of the whole L2, which holds the launch's observation stream (round 4: the cost of the
            // launch grew with the number of workgroups that served resets, profiles/r04_spot_serve.md).
            // Round 6 (a race the round-4 form had, found by tools/vector_soak.py: one reset frame in ~10^7 drawn from the OLD descriptor or
            // from a half-written one): the workgroup-scope release fence this stood on compiles to s_waitcnt lgkmcnt(0) only -- outside
            // tgsplit mode the vector L1 is coherent among a workgroup's waves, so LLVM's memory model leaves vmcnt out -- but the readers here
            // are SCALAR loads, which bypass the vector L1 and could reach the L2 before the stores did.  So: wait for the stores' acknowledgement
            // by hand, and for the invalidation (an SMEM operation, asynchronous like any other) before the first scalar load is issued;
            // the descriptor pointer passes through an opaque copy behind it, so that no load of the (constant-address-space, "invariant")
            // descriptor can be scheduled above the invalidation.
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            __builtin_amdgcn_s_dcache_inv();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            cptr<SpotDesc> fresh = cdescs;
            asm volatile("" : "+s"(fresh));
            for (int k = 0; k < batch && base + k < count; ++k) draw(fresh, io.queue[base + k]);
        }
        const int busy = (count + batch - 1) / batch < SPOT_SVC_WGS ? (count + batch - 1) / batch : SPOT_SVC_WGS;
        if (tid == 0 && atomicAdd(&a.io.qctr[SQ_LEFT], 1) == busy - 1) {  // last service workgroup out
            a.io.qctr[SQ_COUNT] = 0;
            a.io.qctr[SQ_LEFT] = 0;
        }
        return;
    }
    const int stride = (int)gridDim.x - SPOT_SVC_WGS;
    for (int env = (int)blockIdx.x - SPOT_SVC_WGS; env < n; env += stride) {
        if (cdescs[env].valid != 1u) continue;  // masked, or drawn by the workgroup that serves its reset
        draw(cdescs, env);
    }
}

// Debug view: the current descriptors with the agent the reference's debug render shows -- the stored (sprite, rect) pair of
// the last STEP (stale right after a reset; oracle/mgo_spot.c sp_debug), sprite 0 at the agent's rect before any step.
__global__ __launch_bounds__(256) void spot_debug_desc_kernel(SpotParams P0, SpotIO io, SpotDesc* out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P0.n) return;
    const SpotParams& P = io.set_of ? io.sets[set_index(io.set_of, i)] : P0;
    SpotDesc d = io.desc[i];
    const SpotCore s = io.core[i];
    d.valid = 1;
    if (s.pad >> 31) {
        d.sprite = (s.pad >> 16) & 7u;
        d.sx = (int16_t)((int)(s.pad & 0xFFu) - 128 - P.sprite_half);
        d.sy = (int16_t)((int)((s.pad >> 8) & 0xFFu) - 128 - P.sprite_half);
    }
    out[i] = d;
}

// ---------------------------------------------------------------------------------------------------------
static const double SCALE = 0.25;

// sin / cos by two separate libm calls (a merged sincos() differs by 1 ulp for a few integer-degree angles)
static __attribute__((noinline)) double sin_only(double x) { volatile double v = x; return std::sin(v); }
static __attribute__((noinline)) double cos_only(double x) { volatile double v = x; return std::cos(v); }

class SpotFamily : public Family {
   public:
    SpotFamily(int endless, int n) : opt_(one_set()), P_(opt_[0]->P), n_(n) {
        memset(&P_, 0, sizeof(P_));
        P_.endless = endless;
        P_.n = n;
        P_.speed_lo = 0.0025; P_.speed_hi = 0.0075; P_.damage = 1.0;
        P_.visual_feedback = 1; P_.light_threshold = 255;
        coin_scale_ = 1.5 * SCALE; agent_speed_ = 12.0 * SCALE; agent_scale_ = 1.0 * SCALE; exit_scale_ = 2.0 * SCALE;
        P_.sample_agent_position = 1; P_.show_last_action = 1; P_.show_last_positive_reward = 1;
        P_.r_coin = 0.25;
        if (endless) {
            P_.max_steps = -1; P_.steps_per_coin = 160; P_.initial_spawns = 3; P_.spawn_interval = 50;
            P_.coin_enabled = 1; P_.coin_show_duration = 6; P_.agent_health = 10;
        } else {
            P_.max_steps = 256; P_.initial_spawns = 4; P_.num_spawns = 30;
            initial_spawn_interval_ = 30; spawn_interval_threshold_ = 10;
            opt_[0]->st_num_coins.set(P_.num_coins, {1}); P_.agent_health = 5; P_.r_exit = 1.0; P_.use_exit = 1;
        }
        core_.alloc(n);
        for (auto* a : {&sp_t_, &sp_speed_}) a->alloc((size_t)SLOTS * n);
        sp_ang_.alloc((size_t)SLOTS * n);
        sp_r_.alloc((size_t)SLOTS * n);
        flags_.alloc(4);
        exit_hist_.alloc(1 + EXIT_GENS);
        queue_.alloc((size_t)n + SQ_WORDS);

        coins_.alloc((size_t)MAX_COINS * n);
        desc_.alloc(n);
        rng_.alloc(n);
        err_.alloc();
        std::vector<double> ct(360), st(360);
        for (int a = 0; a < 360; ++a) {
            if (a % 90 == 0) {
                static const double C4[4] = {1, 0, -1, 0}, S4[4] = {0, 1, 0, -1};
                ct[a] = C4[a / 90];
                st[a] = S4[a / 90];
            } else {
                double rad = (double)a * M_PI / 180.0;
                ct[a] = cos_only(rad);
                st[a] = sin_only(rad);
            }
        }
        cos_.upload(ct);
        {   // s_k = A^k s_0 + S_k inc for k = 1 .. 16 (PCG64's 128-bit LCG, multiplier as in mg_device.hpp Pcg::advance)
            const u128 A = (((u128)0x2360ED051FC65DA4ull) << 64) | (u128)0x4385DF649FCCF645ull;
            std::vector<uint4> jt(32);
            u128 m = 1, q = 0;
            for (int k = 0; k < 16; ++k) {
                q = q * A + 1;  // S_(k+1) = S_k A + 1
                m = m * A;      // A^(k+1)
                jt[2 * k] = make_uint4((uint32_t)m, (uint32_t)(m >> 32), (uint32_t)(m >> 64), (uint32_t)(m >> 96));
                jt[2 * k + 1] = make_uint4((uint32_t)q, (uint32_t)(q >> 32), (uint32_t)(q >> 64), (uint32_t)(q >> 96));
            }
            jump_.upload(jt);
        }
        sin_.upload(st);
        sets_dev_.alloc(MG_MAX_OPTION_SETS);
        hipLaunchKernelGGL(spot_init_kernel, dim3((n + 255) / 256), dim3(256), 0, 0, n, core_.p);
        MG_HIP(hipDeviceSynchronize());
        rebuild();
        defaults_ = P_;  // (short lists only: no device arrays behind them)
    }

    int action_dim() const override { return 2; }
    int gt_dim() const override { return P_.endless ? 4 : 0; }
    const char* info_name(int k) const override {
        if (k == 0) return "agent_health";
        if (k == 1) return "coins_collected";
        if (k == 2 && !P_.endless) return "success";
        return nullptr;
    }

    // One key of the reset options, for option set `set` (0 = the handle-wide set of mg_set_option).  Sets > 0 hold everything
    // that does not change the geometry (atlases, tables and derived constants are shared by the handle's instances); a geometry
    // option is accepted there when it says what the handle's geometry already is.
    void set_option(const std::string& key, const double* v, int n) override { set_option_set(0, key, v, n); }
    void set_option_set(int set, const std::string& key, const double* v, int n) override {
        if (set < 0 || set >= MG_MAX_OPTION_SETS) throw OptionError{-3, "option set index out of range"};
        while ((int)opt_.size() <= set) {  // a new set starts from the constructor's defaults (= the reference's), geometry from set 0
            opt_.emplace_back(new SpotOpt());
            opt_.back()->P = defaults_;
            copy_geometry(opt_.back()->P, P_);
            derive(*opt_.back());
        }
        SpotOpt& O = *opt_[set];
        SpotParams& P = O.P;
        const bool e = P_.endless;
        // handle-wide values (members of the family) that fix geometry: `member = value` in set 0, "must already be so" elsewhere
        auto G = [&](double& member, double value) {
            if (set == 0) { member = value; dirty_ = true; }
            else if (member != value) throw OptionError{-3, "reset parameter " + key + " changes the geometry shared by the handle's instances: it can only be set for all of them (option set 0)"};
        };
        auto GI = [&](int& member, int value) {
            double m = member;
            G(m, (double)value);
            member = (int)m;
        };
        sets_dirty_ = true;
        auto I = [&](int& dst) { dst = to_int_checked(v[0], key.c_str()); };
        auto B = [&](int& dst) { dst = v[0] != 0.0; };
        auto must_be = [&](bool ok) { if (!ok) throw OptionError{-3, "reset parameter " + key + ": this value is not supported by the MI355X build"}; };
        if (key == "max_steps") I(P.max_steps);
        else if (key == "initial_spawns") { I(P.initial_spawns); must_be(P.initial_spawns >= 0 && P.initial_spawns <= SLOTS); }
        else if (key == "spot_min_radius") { O.min_radius = v[0]; derive(O); }
        else if (key == "spot_max_radius") { O.max_radius = v[0]; derive(O); }
        else if (key == "spot_min_speed") P.speed_lo = v[0];
        else if (key == "spot_max_speed") P.speed_hi = v[0];
        else if (key == "spot_damage") P.damage = v[0];
        else if (key == "visual_feedback") B(P.visual_feedback);
        else if (key == "black_background") {
            B(P.black_background);
            if (P.black_background && !P_.ordered_holes) {  // from now on spotlights may carry a border (sticky, kept in the state)
                P_.ordered_holes = 1;
                const int one = 1;
                MG_HIP(hipMemcpy(flags_.p, &one, sizeof(int), hipMemcpyHostToDevice));
            }
        }
        else if (key == "hide_chessboard") B(P.hide_chessboard);
        else if (key == "light_dim_off_duration") { O.dim_duration = to_int_checked(v[0], key.c_str()); derive(O); }
        else if (key == "light_threshold") I(P.light_threshold);
        else if (key == "coin_scale") G(coin_scale_, v[0]);
        else if (key == "coins_visible") B(P.coins_visible);
        else if (key == "agent_speed") G(agent_speed_, v[0]);
        else if (key == "agent_health") P.agent_health = v[0];
        else if (key == "agent_scale") G(agent_scale_, v[0]);
        else if (key == "agent_visible") P.layer_flags = (P.layer_flags & ~LAYER_AGENT_TOP) | (v[0] != 0.0 ? LAYER_AGENT_TOP : 0);
        else if (key == "sample_agent_position") B(P.sample_agent_position);
        else if (key == "show_last_action") {
            // (the last-reward bar's position and width depend on it, searing_spotlights.py:385-390: geometry)
            GI(P_.show_last_action, v[0] != 0.0 ? 1 : 0);
            // False crashes the ENDLESS reference at its first step (endless_searing_spotlights.py:422 reads action_colors,
            // which :343 only creates when the flag is set); the finite env guards the use (searing_spotlights.py:465)
            if (e) must_be(v[0] != 0.0);
        }
        else if (key == "show_last_positive_reward") B(P.show_last_positive_reward);
        else if (key == "reward_inside_spotlight") P.r_inside = v[0];
        else if (key == "reward_outside_spotlight") P.r_outside = v[0];
        else if (key == "reward_death") P.r_death = v[0];
        else if (key == "reward_coin") P.r_coin = v[0];
        else if (e && key == "steps_per_coin") I(P.steps_per_coin);
        else if (e && key == "spawn_interval") I(P.spawn_interval);
        else if (e && key == "coin_enabled") B(P.coin_enabled);
        else if (e && key == "coin_show_duration") I(P.coin_show_duration);
        else if (!e && key == "num_spawns") { I(P.num_spawns); must_be(P.num_spawns >= 0 && P.num_spawns <= 255); }
        else if (!e && key == "initial_spawn_interval") G(initial_spawn_interval_, v[0]);
        else if (!e && key == "spawn_interval_threshold") G(spawn_interval_threshold_, v[0]);
        else if (!e && key == "spawn_interval_decay") { /* only intervals[0] is ever read (pop() takes the last) */ }
        else if (!e && key == "num_coins") {
            // any length (searing_spotlights.py:408); the empty list is refused by mg_set_option: the reference ends every such
            // episode with a ZeroDivisionError (searing_spotlights.py:553)
            std::vector<int> vals(n);
            for (int k = 0; k < n; ++k) {
                vals[k] = to_int_checked(v[k], key.c_str());
                must_be(vals[k] >= 1 && vals[k] <= MAX_COINS);
            }
            O.st_num_coins.set(P.num_coins, vals);
        }
        // False: legal once the instance has had an exit (its stale one is drawn, spot_reset); before that the reference raises
        // AttributeError and the reset raises error bit 256
        else if (!e && key == "use_exit") B(P.use_exit);
        else if (!e && key == "exit_scale") G(exit_scale_, v[0]);
        else if (!e && key == "exit_visible") P.layer_flags = (P.layer_flags & ~LAYER_EXIT_ABOVE) | (v[0] != 0.0 ? LAYER_EXIT_ABOVE : 0);
        else if (!e && key == "reward_exit") P.r_exit = v[0];
        else if (!e && key == "reward_max_steps") {}
        else throw OptionError{-2, "unknown reset parameter " + key};
    }
    // instance i runs under option set set_of_dev[i] (device array [num_envs], caller-owned; NULL: every instance under set 0)
    void bind_option_sets(const int32_t* set_of_dev) override { set_of_ = set_of_dev; }

    void reset(const int64_t* seeds, const uint8_t* mask, void* obs, float* gt, hipStream_t s) override {
        if (dirty_) rebuild();
        if (!seeds && !seeded_) throw std::runtime_error("reset(seed=None) before any seeded reset");
        for (auto& O : opt_) {   // 16 spotlight slots per instance.  Refuse option sets that overflow them in ANY episode that lasts as long
            // as the fastest spotlight lives (t reaches 1 after ceil(1 / speed) steps, the slot is freed one step later);
            // rarer overflows raise error bit 1, which mg_peek_errors shows without a synchronisation.
            const SpotParams& Q = O->P;
            if (Q.r_hi <= Q.r_lo) throw OptionError{-3, "spot radius range not supported"};
            const int life_min = (int)std::ceil(1.0 / Q.speed_hi) + 1;
            const int interval = P_.endless ? Q.spawn_interval : P_.interval0;
            int later = interval > 0 ? (life_min - 1) / interval : 1 << 20;
            if (!P_.endless && later > Q.num_spawns) later = Q.num_spawns;
            if (Q.initial_spawns + later > SLOTS)
                throw std::runtime_error("these options keep " + std::to_string(Q.initial_spawns + later) +
                                         " spotlights alive at once; this build holds " + std::to_string(SLOTS) +
                                         " per instance (raise spawn_interval / spot_max_speed or lower initial_spawns)");
        }
        if (seeds) seeded_ = true;
        upload_sets(s);
        const dim3 rg((n_ * SLOTS + 255) / 256);
        if (P_.endless) {
            if (per_set()) hipLaunchKernelGGL((spot_reset_kernel<true, true>), rg, dim3(256), 0, s, P_, io(), seeds, mask, gt);
            else hipLaunchKernelGGL((spot_reset_kernel<true, false>), rg, dim3(256), 0, s, P_, io(), seeds, mask, gt);
        } else {
            if (per_set()) hipLaunchKernelGGL((spot_reset_kernel<false, true>), rg, dim3(256), 0, s, P_, io(), seeds, mask, gt);
            else hipLaunchKernelGGL((spot_reset_kernel<false, false>), rg, dim3(256), 0, s, P_, io(), seeds, mask, gt);
        }
        if (mask && sparse_masked_raster()) {  // few frames of many: by the mask, not by a walk over every descriptor (mg_raster.hpp)
            if (P_.ordered_holes) launch_raster_sparse<SpotBorderComposer>(desc_.p, atlas_->dev(), obs, obs_format, n_, s, mask);
            else launch_raster_sparse<SpotComposer>(desc_.p, atlas_->dev(), obs, obs_format, n_, s, mask);
            MG_HIP(hipGetLastError());
        } else raster(obs, s);
    }

    void step(const int32_t* actions, void* obs, float* reward, uint8_t* done, float* gt, const mg_info_buffers* info,
              int autoreset, hipStream_t s) override {
        if (dirty_) throw std::runtime_error("options that change geometry need a reset before the next step");
        mg_info_buffers ib;
        memset(&ib, 0, sizeof(ib));
        if (info) ib = *info;
        upload_sets(s);
        prof.begin(0, s);
        // (resets served inside the raster launch: handles with ONE option set -- the service code takes its parameters from the launch's arguments)
        // (a call that keeps terminal observations: always deferred -- the service workgroup draws the terminal frame from the descriptor this
        // launch leaves, then resets: keeps_final_obs)
        const bool keep_final = autoreset && ib.final_obs_dev && keeps_final_obs(s);
        const int defer = (autoreset && obs_format == MG_OBS_U8_XYC && (fuse_resets() || keep_final) && !per_set()) ? 1 : 0;
        const int sb = step_block(256);
        const SpotStepArgs sa{P_, io(), actions, reward, done, gt, ib, autoreset, defer};
        const dim3 sg((n_ * SLOTS + sb - 1) / sb);
        if (P_.endless) {
            if (per_set()) hipLaunchKernelGGL((spot_step_kernel<true, true>), sg, dim3(sb), 0, s, sa);
            else hipLaunchKernelGGL((spot_step_kernel<true, false>), sg, dim3(sb), 0, s, sa);
        } else {
            if (per_set()) hipLaunchKernelGGL((spot_step_kernel<false, true>), sg, dim3(sb), 0, s, sa);
            else hipLaunchKernelGGL((spot_step_kernel<false, false>), sg, dim3(sb), 0, s, sa);
        }
        end_logic(s);
        prof.begin(1, s);
        if (defer) {
            const int grid = (n_ < raster_grid(n_) ? n_ : raster_grid(n_)) + SPOT_SVC_WGS;
            const int forced_batch = lab_int("MEMGYM_SPOT_SVC_BATCH", 0);  // (lab build: exactly this many)
            const bool fb = forced_batch >= 1 && forced_batch <= SPOT_SVC_BATCH;
            const SpotServeArgs va{desc_.p, atlas_->dev(), obs, n_, P_, io(), gt, fb ? forced_batch : (n_ <= 12288 ? 1 : SPOT_SVC_BATCH), fb ? forced_batch : SPOT_SVC_BATCH,
                                   keep_final ? ib.final_obs_dev : nullptr};
            const bool nt = fused_nt();                // non-temporal: with plain stores the fused launch loses 5-15 us at every occupancy
            const int serve_lds = RASTER_LDS_REQUEST;  // 25 KiB: six per CU
#define SPOT_FUSED2(EN, BO, NT) do { if (keep_final) hipLaunchKernelGGL((spot_raster_serve_kernel<EN, BO, NT, true>), dim3(grid), dim3(256), serve_lds, s, va); \
                                     else hipLaunchKernelGGL((spot_raster_serve_kernel<EN, BO, NT>), dim3(grid), dim3(256), serve_lds, s, va); } while (0)
#define SPOT_FUSED(EN, BO) do { if (nt) SPOT_FUSED2(EN, BO, true); else SPOT_FUSED2(EN, BO, false); } while (0)
            if (P_.endless) { if (P_.ordered_holes) SPOT_FUSED(true, true); else SPOT_FUSED(true, false); }
            else { if (P_.ordered_holes) SPOT_FUSED(false, true); else SPOT_FUSED(false, false); }
#undef SPOT_FUSED
#undef SPOT_FUSED2
            MG_HIP(hipGetLastError());
        } else {
            raster(obs, s);
        }
        prof.end(1, s);
#ifdef MG_LAB
        if (lab_int("MEMGYM_SPOT_WARM", 0)) hipLaunchKernelGGL(spot_warm_kernel, sg, dim3(sb), 0, s, n_, io());
#endif
    }

    std::vector<std::pair<void*, size_t>> state_blobs() override {
        std::vector<std::pair<void*, size_t>> v = {{core_.p, core_.bytes()}, {coins_.p, coins_.bytes()}, {sp_r_.p, sp_r_.bytes()},
                                                  {sp_ang_.p, sp_ang_.bytes()}};
        for (auto* a : {&sp_t_, &sp_speed_}) v.push_back({a->p, a->bytes()});
        v.push_back({flags_.p, flags_.bytes()});
        v.push_back({exit_hist_.p, exit_hist_.bytes()});
        rng_.blobs(v);
        return v;
    }
    void debug_rng(int i, uint64_t out[6]) override { rng_.debug(i, out); }
    void ground_truth64(double* out, hipStream_t s) override {
        if (!gt_dim() || !out) return;
        upload_sets(s);
        hipLaunchKernelGGL(spot_gt64_kernel, dim3((n_ + 255) / 256), dim3(256), 0, s, P_, io(), out);
        MG_HIP(hipGetLastError());
    }
    int poll_errors() override {
        MG_HIP(hipDeviceSynchronize());
        return err_.take();
    }
    int peek_errors() override { return err_.peek(); }

   private:
    SpotIO io() {
        SpotIO o;
        o.core = core_.p;
        o.sp_t = sp_t_.p; o.sp_speed = sp_speed_.p; o.sp_ang = sp_ang_.p; o.sp_r = sp_r_.p;
        o.coins = coins_.p;
        o.rng = rng_.view();
        o.desc = desc_.p;
        o.err = err_.dev;
        o.queue = queue_.p + SQ_WORDS;
        o.qctr = queue_.p;
        o.sets = per_set() ? sets_dev_.p : nullptr;
        o.set_of = per_set() ? set_of_ : nullptr;
        return o;
    }

    void rebuild() {
        int radius = 0;
        std::vector<Stamp> sprites = build_agent_sprites(agent_scale_, &radius);
        P_.agent_radius = radius;
        {
            const int full = sprites[0].w;  // blit position of the un-cropped surface: centre - full / 2
            P_.sprite_half = full / 2 - crop_common_margin(sprites);
        }
        double inv = 1.0 / std::sqrt(2.0);
        P_.v_axis_i = (int)((1.0 / 1.0) * agent_speed_);
        P_.v_diag_i = (int)(inv * agent_speed_);
        for (auto& O : opt_) {
            derive(*O);
            if (O->P.r_hi <= O->P.r_lo) throw OptionError{-3, "spot radius range not supported"};
        }
        P_.coin_radius = (int)(10 * coin_scale_);
        P_.spawn_clamp = (int)(30 * SCALE);
        P_.quarter = (int)(SCREEN / 4);
        P_.bar_h = (int)(16 * SCALE);
        if (P_.show_last_action) { P_.bar_x = (int)(P_.quarter * 2.75); P_.bar_w = (int)(P_.quarter * 0.5); }
        else { P_.bar_x = P_.quarter * 2; P_.bar_w = P_.quarter * 2; }
        P_.half_diag = std::sqrt(std::pow((double)SCREEN, 2) + std::pow((double)SCREEN, 2)) / 2;
        P_.exit_radius = 20.0 / 2 * exit_scale_;
        P_.interval0 = (int)(initial_spawn_interval_ + spawn_interval_threshold_);
        P_.cos_tab = cos_.p;
        P_.sin_tab = sin_.p;
        P_.jump = jump_.p;
        P_.lab_fallback = lab_int("MEMGYM_SPOT_RESET_FALLBACK", 0);

        atlas_.reset(new Atlas());
        // (any size: SpotComposer::Pre holds the first 256 padded pixels of a layer's stamp in registers -- every default stamp in
        // full -- and stamp_apply_lit reads what a *_scale option adds beyond that from the atlas while it composes)
        for (auto& sp : sprites) atlas_->add_stamp(sp);  // 0..7
        atlas_->add_stamp(build_coin(coin_scale_));       // 8
        if (!P_.endless) {  // 9 + 2 g, 10 + 2 g: the exits of generation g (free generations: an empty stamp)
            pick_exit_generation();
            P_.exit_halves = 0;
            for (int g = 0; g < EXIT_GENS; ++g) {
                if (exit_gen_used_[g]) {
                    const int half = (int)(20 * exit_gen_scale_[g]) >> 1;
                    if (half > 255) throw OptionError{-3, "exit_scale beyond 25: the exit would be six screens wide"};
                    P_.exit_halves |= (uint64_t)half << (8 * g);
                    atlas_->add_stamp(build_exit(exit_gen_scale_[g], false));
                    atlas_->add_stamp(build_exit(exit_gen_scale_[g], true));
                } else {
                    atlas_->add_stamp(Stamp(1, 1));
                    atlas_->add_stamp(Stamp(1, 1));
                }
            }
            std::vector<double> hist(1 + EXIT_GENS, 0.0);
            for (int g = 0; g < EXIT_GENS; ++g) hist[1 + g] = exit_gen_used_[g] ? exit_gen_scale_[g] : 0.0;
            hist[0] = (double)P_.exit_gen;
            MG_HIP(hipMemcpy(exit_hist_.p, hist.data(), sizeof(double) * hist.size(), hipMemcpyHostToDevice));
        }
        atlas_->set_templates(build_chessboards(SCALE, SCREEN));
        atlas_->upload();
        dirty_ = false;
        for (size_t k = 1; k < opt_.size(); ++k) copy_geometry(opt_[k]->P, P_);
        copy_geometry(defaults_, P_);
        {   // (the defaults' own radius / dim values, whatever set 0 holds by now)
            SpotOpt D;
            D.P = defaults_;
            derive(D);
            defaults_ = D.P;
        }
        sets_dirty_ = true;
    }

    // The generation new exits belong to = the slot that holds exit_scale_; a new scale takes a free slot.  Slots are only ever
    // freed here, when all eight are taken: the instances' pads are read back and the generations no exit refers to any more
    // are released (rare: eight different exit sizes in the life of one handle).
    void pick_exit_generation() {
        for (int g = 0; g < EXIT_GENS; ++g)
            if (exit_gen_used_[g] && exit_gen_scale_[g] == exit_scale_) { P_.exit_gen = g; return; }
        auto take_free = [&]() {
            for (int g = 0; g < EXIT_GENS; ++g)
                if (!exit_gen_used_[g]) {
                    exit_gen_used_[g] = true;
                    exit_gen_scale_[g] = exit_scale_;
                    P_.exit_gen = g;
                    return true;
                }
            return false;
        };
        if (take_free()) return;
        MG_HIP(hipDeviceSynchronize());
        std::vector<uint32_t> pads(n_);
        MG_HIP(hipMemcpy2D(pads.data(), sizeof(uint32_t), reinterpret_cast<const char*>(core_.p) + offsetof(SpotCore, pad), sizeof(SpotCore),
                           sizeof(uint32_t), n_, hipMemcpyDeviceToHost));
        bool alive[EXIT_GENS] = {false, false, false, false, false, false, false, false};
        for (uint32_t pad : pads)
            if (pad & PAD_HAS_EXIT) alive[(pad & PAD_EXIT_GEN_MASK) >> PAD_EXIT_GEN_SHIFT] = true;
        for (int g = 0; g < EXIT_GENS; ++g) exit_gen_used_[g] = alive[g];
        if (!take_free())
            throw OptionError{-3, "exits of eight different exit_scale values are still on screen (use_exit = False keeps them); a ninth size needs a reset with use_exit = True first"};
    }

    // per-instance option sets
    struct SpotOpt {
        SpotParams P;
        OptListStore st_num_coins;
        double min_radius = 30.0 * 0.25, max_radius = 55.0 * 0.25;  // spot_min_radius / spot_max_radius (defaults x SCALE)
        int dim_duration = 6;                                      // light_dim_off_duration
    };
    // what a set's radius and dim options mean for the kernels (pure logic: the disc span table covers every radius up to DISC_RMAX)
    static void derive(SpotOpt& O) {
        const int r_lo = (int)O.min_radius, r_hi = (int)(O.max_radius + 1);
        // (both bounds travel through set_option one at a time: only a pair that is complete nonsense is refused here, the range
        // as a whole again by rebuild() / the reset that uses it)
        if (r_hi - 1 > DISC_RMAX || r_lo < 1) throw OptionError{-3, "spot radius range not supported"};
        O.P.r_lo = r_lo;
        O.P.r_hi = r_hi;
        O.P.dim_duration = O.dim_duration;
        O.P.dim_step = O.dim_duration > 0 ? (int)(255.0 / O.dim_duration) : 0;
    }
    static std::vector<std::unique_ptr<SpotOpt>> one_set() {
        std::vector<std::unique_ptr<SpotOpt>> v;
        v.emplace_back(new SpotOpt());
        return v;
    }
    // what the shared atlases, tables and derived constants fix for every set of the handle
    static void copy_geometry(SpotParams& d, const SpotParams& s) {
        d.endless = s.endless; d.n = s.n; d.ordered_holes = s.ordered_holes;
        d.show_last_action = s.show_last_action; d.agent_radius = s.agent_radius; d.sprite_half = s.sprite_half;
        d.coin_radius = s.coin_radius; d.v_axis_i = s.v_axis_i; d.v_diag_i = s.v_diag_i; d.spawn_clamp = s.spawn_clamp; d.bar_x = s.bar_x;
        d.bar_w = s.bar_w; d.quarter = s.quarter; d.bar_h = s.bar_h; d.exit_gen = s.exit_gen; d.exit_halves = s.exit_halves; d.half_diag = s.half_diag;
        d.exit_radius = s.exit_radius; d.interval0 = s.interval0; d.cos_tab = s.cos_tab; d.sin_tab = s.sin_tab; d.jump = s.jump; d.lab_fallback = s.lab_fallback;
    }
    bool per_set() const { return set_of_ != nullptr && opt_.size() > 1; }
    void upload_sets(hipStream_t s) {
        if (!per_set() || !sets_dirty_) return;
        SpotParams fresh = defaults_;  // a set that was never written: the reference's defaults under the handle's geometry (include/memgym.h)
        copy_geometry(fresh, P_);
        std::vector<SpotParams> host(MG_MAX_OPTION_SETS, fresh);
        for (size_t k = 0; k < opt_.size(); ++k) {
            copy_geometry(opt_[k]->P, P_);  // (ordered_holes may have been switched on since the last rebuild)
            host[k] = opt_[k]->P;
        }
        MG_HIP(hipMemcpyAsync(sets_dev_.p, host.data(), sizeof(SpotParams) * host.size(), hipMemcpyHostToDevice, s));
        MG_HIP(hipStreamSynchronize(s));  // (rare: only after an option of some set changed)
        sets_dirty_ = false;
    }

    void raster_only(void* obs, const uint8_t* only, hipStream_t s) override {
        if (P_.ordered_holes) launch_raster<SpotBorderComposer>(desc_.p, atlas_->dev(), obs, obs_format, n_, s, only);
        else launch_raster<SpotComposer>(desc_.p, atlas_->dev(), obs, obs_format, n_, s, only);
        MG_HIP(hipGetLastError());
    }

    void raster(void* obs, hipStream_t s) { raster_only(obs, nullptr, s); }
    // (the fused raster / reset launch keeps terminal observations itself; lab MEMGYM_SPOT_FINAL_FUSED=0: the generic path of mg_step)
    bool keeps_final_obs(hipStream_t) override {
        static const bool wanted = lab_int("MEMGYM_SPOT_FINAL_FUSED", 1) != 0;
        return wanted && obs_format == MG_OBS_U8_XYC && !per_set();
    }

    // Resets served inside the raster launch: on for the finite variant up to FUSE_MAX instances (more instances finish per step
    // than in the endless variant and their resets place up to five objects: ~15 us for the 16 lanes of an instance, the tail of
    // the step kernel).  Round 4, with the launch's agent-scope fence gone (profiles/r04_spot_step.md, M env-steps/s fused /
    // not): 16,384: 216 / 182, 32,768: 235 / 206, 65,536: 247 / 228, 131,072: 234 / 241 -- beyond FUSE_MAX the step kernel's reset
    // tail is amortised over several rounds of waves and the plain raster's seven workgroups per CU win.  The endless variant (its
    // step kernel's reset tail is 4 us) the other way round: off up to 16,384 instances (230 / 230; the plain raster stores with
    // the cached policy there), on beyond, where the plain raster stores non-temporally as well (32,768: 242-244 / 233-234,
    // 65,536: 245-248 / 239-242, 131,072: 256-257 / 236-252).  MEMGYM_SPOT_FUSE=0 / 1 (lab build) forces it off / on for both.
    static constexpr int FUSE_MAX = 65536;
    // store flavour of the fused launch: it runs five workgroups per CU (the reset code's registers), where only the
    // non-temporal stream keeps up; MEMGYM_RASTER_NT forces (tuning only)
    bool fused_nt() const {
        static const int forced = [] {
            const char* e = lab_env("MEMGYM_RASTER_NT");
            return e ? (atoi(e) != 0 ? 1 : 0) : -1;
        }();
        return forced >= 0 ? forced != 0 : true;
    }
    bool fuse_resets() const {
        static const int forced = [] {
            const char* e = lab_env("MEMGYM_SPOT_FUSE");
            return e ? (atoi(e) != 0 ? 1 : 0) : -1;
        }();
        return forced >= 0 ? forced != 0 : (P_.endless ? n_ > RASTER_PLAIN_MAX : n_ <= FUSE_MAX);
    }

    std::vector<std::unique_ptr<SpotOpt>> opt_;  // [0] = the handle-wide set (P_ below is its parameter block)
    SpotParams& P_;
    SpotParams defaults_;
    const int32_t* set_of_ = nullptr;
    bool sets_dirty_ = true;
    DevArray<SpotParams> sets_dev_;
    int n_;
    double coin_scale_, agent_speed_, agent_scale_, exit_scale_;
    double initial_spawn_interval_ = 30, spawn_interval_threshold_ = 10;
    bool dirty_ = true, seeded_ = false;

   public:
    void on_state_loaded() override {
        seeded_ = true;
        int f = 0;
        MG_HIP(hipMemcpy(&f, flags_.p, sizeof(int), hipMemcpyDeviceToHost));
        if (f) P_.ordered_holes = 1;
        if (!P_.endless) {  // the exit generations the restored instances refer to; the atlas follows
            std::vector<double> hist(1 + EXIT_GENS, 0.0);
            MG_HIP(hipMemcpy(hist.data(), exit_hist_.p, sizeof(double) * hist.size(), hipMemcpyDeviceToHost));
            for (int g = 0; g < EXIT_GENS; ++g) {
                exit_gen_used_[g] = hist[1 + g] != 0.0;
                exit_gen_scale_[g] = hist[1 + g];
            }
            rebuild();
        }
        sets_dirty_ = true;
    }
    void raster_debug(void* frames, hipStream_t s) override;

   private:
    std::unique_ptr<Atlas> atlas_;
    DevArray<SpotCore> core_;
    DevArray<double> sp_t_, sp_speed_, cos_, sin_;
    DevArray<uint32_t> sp_ang_;
    DevArray<uint4> jump_;  // SpotParams::jump
    DevArray<uint8_t> sp_r_;
    DevArray<int> queue_;  // deferred resets: the counters + n entries
    DevArray<int> flags_;  // [0] = SpotParams::ordered_holes: travels with the state (spotlights with a border may be alive in it)
    // finite variant: [0] = SpotParams::exit_gen, [1 + g] = exit_scale of generation g (0 = free); travels with the state (the
    // instances' exits name their generation, SpotCore::pad)
    DevArray<double> exit_hist_;
    double exit_gen_scale_[EXIT_GENS] = {0, 0, 0, 0, 0, 0, 0, 0};
    bool exit_gen_used_[EXIT_GENS] = {false, false, false, false, false, false, false, false};
    DevArray<uint32_t> coins_;
    DevArray<SpotDesc> desc_;
    ErrorWord err_;
    RngStore rng_;
};

void SpotFamily::raster_debug(void* frames, hipStream_t s) {
    if (dirty_) throw std::runtime_error("options that change geometry need a reset before the next render");
    DevArray<SpotDesc> dbg;
    dbg.alloc(n_, false);
    hipLaunchKernelGGL(spot_debug_desc_kernel, dim3((n_ + 255) / 256), dim3(256), 0, s, P_, io(), dbg.p);
    if (P_.ordered_holes) launch_raster<SpotBorderDebugComposer>(dbg.p, atlas_->dev(), frames, MG_OBS_U8_XYC, n_, s);
    else launch_raster<SpotDebugComposer>(dbg.p, atlas_->dev(), frames, MG_OBS_U8_XYC, n_, s);
    MG_HIP(hipGetLastError());
    MG_HIP(hipStreamSynchronize(s));  // dbg is released on return
}

Family* make_spot(int endless, int num_envs) { return new SpotFamily(endless, num_envs); }

}  // namespace mg

#ifdef MG_LAB_SPOT_CLOCK
extern "C" int mg_lab_spot_clock(unsigned long long* host, int n_waves, int clear) {
    if (clear) {
        void* p = nullptr;
        if (hipGetSymbolAddress(&p, HIP_SYMBOL(mg::g_lab_spot_clock)) != hipSuccess) return -1;
        return hipMemset(p, 0, sizeof(unsigned long long) * 10 * 65536) == hipSuccess ? 0 : -1;
    }
    return hipMemcpyFromSymbol(host, HIP_SYMBOL(mg::g_lab_spot_clock), sizeof(unsigned long long) * 10 * (size_t)n_waves) == hipSuccess ? 0 : -1;
}
#endif
