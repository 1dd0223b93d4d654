// mg_mystery.hip -- Mystery Path family on gfx950: MysteryPath-v0, MysteryPath-Grid-v0 and Endless-MysteryPath-v0.
//
// Reference behaviour reproduced (bit-exact observations, rewards, dones, RNG consumption):
//   memory_gym/mystery_path.py          reset :130-200  step :202-276
//   memory_gym/mystery_path_grid.py     reset :129-199  step :201-277 (GridCharacterController, Discrete(4))
//   memory_gym/endless_mystery_path.py  reset :195-280  step :282-444  drawing :111-160
//   memory_gym/pygame_assets.py         Node :438-493  EndlessMysteryPath :495-604  MysteryPath (noisy A*) :606-736
//   memory_gym/character_controller.py  CharacterController.step :89-146
//
//   mystery_step_kernel : one LANE per instance for the path following / fall-off logic.  The procedural path generator
//                         (33 % inner walls, 4 or 8 outer walls, A* with integers(1,9) noise on every relaxation,
//                         Python-list open/closed-set semantics incl. the reference's tie-breaking and its
//                         `neighbor.g = g` typo) runs in the same kernel for instances that (auto-)reset, but
//                         COOPERATIVELY: requests are served one at a time by the whole wave (coop_path, serve_*),
//                         and only the first 16 (endless: 8) lanes of a wave carry instances (instance_of_lane).
//   raster_kernel<MysteryComposer> : black frame -> goal/origin or past-path tiles -> agent sprite -> fall-off cross.
#include <memory>

#include "mg_atlas_v1.hpp"
#include "mg_lab.hpp"
#include "mg_device.hpp"
#include "mg_family.hpp"
#include "mg_raster_v1.hpp"
#include "mg_stamps.hpp"

namespace mg {
using namespace v1;  // raster generation 1 (see mg_raster_v1.hpp)

constexpr int G = 7;             // grid_dim
constexpr int SEG_STRIDE = 52;   // bytes per stored segment: [0] = length, [1..50] nodes
constexpr int MAX_SEG = 128;
constexpr int MAX_FALL = 128;
constexpr int ST_CROSS = 8;
constexpr int TILE = SCREEN / G;   // 12 px
// Endless: AUX_WORDS words per instance.  The first 128-byte line holds what a step may need besides the state and segment records,
// so that ONE batch of loads fetches it (emp_step_b): words 0..19 the EMP_PRE record (0..12 the segment record, SEG_STRIDE bytes; 13
// the stream's buffered half; 14 = has_buffered | end_y << 8; 16..19 the stream's 128-bit state behind the segment's draws, low word
// first), words 20..31 the first twelve fall-off keys; the list goes on behind them (MAX_FALL keys: x | y << 16, y biased by 1024).
constexpr int AUX_FALL = 20, AUX_WORDS = 160;
static_assert(AUX_FALL + MAX_FALL <= AUX_WORDS && AUX_WORDS % 32 == 0, "the fall-off list must fit behind the record");
constexpr int STAMINA_W = 4;      // int(16 * SCALE)

struct MysteryParams {
    int endless, grid, n;
    int max_steps, show_origin, show_goal, visual_feedback, show_past_path, show_background, show_stamina, stamina_level, depth;
    int agent_radius, sprite_dim, v_axis_i, v_diag_i, tile, cross_dim;
    int camera_offset;  // integral at the supported camera_offset_scale values
    int svc_prio;       // wave priority of the path-service waves inside the fused raster launches (s_setprio)
    int lazy;           // Endless: a reset generates ONE of its three initial segments, the other two are owed (see EMP_OWED)
    int path_help;      // frame workgroups help with long path queues (MEMGYM_PATH_HELP=0: the 128 dedicated workgroups alone, round 2)
    int bg_coop;        // Endless, fused launch: owed segments as queue entries of the service waves (small launches), not one per lane of frame workgroups
    int lazy_append;    // Endless, lazy: a segment appended during an episode is owed too, not a queue entry of the step (emp_step_a)
    int pre;            // Endless, lazy: the NEXT episode's first segment is generated ahead of time as a background job (see EMP_PRE)
    int seg_cap, fall_cap;  // Endless: segments / fall-off cells an episode may reach (MAX_SEG / MAX_FALL; the lab build lowers them for tests)
    OptList cardinal;
    double r_goal, r_fall, r_progress, r_dense, r_step;
};

struct __attribute__((aligned(16))) MysteryCore {
    int16_t ax, ay;
    uint8_t rot8, off, cross_on, path_len;  // path_len: finite variants; endless: segments OWED to the instance (EMP_OWED below)
    uint8_t sx, sy, ex, ey;
    int16_t cross_x, cross_y;           // fall_off_rect centre
    int32_t fails, t, ep_len, stamina;
    int32_t max_x, tiles_visited, cur_seg, num_seg;
    int32_t cur_node_seg, cur_node_idx, camera_x, n_falloff;
    uint64_t path_mask, visited_mask;   // finite: bit (x*7+y)
    double ep_sum;
    uint8_t td[3], have_start;
    int8_t end_y;
    uint8_t gx, gy;       // grid controller position (MysteryPath-Grid-v0)
    uint8_t bg;           // endless: -bg_scroll, the scrolling background's phase in pixels (< tile)
};
static_assert(sizeof(MysteryCore) == 96, "MysteryCore must be 96 bytes");
// endless: [EMP_FLO, EMP_FHI] = range of segments that may hold stamina flags (empty: lo > hi).  Two 32-bit halves of the finite variants'
// visited_mask (rounds 1-5: the grid controller's two position bytes, which capped the segment store at 255 records).
#define EMP_FLO(s) (reinterpret_cast<int32_t*>(&(s).visited_mask)[0])
#define EMP_FHI(s) (reinterpret_cast<int32_t*>(&(s).visited_mask)[1])
#define EMP_OWED(s) ((s).path_len)  // endless: segments the instance is owed ("lazy initial segments" below)
#define EMP_PRE(s) ((s).ex)         // endless: io.aux[i] holds the next episode's first segment (ex / ey: the finite variants' goal)
// The whole record as six 16-byte loads issued together.  Field by field the compiler split it into eleven odd-sized loads
// and issued three of them only after the first uses: a second memory round trip (3-4 us on a cold state array) at the head
// of the one-lane-per-instance step kernel (profiles/r03_emp.md, section 7).
__device__ __forceinline__ MysteryCore load_core(const MysteryCore* p) {
    typedef uint32_t q4 __attribute__((ext_vector_type(4)));
    const q4* src = reinterpret_cast<const q4*>(p);
    union { q4 q[6]; MysteryCore c; } u;
#pragma unroll
    for (int k = 0; k < 6; ++k) u.q[k] = src[k];
    return u.c;
}

struct __attribute__((aligned(16))) MysteryDesc {
    uint8_t valid, sprite, n_tiles, cross_on;
    int16_t sx, sy, cross_x, cross_y;        // top-left of the sprite / of the cross stamp
    uint8_t goal_on, goal_x, goal_y, origin_on, origin_x, origin_y, stamina_on, stamina_red;
    // past-path tiles (endless): bit (col*7 + row) of the 16-column x 7-row window whose column 0 is drawn at tile_x0
    uint64_t tile_mask[2];
    int32_t tile_x0;
    uint8_t bg_on, bg_phase, pad8[2];        // show_background: template = icy columns shifted left by bg_phase pixels
    uint32_t pad[2];
};
static_assert(sizeof(MysteryDesc) == 64, "MysteryDesc must be 64 bytes");

// BIG = false: the agent sprite (up to 1,024 pixels: every agent_scale up to 0.28) is requested with the frame's other loads and held
// in four registers per lane.  BIG = true (an agent_scale whose sprite is larger; chosen per handle by MysteryFamily::rebuild): the
// sprite is blitted from the atlas by the generation-1 stamp() loop, any size; everything else is the same code.
template <bool BIG>
struct MysteryComposerT {
    typedef MysteryDesc Desc;
    static __device__ __forceinline__ bool skip(const Desc* dp) { return dp->valid == 0; }
    static __device__ __forceinline__ void compose(const Desc* dp, const RasterCtx& R) {
        const Desc& d = *dp;
        StampRegs<4> sprite;
        if constexpr (!BIG) sprite = stamp_fetch<4>(R, d.sprite);
        StampRegs<1> cross;
        if (d.cross_on) cross = stamp_fetch<1>(R, ST_CROSS);
        if (d.bg_on) fill_template(R, d.bg_phase);
        else fill_clear(R);
        __syncthreads();
        if (d.goal_on) rect(R, d.goal_x * TILE, d.goal_y * TILE, TILE, TILE, C_GREEN, false);
        if (d.origin_on) rect(R, d.origin_x * TILE, d.origin_y * TILE, TILE, TILE, C_BLUE, false);
        for (int h = 0; h < 2; ++h) {  // distinct path cells: no overlap between them, no barrier needed
            uint64_t m = d.tile_mask[h];
            while (m) {
                int b = __ffsll((unsigned long long)m) - 1;
                m &= m - 1;
                int cell = h * 64 + b, col = cell / G, row = cell - col * G;
                rect(R, d.tile_x0 + TILE * col, TILE * row, TILE, TILE, C_WHITE, true);
            }
        }
        __syncthreads();
        if constexpr (BIG) stamp(R, d.sprite, d.sx, d.sy);
        else stamp_apply<4>(R, sprite, d.sx, d.sy);
        if (d.stamina_on) {
            __syncthreads();
            rect(R, SCREEN - STAMINA_W, 0, STAMINA_W, SCREEN, C_GREEN, false);
            if (d.stamina_red) {
                __syncthreads();
                rect(R, SCREEN - STAMINA_W, 0, STAMINA_W, d.stamina_red, C_RED, false);
            }
        }
        if (d.cross_on) {
            __syncthreads();
            stamp_apply<1>(R, cross, d.cross_x, d.cross_y);
        }
    }
};
typedef MysteryComposerT<false> MysteryComposer;
typedef MysteryComposerT<true> MysteryBigComposer;

// _build_debug_surface (mystery_path.py:103-117, endless_mystery_path.py:162-182).  The descriptor is a debug one
// (mystery_debug_desc_kernel): pad8[0] = 1 finite -- tile_mask[0] = the path between its ends (white), tile_mask[1] = the walls
// (red), goal / origin always on; pad8[0] = 2 endless -- tile_mask = EVERY path cell of the 16-column window, drawn as the
// reference's path surface: white with surface alpha 200 over the background; the stamina bar always.
__device__ __forceinline__ void rect_blend_white(const RasterCtx& R, int x, int y, int w, int h, uint32_t alpha) {
    for (int p = R.tid; p < w * h; p += 256) {
        const int px = p / h, py = p - px * h, X = x + px, Y = y + py;
        if ((unsigned)X < (unsigned)SCREEN && (unsigned)Y < (unsigned)SCREEN) {
            uint8_t* q = R.frame + X * COL_BYTES + Y * 3;
#pragma unroll
            for (int c = 0; c < 3; ++c) q[c] = (uint8_t)(q[c] + ((255 - (int)q[c]) * (int)alpha) / 255);  // SDL: d += (s - d) * A / 255
        }
    }
}
template <bool BIG>
struct MysteryDebugComposerT {
    typedef MysteryDesc Desc;
    static __device__ __forceinline__ bool skip(const Desc*) { return false; }
    static __device__ __forceinline__ void compose(const Desc* dp, const RasterCtx& R) {
        const Desc& d = *dp;
        const bool endless = d.pad8[0] == 2;
        StampRegs<4> sprite;
        if constexpr (!BIG) sprite = stamp_fetch<4>(R, d.sprite);
        StampRegs<1> cross;
        if (d.cross_on) cross = stamp_fetch<1>(R, ST_CROSS);
        if (d.bg_on) fill_template(R, d.bg_phase);
        else fill_clear(R);
        __syncthreads();
        for (int h = 0; h < 2; ++h) {  // distinct cells: no overlap, no barrier
            uint64_t m = d.tile_mask[h];
            while (m) {
                const int b = __ffsll((unsigned long long)m) - 1;
                m &= m - 1;
                const int cell = endless ? h * 64 + b : b, col = cell / G, row = cell - col * G;
                if (endless) rect_blend_white(R, d.tile_x0 + TILE * col, TILE * row, TILE, TILE, 200u);
                else rect(R, TILE * col, TILE * row, TILE, TILE, h == 0 ? C_WHITE : C_RED, false);
            }
        }
        if (!endless) {  // path[0] (the END node) green, path[-1] (the start) blue: they are not in tile_mask[0]
            rect(R, d.goal_x * TILE, d.goal_y * TILE, TILE, TILE, C_GREEN, false);
            rect(R, d.origin_x * TILE, d.origin_y * TILE, TILE, TILE, C_BLUE, false);
        }
        __syncthreads();
        if constexpr (BIG) stamp(R, d.sprite, d.sx, d.sy);
        else stamp_apply<4>(R, sprite, d.sx, d.sy);
        if (d.cross_on) {
            __syncthreads();
            stamp_apply<1>(R, cross, d.cross_x, d.cross_y);
        }
        if (d.stamina_on) {
            __syncthreads();
            rect(R, SCREEN - STAMINA_W, 0, STAMINA_W, SCREEN, C_GREEN, false);
            if (d.stamina_red) {
                __syncthreads();
                rect(R, SCREEN - STAMINA_W, 0, STAMINA_W, d.stamina_red, C_RED, false);
            }
        }
    }
};
typedef MysteryDebugComposerT<false> MysteryDebugComposer;
typedef MysteryDebugComposerT<true> MysteryDebugBigComposer;

// info["ground_truth"] in float64: the one-hot direction of the next path tile (endless_mystery_path.py:92-97)
__global__ __launch_bounds__(256) void mystery_gt64_kernel(int n, const MysteryCore* core, double* out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const MysteryCore s = load_core(core + i);
    out[3 * i + 0] = (double)s.td[0];
    out[3 * i + 1] = (double)s.td[1];
    out[3 * i + 2] = (double)s.td[2];
}

struct MysteryIO {
    MysteryCore* core;
    uint8_t* segs;      // endless: [N][seg_rows][SEG_STRIDE]; node byte = x_rel | y<<3 | rvis<<6 | svis<<7
    int seg_rows;       // segment records per instance (MAX_SEG by default; mg_set_capacity "path_segments")
    RngSoA rng;
    MysteryDesc* desc;
    int* err;
    int* queue;  // endless: instances waiting for a reset, filled by the step / enqueue kernels, drained by emp_serve_kernel
    uint64_t* walls;  // finite: [N] wall cells of the current path generation (bit x*7+y), read by the debug view only
    int* qctr;   // QC_COUNT entries, QC_HEAD pops beyond the static first round, QC_LEFT workgroups that left emp_serve_kernel
    int* bgq;    // endless, bg_coop: instances that are owed a segment nobody waits for yet (QC_BG_COUNT entries, popped by the service waves)
    uint8_t* bgflag;  // endless, larger launches: [N] 1 = the instance has a background job in this step's raster launch (lane-per-path service)
    uint32_t* aux;  // endless: [N][AUX_WORDS] the next episode's first segment, generated ahead of time (EMP_PRE), and the fall-off list
    const uint4* jump;  // [64][2] PCG64 jump constants {A^(k+1), S_(k+1)} (WaveRng)
    // telemetry of the finite variants' path generation inside the step's launches (bench.py: C3's measured reset share):
    // [0] wave-ticks (real-time clock, 10 ns) spent generating paths, [1] paths generated; mg_debug_counter "path_gen_ticks" / "path_gen_paths"
    // endless: [2] resets a step did itself from a record generated ahead of time (counted by the lab build only), [3] such records generated
    unsigned long long* stats;
    // per-instance option sets (mg_set_option_set / mg_bind_option_sets): instance i runs under sets[set_of[i]]; both NULL while the
    // handle has one set.  Read by the <PS = true> forms of the reset / step / queue-server kernels only.
    const struct MysteryParams* sets;
    const int32_t* set_of;
    MysteryDesc* tdesc;  // finite variants, FINAL forms of the step / raster kernels (terminal observations kept): [N] terminal-frame descriptors
};
constexpr int QC_COUNT = 0, QC_HEAD = 32, QC_LEFT = 64, QC_BG_COUNT = 96, QC_WORDS = 160;  // one 128-byte line each
// MysteryDesc::valid: 0 = leave the frame alone (masked reset), 1 = draw, 2 = the instance has a queue entry, 3 = served (and, in
// the fused launch, drawn by the workgroup that served it).  Only emp_raster_serve_kernel's frame workgroups tell 1 from 2 / 3.
constexpr uint8_t DESC_QUEUED = 2, DESC_SERVED = 3;

// ---- MysteryPath.__init__: walls + noisy A* on a 7x7 grid.  Returns the path length; out[] = flat indices
// (x*7+y), END FIRST like the reference's list.  -1 = "No valid path found".
__device__ __forceinline__ int nb_of(int idx, int k) {  // Node.add_neighbors order: x+1, x-1, y+1, y-1
    int x = idx / G, y = idx - x * G;
    if (k == 0) return x < G - 1 ? idx + G : -1;
    if (k == 1) return x > 0 ? idx - G : -1;
    if (k == 2) return y < G - 1 ? idx + 1 : -1;
    return y > 0 ? idx - 1 : -1;
}
__device__ __forceinline__ int diag_of(int idx, int k) {
    int x = idx / G, y = idx - x * G;
    if (k == 0) return (x < G - 1 && y < G - 1) ? idx + G + 1 : -1;
    if (k == 1) return (x > 0 && y > 0) ? idx - G - 1 : -1;
    if (k == 2) return (x < G - 1 && y > 0) ? idx + G - 1 : -1;
    return (x > 0 && y < G - 1) ? idx - G + 1 : -1;
}

// ---- Wave-cooperative path generation -------------------------------------------------------------------------
// The environments are stepped one LANE per instance, but generating a path (walls + noisy A*) is a long serial job:
// run by the single lane that happens to reset it took 70-300 us and was the tail of every launch in which any
// instance reset.  Instead, instances that need a path are served one at a time by their whole WAVE at a converged
// point of the kernel (serve_*): the requesting lane's inputs and RNG state are broadcast, all 64 lanes execute the
// same (uniform) control flow with node n's A* record living in lane n's registers and the ordered open list living
// one position per lane, so that the reference's list operations are O(1):
//   selection  "first index i >= 1 with f(open[i]) < f(open[0]) else 0"  = one shuffle of f + one ballot
//   removal    list.pop(i)                                                = one shuffle down
//   membership / closed / walls                                           = uniform 64-bit masks
// LDS: the heuristic table sqrt(0..79) for the block and 64 staging bytes per wave for the finished path.
constexpr int WS_SQRT = 0;                        // double sqrt_tab[80]
constexpr int WS_STAGE = 80 * 8;                  // uint8 stage[4 waves][64]
constexpr int WS_BYTES = WS_STAGE + 4 * 64;

struct PathWS {
    uint8_t* base;
    const uint4* jump;  // WaveRng's per-lane jump constants
    unsigned long long* stats;  // MysteryIO::stats or NULL
    __device__ __forceinline__ double h(int d2) const { return reinterpret_cast<const double*>(base + WS_SQRT)[d2]; }
    __device__ __forceinline__ uint8_t* stage() const { return base + WS_STAGE + (threadIdx.x >> 6) * 64; }
};

__device__ __forceinline__ void path_ws_init(uint8_t* smem) {  // all threads of the block, before any path is generated
    if (threadIdx.x < 80) reinterpret_cast<double*>(smem + WS_SQRT)[threadIdx.x] = sqrt((double)threadIdx.x);
    __syncthreads();
}

__device__ __forceinline__ int bcast(int v, int lane) { return __builtin_amdgcn_readlane(v, lane); }
__device__ __forceinline__ Pcg bcast(const Pcg& g, int lane) {
    Pcg b;
    uint32_t w[9] = {(uint32_t)g.state, (uint32_t)(g.state >> 32), (uint32_t)(g.state >> 64), (uint32_t)(g.state >> 96),
                     (uint32_t)g.inc,   (uint32_t)(g.inc >> 32),   (uint32_t)(g.inc >> 64),   (uint32_t)(g.inc >> 96), g.buf};
#pragma unroll
    for (int k = 0; k < 9; ++k) w[k] = (uint32_t)__builtin_amdgcn_readlane((int)w[k], lane);
    b.state = ((u128)w[3] << 96) | ((u128)w[2] << 64) | ((u128)w[1] << 32) | w[0];
    b.inc = ((u128)w[7] << 96) | ((u128)w[6] << 64) | ((u128)w[5] << 32) | w[4];
    b.buf = w[8];
    b.has = __builtin_amdgcn_readlane(g.has ? 1 : 0, lane) != 0;
    return b;
}

// ---- the instance's RNG stream, generated 64 outputs at a time by the whole wave ------------------------------------------
// A path draws ~130 32-bit numbers; drawn one by one from wave-uniform state, every PCG64 step is a 128 x 128-bit multiply on
// the scalar unit (~45 scalar instructions per 64-bit output) inside a kernel that is bound by scalar issue -- a third of a
// path's instructions.  An LCG can be jumped: s_k = A^k s_0 + S_k inc with S_k = 1 + A + ... + A^(k-1), so lane k computes
// step k + 1 directly (two 128-bit multiplies on the vector unit, all 64 lanes at once) and a draw is one v_readlane.  The
// 32-bit draws are numpy's: low half, then the buffered high half of each 64-bit output (Pcg::next32), a half buffered
// before the hand-over first.  jump[k] = {A^(k+1), S_(k+1)} is built on the host (MysteryFamily).
struct WaveRng {
    u128 M, S;        // per lane: A^(lane + 1), S_(lane + 1)
    u128 st;          // per lane: the state after lane + 1 steps from `base`
    uint32_t lo, hi;  // per lane: that step's output
    u128 base, inc;   // uniform
    int cursor;       // uniform: 32-bit draws taken from the current batch (0 .. 128)
    bool pre_has;     // uniform: the stream was handed over with a buffered half, not consumed yet
    uint32_t pre_buf;

    __device__ __forceinline__ void load_jump(const uint4* jump) {
        const int lane = threadIdx.x & 63;
        const uint4 m = jump[2 * lane], q = jump[2 * lane + 1];
        M = ((u128)m.w << 96) | ((u128)m.z << 64) | ((u128)m.y << 32) | m.x;
        S = ((u128)q.w << 96) | ((u128)q.z << 64) | ((u128)q.y << 32) | q.x;
    }
    __device__ __forceinline__ void refill() {
        st = M * base + S * inc;
        const uint64_t h = (uint64_t)(st >> 64), l = (uint64_t)st, x = h ^ l;
        const unsigned rot = (unsigned)(h >> 58);
        const uint64_t o = (x >> rot) | (x << ((64 - rot) & 63));
        lo = (uint32_t)o;
        hi = (uint32_t)(o >> 32);
        cursor = 0;
    }
    static __device__ __forceinline__ u128 lane128(u128 v, int lane) {
        const uint32_t a = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, lane);
        const uint32_t b = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), lane);
        const uint32_t c = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 64), lane);
        const uint32_t d = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 96), lane);
        return ((u128)d << 96) | ((u128)c << 64) | ((u128)b << 32) | a;
    }
    // g: wave-uniform (a broadcast copy of the requesting lane's stream)
    __device__ __forceinline__ void take(const Pcg& g) {
        base = g.state;
        inc = g.inc;
        pre_has = g.has;
        pre_buf = g.buf;
        refill();
    }
    // the stream as it stands after the draws taken (uniform)
    __device__ __forceinline__ void give(Pcg& g) const {
        const int m = (cursor + 1) >> 1;  // 64-bit outputs consumed from this batch
        g.inc = inc;
        if (m == 0) {
            g.state = base;
            g.has = pre_has;
            g.buf = pre_buf;
        } else {
            g.state = lane128(st, m - 1);
            g.has = (cursor & 1) != 0;
            g.buf = (uint32_t)__builtin_amdgcn_readlane((int)hi, m - 1);  // numpy keeps the last buffered half also once it is used
        }
    }
    __device__ __forceinline__ uint32_t next32() {
        if (pre_has) {
            pre_has = false;
            return pre_buf;
        }
        if (cursor == 128) {
            base = lane128(st, 63);
            refill();
        }
        const int idx = cursor >> 1;
        const uint32_t v = (cursor & 1) ? (uint32_t)__builtin_amdgcn_readlane((int)hi, idx) : (uint32_t)__builtin_amdgcn_readlane((int)lo, idx);
        ++cursor;
        return v;
    }
    // Generator.integers(lo, hi): Lemire bounded draw on the 32-bit path; span 1 consumes nothing (Pcg::integers)
    __device__ __forceinline__ int integers(int lo_, int hi_) {
        const uint32_t rng = (uint32_t)(hi_ - 1 - lo_);
        if (rng == 0) return lo_;
        const uint32_t n = rng + 1u;
        uint64_t m = (uint64_t)next32() * n;
        uint32_t left = (uint32_t)m;
        if (left < n) {
            const uint32_t thr = (0xFFFFFFFFu - rng) % n;
            while (left < thr) {
                m = (uint64_t)next32() * n;
                left = (uint32_t)m;
            }
        }
        return lo_ + (int)(m >> 32);
    }
};

// MysteryPath.__init__ (pygame_assets.py:606-724) + Node (:438-493), every argument wave-uniform, called by all 64
// lanes.  Returns the path length (-1 = "No valid path found"); lane k < len receives the k-th path node (flat index
// x*7+y, END FIRST like the reference's list) in out_node, path_mask has one bit per path node.
__device__ int coop_path(WaveRng& g, const PathWS& W, int sx, int sy, int ex, int ey, int& out_node, uint64_t& path_mask, uint64_t& wall_out) {
    const int lane = threadIdx.x & 63;
    uint64_t wall = 0, closed = 0, in_open = 0;
    for (int i = 0; i < G; ++i)
        for (int j = 0; j < G; ++j)
            if (i > 0 && i < G - 2 && j > 0 && j < G - 2)
                if (g.integers(0, 100) < 33) wall |= 1ull << (i * G + j);
    const int start = sx * G + sy, end = ex * G + ey;
    // outer wall candidates, in the reference's (i, j) order == increasing flat index: one node per lane
    uint64_t outer;
    {
        const int idx = lane < G * G ? lane : 0;
        const int i = idx / G, j = idx - i * G;
        bool ok = lane < G * G && (i == 0 || i == G - 1 || j == 0 || j == G - 1) && idx != start && idx != end;
        for (int k = 0; k < 4; ++k) ok = ok && nb_of(start, k) != idx && nb_of(end, k) != idx;
        for (int k = 0; k < 4; ++k) {
            int q = nb_of(idx, k);
            if (q >= 0 && ((wall >> q) & 1ull)) ok = false;
            q = diag_of(idx, k);
            if (q >= 0 && ((wall >> q) & 1ull)) ok = false;
        }
        outer = __ballot(ok);
    }
    int n_outer = __popcll(outer);
    const int n_iter = g.integers(0, 2) == 0 ? 4 : 8;  // rng.choice([4, 8])
    for (int it = 0; it < n_iter; ++it) {
        if (n_outer > 0) {
            int k = g.integers(0, n_outer);
            uint64_t m = outer;
            for (int q = 0; q < k; ++q) m &= m - 1;  // k-th remaining candidate (list order)
            const int idx = __ffsll((unsigned long long)m) - 1;
            wall |= 1ull << idx;
            outer &= ~(1ull << idx);
            --n_outer;
        }
    }
    wall_out = wall;  // (the debug view draws the walls)
    // per-node record in lane n; f = g_cost + h is only ever evaluated for nodes in the open set, and the reference's
    // `neighbor.g = g` typo means g_cost never changes once a node has entered it
    int gval = 0, prev = -1;
    double hval = 0.0;
    {
        const int idx = lane < G * G ? lane : 0;
        const int ax = idx / G, ay = idx - ax * G;
        hval = W.h((ax - ex) * (ax - ex) + (ay - ey) * (ay - ey));
    }
    int lst = 0, n_open = 0;  // lane p: node at position p of the ordered open list
    if (lane == 0) lst = start;
    n_open = 1;
    in_open |= 1ull << start;
    for (;;) {
        if (n_open == 0) return -1;
        const double fnode = (double)gval + hval;
        const double f_at = __shfl(fnode, lst & 63);
        const double f0 = __shfl(f_at, 0);
        const uint64_t better = __ballot(lane >= 1 && lane < n_open && f_at < f0);
        const int w = better ? __ffsll((unsigned long long)better) - 1 : 0;  // first strictly better than open[0]
        const int cur = bcast(lst, w);
        if (cur == end) {
            int len = 0, t = cur;
            path_mask = 0;
            for (;;) {
                if (lane == len) out_node = t;
                path_mask |= 1ull << t;
                ++len;
                const int pv = bcast(prev, t);
                if (pv < 0) break;
                t = pv;
            }
            return len;
        }
        {  // open_set.remove(current)
            const int nxt = __shfl_down(lst, 1);
            if (lane >= w) lst = nxt;
            --n_open;
        }
        in_open &= ~(1ull << cur);
        closed |= 1ull << cur;
        const int gcur = bcast(gval, cur);
        // Node.add_neighbors order x+1, x-1, y+1, y-1 (-1 = outside).  Everything about WHICH neighbours are evaluated,
        // the draws and the list order is wave-uniform (scalar); the per-node updates are done by the node's own lane,
        // all four at once.  integers(1, 9) has a span of 8: Lemire never rejects, the draw is 1 + (word >> 29).
        const int cx = cur / G, cy = cur - cx * G;
        const int nb[4] = {cx < G - 1 ? cur + G : -1, cx > 0 ? cur - G : -1, cy < G - 1 ? cur + 1 : -1, cy > 0 ? cur - 1 : -1};
        const uint64_t blocked = closed | wall;
        int gg = 0;
        bool mine = false;
        int pos = n_open;
        const uint64_t was_open = in_open;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const bool valid = nb[k] >= 0 && !((blocked >> (nb[k] & 63)) & 1ull);
            if (!valid) continue;  // uniform
            const int cost = gcur + 1 + (int)(g.next32() >> 29);
            if (lane == nb[k]) {
                mine = true;
                gg = cost;
            }
            if (!((was_open >> nb[k]) & 1ull)) {  // open_set.append(neighbor)
                if (lane == pos) lst = nb[k];
                ++pos;
                in_open |= 1ull << nb[k];
            }
        }
        n_open = pos;
        if (mine) {
            if ((was_open >> lane) & 1ull) {
                if (gg < gval) prev = cur;  // `neighbor.g = g` typo: g_cost is NOT updated
            } else {
                gval = gg;
                prev = cur;
            }
        }
    }
}

__device__ __forceinline__ int floordiv_pos(int a, int b) {
    int q = a / b;
    return (a % b != 0 && a < 0) ? q - 1 : q;
}

// CharacterController.step with an optional clamp to the screen
__device__ __forceinline__ void move_agent(const MysteryParams& P, MysteryCore& s, int a0, int a1, bool clamp) {
    int ax = s.ax, ay = s.ay;
    free_move(a0, a1, P.v_axis_i, P.v_diag_i, ax, ay, s.rot8, clamp, P.agent_radius, SCREEN - P.agent_radius, P.agent_radius,
              SCREEN - P.agent_radius);
    s.ax = (int16_t)ax;
    s.ay = (int16_t)ay;
}

// ============================================ finite ============================================
// MysteryPathEnv.reset (mystery_path.py:130-200) in two halves around the path generation, which is served by the
// whole wave (serve_mp): the draws before it, the bookkeeping after it.
struct PathReq {
    int need;            // this lane wants a path
    int sx, sy, ex, ey;  // start / end tile
};
__device__ __forceinline__ PathReq mp_pre_reset(const MysteryParams& P, MysteryCore& s, Pcg& g) {
    s.t = 0;
    s.ep_sum = 0.0;
    s.ep_len = 0;
    int cardinal = choice(g, P.cardinal);
    PathReq r;
    r.need = 1;
    if (cardinal == 0) { r.sx = 0; r.sy = g.integers(0, G); r.ex = G - 1; r.ey = g.integers(0, G); }
    else if (cardinal == 1) { r.sx = G - 1; r.sy = g.integers(0, G); r.ex = 0; r.ey = g.integers(0, G); }
    else if (cardinal == 2) { r.sx = g.integers(0, G); r.sy = 0; r.ex = g.integers(0, G); r.ey = G - 1; }
    else { r.sx = g.integers(0, G); r.sy = G - 1; r.ex = g.integers(0, G); r.ey = 0; }
    return r;
}
__device__ __forceinline__ void mp_post_reset(const MysteryParams& P, MysteryCore& s, const PathReq& r, int len, uint64_t pm, MysteryDesc& d) {
    const int sx = r.sx, sy = r.sy, ex = r.ex, ey = r.ey;
    s.path_mask = pm;
    s.visited_mask = 0;
    s.path_len = (uint8_t)len;
    s.sx = (uint8_t)sx; s.sy = (uint8_t)sy; s.ex = (uint8_t)ex; s.ey = (uint8_t)ey;
    // free controller: start tile corner + radius; grid controller: cell centre 12*i + 6 -- the same pixel at SCALE 0.25
    s.ax = (int16_t)(P.grid ? sx * P.tile + P.tile / 2 : sx * P.tile + P.agent_radius);
    s.ay = (int16_t)(P.grid ? sy * P.tile + P.tile / 2 : sy * P.tile + P.agent_radius);
    s.gx = (uint8_t)sx;
    s.gy = (uint8_t)sy;
    s.rot8 = 0;
    s.off = 0;
    s.fails = 0;
    s.cross_on = 0;
    memset(&d, 0, sizeof(d));
    d.valid = 1;
    d.sprite = 0;
    d.sx = (int16_t)(s.ax - P.sprite_dim / 2);
    d.sy = (int16_t)(s.ay - P.sprite_dim / 2);
    d.goal_on = P.show_goal ? 1 : 0; d.goal_x = (uint8_t)ex; d.goal_y = (uint8_t)ey;
    d.origin_on = P.show_origin ? 1 : 0; d.origin_x = (uint8_t)sx; d.origin_y = (uint8_t)sy;
}
// All 64 lanes, converged: one path per requesting lane, generated by the whole wave on a broadcast copy of that
// lane's RNG stream; the requester receives the stream back together with the path mask and length.
__device__ void serve_mp(const PathWS& W, const PathReq& req, Pcg& g, int* err, int& len_out, uint64_t& pm_out, uint64_t* walls, int inst) {
    const int lane = threadIdx.x & 63;
    uint64_t todo = __ballot(req.need != 0);
    WaveRng wr;
    if (todo) wr.load_jump(W.jump);
    const unsigned long long t_in = (todo && W.stats) ? wall_clock64() : 0ull;
    const int n_paths = __popcll((unsigned long long)todo);
    while (todo) {
        const int L = __ffsll((unsigned long long)todo) - 1;
        todo &= todo - 1;
        Pcg bg = bcast(g, L);
        int node = 0;
        uint64_t pm = 0;
        uint64_t wl = 0;
        wr.take(bg);
        int len = coop_path(wr, W, bcast(req.sx, L), bcast(req.sy, L), bcast(req.ex, L), bcast(req.ey, L), node, pm, wl);
        wr.give(bg);
        if (len < 0) {
            if (lane == 0) raise_error(err, 2);
            len = 0;
            pm = 0;
        }
        if (lane == L) {
            g = bg;
            len_out = len;
            pm_out = pm;
            if (walls) walls[inst] = wl;  // only the debug view reads them
        }
    }
    if (n_paths && W.stats && lane == 0) {  // (rare path: an instance of this wave finished)
        atomicAdd(W.stats, wall_clock64() - t_in);
        atomicAdd(W.stats + 1, (unsigned long long)n_paths);
    }
}

// the frame descriptor of a finite-variant instance as its state stands
__device__ __forceinline__ void mp_desc(const MysteryParams& P, const MysteryCore& s, MysteryDesc& d) {
    memset(&d, 0, sizeof(d));
    d.valid = 1;
    d.sprite = s.rot8;
    d.sx = (int16_t)(s.ax - P.sprite_dim / 2);
    d.sy = (int16_t)(s.ay - P.sprite_dim / 2);
    d.cross_on = s.cross_on;
    d.cross_x = (int16_t)(s.cross_x - P.cross_dim / 2);
    d.cross_y = (int16_t)(s.cross_y - P.cross_dim / 2);
    d.goal_on = P.show_goal ? 1 : 0; d.goal_x = s.ex; d.goal_y = s.ey;
    d.origin_on = P.show_origin ? 1 : 0; d.origin_x = s.sx; d.origin_y = s.sy;
}

// MysteryPathEnv.step (mystery_path.py:202-276).  Returns true if the instance finished and is to be reset in this call
// (the caller then runs mp_pre_reset / serve_mp / mp_post_reset); otherwise the frame descriptor is filled here.
__device__ bool mp_step(const MysteryParams& P, int i, MysteryCore& s, int act0, int act1, float* reward_out,
                        uint8_t* done_out, const mg_info_buffers& info, int autoreset, MysteryDesc& d) {
    double reward = 0.0;
    bool done = false;
    int success = 0;
    if (P.grid) {  // GridCharacterController.step / reset_position (character_controller.py:177-216)
        int a = s.off ? 0 : act0;
        int gx = s.off ? s.sx : s.gx, gy = s.off ? s.sy : s.gy;
        int rot = s.rot8 * 45;
        if (a == 1) rot = (rot + 90) % 360;
        if (a == 2) rot = (rot + 270) % 360;
        if (a == 3) {
            int face = rot / 90;  // 0 N, 1 W, 2 S, 3 E
            if (face == 0) { if (gy > 0) gy--; }
            else if (face == 3) { if (gx < G - 1) gx++; }
            else if (face == 2) { if (gy < G - 1) gy++; }
            else { if (gx > 0) gx--; }
        }
        s.rot8 = (uint8_t)(rot / 45);
        s.gx = (uint8_t)gx;
        s.gy = (uint8_t)gy;
        s.ax = (int16_t)(gx * P.tile + P.tile / 2);
        s.ay = (int16_t)(gy * P.tile + P.tile / 2);
    } else if (!s.off) {
        move_agent(P, s, act0, act1, true);
    } else {
        s.ax = (int16_t)(s.sx * P.tile + P.agent_radius);
        s.ay = (int16_t)(s.sy * P.tile + P.agent_radius);
        move_agent(P, s, 0, 0, true);
    }
    int nx = s.ax / P.tile, ny = s.ay / P.tile;
    if (nx == s.ex && ny == s.ey) {
        reward += P.r_goal;
        done = true;
        success = 1;
    } else {
        int cell = nx * G + ny;
        bool on_path = (s.path_mask >> cell) & 1ull;
        if (on_path) {
            bool special = (nx == s.sx && ny == s.sy) || (nx == s.ex && ny == s.ey);
            if (!((s.visited_mask >> cell) & 1ull) && !special) {
                reward += P.r_progress;
                s.visited_mask |= 1ull << cell;
            }
            s.cross_on = 0;
            s.off = 0;
        } else {
            reward += P.r_fall;
            s.fails++;
            if (P.visual_feedback) s.cross_on = 1;
            s.off = 1;
        }
        s.cross_x = s.ax;
        s.cross_y = s.ay;
    }
    reward += P.r_step;
    s.t++;
    if (s.t == P.max_steps) done = true;
    s.ep_sum += reward;
    s.ep_len++;
    if (done) {
        if (info.ep_reward_dev) info.ep_reward_dev[i] = s.ep_sum;
        if (info.ep_length_dev) info.ep_length_dev[i] = s.ep_len;
        if (info.aux_dev[0]) info.aux_dev[0][i] = (float)success;
        if (info.aux_dev[1]) info.aux_dev[1][i] = (float)s.fails;
    }
    reward_out[i] = (float)reward;
    if (info.reward64_dev) info.reward64_dev[i] = reward;  // the reference's Python float, unrounded
    done_out[i] = done ? 1 : 0;
    if (info.capacity_dev) info.capacity_dev[i] = 0;  // (the finite variants have no capacity an episode can reach)
    if (done && autoreset) return true;
    mp_desc(P, s, d);
    return false;
}

// ============================================ endless ============================================
__device__ __forceinline__ uint8_t* seg_ptr(const MysteryIO& io, int i, int seg) {
    return io.segs + ((size_t)i * io.seg_rows + seg) * SEG_STRIDE;
}
__device__ __forceinline__ int node_x(int seg, uint8_t b) { return seg * (G + 1) + (b & 7); }
__device__ __forceinline__ int node_y(uint8_t b) { return (b >> 3) & 7; }

// One 52-byte segment record in registers.  The segment store is cold in every step (the observation stream evicts
// it), so walking it byte by byte made each access a dependent ~1 us global round trip; a record is fetched with 13
// dword loads in flight together and then indexed in registers.
struct SegRec {
    uint32_t w[SEG_STRIDE / 4];
    int seg;  // -1: nothing loaded
    __device__ __forceinline__ void load(const MysteryIO& io, int i, int sg) {
        if (sg == seg) return;
        const uint32_t* p = reinterpret_cast<const uint32_t*>(seg_ptr(io, i, sg));
#pragma unroll
        for (int j = 0; j < SEG_STRIDE / 4; ++j) w[j] = p[j];
        seg = sg;
    }
    // the record of segment sg: from `other` if that holds it (prefetched), else from memory
    __device__ __forceinline__ void load_or_take(const MysteryIO& io, int i, int sg, const SegRec& other) {
        if (sg == seg) return;
        if (other.seg == sg) {
#pragma unroll
            for (int j = 0; j < SEG_STRIDE / 4; ++j) w[j] = other.w[j];
            seg = sg;
            return;
        }
        load(io, i, sg);
    }
    __device__ __forceinline__ uint8_t byte(int p) const {  // p = 0: node count, 1..: nodes
        uint32_t v = w[0];
#pragma unroll
        for (int j = 1; j < SEG_STRIDE / 4; ++j) v = (p >> 2) == j ? w[j] : v;
        return (uint8_t)(v >> (8 * (p & 3)));
    }
};

// EndlessMysteryPath.add_path_segment (pygame_assets.py:544-604), served by the whole wave: every lane passes the number
// of segments its instance still needs (3 at reset, 1 when the agent enters the last-but-one segment, else 0).  All 64
// lanes, converged.  The finished record is assembled in LDS and written to the instance's segment store as 13 dwords.
__device__ void serve_emp(const MysteryIO& io, const PathWS& W, int i, int want, MysteryCore& s, Pcg& g) {
    const int lane = threadIdx.x & 63;
    int todo_n = want;
    WaveRng wr;
    if (__ballot(todo_n > 0)) wr.load_jump(W.jump);
    for (;;) {
        const uint64_t todo = __ballot(todo_n > 0);
        if (!todo) break;
        const int L = __ffsll((unsigned long long)todo) - 1;
        Pcg bg = bcast(g, L);
        wr.take(bg);
        const int have = bcast((int)s.have_start, L), endy = bcast((int)s.end_y, L);
        const int sy = have ? endy : wr.integers(0, G);
        const int ey = wr.integers(0, G);
        int node = 0;
        uint64_t pm = 0;
        uint64_t walls_unused = 0;
        int len = coop_path(wr, W, 0, sy, G - 1, ey, node, pm, walls_unused);
        wr.give(bg);
        if (len < 0) {
            if (lane == 0) raise_error(io.err, 2);
            len = 0;
        }
        // the segment record as stored: byte 0 = node count, bytes 1..len = the path START first (our list is END first:
        // position p sits in lane len-1-p), byte len+1 = the transition node at x = 8*seg + 7, zeros after it.
        // Assembled in LDS by 52 lanes, written by 13 lanes as dwords (the requester alone copied it byte by byte before).
        uint8_t* stage = W.stage();
        {
            const int from = len - lane;  // lane b in 1..len holds path position b-1
            const int nd = __shfl(node, from >= 0 && from < 64 ? from : 0);
            const int x = nd / G, y = nd - x * G;
            int b = 0;
            if (lane == 0) b = len + 1;
            else if (lane <= len) b = x | (y << 3);
            else if (lane == len + 1) b = 7 | (ey << 3);
            stage[lane] = (uint8_t)b;
        }
        const int nseg = bcast((int)s.num_seg, L);
        if (nseg < io.seg_rows && lane < SEG_STRIDE / 4)
            reinterpret_cast<uint32_t*>(seg_ptr(io, bcast(i, L), nseg))[lane] = reinterpret_cast<const uint32_t*>(stage)[lane];
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");  // the requester's lane reads the record back (emp_post_reset, SegRec)
        if (lane == L) {
            g = bg;
            s.have_start = 1;
            s.end_y = (int8_t)ey;
            EMP_PRE(s) = 0;  // the stream has moved: a record generated ahead of time no longer continues it
            if (s.num_seg >= io.seg_rows) raise_error(io.err, 4);
            else s.num_seg++;
            todo_n--;
        }
    }
}

// nxt_seg / nxt_w0: dword 0 (node count + first three nodes) of segment nxt_seg if the caller has requested it early, else -1
__device__ void emp_direction(const MysteryIO& io, int i, MysteryCore& s, float* gt, SegRec& R, int nxt_seg = -1, uint32_t nxt_w0 = 0) {
    R.load(io, i, s.cur_node_seg);
    const uint8_t cb = R.byte(1 + s.cur_node_idx);
    int cx = node_x(s.cur_node_seg, cb), cy = node_y(cb);
    int nseg = s.cur_node_seg, nidx = s.cur_node_idx + 1;
    if (nidx >= R.byte(0)) {
        nseg++;
        nidx = 0;
    }
    if (nseg < s.num_seg) {
        uint8_t nb;
        if (nseg == R.seg) {
            nb = R.byte(1 + nidx);
        } else if (nseg == nxt_seg && nidx == 0) {
            nb = (uint8_t)(nxt_w0 >> 8);
        } else {  // first node of the following segment (keeps R on the current one for the past-path walk)
            nb = seg_ptr(io, i, nseg)[1 + nidx];
        }
        int x = node_x(nseg, nb) - cx, y = node_y(nb) - cy;
        if (x == 1) { s.td[0] = 1; s.td[1] = 0; s.td[2] = 0; }
        else if (y == -1) { s.td[0] = 0; s.td[1] = 1; s.td[2] = 0; }
        else if (y == 1) { s.td[0] = 0; s.td[1] = 0; s.td[2] = 1; }
    }
    if (gt) {
        gt[0] = (float)s.td[0];
        gt[1] = (float)s.td[1];
        gt[2] = (float)s.td[2];
    }
}

// ---- the past-path walk on whole segment records (emp_fill_desc) ----
// Highest position p in [1, hi] of a record whose node lies in column x_rel == rel, 0 if none (rel > 7: none).  Four node bytes per word;
// ~((x + 0x7F..) | x | 0x7F..) flags exactly the zero bytes of x (bytes <= 7 here: no carry between bytes).
__device__ __forceinline__ int seg_last_in_column(const uint32_t (&w)[SEG_STRIDE / 4], int hi, int rel) {
    int pos = 0;
    const uint32_t t4 = (uint32_t)(rel & 7) * 0x01010101u;
    const bool possible = rel <= 7;
#pragma unroll
    for (int j = 0; j < SEG_STRIDE / 4; ++j) {  // (upwards: the highest word with a match is taken last)
        const uint32_t x = (w[j] & 0x07070707u) ^ t4;
        uint32_t z = ~((x + 0x7F7F7F7Fu) | x | 0x7F7F7F7Fu);  // 0x80 in every byte of x that is zero
        const int last = hi - 4 * j;                            // bytes 0 .. last of this word are positions <= hi
        if (j >= 2 && !__ballot(last >= 0)) break;              // (no lane of the wave has a node this far into its record: paths are ~10-20 nodes)
        const uint32_t upto = last >= 3 ? 0xFFFFFFFFu : (last < 0 ? 0u : (0xFFFFFFFFu >> (8 * (3 - last))));
        z &= upto;
        if (j == 0) z &= ~0xFFu;  // byte 0 of the record is the node count
        if (z) pos = 4 * j + ((31 - __clz((int)z)) >> 3);
    }
    return possible ? pos : 0;
}
// Occupancy of the nodes at positions lo .. hi of a record: bit 8 x_rel + y (bits 8 x_rel + 7 stay clear).  Bytes outside the range are
// replaced by a node (7, 7) that no path has (y <= 6) before the four bytes of a word are turned into bits.
__device__ __forceinline__ uint64_t seg_occupancy(const uint32_t (&w)[SEG_STRIDE / 4], int lo, int hi) {
    uint64_t m = 0;
#pragma unroll
    for (int j = 0; j < SEG_STRIDE / 4; ++j) {
        const int first = lo - 4 * j, last = hi - 4 * j;  // bytes first .. last of this word are in range
        if (j >= 2 && !__ballot(last >= 0)) break;        // (wave-uniform: nothing of any lane's range lies in this word or behind it)
        const uint32_t from = first <= 0 ? 0xFFFFFFFFu : (first > 3 ? 0u : (0xFFFFFFFFu << (8 * first)));
        const uint32_t upto = last >= 3 ? 0xFFFFFFFFu : (last < 0 ? 0u : (0xFFFFFFFFu >> (8 * (3 - last))));
        const uint32_t keep = from & upto;
        const uint32_t sw = ((w[j] & 0x07070707u) << 3) | ((w[j] >> 3) & 0x07070707u);  // x_rel and y swapped: 8 x_rel + y per byte
        const uint32_t v = (sw & keep) | (0x3F3F3F3Fu & ~keep);
        m |= 1ull << (v & 63u);
        m |= 1ull << ((v >> 8) & 63u);
        m |= 1ull << ((v >> 16) & 63u);
        m |= 1ull << (v >> 24);
    }
    return m & 0x7F7F7F7F7F7F7F7Full;
}
// The occupancy of one segment (8 columns x 7 rows, a byte per column) into the descriptor's mask: bit 7 col + y, col = cbase + x_rel
// (cbase = the segment's first column minus past_x, -8 .. 15; columns outside 0 .. 15 hold no node of the walk)
__device__ __forceinline__ void emp_deposit(uint64_t occ, int cbase, uint64_t& mask0, uint64_t& mask1) {
    uint64_t dense = 0;  // 7 bits per column, column x_rel at bit 7 x_rel
#pragma unroll
    for (int c = 0; c < 8; ++c) dense |= ((occ >> (8 * c)) & 0x7Full) << (7 * c);
    const int sh = 7 * cbase;  // -56 .. 105
    if (sh >= 64) {
        mask1 |= dense << (sh - 64);
    } else if (sh > 0) {
        mask0 |= dense << sh;
        mask1 |= dense >> (64 - sh);
    } else {
        mask0 |= dense >> (-sh);
    }
}

// WHOLE: the past-path walk on whole records (the step kernel, one lane per instance: its longest phase under a path-following agent);
// false: the reference's loop -- the fused raster / service launch calls this for the few instances it resets or finishes, inside a
// register budget that sets how many frame workgroups a CU holds (the whole-record form there: 310 us per launch instead of 135).
template <bool WHOLE>
__device__ void emp_fill_desc(const MysteryParams& P, const MysteryIO& io, int i, const MysteryCore& s, MysteryDesc& d, int nx, SegRec& R,
                              const SegRec Rprev) {  // (by value: as a reference the caller's record stayed in scratch)
    memset(&d, 0, sizeof(d));
    d.valid = 1;
    d.sprite = s.rot8;
    d.sx = (int16_t)((s.sx * P.tile + P.agent_radius - P.sprite_dim / 2) - P.camera_offset);  // agent_draw_x (fixed at reset)
    d.sy = (int16_t)(s.ay - P.sprite_dim / 2);
    d.cross_on = (P.visual_feedback && s.cross_on) ? 1 : 0;
    d.cross_x = (int16_t)(s.cross_x - P.cross_dim / 2);
    d.cross_y = (int16_t)(s.cross_y - P.cross_dim / 2);
    d.bg_on = P.show_background ? 1 : 0;
    d.bg_phase = s.bg;
    if (P.show_stamina) {
        d.stamina_on = 1;
        int st = s.stamina < P.stamina_level ? s.stamina : P.stamina_level;
        d.stamina_red = (uint8_t)(int)(SCREEN * (1 - ((double)st / P.stamina_level)));
    }
    uint64_t mask0 = 0, mask1 = 0;
    if (P.show_past_path) {  // _draw_past_path (endless_mystery_path.py:111-132)
        const int x0 = nx - 1;
        if (x0 >= 0) {
            const int past_x = x0 - P.depth > 0 ? x0 - P.depth : 0;
            d.tile_x0 = past_x * P.tile - s.camera_x;
            // The reference walks the path backwards from the node before the agent's, tile by tile, until it has drawn one in column
            // past_x.  As a loop per lane that was the longest phase of the step under an agent that FOLLOWS its path (up to ~35 tiles,
            // each a run-time indexed byte of a record held in registers: 9.8 us of a wave's 20, profiles/r06_emp.md).  The walk only
            // ever touches the record of the current node's segment and the one before it (the window is at most depth + 2 <= 9 columns,
            // a segment has 8, and the stored path is 4-connected), so it is done on whole records instead: the STOP position = the
            // last node before the agent's in column past_x (four node bytes per word, exact zero-byte flags), the tiles = the nodes
            // between it and the agent's as a 64-bit occupancy (bit 8 x_rel + y), repacked to the descriptor's 7 bits per column.
            const int C = s.cur_node_seg;
            const bool cur_in_R = C == R.seg, cur_in_prev = C == Rprev.seg;
            // (depth < 2: the first node of the walk may already lie left of past_x when the agent has just stepped off the path -- the
            // reference's loop ends there; the whole-record form assumes the walk starts inside the window.  Uniform per handle.)
            const bool generic = !WHOLE || P.depth < 2 || !(cur_in_R || cur_in_prev) || (cur_in_R && C > 0 && Rprev.seg != C - 1);
            bool done_fast = false;
            if constexpr (WHOLE) if (__builtin_expect(!generic, 1)) {
                uint32_t wc[SEG_STRIDE / 4];
#pragma unroll
                for (int j = 0; j < SEG_STRIDE / 4; ++j) wc[j] = cur_in_R ? R.w[j] : Rprev.w[j];
                const int hi_c = s.cur_node_idx;  // positions 1 .. cur_node_idx hold the nodes before the agent's
                const int rel_c = past_x - C * (G + 1);
                const int stop_c = (rel_c >= 0 && hi_c >= 1) ? seg_last_in_column(wc, hi_c, rel_c) : 0;
                const uint64_t occ_c = seg_occupancy(wc, stop_c ? stop_c : 1, hi_c);
                emp_deposit(occ_c, C * (G + 1) - past_x, mask0, mask1);
                done_fast = true;
                if (!stop_c && C > 0) {  // the walk goes on in the segment before
                    if (cur_in_R) {
                        const int n_p = (int)(Rprev.w[0] & 0xFFu);
                        const int rel_p = past_x - (C - 1) * (G + 1);
                        const int stop_p = rel_p >= 0 ? seg_last_in_column(Rprev.w, n_p, rel_p) : 0;
                        const uint64_t occ_p = seg_occupancy(Rprev.w, stop_p ? stop_p : 1, n_p);
                        emp_deposit(occ_p, (C - 1) * (G + 1) - past_x, mask0, mask1);
                        if (!stop_p && C - 1 > 0) done_fast = false;  // (cannot happen: past_x >= 8 C - 8; the loop below is the definition)
                    } else {
                        done_fast = false;  // (the segment before the previous one: cannot happen either, see above)
                    }
                }
            }
            if (__builtin_expect(!done_fast, 0)) {  // the reference's loop, literally
                mask0 = mask1 = 0;
                int x = x0, seg = s.cur_node_seg, idx = s.cur_node_idx - 1;
                while (x >= past_x && x >= 0) {
                    if (idx < 0) {
                        seg--;
                        if (seg < 0) break;
                        R.load_or_take(io, i, seg, Rprev);
                        idx = R.byte(0) - 1;
                    }
                    R.load_or_take(io, i, seg, Rprev);
                    uint8_t b = R.byte(1 + idx);
                    x = node_x(seg, b);
                    int y = node_y(b);
                    int col = x - past_x;
                    if (col >= 0 && col < 16) {
                        const int cell = col * G + y;  // (a run-time index into d.tile_mask would put the descriptor into scratch)
                        const uint64_t bit = 1ull << (cell & 63);
                        if (cell < 64) mask0 |= bit;
                        else mask1 |= bit;
                    } else if (col >= 16) {
                        raise_error(io.err, 16);
                    }
                    if (x == past_x) break;
                    idx--;
                }
            }
        }
    }
    d.tile_mask[0] = mask0;
    d.tile_mask[1] = mask1;
}

// ---- lazy initial segments --------------------------------------------------------------------------------------------
// The reference's reset generates three path segments (endless_mystery_path.py:222-224 -> pygame_assets.py:523-527), ~30 us
// of dependent work each for a wave: the critical path of the step's fused raster / service launch.  Only the FIRST one is
// needed for the reset frame, its ground truth and the next steps (the agent starts eight tiles before the second): with
// P.lazy a reset generates one segment and records two as OWED (MysteryCore::path_len); each of the instance's next steps
// queues ONE owed segment as a background job nobody waits for -- served by the lane-per-path generator beside the frames
// (emp_raster_serve_kernel) -- and everything that could observe the difference generates what is owed first: a step that
// gets near the end of what exists (emp_step_a), the next reset (RNG order: the old episode's owed segments are generated,
// and discarded, before the new episode's first), and every look at the state (Family::sync_state: checkpoints, RNG words,
// the debug view).  The instance's random numbers are consumed in exactly the reference's order; nothing else draws from
// the stream of an Endless Mystery Path instance.
// ---- the next episode's first segment, ahead of time (round 5) ------------------------------------------------------------
// With lazy resets a step's queue still held one entry per finishing instance (~1,200 of 32,768 per step under random
// actions): one path of the wave-cooperative generator each, ~30 us of a wave's time and the reason the fused launch needs ~200
// registers per lane.  But nothing draws from an Endless-MysteryPath instance's stream except its segments, so once an episode's
// segments exist the stream stands exactly where the NEXT reset will find it -- unless the agent reaches the last-but-one segment
// first and a new one is appended.  P.pre: an instance that is owed nothing and has no such record generates the next episode's
// first segment as one more background job (lane-per-path generator, beside the frames, from a COPY of its stream) into
// io.aux[i], with the stream as it stands behind it; EMP_PRE(s) says the record is there.  A step that ends the episode then
// resets the instance itself (emp_step_b<true>): the record becomes segment 0, the instance's stream becomes the record's, two
// segments are owed -- the same draws in the same order as the reference's reset, and no queue entry.  Whatever advances
// the stream first (a due segment, any other reset path) clears the flag; the record is never looked at without it.

// EndlessMysteryPathEnv.reset (endless_mystery_path.py:195-280) around the three initial segments (serve_emp)
__device__ __forceinline__ void emp_pre_reset(MysteryCore& s) {
    s.t = 0;
    s.ep_sum = 0.0;
    s.ep_len = 0;
    s.num_seg = 0;
    s.have_start = 0;
    EMP_PRE(s) = 0;
}
// everything of the reset behind the segments except the frame descriptor; R: segment 0's record, its first node flagged
__device__ __forceinline__ void emp_post_reset_state(const MysteryParams& P, const MysteryIO& io, int i, MysteryCore& s, float* gt, SegRec& R) {
    const uint8_t b1 = R.byte(1);
    s.sx = (uint8_t)node_x(0, b1);
    s.sy = (uint8_t)node_y(b1);
    s.camera_x = P.camera_offset;
    s.bg = 0;
    s.ax = (int16_t)(s.sx * P.tile + P.agent_radius);
    s.ay = (int16_t)(s.sy * P.tile + P.agent_radius);
    s.rot8 = 6;  // 270 degrees
    s.cur_node_seg = 0;
    s.cur_node_idx = 0;
    emp_direction(io, i, s, gt, R);
    s.off = 0;
    s.cross_on = 0;
    s.cross_x = s.cross_y = 0;
    s.cur_seg = 0;
    s.fails = 0;
    s.n_falloff = 0;
    EMP_FLO(s) = 0x7FFFFFFF;  // no segment holds a stamina flag
    EMP_FHI(s) = -1;
    s.stamina = P.stamina_level;
    s.max_x = 0;
    s.tiles_visited = 0;
}
__device__ void emp_post_reset(const MysteryParams& P, const MysteryIO& io, int i, MysteryCore& s, MysteryDesc& d, float* gt) {
    SegRec R;
    R.seg = -1;
    R.load(io, i, 0);
    R.w[0] |= 0x4000u;  // the first node of the path shall not yield any reward (bit 6 of byte 1)
    *reinterpret_cast<uint32_t*>(seg_ptr(io, i, 0)) = R.w[0];
    emp_post_reset_state(P, io, i, s, gt, R);
    SegRec none;
    none.seg = -1;
    emp_fill_desc<false>(P, io, i, s, d, s.ax / P.tile, R, none);
    d.cross_on = 0;
    if (P.show_stamina) d.stamina_red = 0;
}

// EndlessMysteryPathEnv.step (endless_mystery_path.py:282-444), first part: move; returns 1 if a new segment is due
// (`current_segment > num_segments - 2`, :333-335), which the wave then generates before the second part runs.
// Bit 0 of the result: a segment is due; bit 1: the instance has reached the capacity of its segment store (EMP_CAP).
constexpr int EMP_DUE = 1, EMP_CAP = 2;
__device__ int emp_step_a(const MysteryParams& P, int i, MysteryCore& s, int a, int& nx, int& ny, int* io_err) {
    int a0 = a == 1 ? 2 : 0, a1 = a == 2 ? 1 : (a == 3 ? 2 : 0);
    if (!s.off) {
        int before = s.ax;
        move_agent(P, s, a0, a1, false);
        const int vx = s.ax - before;
        s.camera_x += vx;  // camera follows the agent's x velocity
        // bg_scroll -= velocity.x; once |bg_scroll| >= tile it becomes (|bg_scroll| % |velocity.x|) * sign, which is 0:
        // it has only ever moved in steps of the same velocity.x (endless_mystery_path.py:311-316)
        int bg = s.bg + vx;
        s.bg = (uint8_t)(bg >= P.tile ? bg % vx : bg);
    } else {
        s.bg = 0;
        s.ax = (int16_t)(s.sx * P.tile + P.agent_radius);
        s.ay = (int16_t)(s.sy * P.tile + P.agent_radius);
        move_agent(P, s, 0, 0, false);
        s.camera_x = P.camera_offset;
    }
    nx = floordiv_pos(s.ax, P.tile);
    ny = floordiv_pos(s.ay, P.tile);
    s.cur_seg = nx / (G + 1);
    // `current_segment > num_segments - 2` counts the owed segments as the reference has them; and whatever this step could
    // read of a segment that is still owed (the next node's direction at the end of the last generated segment) makes the
    // owed ones due now -- conservative: within two columns of the end of what exists
    int owed = EMP_OWED(s);
    if (s.cur_seg > s.num_seg + owed - 2 && s.num_seg + owed >= P.seg_cap) {
        // the segment store is full (the reference's list is unbounded, pygame_assets.py:559): nothing is appended, this step ends the
        // episode and says why (include/memgym.h: mg_info_buffers.capacity_dev, error bit 4); what is owed is generated if the step can see it
        raise_error(io_err, 4);
        return 2 | ((owed > 0 && nx >= (G + 1) * s.num_seg - 2) ? 1 : 0);
    }
    if (s.cur_seg > s.num_seg + owed - 2) {
        // Round 5: the segment the reference appends now (:333-335) is OWED like a reset's second and third -- the agent has only
        // entered the last but one, the new one starts eight columns ahead -- and generated by the next background job instead of by
        // a queue entry of this step (an agent that follows its path appended one every ~8 steps: thousands of cooperative paths per
        // step at 32,768 instances).  Nothing else draws from the stream, so the order of its draws is the reference's.
        if (!P.lazy_append || owed >= 200) return 1;
        EMP_OWED(s) = (uint8_t)(++owed);
        EMP_PRE(s) = 0;  // (a record ahead of time continued the stream as it stood BEFORE this segment)
    }
    return (owed > 0 && nx >= (G + 1) * s.num_seg - 2) ? 1 : 0;
}
#ifdef MG_LAB_EMP_CLOCK
__global__ void lab_wbl2_kernel() { asm volatile("buffer_wbl2 sc0 sc1\n\ts_waitcnt vmcnt(0)" ::: "memory"); }
#endif
#ifdef MG_LAB_EMP_CLOCK  // measurement builds only: phases of emp_step_kernel per wave (constant-rate clock, 10 ns)
static __device__ unsigned long long g_lab_step_clock[12 * 4096];
#define LAB_STEP_CLOCK(slot) do { const int wv_ = (blockIdx.x * blockDim.x + threadIdx.x) >> 6; \
    __builtin_amdgcn_s_waitcnt(0); /* everything issued so far has completed: the phases are what the wave waited for */ \
    if ((threadIdx.x & 63) == __builtin_ctzll(__ballot(1)) && wv_ < 4096) g_lab_step_clock[12 * wv_ + (slot)] = wall_clock64(); } while (0)
#else
#define LAB_STEP_CLOCK(slot) do { } while (0)
#endif

// second part; returns true if the instance finished and is to be reset in this call by somebody else (a queue entry).
// OWN_RESET (emp_step_kernel): an instance whose next episode's first segment exists already (EMP_PRE) is reset right here, and
// everything the step can need from memory -- the segment records AND the instance's aux line (that record, the head of the
// fall-off list) -- is requested in ONE batch: the kernel is a chain of dependent memory round trips on 512 waves (~2 us each on a
// memory system the observation stream has just swept; rounds 3-4: records, then the fall-off list, then the stamina flags'
// records, then the queue's counter), not a matter of bytes (profiles/r03_emp.md section 7, r05_emp.md).
// FINAL (mg_step with mg_info_buffers.final_obs_dev, round 6): the frame descriptor of a finished instance's TERMINAL state goes to
// io.tdesc[i] before anybody resets it (drawn into final_obs_dev by a sparse raster launch behind the fused one); the tail runs twice for
// such an instance -- one copy of emp_fill_desc either way, and for FINAL = false the code of rounds 3-5 (a loop of exactly one pass).
template <bool OWN_RESET, bool WHOLE_DESC = false, bool FINAL = false>
__device__ bool emp_step_b(const MysteryParams& P, const MysteryIO& io, int i, MysteryCore& s, int nx, int ny, float* reward_out,
                           uint8_t* done_out, float* gt, const mg_info_buffers& info, int autoreset, MysteryDesc& d, bool cap = false) {
    typedef uint32_t q4 __attribute__((ext_vector_type(4)));
    double reward = 0.0;
    bool done = cap;  // cap: the segment store is full and the reference would append now (emp_step_a) -- the episode ends here
    const int seg = s.cur_seg;
    SegRec R, Rprev;
    R.seg = -1;
    Rprev.seg = -1;
    // The segment store is cold (6.6 KB per instance, evicted by the observation stream): every dependent access is a ~2 us
    // round trip.  The records this step can touch -- the agent's segment, the one before it (past-path tiles), the head of
    // the one after it (the direction to the next node) -- are requested together, before the first of them is used.
    // (the next segment's head is read unconditionally, from a clamped index: inside a branch the compiler consumed it there
    // and waited for it before the two records were even requested)
    const int nxt_seg = seg + 1 < s.num_seg ? seg + 1 : -1;
    const int nxt_safe = nxt_seg >= 0 ? nxt_seg : 0;
    uint32_t* const aux = io.aux + (size_t)i * AUX_WORDS;
    uint32_t nxt_w0 = *reinterpret_cast<const uint32_t*>(seg_ptr(io, i, nxt_safe));
    // (both records unconditionally too, from clamped indices -- segment 0's slot always exists: behind a branch the compiler waits
    // for a load where the branches join, i.e. before the next request is issued)
    const bool have_prev = seg >= 1 && seg - 1 < s.num_seg, have_cur = seg >= 0 && seg < s.num_seg;
    Rprev.load(io, i, have_prev ? seg - 1 : 0);
    R.load(io, i, have_cur ? seg : 0);
    Rprev.seg = have_prev ? seg - 1 : -1;
    R.seg = have_cur ? seg : -1;
    q4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0, a4 = a0, f0 = a0, f1 = a0, f2 = a0;
    if (OWN_RESET) {
        const q4* aq = reinterpret_cast<const q4*>(aux);
        a0 = aq[0]; a1 = aq[1]; a2 = aq[2]; a3 = aq[3]; a4 = aq[4];  // the record generated ahead of time
        f0 = aq[5]; f1 = aq[6]; f2 = aq[7];                          // fall-off keys 0..11
        asm volatile("" : "+v"(f2));  // (a use the compiler cannot move: every request above is issued before the first wait)
    }
    asm volatile("" : "+v"(nxt_w0));
    LAB_STEP_CLOCK(5);
    bool on_path = false;
    if (seg < s.num_seg) {
        uint8_t* sp = seg_ptr(io, i, seg);
        const uint32_t* w = R.w;
        const int n = (int)(w[0] & 0xFFu);
        const int dx = nx - seg * (G + 1);
        const bool addressable = (unsigned)dx < 8u && (unsigned)ny < 8u;  // node bytes hold x_rel and y in 3 bits each
        const uint32_t target = (uint32_t)(dx & 7) | ((uint32_t)(ny & 7) << 3);
        // first node (list order) on the agent's tile, four node bytes per word at once (round 5; byte by byte the search was
        // 2.7 us of every wave's 17): a byte's low six bits equal the target iff they XOR to zero, and in (x - 0x01..) & ~x &
        // 0x80.. the LOWEST flag marks the lowest zero byte exactly.  Bytes behind the list are zero and can only match behind
        // every real node: a first match beyond the node count means there is none.
        int hit = 0;
        uint32_t hb = 0;
        const uint32_t t4 = target * 0x01010101u;
#pragma unroll
        for (int j = SEG_STRIDE / 4 - 1; j >= 0; --j) {  // (downwards: the lowest word with a match is taken last)
            uint32_t x = (w[j] & 0x3F3F3F3Fu) ^ t4;
            if (j == 0) x |= 0xFFu;  // byte 0 is the node count
            const uint32_t z = (x - 0x01010101u) & ~x & 0x80808080u;
            if (z) {
                const int k = (__ffs((int)z) - 1) >> 3;
                hit = 4 * j + k;
                hb = (w[j] >> (8 * k)) & 0xFFu;
            }
        }
        if (!(addressable && hit >= 1 && hit <= n)) hit = 0;
        if (hit) {
            uint8_t b = (uint8_t)hb;
            on_path = true;
            s.cur_node_seg = seg;
            s.cur_node_idx = hit - 1;
            bool is_start = nx == s.sx && ny == s.sy;
            if (!(b & 0x40) && !is_start) {
                reward += P.r_progress;
                s.tiles_visited++;
                b |= 0x40;
            }
            if (!(b & 0x80) && !is_start) {
                reward += P.r_dense;
                s.stamina = P.stamina_level;
                b |= 0x80;
                EMP_FLO(s) = seg < EMP_FLO(s) ? seg : EMP_FLO(s);  // segments that may hold stamina flags
                EMP_FHI(s) = seg > EMP_FHI(s) ? seg : EMP_FHI(s);
            }
            sp[hit] = b;
        }
    }
    LAB_STEP_CLOCK(6);
    if (!on_path) {
        reward += P.r_fall;
        s.fails++;
        if (P.visual_feedback) s.cross_on = 1;
        s.off = 1;
        if (nx < s.max_x) {
            done = true;
        } else {
            uint32_t* fl = aux + AUX_FALL;
            uint32_t key = (uint32_t)(nx & 0xFFFF) | ((uint32_t)(ny + 1024) << 16);
            const int nf = s.n_falloff;
            bool found = false;
            int k0 = 0;
            if (OWN_RESET) {  // the first twelve keys came with the batch above; slots >= n_falloff hold stale keys
                found = (0 < nf && f0.x == key) || (1 < nf && f0.y == key) || (2 < nf && f0.z == key) || (3 < nf && f0.w == key) ||
                        (4 < nf && f1.x == key) || (5 < nf && f1.y == key) || (6 < nf && f1.z == key) || (7 < nf && f1.w == key) ||
                        (8 < nf && f2.x == key) || (9 < nf && f2.y == key) || (10 < nf && f2.z == key) || (11 < nf && f2.w == key);
                k0 = 12;
            }
            for (int k = k0; k < nf; k += 4) {  // four entries per load
                const uint4 v = reinterpret_cast<const uint4*>(fl)[k >> 2];
                found = found || v.x == key || (k + 1 < nf && v.y == key) || (k + 2 < nf && v.z == key) || (k + 3 < nf && v.w == key);
            }
            if (found) done = true;
            if (!found) {
                if (s.n_falloff < P.fall_cap) fl[s.n_falloff++] = key;
                else {  // the list of fall-off cells is full (the reference's is unbounded, endless_mystery_path.py:385-393): the episode ends
                    raise_error(io.err, 8);
                    done = cap = true;
                }
            }
        }
        // reset all stamina flags -- only segments visited since the last reset can hold any; whole records at a time
        // (bytes past the node count are unused).  The agent's segment and the one before it are in registers already, as they
        // are in memory (no node was flagged in this step: the agent is not on the path).
        for (int q = EMP_FLO(s); q <= EMP_FHI(s) && q < s.num_seg; ++q) {
            uint32_t* wp = reinterpret_cast<uint32_t*>(seg_ptr(io, i, q));
            uint32_t w[SEG_STRIDE / 4];
            if (q == R.seg) {
#pragma unroll
                for (int j = 0; j < SEG_STRIDE / 4; ++j) w[j] = R.w[j];
            } else if (q == Rprev.seg) {
#pragma unroll
                for (int j = 0; j < SEG_STRIDE / 4; ++j) w[j] = Rprev.w[j];
            } else {
#pragma unroll
                for (int j = 0; j < SEG_STRIDE / 4; ++j) w[j] = wp[j];
            }
            wp[0] = w[0] & 0x7F7F7FFFu;  // byte 0 is the node count
#pragma unroll
            for (int j = 1; j < SEG_STRIDE / 4; ++j) wp[j] = w[j] & 0x7F7F7F7Fu;
        }
        EMP_FLO(s) = 0x7FFFFFFF;
        EMP_FHI(s) = -1;
        s.stamina = P.stamina_level;
    } else {
        s.cross_on = 0;
        s.off = 0;
    }
    LAB_STEP_CLOCK(7);
    s.cross_x = (int16_t)(s.ax - s.camera_x);
    s.cross_y = s.ay;
    reward += P.r_step;
    s.stamina--;
    if (s.stamina == 0) done = true;
    s.t++;
    if (s.t == P.max_steps) done = true;
    emp_direction(io, i, s, gt, R, nxt_seg, nxt_w0);
    LAB_STEP_CLOCK(8);
    if (nx > s.max_x && on_path) s.max_x = nx;
    s.ep_sum += reward;
    s.ep_len++;
    if (done) {
        if (info.ep_reward_dev) info.ep_reward_dev[i] = s.ep_sum;
        if (info.ep_length_dev) info.ep_length_dev[i] = s.ep_len;
        if (info.aux_dev[0]) info.aux_dev[0][i] = (float)s.fails;
        if (info.aux_dev[1]) info.aux_dev[1][i] = (float)s.max_x;
        if (info.aux_dev[2]) info.aux_dev[2][i] = (float)s.tiles_visited;
    }
    reward_out[i] = (float)reward;
    if (info.reward64_dev) info.reward64_dev[i] = reward;  // the reference's Python float, unrounded
    done_out[i] = done ? 1 : 0;
    if (info.capacity_dev) info.capacity_dev[i] = cap ? 1 : 0;
    LAB_STEP_CLOCK(9);
    bool fresh = false;
    auto own_reset = [&]() {
        // EndlessMysteryPathEnv.reset (endless_mystery_path.py:195-280) with the first segment taken from the record that was
        // generated ahead of time; the stream continues behind that segment's draws, the other two segments are owed
        emp_pre_reset(s);
        R.w[0] = a0.x | 0x4000u;  // the first node of the path shall not yield any reward
        R.w[1] = a0.y; R.w[2] = a0.z; R.w[3] = a0.w;
        R.w[4] = a1.x; R.w[5] = a1.y; R.w[6] = a1.z; R.w[7] = a1.w;
        R.w[8] = a2.x; R.w[9] = a2.y; R.w[10] = a2.z; R.w[11] = a2.w;
        R.w[12] = a3.x;
        R.seg = 0;
        Rprev.seg = -1;
        uint32_t* dst = reinterpret_cast<uint32_t*>(seg_ptr(io, i, 0));
#pragma unroll
        for (int j = 0; j < SEG_STRIDE / 4; ++j) dst[j] = R.w[j];
        io.rng.s_lo[i] = (uint64_t)a4.x | ((uint64_t)a4.y << 32);
        io.rng.s_hi[i] = (uint64_t)a4.z | ((uint64_t)a4.w << 32);
        io.rng.buf[i] = (uint64_t)a3.y | ((uint64_t)(a3.z & 1u) << 32);
        s.num_seg = 1;
        s.have_start = 1;
        s.end_y = (int8_t)((a3.z >> 8) & 0xFFu);
        EMP_OWED(s) = 2;
        if (LAB_BUILD && io.stats) atomicAdd(io.stats + 2, 1ull);  // mg_debug_counter "emp_own_resets" (lab build: tests)
        emp_post_reset_state(P, io, i, s, gt, R);
        nx = s.ax / P.tile;
        fresh = true;
    };
    if (!FINAL) {  // (rounds 3-5, as it was)
        if (done && autoreset) {
            if (!(OWN_RESET && P.lazy && EMP_PRE(s) && EMP_OWED(s) == 0)) return true;
            own_reset();
        }
        emp_fill_desc<WHOLE_DESC>(P, io, i, s, d, nx, R, Rprev);
    } else {
        const bool fin = done && autoreset;
#pragma nounroll
        for (int pass = fin ? 0 : 1; pass < 2; ++pass) {  // a finished instance: the terminal descriptor first
            if (fin && pass == 1) {
                if (!(OWN_RESET && P.lazy && EMP_PRE(s) && EMP_OWED(s) == 0)) return true;
                own_reset();
            }
            emp_fill_desc<WHOLE_DESC>(P, io, i, s, d, nx, R, Rprev);
            if (pass == 0) {
                io.tdesc[i] = d;
                if (LAB_BUILD && !OWN_RESET && io.stats) atomicAdd(io.stats + 4, 1ull);  // mg_debug_counter "emp_final_served" (lab build: tests)
            }
        }
    }
    if (fresh) {
        d.cross_on = 0;
        if (P.show_stamina) d.stamina_red = 0;
    }
    LAB_STEP_CLOCK(10);
    return false;
}

// Debug descriptors from the state and the current frame descriptors (see MysteryDebugComposer; oracle/mgo_mystery.c
// mpf_debug / emp_debug).
template <bool PS>
__global__ __launch_bounds__(256) void mystery_debug_desc_kernel(MysteryParams P0, MysteryIO io, MysteryDesc* out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P0.n) return;
    const MysteryParams& P = PS ? io.sets[set_index(io.set_of, i)] : P0;
    const MysteryCore s = io.core[i];
    MysteryDesc d = io.desc[i];
    d.valid = 1;
    d.tile_mask[0] = d.tile_mask[1] = 0;
    if (!P.endless) {
        d.pad8[0] = 1;
        const uint64_t ends = (1ull << (s.sx * G + s.sy)) | (1ull << (s.ex * G + s.ey));
        d.tile_mask[0] = s.path_mask & ~ends;
        d.tile_mask[1] = io.walls[i];
        d.goal_on = d.origin_on = 1;
        d.goal_x = s.ex; d.goal_y = s.ey;
        d.origin_x = s.sx; d.origin_y = s.sy;
        d.tile_x0 = 0;
        d.cross_on = s.cross_on;  // the cross surface keeps its alpha: the observation's visibility
    } else {
        d.pad8[0] = 2;
        const int x0 = floordiv_pos(s.camera_x, P.tile);
        d.tile_x0 = x0 * P.tile - s.camera_x;
        const int seg_lo = x0 / (G + 1) - 1, seg_hi = (x0 + 16) / (G + 1) + 1;
        for (int seg = seg_lo < 0 ? 0 : seg_lo; seg <= seg_hi && seg < s.num_seg; ++seg) {
            const uint8_t* sp = seg_ptr(io, i, seg);
            const int cnt = sp[0];
            for (int k = 1; k <= cnt; ++k) {
                const int col = node_x(seg, sp[k]) - x0, y = node_y(sp[k]);
                if (col >= 0 && col < 16) {
                    const int cell = col * G + y;
                    d.tile_mask[cell >> 6] |= 1ull << (cell & 63);
                }
            }
        }
        d.stamina_on = 1;  // blitted whatever show_stamina says
        const int st = s.stamina < P.stamina_level ? s.stamina : P.stamina_level;
        d.stamina_red = (uint8_t)(int)(SCREEN * (1 - ((double)st / P.stamina_level)));
    }
    out[i] = d;
}

__global__ __launch_bounds__(256) void mystery_init_kernel(int n, MysteryCore* core) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    MysteryCore s;
    memset(&s, 0, sizeof(s));
    core[i] = s;
}

// Both kernels keep every lane of a wave alive to the end (lanes beyond n or masked out simply request nothing):
// the path service needs converged waves.
// Instance -> lane mapping: only the first `lpw` lanes of a wave carry instances (lpw = 64, 32, ..., 4), the others are
// pure helpers of the path service.  A wave serves its requests one after another, so a workload whose instances
// reset often (Endless Mystery Path under a random policy: ~2.4 resets x 3 segments per 64 instances and step) is
// spread over more, shorter-lived waves; 32,768 instances are only 512 full waves on 1,024 SIMDs anyway.
__device__ __forceinline__ int instance_of_lane(int lpw, bool& worker) {
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    worker = lane < lpw;
    return wave * lpw + lane;
}

template <bool PS>
__global__ __launch_bounds__(256) void mystery_reset_kernel(MysteryParams P0, MysteryIO io, const int64_t* seeds,
                                                            const uint8_t* mask, float* gt, int lpw) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    path_ws_init(smem);
    const PathWS W{smem, io.jump, io.stats};
    bool worker;
    const int i = instance_of_lane(lpw, worker);
    const bool in_range = worker && i < P0.n;
    const MysteryParams& P = (PS && in_range) ? io.sets[set_index(io.set_of, i)] : P0;  // (PS: per-instance option sets)
    const bool active = in_range && !(mask && !mask[i]);
    if (in_range && !active) io.desc[i].valid = 0;
    Pcg g;
    MysteryCore s;
    MysteryDesc d;
    if (active) {
        if (seeds) g.seed((uint64_t)seeds[i]);
        else g.load(io.rng, i);
        s = io.core[i];
    } else {
        g.state = g.inc = 0; g.buf = 0; g.has = false;
        memset(&s, 0, sizeof(s));
    }
    {
        PathReq req;
        req.need = 0; req.sx = req.sy = req.ex = req.ey = 0;
        if (active) req = mp_pre_reset(P, s, g);
        int len = 0;
        uint64_t pm = 0;
        serve_mp(W, req, g, io.err, len, pm, io.walls, i);
        if (active) mp_post_reset(P, s, req, len, pm, d);
    }
    if (active) {
        io.core[i] = s;
        g.store(io.rng, i);
        io.desc[i] = d;
    }
}

// defer != 0: an instance that resets in this call gets everything but its path here (the draws in front of it, the agent's
// start, the frame descriptor -- a reset frame shows nothing of the path) and is queued; the queue is served by the first
// workgroups of the raster launch that follows (mystery_raster_paths_kernel).  The launch no longer lasts as long as one noisy A* (23 us) whenever any of
// its instances resets (MysteryPath-Grid: 0.5 % of them per step).
constexpr int HYBRID_INLINE = 2;
// FINAL (round 6): a call that keeps terminal observations (mg_info_buffers.final_obs_dev).  An instance that finishes leaves the descriptor
// of its TERMINAL frame in io.tdesc[i] and marks the reset frame's descriptor (pad8[1]; pad8[0] is the debug view's); the frame workgroup of the raster launch draws
// the terminal frame into the caller's final-observation buffer first (mystery_raster_paths_kernel<FMT, true>).
template <bool PS, bool FINAL = false>
__global__ __launch_bounds__(256) void mystery_step_kernel(MysteryParams P0, MysteryIO io, const int32_t* actions,
                                                           float* reward_out, uint8_t* done_out, float* gt,
                                                           mg_info_buffers info, int autoreset, int lpw, int defer) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    path_ws_init(smem);
    const PathWS W{smem, io.jump, io.stats};
    bool worker;
    const int i = instance_of_lane(lpw, worker);
    const bool active = worker && i < P0.n;
    const MysteryParams& P = (PS && active) ? io.sets[set_index(io.set_of, i)] : P0;  // (PS: per-instance option sets)
    MysteryCore s;
    Pcg g;
    MysteryDesc d;
    int act0 = 0, act1 = 0;
    if (active) {
        // the action is requested together with the state record (read where it is used -- behind a test of the state -- it
        // was a second memory round trip at the head of the kernel)
        // (both reads unconditional, the grid variant's second one a repeat of the first: a load inside the variant's branch
        // is waited for at the end of that branch)
        act0 = actions[P.grid ? i : 2 * i];
        act1 = actions[P.grid ? i : 2 * i + 1];
        s = load_core(&io.core[i]);
        g.load(io.rng, i);
        asm volatile("" : "+v"(act0), "+v"(act1));  // (a use the compiler cannot move below the record's first use)
    } else {
        memset(&s, 0, sizeof(s));
        g.state = g.inc = 0; g.buf = 0; g.has = false;
    }
    bool reset_me = false;
    {
        if (active) reset_me = mp_step(P, i, s, act0, act1, reward_out, done_out, info, autoreset, d);
        if constexpr (FINAL) {
            if (reset_me) {
                MysteryDesc td;
                mp_desc(P, s, td);
                io.tdesc[i] = td;
            }
        }
        PathReq req;
        req.need = 0; req.sx = req.sy = req.ex = req.ey = 0;
        if (reset_me) req = mp_pre_reset(P, s, g);
        int len = 0;
        uint64_t pm = 0;
        // defer 2 (hybrid): a wave serves up to HYBRID_INLINE requests itself and queues them all once it has more -- the
        // steps in which nearly every instance is truncated at once (t == max_steps for everybody who survived: all 16
        // instances of every wave, 16 x 25 us in a row) go to the raster launch, where every resident workgroup helps
        const bool queue_mine = defer == 1 || (defer == 2 && __popcll(__ballot(req.need != 0)) > HYBRID_INLINE);
        if (!queue_mine) serve_mp(W, req, g, io.err, len, pm, io.walls, i);
        if (reset_me) {
            mp_post_reset(P, s, req, len, pm, d);  // (queued: path_mask / path_len are filled in by the raster launch's path service)
            if constexpr (FINAL) d.pad8[1] = 1;
            if (queue_mine) queue_push(io.queue, &io.qctr[QC_COUNT], P.n, i, io.err);
        }
    }
    if (active) {
        g.store(io.rng, i);  // unchanged streams are rewritten with the same words
        io.core[i] = s;
        io.desc[i] = d;
    }
}

// Endless Mystery Path: nothing that generates a path is served by the wave that carries the instance.  A reset needs
// three path generations in a row (~70 us of dependent work for one wave), a new segment one, and a wave that happened
// to hold two or three such instances set the duration of the whole launch (profiles/r01e_logic_tails.md).
// emp_step_kernel (one lane per instance, no LDS) only queues those instances; emp_serve_kernel spreads the queue over
// the chip, one wave per entry at a time, lane 0 playing the instance's lane for the unchanged serve_emp /
// emp_step_b / emp_post_reset.  The last workgroup out clears the counters, so the launches can be replayed from a HIP
// graph.  Entries: instance | EMP_Q_SEGMENT = "append one segment, then finish the step (which may end in a reset)";
// plain instance = "reset".
constexpr int EMP_Q_SEGMENT = 1 << 30;
constexpr int EMP_Q_OWED = 1 << 29;  // "generate one of the segments this instance is owed" (a background job served like an entry: bg_coop)
constexpr int EMP_Q_INST = EMP_Q_OWED - 1;

template <bool PS, bool FINAL = false>
__global__ __launch_bounds__(256) void emp_step_kernel(MysteryParams P0, MysteryIO io, const int32_t* actions, float* reward_out,
                                                       uint8_t* done_out, float* gt, mg_info_buffers info, int autoreset) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P0.n) return;
    // (fewer instance-carrying lanes per wave -- 32 / 16 / 8, as the finite variants' kernel has them -- measured slower: 34-40 us
    // against 25-29, profiles/r05_emp.md)
    const MysteryParams& P = PS ? io.sets[set_index(io.set_of, i)] : P0;  // (PS: per-instance option sets)
    LAB_STEP_CLOCK(0);
    int act = actions[i];  // requested together with the state record ...
    MysteryCore s = load_core(&io.core[i]);
    asm volatile("" : "+v"(act));  // ... (a use the compiler cannot move: without it the request is issued after the record has arrived)
    int nx = 0, ny = 0;
    const int ra = emp_step_a(P, i, s, act, nx, ny, io.err);
    const int due = ra & EMP_DUE;
    LAB_STEP_CLOCK(1);
    MysteryDesc d;
    bool q = false, bg = false;
    if (due) {  // the agent entered the last-but-one segment: the rest of its step needs the new one
        queue_push(io.queue, &io.qctr[QC_COUNT], P.n, i | EMP_Q_SEGMENT, io.err);
    } else {
        q = emp_step_b<!PS, true, FINAL>(P, io, i, s, nx, ny, reward_out, done_out, gt ? gt + 3 * i : nullptr, info, autoreset, d, (ra & EMP_CAP) != 0);
        LAB_STEP_CLOCK(2);
        if (q) {
            queue_push(io.queue, &io.qctr[QC_COUNT], P.n, i, io.err);
            d.valid = DESC_QUEUED;
        } else {
            // one owed segment per step -- or, when nothing is owed, the next episode's first (EMP_PRE) -- as a job nobody waits for.
            // The record ahead of time only once the episode CAN end soon: the agent is off the path or behind its frontier (a
            // fall-off there ends the episode, endless_mystery_path.py:385-393, and no new tile refills its stamina); an agent AT its
            // frontier appends segments, each of which would drop the record again (a path-following agent: one more path per
            // appended segment for nothing, tools/emp_policy_bench.py).
            bg = P.lazy && (EMP_OWED(s) > 0 || (P.pre && !EMP_PRE(s) && (s.off || nx < s.max_x)));
        }
    }
    LAB_STEP_CLOCK(3);
    io.core[i] = s;
    if (due) io.desc[i].valid = DESC_QUEUED;  // (the rest of the descriptor is last step's)
    else io.desc[i] = d;
    // Background jobs.  Small launches (bg_coop): entries of a queue the service waves pop behind the step's own entries.  The others:
    // a FLAG per instance, collected by the background workgroups of the raster launch (round 5; rounds 3-4 pushed there too -- one more
    // atomic on a counter all 512 waves share, ~1.5 us in every wave's path).
    if (P.lazy) {
        if (P.bg_coop) {
            if (bg) queue_push(io.bgq, &io.qctr[QC_BG_COUNT], P.n, i, io.err);
        } else {
            io.bgflag[i] = bg ? 1 : 0;
        }
    }
    LAB_STEP_CLOCK(4);
}

__global__ __launch_bounds__(256) void emp_enqueue_kernel(int n, MysteryIO io, const uint8_t* mask) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (mask[i]) queue_push(io.queue, &io.qctr[QC_COUNT], n, i, io.err);
    else io.desc[i].valid = 0;
}

// reset(seed=None, mask) the way the auto-reset STEP resets (round 6): an instance whose next episode's first segment exists already
// (EMP_PRE, nothing owed) is reset right here from that record -- the same stores as emp_step_b<true>'s own reset: the record becomes segment
// 0, the instance's stream becomes the record's, two segments are owed -- and everybody else becomes a queue entry (served lazily: one
// segment, two owed).  A step in the gymnasium vector convention (mg_step with final_obs_dev) is a step without auto-reset plus this masked
// reset: through emp_enqueue_kernel every finishing instance was three cooperative paths of the queue server, 92 us per step at 32,768.
__global__ __launch_bounds__(256) void emp_masked_reset_kernel(MysteryParams P, MysteryIO io, const uint8_t* mask, float* gt) {
    typedef uint32_t q4 __attribute__((ext_vector_type(4)));
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P.n) return;
    if (!mask[i]) {
        io.desc[i].valid = 0;
        return;
    }
    MysteryCore s = load_core(&io.core[i]);
    if (!(P.pre && EMP_PRE(s) && EMP_OWED(s) == 0)) {
        queue_push(io.queue, &io.qctr[QC_COUNT], P.n, i, io.err);
        return;
    }
    const q4* aq = reinterpret_cast<const q4*>(io.aux + (size_t)i * AUX_WORDS);
    const q4 a0 = aq[0], a1 = aq[1], a2 = aq[2], a3 = aq[3], a4 = aq[4];  // the record generated ahead of time (AUX_WORDS layout above)
    emp_pre_reset(s);
    SegRec R, none;
    none.seg = -1;
    R.w[0] = a0.x | 0x4000u;  // the first node of the path shall not yield any reward
    R.w[1] = a0.y; R.w[2] = a0.z; R.w[3] = a0.w;
    R.w[4] = a1.x; R.w[5] = a1.y; R.w[6] = a1.z; R.w[7] = a1.w;
    R.w[8] = a2.x; R.w[9] = a2.y; R.w[10] = a2.z; R.w[11] = a2.w;
    R.w[12] = a3.x;
    R.seg = 0;
    uint32_t* dst = reinterpret_cast<uint32_t*>(seg_ptr(io, i, 0));
#pragma unroll
    for (int j = 0; j < SEG_STRIDE / 4; ++j) dst[j] = R.w[j];
    io.rng.s_lo[i] = (uint64_t)a4.x | ((uint64_t)a4.y << 32);
    io.rng.s_hi[i] = (uint64_t)a4.z | ((uint64_t)a4.w << 32);
    io.rng.buf[i] = (uint64_t)a3.y | ((uint64_t)(a3.z & 1u) << 32);
    s.num_seg = 1;
    s.have_start = 1;
    s.end_y = (int8_t)((a3.z >> 8) & 0xFFu);
    EMP_OWED(s) = 2;
    if (LAB_BUILD && io.stats) atomicAdd(io.stats + 2, 1ull);  // mg_debug_counter "emp_own_resets" (lab build: tests)
    emp_post_reset_state(P, io, i, s, gt ? gt + 3 * i : nullptr, R);
    MysteryDesc d;
    emp_fill_desc<false>(P, io, i, s, d, s.ax / P.tile, R, none);
    d.cross_on = 0;
    if (P.show_stamina) d.stamina_red = 0;
    io.core[i] = s;
    io.desc[i] = d;
}

// One queue entry, served by one converged wave whose lane 0 plays the instance's lane: "append a segment, finish the step"
// and / or "reset" (three segments), state, stream and frame descriptor written back.
template <bool FINAL = false>
__device__ void emp_serve_entry(const MysteryParams& P, const MysteryIO& io, const PathWS& W, int entry, const int64_t* seeds, float* reward_out,
                                uint8_t* done_out, float* gt, const mg_info_buffers& info, int autoreset, MysteryDesc* d_out = nullptr) {
    const bool me = (threadIdx.x & 63) == 0;
    const int i = entry & EMP_Q_INST;
    float* gti = gt ? gt + 3 * i : nullptr;
    Pcg g;
    MysteryCore s;
    MysteryDesc d;
    if (me) {
        if (seeds) g.seed((uint64_t)seeds[i]);
        else g.load(io.rng, i);
        s = io.core[i];
    } else {
        g.state = g.inc = 0; g.buf = 0; g.has = false;
        memset(&s, 0, sizeof(s));
    }
    int reset_me = 1;
    if (entry & (EMP_Q_SEGMENT | EMP_Q_OWED)) {
        // the new segment is due: whatever is still owed comes first (stream order), or -- emp_step_a's conservative test --
        // only what is owed is due and `current_segment > num_segments - 2` does not hold yet
        int want = 0;
        bool cap = false;
        if (me) {
            if (entry & EMP_Q_OWED) {  // a background job: one owed segment, nothing else
                want = EMP_OWED(s) > 0 ? 1 : 0;
                EMP_OWED(s) = (uint8_t)(EMP_OWED(s) - want);
            } else {
                const bool append = s.cur_seg > s.num_seg + EMP_OWED(s) - 2;
                cap = append && s.num_seg + EMP_OWED(s) >= P.seg_cap;  // (emp_step_a has raised the error bit; the step below ends the episode)
                want = EMP_OWED(s) + ((append && !cap) ? 1 : 0);
                EMP_OWED(s) = 0;
            }
        }
        serve_emp(io, W, i, want, s, g);
        if (entry & EMP_Q_OWED) {  // only the fields a segment changes: the frame and the rest of the record are this step's already
            if (me && want) {
                io.core[i] = s;
                g.store(io.rng, i);
            }
            return;
        }
        if (me)
            reset_me = emp_step_b<false, false, FINAL>(P, io, i, s, floordiv_pos(s.ax, P.tile), floordiv_pos(s.ay, P.tile), reward_out, done_out,
                                         gti, info, autoreset, d, cap) ? 1 : 0;
        reset_me = bcast(reset_me, 0);
    }
    if (reset_me) {
        // segments the finished episode is still owed are generated first (and discarded): they come first in the stream
        const int owed_old = bcast((me && !seeds) ? (int)EMP_OWED(s) : 0, 0);  // (a re-seeded instance starts a new stream)
        if (owed_old) serve_emp(io, W, i, me ? owed_old : 0, s, g);
        if (me) emp_pre_reset(s);
        serve_emp(io, W, i, me ? (P.lazy ? 1 : 3) : 0, s, g);
        if (me) {
            EMP_OWED(s) = P.lazy ? 2 : 0;
            emp_post_reset(P, io, i, s, d, gti);
        }
    }
    if (me) {
        d.valid = DESC_SERVED;
        io.core[i] = s;
        g.store(io.rng, i);
        io.desc[i] = d;
        if (d_out) *d_out = d;  // (the fused kernel composes the frame from this copy)
    }
}

// ---- Lane-per-job path generation ---------------------------------------------------------------------------------
// The wave-cooperative generator above finishes ONE path in ~25-33 us, as a chain of ~10,000 wave-uniform (scalar)
// instructions; a full reset of 32,768 instances (98,304 paths) keeps every SIMD's scalar issue busy for 1.25 ms.  Here
// every LANE generates its own path: the same algorithm with the per-node records and the open list in LDS (one column
// per lane) -- ~105 us per path (~1,100 vector instructions per expansion, one wave per SIMD), 64 paths per wave: a
// full reset in three rounds of 512 waves.  Used where many paths are due at once and nothing else runs (mg_reset of all
// instances); a step's few thousand queue entries stay with the cooperative generator, whose latency is lower
// (profiles/r02_emp.md).
//   * open list: every node enters it at most once and its f never changes afterwards (the reference's `neighbor.g = g`
//     typo), so the list is append-only with a 64-bit mask of the positions still in it; "first i >= 1 with f[i] < f[0]"
//     walks the set bits.
//   * f = g_cost + sqrt(d2) is compared through an integer key (g_cost << 17) + round(sqrt(d2) * 2^17): over all g_cost
//     <= 459 and all 27 values of d2 the keys order exactly like the doubles and are equal exactly where those are
//     (distinct sums differ by >= 2.5e-3; tests/test_path_keys.py checks every pair).
constexpr int LW_KEY = 0;                          // uint32 key[52][64]: (fkey << 6) | node, by list position; later the path
constexpr int LW_NODE = LW_KEY + 52 * 64 * 4;      // uint16 rec[49][64]: g_cost | previous_node << 9 (63 = none), by node
constexpr int LW_HFIX = LW_NODE + 49 * 64 * 2;     // uint32 hfix[80]
constexpr int LW_BYTES = LW_HFIX + 80 * 4;
struct LaneWS {
    uint8_t* base;
    int lane;
    __device__ __forceinline__ uint32_t& key(int p) const { return reinterpret_cast<uint32_t*>(base + LW_KEY)[p * 64 + lane]; }
    __device__ __forceinline__ uint16_t& rec(int n) const { return reinterpret_cast<uint16_t*>(base + LW_NODE)[n * 64 + lane]; }
    __device__ __forceinline__ uint32_t hfix(int d2) const { return reinterpret_cast<const uint32_t*>(base + LW_HFIX)[d2]; }
};
__device__ __forceinline__ void lane_ws_init(uint8_t* smem) {  // all threads of the block
    for (int d = threadIdx.x; d < 80; d += blockDim.x)
        reinterpret_cast<uint32_t*>(smem + LW_HFIX)[d] = (uint32_t)__double2ll_rn(sqrt((double)d) * 131072.0);
    __syncthreads();
}
constexpr uint64_t grid_mask(int which) {  // 0: y == 0, 1: y == 6, 2: border
    uint64_t m = 0;
    for (int x = 0; x < G; ++x)
        for (int y = 0; y < G; ++y)
            if ((which == 0 && y == 0) || (which == 1 && y == G - 1) || (which == 2 && (x == 0 || x == G - 1 || y == 0 || y == G - 1)))
                m |= 1ull << (x * G + y);
    return m;
}
constexpr uint64_t GM_Y0 = grid_mask(0), GM_Y6 = grid_mask(1), GM_BORDER = grid_mask(2), GM_ALL = (1ull << (G * G)) - 1;
__device__ __forceinline__ uint64_t cells_around4(uint64_t m) {
    return (((m << 1) & ~GM_Y0) | ((m >> 1) & ~GM_Y6) | (m << G) | (m >> G)) & GM_ALL;
}
__device__ __forceinline__ uint64_t cells_around8(uint64_t m) {  // m itself included
    const uint64_t v = m | ((m << 1) & ~GM_Y0) | ((m >> 1) & ~GM_Y6);
    return (v | (v << G) | (v >> G)) & GM_ALL;
}

// MysteryPath.__init__ (pygame_assets.py:606-724) by one lane.  Returns the path length (-1: none); W.key(k), k < len, is
// the k-th path node (flat index x*7+y, END first like the reference's list).
__device__ int lane_path(Pcg& g, const LaneWS& W, int sx, int sy, int ex, int ey, uint64_t& path_mask, uint64_t& wall_out) {
    uint64_t wall = 0;
    for (int i = 1; i < G - 2; ++i)
        for (int j = 1; j < G - 2; ++j)
            if (g.integers(0, 100) < 33) wall |= 1ull << (i * G + j);
    const int start = sx * G + sy, end = ex * G + ey;
    const uint64_t ends = (1ull << start) | (1ull << end);
    uint64_t outer = GM_BORDER & ~ends & ~cells_around4(ends) & ~cells_around8(wall);
    int n_outer = __popcll(outer);
    const int n_iter = g.integers(0, 2) == 0 ? 4 : 8;  // rng.choice([4, 8])
    for (int it = 0; it < n_iter; ++it) {
        if (n_outer > 0) {
            const int k = g.integers(0, n_outer);
            uint64_t m = outer;
            for (int q = 0; q < k; ++q) m &= m - 1;
            const uint64_t bit = m & (~m + 1);
            wall |= bit;
            outer &= ~bit;
            --n_outer;
        }
    }
    wall_out = wall;
    uint64_t closed = 0, in_open = 1ull << start, live = 1;
    int n_pos = 1;
    W.key(0) = (uint32_t)start;  // (only the order of the keys matters: the start is alone in the list when it is taken)
    W.rec(start) = (uint16_t)(63u << 9);
    // "first i >= 1 with f[i] < f[0], else 0": while open[0] stays, the positions before the last hit are known not to beat
    // it (keys never change), so the walk resumes behind the hit; four keys are fetched per round trip to LDS
    int head = -1, scan = 0;
    uint32_t khead = 0;
    for (;;) {
        if (!live) return -1;
        const int p0 = __ffsll((unsigned long long)live) - 1;
        if (p0 != head) {
            head = p0;
            khead = W.key(p0);
            scan = p0 + 1;
        }
        const uint32_t k0 = khead >> 6;
        int w = p0;
        uint32_t kw = khead;
        for (int p = scan; p < n_pos && w == p0; p += 4) {
            const uint32_t a[4] = {W.key(p), W.key(p + 1), W.key(p + 2), W.key(p + 3)};
#pragma unroll
            for (int q = 3; q >= 0; --q)  // the lowest qualifying position wins
                if (p + q < n_pos && ((live >> (p + q)) & 1ull) && (a[q] >> 6) < k0) {
                    w = p + q;
                    kw = a[q];
                }
        }
        scan = w != p0 ? w + 1 : n_pos;
        const int cur = (int)(kw & 63u);
        if (cur == end) {
            int len = 0, t = cur;
            path_mask = 0;
            for (;;) {
                path_mask |= 1ull << t;
                const int pv = W.rec(t) >> 9;
                W.key(len++) = (uint32_t)t;
                if (pv == 63) break;
                t = pv;
            }
            return len;
        }
        live &= ~(1ull << w);
        in_open &= ~(1ull << cur);
        closed |= 1ull << cur;
        const int gcur = W.rec(cur) & 511;
        const int cx = cur / G, cy = cur - cx * G;
        const uint64_t blocked = closed | wall;
#pragma unroll
        for (int k = 0; k < 4; ++k) {  // Node.add_neighbors order x+1, x-1, y+1, y-1
            const int nb = k == 0 ? (cx < G - 1 ? cur + G : -1) : k == 1 ? (cx > 0 ? cur - G : -1) : k == 2 ? (cy < G - 1 ? cur + 1 : -1) : (cy > 0 ? cur - 1 : -1);
            if (nb < 0 || ((blocked >> nb) & 1ull)) continue;
            const int cost = gcur + 1 + (int)(g.next32() >> 29);  // integers(1, 9): span 8, never rejects
            if ((in_open >> nb) & 1ull) {
                const uint16_t r = W.rec(nb);
                if (cost < (int)(r & 511)) W.rec(nb) = (uint16_t)((r & 511) | (cur << 9));  // `neighbor.g = g` typo: g_cost stays
            } else {
                W.rec(nb) = (uint16_t)(cost | (cur << 9));
                const int ax = nb / G - ex, ay = nb % G - ey;
                W.key(n_pos) = ((((uint32_t)cost << 17) + W.hfix(ax * ax + ay * ay)) << 6) | (uint32_t)nb;
                live |= 1ull << n_pos;
                ++n_pos;
                in_open |= 1ull << nb;
            }
        }
    }
}

// EndlessMysteryPath.add_path_segment (pygame_assets.py:544-604) by one lane: the draws and the path; the record goes to dst
// (13 dwords: byte 0 = node count, then the path START first, then the transition node, see serve_emp).  Returns the end row.
__device__ int lane_segment_record(const MysteryIO& io, const LaneWS& W, bool have_start, int end_y, Pcg& g, uint32_t* dst) {
    const int sy = have_start ? end_y : g.integers(0, G);
    const int ey = g.integers(0, G);
    uint64_t pm = 0, wl = 0;
    int len = lane_path(g, W, 0, sy, G - 1, ey, pm, wl);
    if (len < 0) {
        raise_error(io.err, 2);
        len = 0;
    }
    if (dst) {
        for (int j = 0; j < SEG_STRIDE / 4; ++j) {
            uint32_t word = 0;
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const int p = 4 * j + b;
                uint32_t v = 0;
                if (p == 0) v = (uint32_t)(len + 1);
                else if (p <= len) {
                    const int nd = (int)W.key(len - p);
                    v = (uint32_t)((nd / G) | ((nd % G) << 3));
                } else if (p == len + 1) v = (uint32_t)(7 | (ey << 3));
                word |= v << (8 * b);
            }
            dst[j] = word;
        }
    }
    return ey;
}
__device__ void lane_segment(const MysteryIO& io, const LaneWS& W, int i, MysteryCore& s, Pcg& g) {
    const bool room = s.num_seg < io.seg_rows;
    const int ey = lane_segment_record(io, W, s.have_start != 0, (int)s.end_y, g, room ? reinterpret_cast<uint32_t*>(seg_ptr(io, i, s.num_seg)) : nullptr);
    if (room) s.num_seg++;
    else raise_error(io.err, 4);
    s.have_start = 1;
    s.end_y = (int8_t)ey;
    EMP_PRE(s) = 0;  // the stream has moved: a record generated ahead of time no longer continues it
}

// The queued resets of a deferred step are served INSIDE the raster launch: its first PATH_WGS workgroups do not draw frames
// but drain the queue, one wave per entry (entry w is wave w's first job, later ones come from a shared counter, the
// last of them out clears the counters: see emp_serve_kernel), then leave their slots to frame workgroups.  Lane 0 plays the
// instance: its stream stands right behind the draws of mp_pre_reset, the path's ends are in its record; the path, the walls
// and the stream come back.  No second stream, no events: a fork/join around a side-stream kernel cost 10 us per step.
constexpr int PATH_WGS = 128;
// Long queues (a step in which nearly every instance is truncated at once: t == max_steps for all survivors of a batch that
// was reset together -- every 128 steps for MysteryPath-Grid's defaults, every 512 for MysteryPath-v0): the first
// PATH_HELP_MAX FRAME workgroups serve entries as well before they start on their frames (a reset frame shows nothing of
// the path, so no frame waits for one).  128 + 1,664 = 1,792 = the workgroups resident at once (7 per CU): every wave of
// the chip's first round takes entries w, w + SW, w + 2 SW, ... (static striding: thousands of pops from one counter are
// 22 ns each, in series).  32,768 paths then take what the cooperative generator's scalar-issue bound allows (~80 paths
// per us chip-wide, profiles/r03_mass_resets.md) instead of 64 paths in a row on 512 waves (1.3 ms).
constexpr int PATH_HELP_MAX = 1664;
constexpr int PATH_MASS = 2 * 4 * PATH_WGS;  // more than two entries per dedicated wave: call for help
// ... and with the LANE-per-path generator (lane_path: 64 paths per wave, twice the cooperative generator's throughput when
// there are enough paths to fill the lanes -- 107 vs 215 us per 32,768 paths, profiles/r03_mass_resets.md): wave 0 of every
// participating workgroup takes 64 entries, one per lane, its workspace (LW_BYTES = 19,904 B) is the workgroup's frame.
static_assert(LW_BYTES <= RASTER_LDS, "the lane generator's workspace must fit into the raster workgroup's LDS");
template <int FMT, bool FINAL = false>
__global__ __launch_bounds__(256, 7) void mystery_raster_paths_kernel(const MysteryDesc* __restrict__ descs, RasterAtlas A, void* __restrict__ obs,
                                                                      int n, MysteryParams P, MysteryIO io, void* __restrict__ final_obs) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    if (blockIdx.x < PATH_WGS + PATH_HELP_MAX) {
        // (every one of these workgroups reads the count BEFORE the last participant can clear it: in the long-queue case the
        // clearing waits for all of them, in the short-queue case whatever a late frame workgroup reads -- the count or 0 -- tells
        // it not to take part)
        const int count = queue_count(&io.qctr[QC_COUNT], n);
        const bool mass = P.path_help && count > PATH_MASS;
        const bool by_lanes = mass && P.path_help == 1;  // MEMGYM_PATH_HELP=2: helpers with the cooperative generator
        const int want = by_lanes ? (count + 63) / 64 : (count + 3) / 4;  // workgroups for one entry per lane / per wave
        const int helpers = mass ? max(0, min(PATH_HELP_MAX, want - PATH_WGS)) : 0;
        const int busy = PATH_WGS + helpers;
        if ((int)blockIdx.x < busy) {
            if (by_lanes) {
                lane_ws_init(smem);
                if (threadIdx.x < 64) {
                    const LaneWS LW{smem, (int)threadIdx.x};
                    const unsigned long long t_in = io.stats ? wall_clock64() : 0ull;
                    int mine = 0;
                    for (int idx = blockIdx.x * 64 + threadIdx.x; idx < count; idx += busy * 64) {
                        ++mine;
                        const int i = io.queue[idx];
                        Pcg g;
                        g.load(io.rng, i);
                        const MysteryCore c = io.core[i];
                        uint64_t pm = 0, wl = 0;
                        int len = lane_path(g, LW, c.sx, c.sy, c.ex, c.ey, pm, wl);
                        if (len < 0) {
                            raise_error(io.err, 2);
                            len = 0;
                            pm = 0;
                        }
                        io.core[i].path_mask = pm;
                        io.core[i].path_len = (uint8_t)len;
                        if (io.walls) io.walls[i] = wl;
                        g.store(io.rng, i);
                    }
                    if (io.stats) {  // one wave's time for up to 64 paths side by side
                        const unsigned long long n_here = (unsigned long long)__popcll(__ballot(mine > 0)) ;
                        int total = mine;
                        for (int o = 32; o > 0; o >>= 1) total += __shfl_down(total, o);
                        if (threadIdx.x == 0 && n_here) {
                            atomicAdd(io.stats, wall_clock64() - t_in);
                            atomicAdd(io.stats + 1, (unsigned long long)total);
                        }
                    }
                }
            } else if (count > 0) {
                if (P.svc_prio) __builtin_amdgcn_s_setprio(3);
                path_ws_init(smem);
                const PathWS W{smem, io.jump, io.stats};
                const bool me = (threadIdx.x & 63) == 0;
                const int waves = busy * 4;
                for (int idx = bcast((int)(blockIdx.x * 4 + (threadIdx.x >> 6)), 0); idx < count; idx += waves) {
                    const int i = bcast(io.queue[idx], 0);
                    Pcg g;
                    PathReq req;
                    req.need = 0; req.sx = req.sy = req.ex = req.ey = 0;
                    if (me) {
                        g.load(io.rng, i);
                        const MysteryCore c = io.core[i];
                        req.need = 1; req.sx = c.sx; req.sy = c.sy; req.ex = c.ex; req.ey = c.ey;
                    } else {
                        g.state = g.inc = 0; g.buf = 0; g.has = false;
                    }
                    int len = 0;
                    uint64_t pm = 0;
                    serve_mp(W, req, g, io.err, len, pm, io.walls, i);
                    if (me) {
                        io.core[i].path_mask = pm;
                        io.core[i].path_len = (uint8_t)len;
                        g.store(io.rng, i);
                    }
                }
            }
            __builtin_amdgcn_s_setprio(0);
            __syncthreads();
            if (threadIdx.x == 0 && atomicAdd(&io.qctr[QC_LEFT], 1) == busy - 1) {  // last participant out
                io.qctr[QC_COUNT] = 0;
                io.qctr[QC_HEAD] = 0;
                io.qctr[QC_LEFT] = 0;
            }
            if (blockIdx.x < PATH_WGS) return;
            __syncthreads();  // a helper goes on to its frames: the path workspace in LDS is the frame from here on
        }
    }
    RasterCtx R;
    R.frame = smem;
    R.mask = reinterpret_cast<uint32_t*>(smem + FRAME_BYTES);
    R.A = A;
    R.T = A.tables;
    R.tid = threadIdx.x;
    const int tid = threadIdx.x, stride = (int)gridDim.x - PATH_WGS;
    for (int env = (int)blockIdx.x - PATH_WGS; env < n; env += stride) {
        const MysteryDesc* d = descs + env;
        if constexpr (FINAL) {
            if (d->pad8[1]) {  // the instance finished in this step: its terminal frame first, into the caller's final-observation buffer
                MysteryComposer::compose(io.tdesc + env, R);
                __syncthreads();
                store_frame<FMT, false>(smem, final_obs, env, tid);
                __syncthreads();
            }
        }
        if (MysteryComposer::skip(d)) continue;
        MysteryComposer::compose(d, R);
        __syncthreads();
        store_frame<FMT, false>(smem, obs, env, tid);  // (plain stores: non-temporal ones cost this launch 20 %, profiles/r04_emp.md)
        __syncthreads();
    }
}

// One background job by one lane: the instance's next owed segment (lazy initial segments, EMP_OWED) -- or, ahead = true and
// nothing owed, the NEXT episode's first segment from a copy of the stream (EMP_PRE).  (One call site of the generator for both:
// with two the compiler turns it into a real function call, 1,300 B of stack per lane in the fused launch.)
__device__ __forceinline__ void lane_owed_segment(const MysteryIO& io, const LaneWS& W, int i, int how_many, bool ahead = false) {
    MysteryCore s = io.core[i];
    int owed = EMP_OWED(s);
    const bool pre_job = owed <= 0;
    if (pre_job && (!ahead || EMP_PRE(s))) return;
    Pcg g;
    g.load(io.rng, i);
    uint32_t* const rec = io.aux + (size_t)i * AUX_WORDS;
    for (int k = 0; k < how_many && (owed > 0 || pre_job); ++k) {
        const bool room = s.num_seg < io.seg_rows;
        uint32_t* dst = pre_job ? rec : (room ? reinterpret_cast<uint32_t*>(seg_ptr(io, i, s.num_seg)) : nullptr);
        // (a reset's first segment draws its start row)
        const int ey = lane_segment_record(io, W, !pre_job && s.have_start != 0, (int)s.end_y, g, dst);
        if (pre_job) {
            rec[13] = g.buf;
            rec[14] = (g.has ? 1u : 0u) | ((uint32_t)ey << 8);
            rec[16] = (uint32_t)g.state;
            rec[17] = (uint32_t)(g.state >> 32);
            rec[18] = (uint32_t)(g.state >> 64);
            rec[19] = (uint32_t)(g.state >> 96);
            EMP_PRE(s) = 1;  // (the instance's own stream stays where it is)
            io.core[i] = s;
            if (io.stats) atomicAdd(io.stats + 3, 1ull);  // mg_debug_counter "emp_ahead_records"
            return;
        }
        if (room) s.num_seg++;
        else raise_error(io.err, 4);
        s.have_start = 1;
        s.end_y = (int8_t)ey;
        EMP_PRE(s) = 0;
        --owed;
    }
    // only the fields a segment changes: the instance's record belongs to nobody else between its step and its next step
    EMP_OWED(s) = (uint8_t)owed;
    io.core[i] = s;
    g.store(io.rng, i);
}

// Everything still owed, for every instance (Family::sync_state: before the state is looked at)
__global__ __launch_bounds__(64) void emp_flush_owed_kernel(MysteryParams P, MysteryIO io) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    lane_ws_init(smem);
    const LaneWS W{smem, (int)threadIdx.x};
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i < P.n) lane_owed_segment(io, W, i, 255);
}

// mg_reset of every Endless-MysteryPath instance: one LANE per instance (emp_serve_kernel: one wave per instance)
__global__ __launch_bounds__(64) void emp_reset_lanes_kernel(MysteryParams P, MysteryIO io, const int64_t* seeds, float* gt) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    lane_ws_init(smem);
    const LaneWS W{smem, (int)threadIdx.x};
    const int i = blockIdx.x * 64 + threadIdx.x;
    const bool active = i < P.n;
    Pcg g;
    MysteryCore s;
    g.state = g.inc = 0; g.buf = 0; g.has = false;
    memset(&s, 0, sizeof(s));
    int owed_old = 0;
    if (active) {
        s = io.core[i];
        if (seeds) g.seed((uint64_t)seeds[i]);
        else {
            g.load(io.rng, i);
            owed_old = EMP_OWED(s);  // reset(seed=None): what the old episode is owed comes first in the stream
        }
    }
    for (int k = 0; k < owed_old; ++k) lane_segment(io, W, i, s, g);  // (two of a lazy reset's, and appended ones: emp_step_a)
    if (active) {
        emp_pre_reset(s);
        EMP_OWED(s) = 0;
    }
    for (int k = 0; k < 3; ++k)
        if (active) lane_segment(io, W, i, s, g);
    if (active) {
        MysteryDesc d;
        emp_post_reset(P, io, i, s, d, gt ? gt + 3 * i : nullptr);
        io.core[i] = s;
        g.store(io.rng, i);
        io.desc[i] = d;
    }
}

// all != 0: mg_reset of every instance (entry k = instance k, seeds may be given); otherwise the queue is drained
template <bool PS>
__global__ __launch_bounds__(256) void emp_serve_kernel(MysteryParams P, MysteryIO io, const int64_t* seeds, int all, float* reward_out,
                                                        uint8_t* done_out, float* gt, mg_info_buffers info, int autoreset) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    path_ws_init(smem);
    const PathWS W{smem, io.jump, io.stats};
    const bool me = (threadIdx.x & 63) == 0;
    const int count = all ? P.n : queue_count(&io.qctr[QC_COUNT], P.n);
    // the first entry of wave w is entry w (no atomic: with thousands of idle waves the same-address atomics of their
    // failing pops were the launch time); later ones are popped from a shared counter that starts after the last wave
    const int waves = gridDim.x * (blockDim.x >> 6);
    int idx = bcast((int)(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)), 0);
    while (idx < count) {
        const int entry = all ? idx : bcast(io.queue[idx], 0);
        emp_serve_entry(PS ? io.sets[set_index(io.set_of, entry & EMP_Q_INST)] : P, io, W, entry, seeds, reward_out, done_out, gt, info, autoreset);
        if (me) {
            idx = waves + atomicAdd(&io.qctr[QC_HEAD], 1);
        }
        idx = bcast(idx, 0);
    }
    __syncthreads();
    if (threadIdx.x == 0 && atomicAdd(&io.qctr[QC_LEFT], 1) == (int)gridDim.x - 1) {  // last workgroup out
        io.qctr[QC_COUNT] = 0;
        io.qctr[QC_HEAD] = 0;
        io.qctr[QC_LEFT] = 0;
    }
}

// mg_step of the endless variant, second launch: raster AND queue service in one.  The first EMP_SVC_WGS workgroups do what
// emp_serve_kernel does -- one queue entry per wave at a time -- and then draw the frames of the instances they served
// themselves (from the descriptor their wave just produced, kept in LDS); all other workgroups walk the frames of the
// instances that were NOT queued (MysteryDesc::valid == 1; emp_step_kernel marks the queued ones).  The 100 us of dependent path generation that a
// reset costs no longer stand in front of the raster: they run next to it, on a quarter of the resident workgroups.
// Measured (32,768 instances, us per step incl. the 26 us of emp_step_kernel; separate launches: 240): 384 / 512 / 768 /
// 1,024 / 1,536 service workgroups at 7 workgroups per CU (72 VGPRs, the path generator spills) 218 / 217 / 216 / 224 / 234;
// at 5 per CU (96 VGPRs) 212 / 211 / 215 / 224 / 227; at 4 per CU 210 / 212 / 213 / 219 / 223.
#ifndef MG_LAB_EMP_SVC  // measurement builds: -DMG_LAB_EMP_SVC=<workgroups> -DMG_LAB_EMP_LB=<workgroups per CU>
#define MG_LAB_EMP_SVC 256  // round 4, with non-temporal frame stores (round 3: 384, with lazy initial segments: profiles/r03_emp.md)
#endif
#ifndef MG_LAB_EMP_LB
#define MG_LAB_EMP_LB 6  // (round 5; rounds 3-4: 5)
#endif
#ifndef MG_LAB_EMP_SVC_SMALL
#define MG_LAB_EMP_SVC_SMALL 768
#endif
constexpr int EMP_SVC_WGS = MG_LAB_EMP_SVC, EMP_SVC_WGS_SMALL = MG_LAB_EMP_SVC_SMALL;
// Frame stores of the fused launch: NON-TEMPORAL (round 4).  Alone, a plain store stream is the faster one for these frames (110 us
// against 131 us for 32,768 of them, and every other launch of the mortar / mystery families keeps plain stores: -15 to -20 % with
// nt); beside the path service the plain stream takes 149-152 us and the non-temporal one still 129-136 us -- it does not push the
// service waves' working set (segment stores, queue, the generator's spills) out of the L2.  183-190 -> 203-209 M env-steps/s at
// 32,768 instances; buffer-addressed stores, 4 / 6 workgroups per CU, 256 / 512 / 768 service workgroups: all within 2 % of it
// (profiles/r04_emp.md).  Lab switch MEMGYM_EMP_NT=0 / 1 forces plain / non-temporal stores.
// Round 5: with the next episode's first segment generated ahead of time (EMP_PRE) the service queue is all but empty (a due segment
// now and then) and the PLAIN stream is the faster one again: same box, 32,768 instances, fused launch 123.5-123.9 us plain against
// 142-161 us non-temporal (without EMP_PRE: 159 plain, 128 non-temporal); 64 service workgroups instead of 256: 121.7 us.  The
// kernel therefore exists in both forms and the host picks (profiles/r05_emp.md).
#ifndef MG_LAB_EMP_SVC_PRE
#define MG_LAB_EMP_SVC_PRE 64
#endif
constexpr int EMP_SVC_WGS_PRE = MG_LAB_EMP_SVC_PRE;
#ifdef MG_LAB_EMP_CLOCK  // measurement builds only: per-workgroup start / end of service / end, constant-rate clock (10 ns)
static __device__ unsigned long long g_lab_emp_clock[3 * 16384];
#define LAB_CLOCK(slot) do { if (threadIdx.x == 0 && blockIdx.x < 16384) g_lab_emp_clock[3 * blockIdx.x + (slot)] = wall_clock64(); } while (0)
#else
#define LAB_CLOCK(slot) do { } while (0)
#endif
constexpr int EMP_BG_WGS = 512;  // at most so many workgroups behind the service workgroups take background jobs (EMP_BG_SPAN instances' flags each)
#ifndef MG_LAB_EMP_BG_SPAN
#define MG_LAB_EMP_BG_SPAN 256
#endif
constexpr int EMP_BG_SPAN = MG_LAB_EMP_BG_SPAN;
static_assert(EMP_BG_SPAN <= 256 || EMP_BG_SPAN % 256 == 0, "a background workgroup reads its span's flags 256 at a time");
static_assert(LW_BYTES <= v1::RASTER_LDS, "the lane generator's workspace must fit into the raster workgroup's LDS");
// (Round 4 tried the service and background workgroups as a launch of their own on a side stream beside a plain raster launch:
// bit-exact, 185 M env-steps/s against 189-192 M for this fused launch -- the raster alone takes 110 us, beside the service
// 136-144 us, and the fork / join costs ~10 us of stream time: profiles/r04_emp.md.  Taken out again.  So was the arguments-as-one-
// struct form that helped the spotlight family's fused kernel (service loop reading them through an opaque pointer where it uses
// them): scratch 672 -> 624 B only -- the path generator wants ~200 VGPRs whatever the scalar side does -- and the launch got
// SLOWER, 149-151 -> 156-164 us.)
template <int FMT, bool EMP_NT, bool FINAL = false>
__global__ __launch_bounds__(256, MG_LAB_EMP_LB) void emp_raster_serve_kernel(const MysteryDesc* __restrict__ descs, RasterAtlas A, void* __restrict__ obs, int n,
                                                                  MysteryParams P, MysteryIO io, float* reward_out, uint8_t* done_out,
                                                                  float* gt, mg_info_buffers info, int autoreset, int svc, int bgw, int turn) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    __shared__ MysteryDesc sdesc[4];
    __shared__ int served[4];
    RasterCtx R;
    R.frame = smem;
    R.mask = reinterpret_cast<uint32_t*>(smem + FRAME_BYTES);
    R.A = A;
    R.T = A.tables;
    R.tid = threadIdx.x;
    const int tid = threadIdx.x;
    LAB_CLOCK(0);
    if ((int)blockIdx.x < svc) {
        // the service waves run a long dependent instruction chain next to memory-bound raster waves: let them issue first
        if (P.svc_prio) __builtin_amdgcn_s_setprio(3);
        uint8_t* ws = smem + FRAME_BYTES;  // the path workspace lives in the (unused) mask words behind the frame
        path_ws_init(ws);
        const PathWS W{ws, io.jump, io.stats};
        const int wv = tid >> 6;
        const bool me = (tid & 63) == 0;
        const int count = queue_count(&io.qctr[QC_COUNT], n);
        // bg_coop (launches of up to ~20,000 instances): the owed segments are entries count .. count + bg - 1 of the same queue --
        // with the lane-per-path generator of the frame workgroups (below) such a launch lasts as long as that generator's one
        // path, ~105 us, whatever its frames take
        const int bg = P.bg_coop ? queue_count(&io.qctr[QC_BG_COUNT], n) : 0;
        const int waves = svc * 4;
        int idx = bcast((int)(blockIdx.x * 4 + wv), 0);
        for (;;) {
            int inst = -1;
            if (idx < count + bg) {
                const int entry = idx < count ? bcast(io.queue[idx], 0) : (bcast(io.bgq[idx - count], 0) | EMP_Q_OWED);
#ifndef MG_LAB_EMP_NOSVC  // (measurement builds: what the launch costs without the cooperative generator's registers; entries are dropped)
                emp_serve_entry<FINAL>(P, io, W, entry, nullptr, reward_out, done_out, gt, info, autoreset, &sdesc[wv]);
#endif
                if (!(entry & EMP_Q_OWED)) inst = entry & EMP_Q_INST;
                if (me) idx = waves + atomicAdd(&io.qctr[QC_HEAD], 1);
                idx = bcast(idx, 0);
            }
            if (me) served[wv] = inst;
            __syncthreads();
            LAB_CLOCK(1);
            bool any = false;
            for (int w = 0; w < 4; ++w) {
                const int e = served[w];
                if (e < 0) continue;
                any = true;
                MysteryComposer::compose(&sdesc[w], R);
                __syncthreads();
                store_frame<FMT, EMP_NT>(smem, obs, e, tid);
                __syncthreads();
            }
            // (a round in which every wave served a background job draws nothing and goes on)
            const bool more = __syncthreads_or(idx < count + bg);
            if (!any && !more) break;
            __syncthreads();  // served[] / sdesc[] are rewritten by the next round: every wave has finished reading them
        }
        if (tid == 0 && atomicAdd(&io.qctr[QC_LEFT], 1) == svc - 1) {  // last service workgroup out
            io.qctr[QC_COUNT] = 0;
            io.qctr[QC_HEAD] = 0;
            io.qctr[QC_LEFT] = 0;
            if (P.bg_coop) io.qctr[QC_BG_COUNT] = 0;
        }
        LAB_CLOCK(2);
        return;
    }
    // Background jobs (owed segments, lazy initial segments; records ahead of time, EMP_PRE): the `bgw` workgroups behind the service
    // workgroups.  Workgroup b looks at the flags of instances b * EMP_BG_SPAN .. (emp_step_kernel wrote them), compacts the flagged
    // ones (~30 of 256 under random actions) into a list in LDS and its wave 0 takes up to 64 of them, one per lane, with the
    // lane-per-path generator in the workgroup's frame buffer; ~105 us that run beside the other workgroups' frames, and nothing of
    // this launch depends on them.  What does not fit a wave waits for the instance's next step (its flag is set again; the
    // list is entered at a position that moves with the launches, so no instance waits for ever): a workgroup that generates paths
    // holds a frame workgroup's slot for the whole launch and costs the store stream in proportion -- the same launch 124 us with
    // 128 such workgroups, 111 us with the jobs moved out of it (profiles/r05_emp.md) -- so they are few and full.  These workgroups
    // draw no frames (rounds 3-4: frame workgroups carried the jobs and went on to the frames of their stride afterwards).
    const int fb = svc + bgw;  // first frame workgroup
    if ((int)blockIdx.x < fb) {
        // (the list lives behind the lane generator's workspace in the frame buffer: 1 KB more of static LDS and the seventh workgroup
        // no longer fits a CU)
        int* const bg_jobs = reinterpret_cast<int*>(smem + LW_BYTES);
        int* const bg_cnt = bg_jobs + EMP_BG_SPAN;
        static_assert(LW_BYTES + EMP_BG_SPAN * 4 + 16 <= v1::RASTER_LDS && LW_BYTES % 16 == 0, "the job list must fit behind the lane generator's workspace");
        const int b = (int)blockIdx.x - svc;
#ifdef MG_LAB_EMP_NOBG  // (measurement builds: what the launch costs without the background jobs; owed segments are never generated here)
        if (b >= 0) return;
#endif
        bool ws = false;
        for (int base = b * EMP_BG_SPAN; base < n; base += bgw * EMP_BG_SPAN) {
            int total = 0;
            for (int c = 0; c < EMP_BG_SPAN; c += 256) {  // the span's flags, 256 at a time
                const int inst = base + c + tid;
                const bool want = c + tid < EMP_BG_SPAN && inst < n && io.bgflag[inst] != 0;
                const uint64_t m = __ballot(want);
                if ((tid & 63) == 0) bg_cnt[tid >> 6] = __popcll(m);
                __syncthreads();
                int off = total;
                for (int w = 0; w < (tid >> 6); ++w) off += bg_cnt[w];
                total += bg_cnt[0] + bg_cnt[1] + bg_cnt[2] + bg_cnt[3];
                if (want) bg_jobs[off + __popcll(m & ((1ull << (tid & 63)) - 1ull))] = inst;
                __syncthreads();  // (the counts are rewritten by the next chunk; the list is complete behind the last one)
            }
            if (total && !ws) {
                lane_ws_init(smem);
                ws = true;
            }
            if (tid < 64) {
                if (P.svc_prio) __builtin_amdgcn_s_setprio(3);  // a long dependent chain next to memory-bound raster waves
                const LaneWS LW{smem, tid};
                const int rot = total > 64 ? (int)((unsigned)turn * 61u % (unsigned)total) : 0;
                if (tid < total) lane_owed_segment(io, LW, bg_jobs[(tid + rot) % total], 1, P.pre != 0);
            }
            __syncthreads();  // (the list is rewritten by the next round)
        }
        LAB_CLOCK(1);
        LAB_CLOCK(2);
        return;
    }
    const int stride = (int)gridDim.x - fb;
    for (int env = (int)blockIdx.x - fb; env < n; env += stride) {
        const MysteryDesc* d = descs + env;
        if (d->valid != 1) continue;  // masked, or drawn by the workgroup that serves its queue entry
        MysteryComposer::compose(d, R);
        __syncthreads();
        store_frame<FMT, EMP_NT>(smem, obs, env, tid);
        __syncthreads();
    }
    LAB_CLOCK(2);
}

// ---------------------------------------------------------------------------------------------------------
static const double SCALE = 0.25;

class MysteryFamily : public Family {
   public:
    MysteryFamily(int variant, int n) : n_(n) {  // 0 MysteryPath-v0, 1 Endless-MysteryPath-v0, 2 MysteryPath-Grid-v0
        const int endless = variant == 1;
        memset(&P_, 0, sizeof(P_));
        P_.endless = endless;
        P_.grid = variant == 2;
        P_.n = n;
        agent_scale_ = 1.0 * SCALE;
        agent_speed_ = 12.0 * SCALE;
        P_.visual_feedback = 1;
        // Endless-MysteryPath: 187-189 -> 175-182 us per fused launch (profiles/r03_emp.md); no effect on MysteryPath-Grid's
        P_.svc_prio = [endless] { const char* e = lab_env("MEMGYM_SVC_PRIO"); return e ? atoi(e) : (endless ? 1 : 0); }();
        P_.path_help = [] { const char* e = lab_env("MEMGYM_PATH_HELP"); return e ? atoi(e) : 1; }();
        P_.bg_coop = lab_int("MEMGYM_EMP_BG_COOP", n <= 20480 ? 1 : 0);
        P_.seg_cap = std::min(MAX_SEG, std::max(4, lab_int("MEMGYM_EMP_SEG_CAP", MAX_SEG)));    // (lab build: tests reach the capacities in a few
        P_.fall_cap = std::min(MAX_FALL, std::max(1, lab_int("MEMGYM_EMP_FALL_CAP", MAX_FALL)));  // hundred steps, tests/test_gpu_capacity.py)
        lazy_wanted_ = endless && [] { const char* e = lab_env("MEMGYM_EMP_LAZY"); return e ? atoi(e) != 0 : true; }();
        // the next episode's first segment ahead of time (EMP_PRE): with the lane-per-path background jobs of the larger launches
        // (as entries of the service queue -- bg_coop -- a record ahead of time costs the path it saves)
        pre_wanted_ = endless && !P_.bg_coop && lab_int("MEMGYM_EMP_PRE", 1) != 0;
        P_.r_fall = 0.0; P_.r_progress = 0.1; P_.r_step = 0.0;
        if (endless) {
            P_.max_steps = -1; P_.show_past_path = 1; camera_offset_scale_ = 5.0; P_.stamina_level = 20;
        } else {
            P_.max_steps = P_.grid ? 128 : 512;
            if (P_.grid) P_.r_progress = 0.0;
            st_cardinal_.set(P_.cardinal, {0, 1, 2, 3});
            P_.r_goal = 1.0;
        }
        core_.alloc(n);
        walls_.alloc(n);
        desc_.alloc(n);
        tdesc_.alloc(n);  // (terminal-frame descriptors of the FINAL kernels: 64 B per instance; allocated here so that no step allocates)
        rng_.alloc(n);
        err_.alloc();
        queue_.alloc((size_t)n + 32 + QC_WORDS);
        bgq_.alloc(endless ? (size_t)n : 1);
        bgflag_.alloc(endless ? (size_t)n : 1);
        {   // WaveRng: s_k = A^k s_0 + S_k inc for k = 1 .. 64 (PCG64's 128-bit LCG, multiplier as in mg_device.hpp Pcg::advance)
            const u128 A = (((u128)0x2360ED051FC65DA4ull) << 64) | (u128)0x4385DF649FCCF645ull;
            std::vector<uint4> jt(128);
            u128 m = 1, q = 0;
            for (int k = 0; k < 64; ++k) {
                q = q * A + 1;  // S_(k+1) = S_k A + 1
                m = m * A;      // A^(k+1)
                jt[2 * k] = make_uint4((uint32_t)m, (uint32_t)(m >> 32), (uint32_t)(m >> 64), (uint32_t)(m >> 96));
                jt[2 * k + 1] = make_uint4((uint32_t)q, (uint32_t)(q >> 32), (uint32_t)(q >> 64), (uint32_t)(q >> 96));
            }
            jump_.upload(jt);
            stats_.alloc(8);
        }
        if (endless) {
            seg_rows_ = MAX_SEG;
            segs_.alloc((size_t)n * seg_rows_ * SEG_STRIDE);
            aux_.alloc((size_t)n * AUX_WORDS);
        } else {
            segs_.alloc(16);
            aux_.alloc(4);
        }
        sets_dev_.alloc(MG_MAX_OPTION_SETS);
        hipLaunchKernelGGL(mystery_init_kernel, dim3((n + 255) / 256), dim3(256), 0, 0, n, core_.p);
        MG_HIP(hipDeviceSynchronize());
        rebuild();
        defaults_ = P_;  // (the cardinal list is short: no device array behind it)
    }

    // include/memgym.h: mg_set_capacity.  "path_segments" (Endless-MysteryPath-v0): records of the segment store per instance -- the
    // reference's path grows without limit (pygame_assets.py:559); an episode that needs one more segment than this ends (capacity_dev)
    void set_capacity(const std::string& what, int64_t v) override {
        if (!(P_.endless && what == "path_segments")) return Family::set_capacity(what, v);
        if (v < 4 || v > 32767) throw OptionError{-3, "path_segments: 4 .. 32,767"};
        if (seeded_) throw std::runtime_error("mg_set_capacity: before the first reset");
        MG_HIP(hipDeviceSynchronize());
        seg_rows_ = (int)v;
        segs_.alloc((size_t)n_ * seg_rows_ * SEG_STRIDE);
        P_.seg_cap = seg_rows_;
        for (auto& e : extra_) e->P.seg_cap = seg_rows_;
        defaults_.seg_cap = seg_rows_;
        sets_dirty_ = true;
    }
    int64_t capacity(const std::string& what) const override {
        if (P_.endless && what == "path_segments") return P_.seg_cap;
        if (P_.endless && what == "fall_off_cells") return P_.fall_cap;
        return Family::capacity(what);
    }
    int action_dim() const override { return (P_.endless || P_.grid) ? 1 : 2; }
    int gt_dim() const override { return P_.endless ? 3 : 0; }
    const char* info_name(int k) const override {
        if (P_.endless) return k == 0 ? "num_fails" : (k == 1 ? "max_x" : (k == 2 ? "tiles_visited" : nullptr));
        return k == 0 ? "success" : (k == 1 ? "num_fails" : nullptr);
    }

    // One key of the reset options, for option set `set` (0 = the handle-wide set of mg_set_option).  Sets > 0 hold everything that
    // does not change the geometry (sprites, camera offset and speeds are shared by the handle's instances).
    void set_option(const std::string& key, const double* v, int n) override { set_option_set(0, key, v, n); }
    void set_option_set(int set, const std::string& key, const double* v, int n) override {
        if (set < 0 || set >= MG_MAX_OPTION_SETS) throw OptionError{-3, "option set index out of range"};
        while ((int)extra_.size() < set) {  // a new set starts from the constructor's defaults (= the reference's), geometry from set 0
            extra_.emplace_back(new MysteryOpt());
            extra_.back()->P = defaults_;
            copy_geometry(extra_.back()->P, P_);
        }
        MysteryParams& P = set == 0 ? P_ : extra_[set - 1]->P;
        OptListStore& st_cardinal = set == 0 ? st_cardinal_ : extra_[set - 1]->st_cardinal;
        const bool e = P_.endless;
        auto I = [&](int& dst) { dst = to_int_checked(v[0], key.c_str()); };
        auto B = [&](int& dst) { dst = v[0] != 0.0; };
        auto must_be = [&](bool ok) { if (!ok) throw OptionError{-3, "reset parameter " + key + ": this value is not supported by the MI355X build"}; };
        // (a geometry option in a set > 0 is accepted when it says what the handle's geometry already is)
        auto geometry = [&](double& mine) {
            if (set != 0) {
                if (v[0] != mine)
                    throw OptionError{-3, "reset parameter " + key + " changes the geometry shared by the handle's instances: it can only be set for all of them (option set 0)"};
            } else {
                mine = v[0];
                dirty_ = true;
            }
        };
        if (key == "max_steps") I(P.max_steps);
        else if (key == "agent_scale") geometry(agent_scale_);
        else if (!P_.grid && key == "agent_speed") geometry(agent_speed_);
        else if (key == "show_origin") { B(P.show_origin); if (e) P.show_origin = 0; /* dead branch in the reference (:150) */ }
        else if (key == "visual_feedback") B(P.visual_feedback);
        else if (key == "reward_fall_off") P.r_fall = v[0];
        else if (key == "reward_path_progress") P.r_progress = v[0];
        else if (key == "reward_step") P.r_step = v[0];
        else if (e && key == "show_past_path") B(P.show_past_path);
        else if (e && key == "show_background") B(P.show_background);
        else if (e && key == "show_stamina") B(P.show_stamina);
        else if (e && key == "camera_offset_scale") geometry(camera_offset_scale_);
        else if (e && key == "stamina_level") { I(P.stamina_level); must_be(P.stamina_level > 0); }
        else if (e && key == "reward_path_progress_dense") P.r_dense = v[0];
        else if (!e && key == "cardinal_origin_choice") {
            must_be(n >= 1);  // any length; every value other than 0, 1, 2 takes the reference's `else` branch (mystery_path.py:155-166)
            std::vector<int> vals(n);
            for (int k = 0; k < n; ++k) {
                const int c = to_int_checked(v[k], key.c_str());
                vals[k] = (c >= 0 && c <= 2) ? c : 3;
            }
            st_cardinal.set(P.cardinal, vals);
        }
        else if (!e && key == "show_goal") B(P.show_goal);
        else if (!e && key == "reward_goal") P.r_goal = v[0];
        else throw OptionError{-2, "unknown reset parameter " + key};
        sets_dirty_ = true;
    }
    // instance i runs under option set set_of_dev[i] (device array [num_envs], caller-owned; NULL: every instance under set 0)
    void bind_option_sets(const int32_t* set_of_dev) override { set_of_ = set_of_dev; }

    void reset(const int64_t* seeds, const uint8_t* mask, void* obs, float* gt, hipStream_t s) override {
        if (dirty_) rebuild();
        if (!seeds && !seeded_) throw std::runtime_error("reset(seed=None) before any seeded reset");
        if (seeds) seeded_ = true;
        const bool ps = per_set();
        if (P_.endless) {
            mg_info_buffers none;
            memset(&none, 0, sizeof(none));
            // a masked reset(seed=None) of a handle whose steps run the fused arrangement: reset like the auto-reset step resets (lazy
            // segments, records ahead of time: emp_masked_reset_kernel); lab MEMGYM_EMP_MASKED_FAST=0: through the queue server like any other
            static const bool fast_wanted = lab_int("MEMGYM_EMP_MASKED_FAST", 1) != 0;
            const bool fast = fast_wanted && mask && !seeds && !ps && lazy_wanted_ && fuse_serve() && obs_format == MG_OBS_U8_XYC && !big_sprites_;
            P_.lazy = fast ? 1 : 0;  // (otherwise an explicit reset generates all three segments; whatever an old episode is owed comes first)
            P_.pre = (fast && pre_wanted_) ? 1 : 0;
            P_.lazy_append = 0;
            if (fast) owed_possible_ = true;
            upload_sets(s);
            if (fast) {
                hipLaunchKernelGGL(emp_masked_reset_kernel, dim3((n_ + 255) / 256), dim3(256), 0, s, P_, io(), mask, gt);
                hipLaunchKernelGGL(emp_serve_kernel<false>, dim3(servers(false)), dim3(256), WS_BYTES, s, P_, io(), seeds, 0, (float*)nullptr,
                                   (uint8_t*)nullptr, gt, none, 0);
            } else if (mask) {
                hipLaunchKernelGGL(emp_enqueue_kernel, dim3((n_ + 255) / 256), dim3(256), 0, s, n_, io(), mask);
                if (ps)
                    hipLaunchKernelGGL(emp_serve_kernel<true>, dim3(servers(false)), dim3(256), WS_BYTES, s, P_, io(), seeds, 0, (float*)nullptr,
                                       (uint8_t*)nullptr, gt, none, 0);
                else
                    hipLaunchKernelGGL(emp_serve_kernel<false>, dim3(servers(false)), dim3(256), WS_BYTES, s, P_, io(), seeds, 0, (float*)nullptr,
                                       (uint8_t*)nullptr, gt, none, 0);
            } else if (n_ >= 1024 && reset_by_lanes() && !ps) {  // many paths at once: one lane per instance
                hipLaunchKernelGGL(emp_reset_lanes_kernel, dim3((n_ + 63) / 64), dim3(64), LW_BYTES, s, P_, io(), seeds, gt);
            } else if (ps) {
                hipLaunchKernelGGL(emp_serve_kernel<true>, dim3(servers(true)), dim3(256), WS_BYTES, s, P_, io(), seeds, 1, (float*)nullptr,
                                   (uint8_t*)nullptr, gt, none, 0);
            } else {
                hipLaunchKernelGGL(emp_serve_kernel<false>, dim3(servers(true)), dim3(256), WS_BYTES, s, P_, io(), seeds, 1, (float*)nullptr,
                                   (uint8_t*)nullptr, gt, none, 0);
            }
        } else {
            upload_sets(s);
            if (ps) hipLaunchKernelGGL(mystery_reset_kernel<true>, dim3(blocks()), dim3(256), WS_BYTES, s, P_, io(), seeds, mask, nullptr, lpw());
            else hipLaunchKernelGGL(mystery_reset_kernel<false>, dim3(blocks()), dim3(256), WS_BYTES, s, P_, io(), seeds, mask, nullptr, lpw());
        }
        if (mask && sparse_masked_raster()) {  // few frames of many: by the mask, not by a walk over every descriptor (mg_raster_v1.hpp)
            if (big_sprites_) launch_raster_sparse<MysteryBigComposer>(desc_.p, atlas_->dev(), obs, obs_format, n_, s, mask);
            else launch_raster_sparse<MysteryComposer>(desc_.p, atlas_->dev(), obs, obs_format, n_, s, mask);
            MG_HIP(hipGetLastError());
        } else raster(obs, s);
    }

    void step(const int32_t* actions, void* obs, float* reward, uint8_t* done, float* gt, const mg_info_buffers* info,
              int autoreset, hipStream_t s) override {
        if (dirty_) throw std::runtime_error("options that change geometry need a reset before the next step");
        mg_info_buffers ib;
        memset(&ib, 0, sizeof(ib));
        if (info) ib = *info;
        prof.begin(0, s);
        const bool ps = per_set();
        if (P_.endless) {
            // lazy initial segments need the fused launch (its frame workgroups carry the background jobs); any other path
            // first generates what earlier fused steps left owed
            // (per-instance option sets: the plain arrangement -- step kernel, queue server, raster -- whose kernels have a <PS> form)
            const bool fused = fuse_serve() && obs_format == MG_OBS_U8_XYC && !ps && !big_sprites_;
            P_.lazy = (fused && lazy_wanted_) ? 1 : 0;
            P_.pre = (P_.lazy && pre_wanted_) ? 1 : 0;
            static const int lazy_append = lab_int("MEMGYM_EMP_LAZY_APPEND", 1);
            P_.lazy_append = (P_.lazy && lazy_append) ? 1 : 0;
            if (!P_.lazy && owed_possible_) flush_owed(s);
            if (P_.lazy) owed_possible_ = true;
            upload_sets(s);
            const int sb = step_block(256);
            // terminal observations (mg_step, mg_info_buffers.final_obs_dev): the FINAL forms of the two launches leave the terminal frame
            // descriptors in tdesc_, a sparse raster launch behind them draws those frames (round 6; keeps_final_obs)
            const bool keep_final = autoreset && ib.final_obs_dev && fused && keeps_final_obs(s);
            if (ps) hipLaunchKernelGGL(emp_step_kernel<true>, dim3((n_ + sb - 1) / sb), dim3(sb), 0, s, P_, io(), actions, reward, done, gt, ib, autoreset);
            else if (keep_final) hipLaunchKernelGGL((emp_step_kernel<false, true>), dim3((n_ + sb - 1) / sb), dim3(sb), 0, s, P_, io(), actions, reward, done, gt, ib, autoreset);
            else hipLaunchKernelGGL(emp_step_kernel<false>, dim3((n_ + sb - 1) / sb), dim3(sb), 0, s, P_, io(), actions, reward, done, gt, ib, autoreset);
            if (fused) {  // the queue is served inside the raster launch
                end_logic(s);
                prof.begin(1, s);
                // (small launches: the owed segments are entries too; with records ahead of time: next to no entries)
                const int svc = P_.bg_coop ? EMP_SVC_WGS_SMALL : (P_.pre ? EMP_SVC_WGS_PRE : EMP_SVC_WGS);
                static const int nt_forced = lab_int("MEMGYM_EMP_NT", -1);
                const bool nt = nt_forced >= 0 ? nt_forced != 0 : !P_.pre;
                int bgw = P_.bg_coop ? 0 : std::min(EMP_BG_WGS, (n_ + EMP_BG_SPAN - 1) / EMP_BG_SPAN);   // background workgroups: EMP_BG_SPAN instances' flags each
                ++turn_;
                // lab: MEMGYM_EMP_BG_SEPARATE=1 runs the background jobs as a launch of their own BEHIND the raster (what they cost it)
                static const int bg_separate = lab_int("MEMGYM_EMP_BG_SEPARATE", 0);
                const int bgw_later = bg_separate ? bgw : 0;
                if (bg_separate) bgw = 0;
                const int grid = (n_ < raster_grid(n_) ? n_ : raster_grid(n_)) + svc + bgw;  // (service, background, frames)
                if (keep_final && nt)
                    hipLaunchKernelGGL((emp_raster_serve_kernel<MG_OBS_U8_XYC, true, true>), dim3(grid), dim3(256), RASTER_LDS, s, desc_.p, atlas_->dev(), obs, n_,
                                       P_, io(), reward, done, gt, ib, autoreset, svc, bgw, turn_);
                else if (keep_final)
                    hipLaunchKernelGGL((emp_raster_serve_kernel<MG_OBS_U8_XYC, false, true>), dim3(grid), dim3(256), RASTER_LDS, s, desc_.p, atlas_->dev(), obs, n_,
                                       P_, io(), reward, done, gt, ib, autoreset, svc, bgw, turn_);
                else if (nt)
                    hipLaunchKernelGGL((emp_raster_serve_kernel<MG_OBS_U8_XYC, true>), dim3(grid), dim3(256), RASTER_LDS, s, desc_.p, atlas_->dev(), obs, n_,
                                       P_, io(), reward, done, gt, ib, autoreset, svc, bgw, turn_);
                else
                    hipLaunchKernelGGL((emp_raster_serve_kernel<MG_OBS_U8_XYC, false>), dim3(grid), dim3(256), RASTER_LDS, s, desc_.p, atlas_->dev(), obs, n_,
                                       P_, io(), reward, done, gt, ib, autoreset, svc, bgw, turn_);
                MG_HIP(hipGetLastError());
                if (keep_final)  // every finished instance's flag is in `done` by now (the service workgroups wrote the last of them)
                    launch_raster_sparse<MysteryComposer>(tdesc_.p, atlas_->dev(), ib.final_obs_dev, MG_OBS_U8_XYC, n_, s, done);
                prof.end(1, s);
                if (bgw_later)  // (grid = service + background workgroups only: no frames; the queue is empty by now)
                    hipLaunchKernelGGL((emp_raster_serve_kernel<MG_OBS_U8_XYC, false>), dim3(svc + bgw_later), dim3(256), RASTER_LDS, s, desc_.p, atlas_->dev(), obs, n_,
                                       P_, io(), reward, done, gt, ib, autoreset, svc, bgw_later, turn_);
#ifdef MG_LAB_EMP_CLOCK  // diagnosis: is the next logic kernel slow because the L2 is full of dirty observation lines?
                static const int wb = [] { const char* e = lab_env("MEMGYM_LAB_WBL2"); return e ? atoi(e) : 0; }();
                if (wb) hipLaunchKernelGGL(lab_wbl2_kernel, dim3(wb), dim3(64), 0, s);
#endif
                return;
            }
            if (ps)
                hipLaunchKernelGGL(emp_serve_kernel<true>, dim3(servers(false)), dim3(256), WS_BYTES, s, P_, io(), (const int64_t*)nullptr, 0,
                                   reward, done, gt, ib, autoreset);
            else
                hipLaunchKernelGGL(emp_serve_kernel<false>, dim3(servers(false)), dim3(256), WS_BYTES, s, P_, io(), (const int64_t*)nullptr, 0,
                                   reward, done, gt, ib, autoreset);
        } else {
            // (per-instance option sets: only this kernel has a <PS> form -- the raster launch's path service reads nothing of the options)
            const int defer = (autoreset && !big_sprites_) ? defer_mode() : 0;
            upload_sets(s);
            if (autoreset && ib.final_obs_dev && keeps_final_obs(s)) {  // terminal observations kept by these two launches (defer != 0, one option set)
                hipLaunchKernelGGL((mystery_step_kernel<false, true>), dim3(blocks()), dim3(256), WS_BYTES, s, P_, io(), actions, reward, done,
                                   (float*)nullptr, ib, autoreset, lpw(), defer);
                end_logic(s);
                prof.begin(1, s);
                raster_with_paths(obs, s, ib.final_obs_dev);
                prof.end(1, s);
                return;
            }
            if (ps)
                hipLaunchKernelGGL(mystery_step_kernel<true>, dim3(blocks()), dim3(256), WS_BYTES, s, P_, io(), actions, reward, done,
                                   (float*)nullptr, ib, autoreset, lpw(), defer);
            else
                hipLaunchKernelGGL(mystery_step_kernel<false>, dim3(blocks()), dim3(256), WS_BYTES, s, P_, io(), actions, reward, done,
                                   (float*)nullptr, ib, autoreset, lpw(), defer);
            if (defer) {  // the paths of this step's resets are generated by the first workgroups of the raster launch
                end_logic(s);
                prof.begin(1, s);
                raster_with_paths(obs, s);
                prof.end(1, s);
                return;
            }
        }
        end_logic(s);
        prof.begin(1, s);
        raster(obs, s);
        prof.end(1, s);
    }

    std::vector<std::pair<void*, size_t>> state_blobs() override {
        // (aux: fall-off lists, and the records generated ahead of time -- they belong to the state: the EMP_PRE flags travel in `core`)
        std::vector<std::pair<void*, size_t>> v = {{core_.p, core_.bytes()}, {segs_.p, segs_.bytes()}, {aux_.p, aux_.bytes()},
                                                  {walls_.p, walls_.bytes()}};
        rng_.blobs(v);
        return v;
    }
    void debug_rng(int i, uint64_t out[6]) override { rng_.debug(i, out); }
    void ground_truth64(double* out, hipStream_t s) override {
        if (!gt_dim() || !out) return;
        hipLaunchKernelGGL(mystery_gt64_kernel, dim3((n_ + 255) / 256), dim3(256), 0, s, n_, core_.p, out);
        MG_HIP(hipGetLastError());
    }
    int poll_errors() override {
        MG_HIP(hipDeviceSynchronize());
        return err_.take();
    }
    int peek_errors() override { return err_.peek(); }
    bool debug_counter(const std::string& name, int64_t* out) override {
        if (name == "emp_segments_sum" || name == "emp_segments_max" || name == "emp_falloff_max") {  // a scan of the state records as they stand
            std::vector<MysteryCore> h(n_);
            MG_HIP(hipDeviceSynchronize());
            MG_HIP(hipMemcpy(h.data(), core_.p, sizeof(MysteryCore) * (size_t)n_, hipMemcpyDeviceToHost));
            int64_t sum = 0, mx = 0, fmx = 0;
            for (const MysteryCore& c : h) {  // (segments generated so far; the ones still owed, EMP_OWED, are not counted)
                sum += c.num_seg;
                mx = c.num_seg > mx ? c.num_seg : mx;
                fmx = c.n_falloff > fmx ? c.n_falloff : fmx;
            }
            *out = name == "emp_segments_sum" ? sum : (name == "emp_segments_max" ? mx : fmx);
            return true;
        }
        const int k = name == "path_gen_ticks" ? 0 : (name == "path_gen_paths" ? 1 : (name == "emp_own_resets" ? 2 : (name == "emp_ahead_records" ? 3 : (name == "emp_final_served" ? 4 : -1))));
        if (k < 0 || !stats_.p) return false;
        unsigned long long v = 0;
        MG_HIP(hipMemcpy(&v, stats_.p + k, sizeof v, hipMemcpyDeviceToHost));
        *out = (int64_t)v;
        return true;
    }

   private:
    // instance-carrying lanes per wave (see instance_of_lane); MEMGYM_MYSTERY_LPW overrides for tuning
    int lpw() const {
        static const int forced = [] {
            const char* e = lab_env("MEMGYM_MYSTERY_LPW");
            return e ? atoi(e) : 0;
        }();
        if (forced == 4 || forced == 8 || forced == 16 || forced == 32 || forced == 64) return forced;
        return 16;  // measured: profiles/r01e_logic_tails.md (the endless variant has its own kernels)
    }
    // workgroups (4 waves each) of emp_serve_kernel; MEMGYM_EMP_SERVERS overrides for tuning
    int servers(bool all) const {
        static const int forced = [] {
            const char* e = lab_env("MEMGYM_EMP_SERVERS");
            return e ? atoi(e) : 0;
        }();
        const int want = forced > 0 ? forced : (all ? 1024 : 512);  // measured: profiles/r01e_logic_tails.md section 4
        const int cap = (n_ + 3) / 4;
        return want < cap ? want : cap;
    }
    int blocks() const { const int per_block = 4 * lpw(); return (n_ + per_block - 1) / per_block; }
    // finite variants: generate the paths of auto-resets on a side stream under the raster (MEMGYM_MYSTERY_DEFER=0: in the step kernel)
    // Measured (MysteryPath-Grid, 32,768 instances, 0.5 % of them reset per step): logic 32.7 -> 12.1 us, 231 -> 252 M
    // env-steps/s; MysteryPath-v0 with its default 512-step episodes resets too rarely to pay for the 128 service workgroups
    // (280 -> 275 M), so only the grid variant defers by default.  MEMGYM_MYSTERY_DEFER=0 / 1 forces it off / on.
    // MEMGYM_EMP_FUSE=0: separate queue-server launch in front of the raster (the round-1 arrangement)
    bool fuse_serve() const {
        static const bool on = [] {
            const char* e = lab_env("MEMGYM_EMP_FUSE");
            return !(e && atoi(e) == 0);
        }();
        return on;
    }
    // MEMGYM_EMP_RESET_LANES=0: a full reset through the queue server, one wave per instance (round 1)
    bool reset_by_lanes() const {
        static const bool on = [] {
            const char* e = lab_env("MEMGYM_EMP_RESET_LANES");
            return !(e && atoi(e) == 0);
        }();
        return on;
    }
    // 0: paths of auto-resets in the step kernel; 1: all of them queued for the raster launch (MysteryPath-Grid: 0.5 % of the
    // instances reset per step, logic 32.7 -> 12.1 us); 2: hybrid, a wave queues its requests only when it has more than
    // HYBRID_INLINE of them (MysteryPath-v0: its 512-step episodes reset too rarely to pay for always queueing, but the step
    // in which all survivors are truncated at once was a 450-us launch).  MEMGYM_MYSTERY_DEFER=0 / 1 / 2 forces a mode.
    int defer_mode() const {
        static const int forced = [] {
            const char* e = lab_env("MEMGYM_MYSTERY_DEFER");
            return e ? atoi(e) : -1;
        }();
        return forced >= 0 && forced <= 2 ? forced : (P_.grid != 0 ? 1 : 2);
    }
    void raster_with_paths(void* obs, hipStream_t s, void* final_obs = nullptr) {
        const int grid = (n_ < raster_grid(n_) ? n_ : raster_grid(n_)) + PATH_WGS;
        void* const none = nullptr;
        if (final_obs)  // (uint8 format: keeps_final_obs)
            hipLaunchKernelGGL((mystery_raster_paths_kernel<MG_OBS_U8_XYC, true>), dim3(grid), dim3(256), RASTER_LDS, s, desc_.p, atlas_->dev(), obs, n_, P_, io(), final_obs);
        else if (obs_format == MG_OBS_F32_CYX)
            hipLaunchKernelGGL((mystery_raster_paths_kernel<MG_OBS_F32_CYX>), dim3(grid), dim3(256), RASTER_LDS, s, desc_.p, atlas_->dev(), obs, n_, P_, io(), none);
        else if (obs_format == MG_OBS_BF16_CYX)
            hipLaunchKernelGGL((mystery_raster_paths_kernel<MG_OBS_BF16_CYX>), dim3(grid), dim3(256), RASTER_LDS, s, desc_.p, atlas_->dev(), obs, n_, P_, io(), none);
        else if (obs_format == MG_OBS_F16_CYX)
            hipLaunchKernelGGL((mystery_raster_paths_kernel<MG_OBS_F16_CYX>), dim3(grid), dim3(256), RASTER_LDS, s, desc_.p, atlas_->dev(), obs, n_, P_, io(), none);
        else
            hipLaunchKernelGGL((mystery_raster_paths_kernel<MG_OBS_U8_XYC>), dim3(grid), dim3(256), RASTER_LDS, s, desc_.p, atlas_->dev(), obs, n_, P_, io(), none);
        MG_HIP(hipGetLastError());
    }
    // (the finite variants' step + raster / path-service launches keep terminal observations themselves; lab MEMGYM_MYSTERY_FINAL_FUSED=0: the
    // generic path of mg_step.  Endless Mystery Path: its step kernel and the service waves of its fused launch leave the terminal frame
    // DESCRIPTORS behind, one sparse raster launch draws them -- lab MEMGYM_EMP_FINAL_FUSED=0: the generic path.)
    bool keeps_final_obs(hipStream_t) override {
        static const bool wanted = lab_int("MEMGYM_MYSTERY_FINAL_FUSED", 1) != 0, emp_wanted = lab_int("MEMGYM_EMP_FINAL_FUSED", 1) != 0;
        if (P_.endless) return emp_wanted && fuse_serve() && obs_format == MG_OBS_U8_XYC && !big_sprites_ && !per_set();
        return wanted && obs_format == MG_OBS_U8_XYC && !big_sprites_ && !per_set() && defer_mode() != 0;
    }
    MysteryIO io() {
        MysteryIO o;
        o.core = core_.p;
        o.segs = segs_.p;
        o.rng = rng_.view();
        o.seg_rows = seg_rows_;
        o.desc = desc_.p;
        o.err = err_.dev;
        o.queue = queue_.p;
        o.walls = P_.endless ? nullptr : walls_.p;
        o.qctr = queue_.p + ((n_ + 31) & ~31);
        o.bgq = bgq_.p;
        o.bgflag = bgflag_.p;
        o.aux = aux_.p;
        o.jump = jump_.p;
        o.stats = stats_.p;
        o.sets = per_set() ? sets_dev_.p : nullptr;
        o.set_of = per_set() ? set_of_ : nullptr;
        o.tdesc = tdesc_.p;
        return o;
    }

    // per-instance option sets
    struct MysteryOpt {
        MysteryParams P;
        OptListStore st_cardinal;
    };
    // what the shared atlas, the camera and the launch arrangement fix for every set of the handle
    static void copy_geometry(MysteryParams& d, const MysteryParams& s) {
        d.endless = s.endless; d.grid = s.grid; d.n = s.n; d.depth = s.depth; d.agent_radius = s.agent_radius; d.sprite_dim = s.sprite_dim;
        d.v_axis_i = s.v_axis_i; d.v_diag_i = s.v_diag_i; d.tile = s.tile; d.cross_dim = s.cross_dim; d.camera_offset = s.camera_offset;
        d.svc_prio = s.svc_prio; d.lazy = s.lazy; d.path_help = s.path_help; d.bg_coop = s.bg_coop; d.pre = s.pre; d.lazy_append = s.lazy_append;
        d.seg_cap = s.seg_cap; d.fall_cap = s.fall_cap;
    }
    bool per_set() const { return set_of_ != nullptr && !extra_.empty(); }
    // the sets as the kernels read them, stream-ordered behind what the stream holds (pageable source: staged before the call returns)
    void upload_sets(hipStream_t s) {
        if (!per_set() || !sets_dirty_) return;
        MysteryParams fresh = defaults_;  // a set that was never written: the reference's defaults under the handle's geometry (include/memgym.h)
        copy_geometry(fresh, P_);
        std::vector<MysteryParams> host(MG_MAX_OPTION_SETS, fresh);
        host[0] = P_;
        for (size_t k = 0; k < extra_.size(); ++k) {
            host[k + 1] = extra_[k]->P;
            copy_geometry(host[k + 1], P_);  // (incl. lazy = 0: the plain arrangement generates every segment when it is due)
        }
        MG_HIP(hipMemcpyAsync(sets_dev_.p, host.data(), sizeof(MysteryParams) * host.size(), hipMemcpyHostToDevice, s));
        MG_HIP(hipStreamSynchronize(s));  // (rare: only after an option of some set changed)
        sets_dirty_ = false;
    }

    void rebuild() {
        int radius = 0;
        // (MysteryPath-Grid-v0 accepts agent_scale and never reads it: GridCharacterController(SCALE, ...), mystery_path_grid.py:188)
        std::vector<Stamp> sprites = build_agent_sprites(P_.grid ? 1.0 * SCALE : agent_scale_, &radius);
        P_.agent_radius = radius;
        P_.sprite_dim = sprites[0].w;
        double inv = 1.0 / std::sqrt(2.0);
        P_.v_axis_i = (int)((1.0 / 1.0) * agent_speed_);
        P_.v_diag_i = (int)(inv * agent_speed_);
        P_.tile = SCREEN / G;
        double cos_ = camera_offset_scale_ < 0 ? 0 : (camera_offset_scale_ > 5.5 ? 5.5 : camera_offset_scale_);
        double cam = -P_.tile * cos_;
        if (cam != std::floor(cam)) throw OptionError{-3, "camera_offset_scale must give an integral pixel offset (multiples of 1/12)"};
        P_.camera_offset = (int)cam;
        P_.depth = (int)camera_offset_scale_;
        if (P_.depth > 7) throw OptionError{-3, "camera_offset_scale too large"};
        Stamp cross = build_cross(SCALE);
        P_.cross_dim = cross.w;
        atlas_.reset(new Atlas());
        // MysteryComposer holds a sprite of up to 1,024 pixels in StampRegs<4>; a larger one (agent_scale beyond 0.28) switches the
        // handle to MysteryBigComposer and to the plain launch arrangement (the fused launches compose with the register form)
        big_sprites_ = sprites[0].w * sprites[0].h > 1024;
        for (auto& sp : sprites) atlas_->add_stamp(sp);  // 0..7
        atlas_->add_stamp(cross, 256);                    // 8      (StampRegs<1>; build_cross(SCALE): no option scales it)
        if (P_.endless) {
            // show_background: draw_column_tile_surface / draw_icy_surface (pygame_assets.py:780-817) blitted every `tile`
            // pixels from x = bg_scroll - tile on (endless_mystery_path.py:141-143) = one template per scroll phase
            const uint8_t ice[3] = {125, 177, 250}, edge[3] = {210, 210, 210};
            std::vector<uint8_t> t((size_t)P_.tile * FRAME_BYTES);
            for (int ph = 0; ph < P_.tile; ++ph)
                for (int x = 0; x < SCREEN; ++x)
                    for (int y = 0; y < SCREEN; ++y) {
                        const int u = (x + ph) % P_.tile, w = y % P_.tile;
                        const bool on_edge = u == 0 || w == 0 || u == P_.tile - 1 || w == P_.tile - 1;
                        for (int c = 0; c < 3; ++c) t[(size_t)ph * FRAME_BYTES + ((size_t)x * SCREEN + y) * 3 + c] = on_edge ? edge[c] : ice[c];
                    }
            atlas_->set_templates(t);
        }
        atlas_->upload();
        dirty_ = false;
    }

    void raster_only(void* obs, const uint8_t* only, hipStream_t s) override {
        if (big_sprites_) launch_raster<MysteryBigComposer>(desc_.p, atlas_->dev(), obs, obs_format, n_, s, only);
        else launch_raster<MysteryComposer>(desc_.p, atlas_->dev(), obs, obs_format, n_, s, only);
        MG_HIP(hipGetLastError());
    }

    void raster(void* obs, hipStream_t s) { raster_only(obs, nullptr, s); }

    int n_;
    MysteryParams P_;       // option set 0 (the handle-wide set of mg_set_option)
    MysteryParams defaults_;
    std::vector<std::unique_ptr<MysteryOpt>> extra_;  // option sets 1 ..
    const int32_t* set_of_ = nullptr;
    bool sets_dirty_ = true;
    DevArray<MysteryParams> sets_dev_;
    double agent_scale_, agent_speed_, camera_offset_scale_ = 5.0;
    bool dirty_ = true, seeded_ = false;
    bool big_sprites_ = false;  // rebuild(): the agent sprites exceed MysteryComposer's registers

   public:
    void on_state_loaded() override {
        seeded_ = true;
        owed_possible_ = P_.endless != 0;  // the blob may carry owed segments
    }
    void sync_state() override {
        if (P_.endless && owed_possible_) {
            MG_HIP(hipDeviceSynchronize());  // steps in flight on the caller's streams come first
            flush_owed(0);
            MG_HIP(hipDeviceSynchronize());
        }
    }
    void flush_owed(hipStream_t s) {
        hipLaunchKernelGGL(emp_flush_owed_kernel, dim3((n_ + 63) / 64), dim3(64), LW_BYTES, s, P_, io());
        MG_HIP(hipGetLastError());
        owed_possible_ = false;
    }
    void raster_debug(void* frames, hipStream_t s) override;

   private:
    std::unique_ptr<Atlas> atlas_;
    DevArray<MysteryCore> core_;
    DevArray<uint8_t> segs_;
    int seg_rows_ = MAX_SEG;
    DevArray<MysteryDesc> desc_, tdesc_;  // tdesc_: terminal-frame descriptors of the FINAL kernels (finite variants)
    DevArray<int> queue_;  // n entries + the counters
    DevArray<int> bgq_;    // endless: background jobs (owed segments), small launches
    DevArray<uint8_t> bgflag_;  // ... larger launches: one flag per instance
    bool lazy_wanted_ = false, owed_possible_ = false, pre_wanted_ = false;
    int turn_ = 0;  // fused launches so far (where a background workgroup enters an over-long job list)
    DevArray<uint32_t> aux_;  // endless: per instance, the next episode's first segment + the stream behind it (EMP_PRE) and the fall-off list
    DevArray<uint4> jump_;  // WaveRng jump constants
    DevArray<unsigned long long> stats_;  // MysteryIO::stats
    DevArray<uint64_t> walls_;  // finite: wall cells of every instance's path generation (debug view)
    ErrorWord err_;
    RngStore rng_;
    OptListStore st_cardinal_;
};

void MysteryFamily::raster_debug(void* frames, hipStream_t s) {
    if (dirty_) throw std::runtime_error("options that change geometry need a reset before the next render");
    DevArray<MysteryDesc> dbg;
    dbg.alloc(n_, false);
    upload_sets(s);
    if (per_set()) hipLaunchKernelGGL(mystery_debug_desc_kernel<true>, dim3((n_ + 255) / 256), dim3(256), 0, s, P_, io(), dbg.p);
    else hipLaunchKernelGGL(mystery_debug_desc_kernel<false>, dim3((n_ + 255) / 256), dim3(256), 0, s, P_, io(), dbg.p);
    if (big_sprites_) launch_raster<MysteryDebugBigComposer>(dbg.p, atlas_->dev(), frames, MG_OBS_U8_XYC, n_, s);
    else launch_raster<MysteryDebugComposer>(dbg.p, atlas_->dev(), frames, MG_OBS_U8_XYC, n_, s);
    MG_HIP(hipGetLastError());
    MG_HIP(hipStreamSynchronize(s));  // dbg is released on return
}

Family* make_mystery(int variant, int num_envs) { return new MysteryFamily(variant, num_envs); }

}  // namespace mg

#ifdef MG_LAB_EMP_CLOCK
extern "C" int mg_lab_step_clock(unsigned long long* host, int n_waves) {
    return hipMemcpyFromSymbol(host, HIP_SYMBOL(mg::g_lab_step_clock), sizeof(unsigned long long) * 12 * (size_t)n_waves) == hipSuccess ? 0 : -1;
}
extern "C" int mg_lab_emp_clock(unsigned long long* host, int n_wgs) {
    return hipMemcpyFromSymbol(host, HIP_SYMBOL(mg::g_lab_emp_clock), sizeof(unsigned long long) * 3 * (size_t)n_wgs) == hipSuccess ? 0 : -1;
}
#endif
