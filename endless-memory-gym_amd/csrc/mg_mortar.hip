// mg_mortar.hip -- Mortar Mayhem family on gfx950: MortarMayhem-Grid-v0, MortarMayhem-v0, Endless-MortarMayhem-v0.
//
// Reference behaviour reproduced (bit-exact observations, rewards, dones, RNG consumption):
//   memory_gym/mortar_mayhem_grid.py     reset :213-278  step :280-375
//   memory_gym/mortar_mayhem.py          reset :206-272  step :274-369
//   memory_gym/endless_mortar_mayhem.py  reset :194-259  step :261-373
//   memory_gym/character_controller.py   free :89-146  grid :177-210  screen-wrap :226-283
//   memory_gym/pygame_assets.py          Command :241-304  MortarTile/MortarArena :306-418
//
// The step is ONE launch for uint8 observations (mortar_step_raster_kernel: the step's workgroups lead the raster's grid and a
// frame waits for its own descriptor), two launches otherwise (float formats, HIP-graph capture):
//   mortar_step_kernel : one LANE per environment instance.  Episode state machine, RNG, reward/done/info; emits a
//                   16-byte frame descriptor per instance.  State is small fixed-size records in HBM, read and
//                   written fully coalesced (lane i <-> record i).
//   raster_kernel<MortarComposer> : (mg_raster_v1.hpp) persistent workgroups, one frame at a time in LDS: arena
//                   template (selected by tiles-on / target tile) -> agent sprite stamp -> command glyph stamp.
#include <memory>

#include "mg_atlas_v1.hpp"
#include "mg_device.hpp"
#include "mg_family.hpp"
#include "mg_lab.hpp"
#include "mg_raster_v1.hpp"
#include "mg_stamps.hpp"

namespace mg {
using namespace v1;  // raster generation 1 (see mg_raster_v1.hpp)

enum { V_GRID = 0, V_FREE = 1, V_ENDLESS = 2 };

struct MortarParams {
    int variant, N, allowed, visual_feedback, max_steps, initial_count;
    int taskb;                   // MortarMayhemB*: no display phase, spawn offset for the free controller, vector obs
    int cmd_cap;                 // per-instance command list capacity
    int arena_x0, tile;          // arena top-left (x == y) and tile size in px
    int radius, sprite_dim;      // agent radius, sprite box
    int glyph_x0;                // blit position of the command glyph (x == y)
    int v_axis_i, v_diag_i;      // free controller: int(speed), int(speed/sqrt2)
    int off_lo, off_hi;          // endless: spawn offset = integers(off_lo, off_hi)
    double v_axis, v_diag;       // screen-wrap controller: un-truncated velocities
    OptList command_count, show_dur, show_delay, expl_dur, expl_delay;
    double r_fail, r_succ, r_ep_succ, r_new;
};

// 64-byte per-instance record
struct __attribute__((aligned(16))) MortarState {
    int16_t ax, ay;          // agent rect centre
    int16_t disp_x, disp_y;  // centre of the rect the frame shows (differs from ax/ay only through the Endless stale-sprite quirk)
    uint8_t rot8 : 3;        // agent.rotation / 45
    uint8_t disp_is_agent : 1;  // rotated_agent_rect is the live agent's rect
    uint8_t tiles_on : 1;
    uint8_t disp_sprite;     // sprite index the frame shows, 0xFF = none yet
    int8_t tx, ty;           // target tile
    int8_t nx, ny;           // normalized agent position
    uint16_t num_cmds, cur_cmd;
    uint16_t vis_pos, vis_len, vis_base;  // display schedule: next entry, length, first command it covers
    uint16_t cmd_steps, verify_step;
    // this episode's draws from the "sample one per episode" lists: 16 bits each (round 5; bytes before -- the reference takes
    // any int, mortar_mayhem_grid.py:253-254,268-269); the host refuses only what overflows the 16-bit display schedule
    uint16_t show_dur, show_delay, expl_dur, expl_delay;
    uint8_t gx, gy;          // grid controller position
    int32_t ep_len, t, total_completed;
    uint32_t dbg_pops;       // debug view only: entries popped from the reference's CLONE of the display schedule (one per debug
                             // render while the real schedule holds entries; copied anew at reset and at an endless regeneration,
                             // mortar_mayhem_grid.py:122,257, endless_mortar_mayhem.py:321)
    double ep_sum;
};
static_assert(sizeof(MortarState) == 64, "MortarState must be 64 bytes");

// per-instance frame descriptor: what the raster kernel composes (template -> agent sprite -> command glyph)
struct __attribute__((aligned(16))) MortarDesc {
    int16_t sx, sy;    // sprite top-left on screen
    uint16_t tmpl;     // background template index, 0xFFFF = leave the frame untouched (masked reset)
    uint8_t sprite;    // 0..7, 0xFF none
    uint8_t glyph;     // 0..9 (9 = blank), 0xFF none
    int16_t glyph_x0;  // blit position of the glyph (x == y)
    int16_t ring_x, ring_y;  // debug view only: top-left of the target ring stamp
    uint8_t ring_on;
    uint8_t epoch;     // one-launch step (mortar_step_raster_kernel): the step this descriptor belongs to, mod 256; the LAST
                       // byte of the record, so that the word that carries it can be published last
};
static_assert(sizeof(MortarDesc) == 16, "MortarDesc must be 16 bytes");
constexpr int STAMP_SPRITE0 = 0, STAMP_GLYPH0 = 8, STAMP_RING = 18;

struct MortarComposer {
    typedef MortarDesc Desc;
    static __device__ __forceinline__ bool skip(const Desc* dp) { return dp->tmpl == 0xFFFF; }
    static __device__ __forceinline__ void compose(const Desc* dp, const RasterCtx& R) {
        const Desc& d = *dp;
#if defined(MG_LAB_NO_TEMPLATE) && MG_LAB_NO_TEMPLATE == 2  // measurement builds (profiles/r06_raster_limits.md): no template at all (stale LDS)
#elif defined(MG_LAB_NO_TEMPLATE)                           // ... a cleared frame instead of the template: no global loads, the LDS writes stay
        fill_clear(R);
#else
        fill_template(R, d.tmpl);
#endif
        __syncthreads();
#ifndef MG_LAB_NO_STAMPS
        if (d.sprite != 0xFF) stamp(R, STAMP_SPRITE0 + d.sprite, d.sx, d.sy);
        if (d.glyph < 9) {
            __syncthreads();
            stamp(R, STAMP_GLYPH0 + d.glyph, d.glyph_x0, d.glyph_x0);
        }
#endif
    }
};

// _build_debug_surface (mortar_mayhem_grid.py:104-135): the observation's layers plus a green ring around the target tile
struct MortarDebugComposer {
    typedef MortarDesc Desc;
    static __device__ __forceinline__ bool skip(const Desc*) { return false; }
    static __device__ __forceinline__ void compose(const Desc* dp, const RasterCtx& R) {
        MortarComposer::compose(dp, R);
        __syncthreads();
        if (dp->ring_on) stamp(R, STAMP_RING, dp->ring_x, dp->ring_y);
    }
};

// (dx, dy) of command c: {1, 0, -1, 0, 0, 1, 1, -1, -1} / {0, 1, 0, -1, 0, 1, -1, 1, -1}, two bits each (value + 1) in a constant
// -- a table in memory is a dependent load per command in the reset's serial loop, with a lane-dependent index
constexpr uint32_t pack_deltas(const int (&v)[9]) {
    uint32_t m = 0;
    for (int c = 0; c < 9; ++c) m |= (uint32_t)(v[c] + 1) << (2 * c);
    return m;
}
constexpr int kDxHost[9] = {1, 0, -1, 0, 0, 1, 1, -1, -1}, kDyHost[9] = {0, 1, 0, -1, 0, 1, -1, 1, -1};
constexpr uint32_t CMD_DX_BITS = pack_deltas(kDxHost), CMD_DY_BITS = pack_deltas(kDyHost);
__device__ __forceinline__ int cmd_dx(int c) { return (int)((CMD_DX_BITS >> (2 * c)) & 3u) - 1; }
__device__ __forceinline__ int cmd_dy(int c) { return (int)((CMD_DY_BITS >> (2 * c)) & 3u) - 1; }

__device__ __forceinline__ int floordiv(int a, int b) {  // b > 0
    int q = a / b;
    return (a % b != 0 && a < 0) ? q - 1 : q;
}
__device__ __forceinline__ int mod6(int a) { return ((a % 6) + 6) % 6; }
__device__ __forceinline__ int round_haz(double v) { return v >= 0 ? (int)floor(v + 0.5) : -(int)floor(-v + 0.5); }

// Env.reset body (RNG draw order: spawn tile, [offset x2], [command_count], commands, show dur/delay, explosion dur/delay)
// _encode_commands_one_hot (mortar_mayhem_b_grid.py:100-129): slot of a Command.COMMANDS id inside its block of 9
__constant__ int8_t kCmdOneHot[9] = {1, 4, 2, 3, 0, 5, 6, 7, 8};
constexpr int VEC_DIM = 180;  // max_num_commands (20) * 9

__device__ void mortar_reset(const MortarParams& P, MortarState& s, Pcg& g, uint8_t* cmds, MortarDesc& d, float* gt, float* vec) {
    // the frame keeps showing the previous agent's rect until the first execution step (Endless only can observe it)
    if (s.disp_sprite != 0xFF && s.disp_is_agent) {
        s.disp_x = s.ax;
        s.disp_y = s.ay;
        s.disp_is_agent = 0;
    }
    int half = P.tile / 2;
    int tile_id = g.integers(0, P.N * P.N);
    int cx = P.arena_x0 + P.tile * (tile_id / P.N) + half;
    int cy = P.arena_x0 + P.tile * (tile_id % P.N) + half;
    if (P.variant == V_ENDLESS || (P.taskb && P.variant == V_FREE)) {  // mortar_mayhem_b.py:167
        cx += g.integers(P.off_lo, P.off_hi);
        cy += g.integers(P.off_lo, P.off_hi);
    }
    s.ax = (int16_t)cx;
    s.ay = (int16_t)cy;
    s.rot8 = 0;
    int nx = floordiv(cx - P.arena_x0, P.tile), ny = floordiv(cy - P.arena_x0, P.tile);
    s.nx = (int8_t)nx;
    s.ny = (int8_t)ny;
    s.gx = (uint8_t)nx;
    s.gy = (uint8_t)ny;

    int n;
    if (P.variant == V_ENDLESS) {
        n = P.initial_count;
        for (int i = 0; i < n; ++i) cmds[i] = (uint8_t)g.integers(0, P.allowed);
    } else {
        n = choice(g, P.command_count);
        int px = nx, py = ny;
        for (int i = 0; i < n; ++i) {
            uint32_t valid = 0;  // bit c: command c keeps the agent inside the arena (the reference's list, in order)
#pragma unroll
            for (int c = 0; c < 9; ++c) {
                const int qx = px + cmd_dx(c), qy = py + cmd_dy(c);
                if (c < P.allowed && qx >= 0 && qx < P.N && qy >= 0 && qy < P.N) valid |= 1u << c;
            }
            const int pick = g.integers(0, __popc(valid));
            uint32_t m = valid;
            for (int k = 0; k < pick; ++k) m &= m - 1;  // pick-th entry of the list
            const int c = __ffs(m) - 1;
            cmds[i] = (uint8_t)c;
            px += cmd_dx(c);
            py += cmd_dy(c);
        }
    }
    s.num_cmds = (uint16_t)n;
    if (P.taskb) {  // mortar_mayhem_b_grid.py:172 `_command_visualization = None`: nothing is drawn (no draws either)
        s.show_dur = s.show_delay = 0;
    } else {
        s.show_dur = (uint16_t)choice(g, P.show_dur);
        s.show_delay = (uint16_t)choice(g, P.show_delay);
    }
    s.vis_len = (uint16_t)(n * (s.show_dur + s.show_delay));
    s.vis_base = 0;
    s.vis_pos = 1;  // reset pops the first entry for its own frame
    s.dbg_pops = 0;
    int first = cmds[0];
    uint8_t glyph = s.show_dur > 0 ? (uint8_t)first : (uint8_t)9;
    if (P.variant == V_ENDLESS) {
        s.tx = (int8_t)mod6(nx + cmd_dx(first));
        s.ty = (int8_t)mod6(ny + cmd_dy(first));
    } else {
        s.tx = (int8_t)(nx + cmd_dx(first));
        s.ty = (int8_t)(ny + cmd_dy(first));
    }
    s.cur_cmd = 0;
    s.cmd_steps = 0;
    s.verify_step = 0;
    s.total_completed = 0;
    s.tiles_on = 0;
    s.t = 0;
    s.ep_len = 0;
    s.ep_sum = 0.0;
    s.expl_dur = (uint16_t)choice(g, P.expl_dur);
    s.expl_delay = (uint16_t)choice(g, P.expl_delay);

    // reset frame: blue arena, sprite 0 at the NEW agent position, first glyph
    d.tmpl = 0;
    d.sprite = 0;
    d.sx = (int16_t)(cx - P.sprite_dim / 2);
    d.sy = (int16_t)(cy - P.sprite_dim / 2);
    d.glyph = glyph;
    if (gt) {
        gt[0] = (float)(s.tx / 5.0);
        gt[1] = (float)(s.ty / 5.0);
    }
    if (vec) {  // obs["vector_observation"]: constant over the episode, written once per reset
        for (int k = 0; k < VEC_DIM; ++k) vec[k] = 0.0f;
        for (int c = 0; c < n && c < VEC_DIM / 9; ++c) vec[9 * c + kCmdOneHot[cmds[c]]] = 1.0f;
    }
}

struct MortarIO {
    MortarState* state;
    uint8_t* cmds;
    RngSoA rng;
    MortarDesc* desc;
    float* vec;  // [N][180] caller buffer bound with mg_bind_vector_obs (MortarMayhemB*), or NULL
    int* err;    // sticky error bits (mg_poll_errors / mg_peek_errors)
    // per-instance option sets (mg_set_option_set / mg_bind_option_sets): instance i runs under sets[set_of[i]]; both NULL while
    // the handle has ONE set -- the kernels then take the parameters from their arguments (scalar registers) as ever
    const MortarParams* sets;
    const int32_t* set_of;
};
constexpr int ERR_CMD_OVERFLOW = 32;  // include/memgym.h: Endless Mortar Mayhem command list longer than its capacity

// PS: per-instance option sets -- the parameters come from memory, io.sets[set_index(io.set_of, i)], instead of from the kernel arguments
template <bool PS>
__global__ __launch_bounds__(256) void mortar_reset_kernel(MortarParams P0, int n, MortarIO io, const int64_t* seeds,
                                                           const uint8_t* mask, float* gt) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const MortarParams& P = PS ? io.sets[set_index(io.set_of, i)] : P0;
    MortarDesc d;
    memset(&d, 0, sizeof(d));
    d.glyph_x0 = (int16_t)P.glyph_x0;
    if (mask && !mask[i]) {
        d.tmpl = 0xFFFF;
        io.desc[i] = d;
        return;
    }
    Pcg g;
    if (seeds) g.seed((uint64_t)seeds[i]);
    else g.load(io.rng, i);
    MortarState s = io.state[i];
    mortar_reset(P, s, g, io.cmds + (size_t)i * P.cmd_cap, d, gt ? gt + 2 * i : nullptr, io.vec ? io.vec + (size_t)i * VEC_DIM : nullptr);
    io.state[i] = s;
    g.store(io.rng, i);
    io.desc[i] = d;
}

__global__ __launch_bounds__(256) void mortar_init_kernel(int n, MortarState* state) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    MortarState s;
    memset(&s, 0, sizeof(s));
    s.disp_sprite = 0xFF;
    state[i] = s;
}

// The step of instance i.  FUSED (the one-launch step, mortar_step_raster_kernel): the RNG stream is read where it is drawn
// (ten registers less: that kernel must fit the raster's 72 VGPRs without scratch) and the descriptor is published for the
// frame workgroups of the SAME launch: agent-scope (write-through) stores, the word that carries the epoch last.
// What a step needs besides the instance index: ONE struct, so that it is the head of the kernel-argument segment of both
// step kernels (mortar_step_raster_kernel reads it a second time through the segment pointer, see there).
struct MortarStepArgs {
    MortarParams P;
    int n;
    MortarIO io;
    const int32_t* actions;
    float* reward_out;
    uint8_t* done_out;
    float* gt;
    mg_info_buffers info;
    int autoreset;
    MortarDesc* tdesc;  // FINAL form of the one-launch step (terminal observations kept): [N] descriptors of the terminal frames
};

// CLAIM (the step workgroups of the one-launch step): the wave steps its 64 instances only if it is the first to exchange this
// step's ticket into `claim_word` (see mortar_step_raster_kernel).  The exchange is ISSUED first and its answer awaited together
// with the state record: as a round trip of its own in front of the loads it delayed every descriptor, i.e. the whole launch,
// by 5-8 us (16,384 instances: 65 -> 73 us).
// FINAL (the one-launch step of a call that keeps terminal observations, mg_info_buffers.final_obs_dev): an instance that finishes
// publishes the descriptor of its TERMINAL frame in a.tdesc[i] before it resets, and says so in the reset frame's descriptor (ring_on, a
// field only the debug view uses otherwise): the frame workgroup draws the terminal frame into final_obs_dev first.
template <bool FUSED, bool CLAIM = false, bool PS = false, bool FINAL = false>
__device__ __forceinline__ void mortar_step_body(int i, const MortarStepArgs& a, uint32_t epoch, uint32_t* claim_word = nullptr,
                                                 uint32_t ticket = 0u) {
    uint32_t claimed_by = 0u;
    if constexpr (CLAIM) {
        claimed_by = ticket + 1u;  // lanes other than the wave's first: any value but the ticket
        if ((threadIdx.x & 63) == 0) claimed_by = __hip_atomic_exchange(claim_word, ticket, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // (the caller has dropped lanes with i >= n: the wave's first lane has its smallest i, so it is active whenever any lane is)
    }
    const MortarIO& io = a.io;
    const MortarParams& P = PS ? io.sets[set_index(io.set_of, i)] : a.P;  // (PS: per-instance option sets)
    const int32_t* const actions = a.actions;
    float* const reward_out = a.reward_out;
    uint8_t* const done_out = a.done_out;
    float* const gt = a.gt;
    const mg_info_buffers& info = a.info;
    const int autoreset = a.autoreset;
    // the action is requested together with the state record (read where it is used -- behind a test of the state -- it was
    // a second memory round trip at the head of the kernel)
    // (both reads unconditional, the grid variant's second one a repeat of the first: a load inside the variant's branch was
    // waited for at the end of that branch)
    const bool one_action = P.variant == V_GRID;
    int act0 = actions[one_action ? i : 2 * i], act1 = actions[one_action ? i : 2 * i + 1];
    // ... and so is the instance's RNG stream (40 bytes): only a finishing instance or an Endless list extension draws, but
    // read where it is drawn it was a third round trip, in the reset's tail of every launch
    Pcg g;
    bool rng_loaded = !FUSED;
    if constexpr (!FUSED) g.load(io.rng, i);
    MortarState s = io.state[i];
    asm volatile("" : "+v"(act0), "+v"(act1));  // (a use the compiler cannot move below the record's first use)
    if constexpr (!FUSED) g.pin();
    if constexpr (CLAIM) {
        asm volatile("" : "+v"(claimed_by));
        if ((uint32_t)__builtin_amdgcn_readfirstlane((int)claimed_by) == ticket) return;  // a frame wave has stepped this slot already
    }
    uint8_t* cmds = io.cmds + (size_t)i * P.cmd_cap;
    double reward = 0.0;
    bool done = false, cap = false;
    int success = 0;
    uint8_t glyph = 0xFF;
    bool rng_used = false;

    if (s.vis_pos < s.vis_len) {
        // display phase: pop the next schedule entry, agent frozen
        int period = s.show_dur + s.show_delay;
        int k = s.vis_pos / period, w = s.vis_pos % period;
        glyph = (w < s.show_dur) ? cmds[s.vis_base + k] : (uint8_t)9;
        s.vis_pos++;
        if (P.variant != V_ENDLESS || s.disp_sprite == 0xFF) {
            s.disp_sprite = 0;  // get_rotated_sprite(0) with the live agent's rect
            s.disp_is_agent = 1;
        }
    } else {
        int ax = s.ax, ay = s.ay;
        if (P.variant == V_GRID) {
            int a = act0;
            int rot = s.rot8 * 45;
            if (a == 1) rot = (rot + 90) % 360;
            if (a == 2) rot = (rot + 270) % 360;
            int gx = s.gx, gy = s.gy;
            if (a == 3) {
                int face = rot / 90;  // 0 N, 1 W, 2 S, 3 E
                if (face == 0) { if (gy > 0) gy--; }
                else if (face == 3) { if (gx < P.N - 1) gx++; }
                else if (face == 2) { if (gy < P.N - 1) gy++; }
                else { if (gx > 0) gx--; }
                ax = P.arena_x0 + P.tile * gx + P.tile / 2;
                ay = P.arena_x0 + P.tile * gy + P.tile / 2;
            }
            s.gx = (uint8_t)gx;
            s.gy = (uint8_t)gy;
            s.rot8 = (uint8_t)(rot / 45);
        } else {
            int a0 = act0, a1 = act1;
            int dxs = a0 == 1 ? -1 : (a0 == 2 ? 1 : 0), dys = a1 == 1 ? -1 : (a1 == 2 ? 1 : 0);
            int rot = s.rot8 * 45;
            if (a0 == 1) rot = 90;
            if (a0 == 2) rot = 270;
            if (a1 == 1) rot = 0;
            if (a1 == 2) rot = 180;
            if (dxs < 0 && dys < 0) rot = 45;
            if (dxs < 0 && dys > 0) rot = 135;
            if (dxs > 0 && dys < 0) rot = 315;
            if (dxs > 0 && dys > 0) rot = 225;
            s.rot8 = (uint8_t)(rot / 45);
            bool diag = dxs != 0 && dys != 0;
            if (P.variant == V_FREE) {
                int v = diag ? P.v_diag_i : P.v_axis_i;
                ax += dxs * v;
                ay += dys * v;
                int lo = P.arena_x0 + P.radius, hi = P.arena_x0 + P.tile * P.N - P.radius;
                ax = ax > hi ? hi : ax;
                ax = ax < lo ? lo : ax;
                ay = ay > hi ? hi : ay;
                ay = ay < lo ? lo : ay;
            } else {
                double v = diag ? P.v_diag : P.v_axis;
                ax = round_haz((double)ax + dxs * v);
                ay = round_haz((double)ay + dys * v);
                // wrap once the centre passes the arena edge by radius * 0.5 (character_controller.py:269-281)
                double left = P.arena_x0, right = P.arena_x0 + P.tile * P.N, off = P.radius * 0.5;
                double x = ax, y = ay;
                if (x > right + off) x = left - off;
                if (x < left - off) x = right + off;
                if (y > right + off) y = left - off;
                if (y < left - off) y = right + off;
                ax = round_haz(x);
                ay = round_haz(y);
            }
        }
        s.ax = (int16_t)ax;
        s.ay = (int16_t)ay;
        s.disp_sprite = s.rot8;
        s.disp_is_agent = 1;
        int nx = floordiv(ax - P.arena_x0, P.tile), ny = floordiv(ay - P.arena_x0, P.tile);
        s.nx = (int8_t)nx;
        s.ny = (int8_t)ny;
        bool on_target = (nx == s.tx) && (ny == s.ty);

        bool verify = (s.cmd_steps % s.expl_delay == 0) && s.cmd_steps > 0;
        if (verify && !s.tiles_on) {
            if (s.cur_cmd < s.num_cmds) {
                s.cur_cmd++;
                s.tiles_on = 1;
                if (on_target) {
                    reward += P.r_succ;
                    if (P.variant == V_ENDLESS) {
                        s.total_completed++;
                        if (s.cur_cmd == s.num_cmds) reward += P.r_new;
                    }
                } else {
                    done = true;
                    reward += P.r_fail;
                }
            }
            if (s.cur_cmd >= s.num_cmds) {
                if (P.variant == V_ENDLESS) {
                    if (!rng_loaded) g.load(io.rng, i);
                    rng_loaded = true;
                    rng_used = true;
                    int nc = g.integers(0, P.allowed);
                    if (s.num_cmds < P.cmd_cap) {
                        cmds[s.num_cmds] = (uint8_t)nc;
                        s.vis_base = s.num_cmds;
                        s.num_cmds++;
                    } else {  // capacity reached (512 commands = 131,328 correct tile visits in one episode; the reference's
                        // list is unbounded, endless_mortar_mayhem.py:316-318): end the episode AND say so
                        raise_error(io.err, ERR_CMD_OVERFLOW);
                        done = true;
                        cap = true;
                        s.vis_base = (uint16_t)(s.num_cmds - 1);
                    }
                    s.cur_cmd = 0;
                    s.verify_step = 0;
                    s.vis_pos = 0;
                    s.vis_len = (uint16_t)(s.show_dur + s.show_delay);
                    s.dbg_pops = 0;
                } else {
                    done = true;
                    success = 1;
                    reward += P.r_ep_succ;
                }
            }
            s.cmd_steps = 1;
        }
        if (s.tiles_on) {
            if (s.verify_step % s.expl_dur == 0 && s.verify_step > 0) {
                s.tiles_on = 0;
                s.verify_step = 0;
                if (s.cur_cmd < s.num_cmds) {
                    int c = cmds[s.cur_cmd];
                    if (P.variant == V_ENDLESS) {
                        s.tx = (int8_t)mod6(s.tx + cmd_dx(c));
                        s.ty = (int8_t)mod6(s.ty + cmd_dy(c));
                    } else {
                        s.tx = (int8_t)(s.tx + cmd_dx(c));
                        s.ty = (int8_t)(s.ty + cmd_dy(c));
                    }
                }
            } else {
                if (!on_target) {
                    done = true;
                    reward = P.r_fail;  // overwrite (mortar_mayhem_grid.py:348)
                }
                s.verify_step++;
            }
        } else {
            s.cmd_steps++;
        }
    }

    if (P.variant == V_ENDLESS) {
        s.t++;
        if (s.t == P.max_steps) done = true;
    }
    s.ep_sum += reward;
    s.ep_len++;

    if (done) {
        if (info.ep_reward_dev) info.ep_reward_dev[i] = s.ep_sum;
        if (info.ep_length_dev) info.ep_length_dev[i] = s.ep_len;
        if (P.variant == V_ENDLESS) {
            if (info.aux_dev[0]) info.aux_dev[0][i] = (float)s.total_completed;
            if (info.aux_dev[1]) info.aux_dev[1][i] = (float)(s.num_cmds > 1 ? s.num_cmds - 1 : 0);
        } else {
            if (info.aux_dev[0]) info.aux_dev[0][i] = (float)success;
            if (info.aux_dev[1]) info.aux_dev[1][i] = (float)((double)((int)s.cur_cmd - 1 + success) / (double)s.num_cmds);
        }
    }
    reward_out[i] = (float)reward;
    if (info.reward64_dev) info.reward64_dev[i] = reward;  // the reference's Python float, unrounded
    done_out[i] = done ? 1 : 0;
    if (info.capacity_dev) info.capacity_dev[i] = cap ? 1 : 0;  // (include/memgym.h: the episode ended on a capacity of this build)

    MortarDesc d;
    memset(&d, 0, sizeof(d));
    d.glyph_x0 = (int16_t)P.glyph_x0;
    if (done && autoreset) {
        if constexpr (FINAL) {  // the terminal frame's descriptor (the else branch below), published like the frame descriptor's first words
            MortarDesc td;
            memset(&td, 0, sizeof(td));
            td.glyph_x0 = (int16_t)P.glyph_x0;
            const int tcx = s.disp_is_agent ? s.ax : s.disp_x, tcy = s.disp_is_agent ? s.ay : s.disp_y;
            td.sx = (int16_t)(tcx - P.sprite_dim / 2);
            td.sy = (int16_t)(tcy - P.sprite_dim / 2);
            td.sprite = s.disp_sprite;
            td.glyph = glyph;
            td.tmpl = (uint16_t)((s.tiles_on && P.visual_feedback) ? 1 + s.tx * P.N + s.ty : 0);
            uint32_t tw[4];
            memcpy(tw, &td, sizeof(tw));
            uint32_t* tdst = reinterpret_cast<uint32_t*>(&a.tdesc[i]);
            __hip_atomic_store(tdst + 0, tw[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(tdst + 1, tw[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(tdst + 2, tw[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(tdst + 3, tw[3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // (all four in front of the wait below)
        }
        if (!rng_loaded) g.load(io.rng, i);
        rng_loaded = true;
        rng_used = true;
        mortar_reset(P, s, g, cmds, d, (gt && P.variant == V_ENDLESS) ? gt + 2 * i : nullptr, io.vec ? io.vec + (size_t)i * VEC_DIM : nullptr);
        if constexpr (FINAL) d.ring_on = 1;
    } else {
        int cx = s.disp_is_agent ? s.ax : s.disp_x, cy = s.disp_is_agent ? s.ay : s.disp_y;
        d.sx = (int16_t)(cx - P.sprite_dim / 2);
        d.sy = (int16_t)(cy - P.sprite_dim / 2);
        d.sprite = s.disp_sprite;
        d.glyph = glyph;
        d.tmpl = (uint16_t)((s.tiles_on && P.visual_feedback) ? 1 + s.tx * P.N + s.ty : 0);
        if (gt && P.variant == V_ENDLESS) {
            gt[2 * i] = (float)(s.tx / 5.0);
            gt[2 * i + 1] = (float)(s.ty / 5.0);
        }
    }
    if (rng_used) g.store(io.rng, i);
    io.state[i] = s;
    if constexpr (FUSED) {
        d.epoch = (uint8_t)epoch;
        uint32_t w[4];
        memcpy(w, &d, sizeof(w));
        uint32_t* dst = reinterpret_cast<uint32_t*>(&io.desc[i]);
        __hip_atomic_store(dst + 0, w[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(dst + 1, w[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(dst + 2, w[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the three words have reached the coherence point before the fourth leaves
        __hip_atomic_store(dst + 3, w[3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
        io.desc[i] = d;
    }
}

template <bool PS>
__global__ __launch_bounds__(256) void mortar_step_kernel(MortarStepArgs a) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < a.n) mortar_step_body<false, false, PS>(i, a, 0u);
}

// ONE launch per step (uint8 observations).  `logic_wgs` workgroups of the grid run the step (one lane per instance), all others
// are the raster's persistent workgroups; a frame's workgroup waits for ITS descriptor -- the epoch in the descriptor's last
// word, read at agent scope past the caches -- instead of for the slowest wave of a separate logic launch plus that launch's
// fixed cost: the first frames leave ~8 us earlier (MortarMayhem-Grid 65,536: 233 -> 224 us per step, 281 -> 292 M env-steps/s;
// 16,384: 69 -> 65 us; profiles/r03_one_launch.md).
//
// Liveness does NOT rest on the order in which the hardware dispatches workgroups (round 4).  The instances are stepped in
// slots of 64 (one wave); a slot belongs to whichever wave first exchanges this step's ticket into its claim word.  Normally
// that is the step workgroup's wave (the step workgroups come first in the grid and are resident before the frame workgroups
// fill the chip).  A frame wave whose descriptor has not shown the epoch after RESCUE_AFTER_TICKS (200 us) tries the claim of the
// slot its frame belongs to ITSELF: if it wins, the step wave has not started yet (e.g. no free slot on the chip because frame
// workgroups were dispatched first) and the frame wave steps those 64 instances with its own lanes, then draws; if it loses,
// the slot's owner is a resident wave that never waits for anything, so the descriptor is on its way.  Every wait therefore
// ends, no frame is ever drawn from a stale descriptor, and there is no time-out to report (error bit 128 of rounds <= 3 is
// gone).  tests/test_gpu_one_launch.py runs the launch with the step workgroups LAST in the grid (lab build) -- every frame
// workgroup resident before any step workgroup -- and under a concurrent stream.
//
// Hand-over of the 16-byte descriptor: the publisher writes words 0..2 with agent-scope (write-through) stores, waits until they
// have reached the coherence point (s_waitcnt vmcnt(0)) and only then writes word 3, which carries the epoch; the reader polls
// word 3 with agent-scope loads and, once it shows the epoch, reads words 0..2 with agent-scope loads issued AFTER that
// observation.  Release / acquire atomics would be the textbook form; at agent scope on gfx950 they write back / invalidate
// the whole L2 of the XCD around every hand-over (buffer_wbl2 / buffer_inv sc1), with the observation stream in that L2.
// The two-launch form is used while a stream is being captured into a HIP graph (epoch and ticket are launch arguments: a
// replay would find them satisfied already) and for handles with instance groups (their stagger needs the logic launch's end).
#define MG_KERNARG_AS __attribute__((address_space(4)))
// A frame wave tries the claim after it has waited this long (real-time clock, 100 MHz).  The step workgroups normally publish
// within 15-20 us; the first version counted 32 polls (~15 us as it turned out): every early frame wave then sent its one
// exchange at the few cache lines of claim words, and those ~7,000 serialised atomics cost the 16,384-instance launch 6 of
// its 66 us (profiles/r04_one_launch.md).
constexpr unsigned long long RESCUE_AFTER_TICKS = 20000;  // 200 us
// DONE_FLAG (the single-instance fast path, mg_single_step: ONE frame workgroup): when the frame is out, the workgroup stores `done_ticket`
// to `done_flag` -- a word in the caller's pinned block that the host polls -- at system scope: 2.7 us less per step than a stream memory
// operation behind the launch, 4.5 us less than hipStreamSynchronize (tools/microbench/launch_wait.hip).  Everything else the host reads
// (reward, done, the episode record) was stored by the step's wave BEFORE it published the descriptor this workgroup waited for.
// FINAL: the call keeps terminal observations (see mortar_step_body) -- a kernel of its own, the measured one (FINAL = false) is as it was.
template <bool DONE_FLAG, bool FINAL = false>
__global__ __launch_bounds__(256, 7) void mortar_step_raster_kernel(MortarStepArgs a, int logic_wgs, int logic_base, uint32_t epoch,
                                                                    uint32_t ticket, uint32_t* claims, uint32_t* rescues,
                                                                    RasterAtlas A, void* __restrict__ obs, uint32_t* done_flag, uint32_t done_ticket) {
    const int n = a.n;
    const int tid = threadIdx.x, lane = tid & 63;
    const int rel = (int)blockIdx.x - logic_base;
    const bool is_logic = rel >= 0 && rel < logic_wgs;
    // true: this wave owns slot `q` (instances 64 q .. 64 q + 63) for this step
    auto claim = [&](int q) -> bool {
        uint32_t old = 0;
        if (lane == 0) old = __hip_atomic_exchange(claims + q, ticket, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return (uint32_t)__builtin_amdgcn_readfirstlane((int)old) != ticket;
    };
    if (is_logic) {  // a step workgroup: wave w steps slot 4 rel + w unless a frame wave got there first
        const int q = rel * 4 + (tid >> 6), i = q * 64 + lane;
#if defined(MG_LABV) && MG_LABV >= 1
        if (i < n) mortar_step_body<true, false, false, FINAL>(i, a, epoch);
#else
        if (i < n) mortar_step_body<true, true, false, FINAL>(i, a, epoch, claims + q, ticket);
#endif
        return;
    }
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    RasterCtx R;
    R.frame = smem;
    R.mask = reinterpret_cast<uint32_t*>(smem + FRAME_BYTES);
    R.A = A;
    R.T = A.tables;
    R.tid = tid;
    const int stride = (int)gridDim.x - logic_wgs;
    for (int v = (int)blockIdx.x < logic_base ? (int)blockIdx.x : (int)blockIdx.x - logic_wgs; v < n; v += stride) {
        const int env = xcd_grouped_frame(v, n);
        // every lane reads the same words (one transaction per wave); no barrier: the waves of a workgroup wait separately
        const uint32_t* src = reinterpret_cast<const uint32_t*>(a.io.desc + env);
        uint32_t w[4];
        bool tried = false;
        unsigned long long t0 = 0;
        for (int polls = 0;; ++polls) {
            // (all lanes read the same word; readfirstlane tells the compiler so: the wait loop's control stays scalar)
            w[3] = (uint32_t)__builtin_amdgcn_readfirstlane((int)__hip_atomic_load(src + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
            if ((w[3] >> 24) == epoch) break;
#if defined(MG_LABV) && MG_LABV >= 2
            if (false) {
#else
            if (polls == 0) t0 = wall_clock64();
            if (!tried && (polls & 15) == 15 && wall_clock64() - t0 >= RESCUE_AFTER_TICKS) {  // (the clock is read every 16th poll)
#endif
                tried = true;  // (a lost claim is not retried: its owner is running)
                if (claim(env >> 6)) {  // rare: step the 64 instances around this frame here; the next poll finds the epoch
                    // The step's arguments are read AGAIN, from the kernel-argument segment, through a pointer the compiler
                    // cannot see through: as loop invariants they were hoisted out of the frame loop and kept in ~80 scalar
                    // registers for its whole length (spilled to vector lanes, those to scratch: 232 B per lane).
                    const MortarStepArgs MG_KERNARG_AS* ka = (const MortarStepArgs MG_KERNARG_AS*)__builtin_amdgcn_kernarg_segment_ptr();
                    asm volatile("" : "+s"(ka));
                    int i = (env >> 6) * 64 + lane;
                    asm volatile("" : "+v"(i));  // (nor may what the step derives from `i` be computed at the head of every frame)
                    if (i < n) mortar_step_body<true, false, false, FINAL>(i, *(const MortarStepArgs*)ka, epoch);
                    if (lane == 0) atomicAdd(rescues, 1u);
                    continue;
                }
            }
            __builtin_amdgcn_s_sleep(4);
        }
        asm volatile("" ::: "memory");  // the loads below stay behind the observation of the epoch
        w[0] = __hip_atomic_load(src + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        w[1] = __hip_atomic_load(src + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        w[2] = __hip_atomic_load(src + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        MortarDesc d;
        memcpy(&d, w, sizeof(d));
        if constexpr (FINAL) {
            if (d.ring_on) {  // the instance finished in this step: its terminal frame first, into the caller's final-observation buffer
                // (a.tdesc[env] was published in front of the descriptor whose epoch has just been observed)
                const uint32_t* tsrc = reinterpret_cast<const uint32_t*>(a.tdesc + env);
                uint32_t tw[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) tw[k] = __hip_atomic_load(tsrc + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                MortarDesc td;
                memcpy(&td, tw, sizeof(td));
                int tt = tid;
                asm volatile("" : "+v"(tt));
                R.tid = tt;
                MortarComposer::compose(&td, R);
                __syncthreads();
                store_frame<MG_OBS_U8_XYC, false>(smem, a.info.final_obs_dev, env, tt);
                __syncthreads();
            }
        }
        if (MortarComposer::skip(&d)) continue;
        // the lane's frame offsets are derived from an opaque copy of its index, i.e. inside the iteration: as loop invariants
        // they were live across the (rare) step code above, which needs every register the kernel has
        int t = tid;
        asm volatile("" : "+v"(t));
        R.tid = t;
        MortarComposer::compose(&d, R);
        __syncthreads();
        store_frame<MG_OBS_U8_XYC, false>(smem, obs, env, t);  // (plain stores: non-temporal ones 281 -> 226-241 M at 65,536, round 4)
        __syncthreads();
    }
    if constexpr (DONE_FLAG) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");  // this wave's stores (the frame; after a rescue also the step's results) are performed system-wide
        __syncthreads();
        if (tid == 0) __hip_atomic_store(done_flag, done_ticket, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// Debug view: the frame descriptors of the current frames with (a) the glyph the reference's CLONE of the display schedule
// info["ground_truth"] in float64: target tile / 5.0 (endless_mortar_mayhem.py:259,358,362)
__global__ __launch_bounds__(256) void mortar_gt64_kernel(int n, const MortarState* state, double* out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const MortarState s = state[i];
    out[2 * i] = s.tx / 5.0;
    out[2 * i + 1] = s.ty / 5.0;
}

// yields -- its next entry, popped (dbg_pops, the only state a debug render changes), only while the real schedule still
// holds entries (oracle/mgo_mortar.c mm_debug) -- and (b) the ring around the target tile.
__global__ __launch_bounds__(256) void mortar_debug_desc_kernel(MortarParams P0, int n, MortarIO io, MortarDesc* out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const MortarParams& P = io.set_of ? io.sets[set_index(io.set_of, i)] : P0;
    const MortarState s = io.state[i];
    const uint8_t* cmds = io.cmds + (size_t)i * P.cmd_cap;
    MortarDesc d = io.desc[i];
    {   // the agent the debug view shows is the stored (rotated_agent_surface, rotated_agent_rect) pair, which no reset clears
        // (mortar_mayhem_grid.py:115-118): stale from the previous episode until the first step; sprite 0 before any step
        const bool have = s.disp_sprite != 0xFF;
        const int cx = (have && !s.disp_is_agent) ? s.disp_x : s.ax, cy = (have && !s.disp_is_agent) ? s.disp_y : s.ay;
        d.sx = (int16_t)(cx - P.sprite_dim / 2);
        d.sy = (int16_t)(cy - P.sprite_dim / 2);
        d.sprite = have ? s.disp_sprite : (uint8_t)0;
        d.tmpl = (uint16_t)((s.tiles_on && P.visual_feedback) ? 1 + s.tx * P.N + s.ty : 0);
        d.glyph_x0 = (int16_t)P.glyph_x0;
    }
    d.glyph = 0xFF;
    if (s.vis_pos < s.vis_len) {
        const int idx = (int)s.dbg_pops, period = s.show_dur + s.show_delay;
        io.state[i].dbg_pops = s.dbg_pops + 1;
        if (idx >= 0 && idx < (int)s.vis_len && period > 0) {
            const int k = idx / period, w = idx % period;
            d.glyph = (w < s.show_dur) ? cmds[s.vis_base + k] : (uint8_t)9;
        }
    }
    const int r = P.tile / 2;  // pygame.draw.circle(surface, green, tile centre, tile_dim // 2, int(8 * SCALE))
    d.ring_x = (int16_t)(P.arena_x0 + P.tile * s.tx + P.tile / 2 - r);
    d.ring_y = (int16_t)(P.arena_x0 + P.tile * s.ty + P.tile / 2 - r);
    d.ring_on = 1;
    out[i] = d;
}

// ---------------------------------------------------------------------------------------------------------
// Host side
// ---------------------------------------------------------------------------------------------------------
static const double SCALE = 0.25;  // the reference's module constant (e.g. mortar_mayhem_grid.py:13)

class MortarFamily : public Family {
   public:
    // variant 3 / 4 = MortarMayhemB-Grid-v0 / MortarMayhemB-v0: the Grid / free machine with taskb set
    MortarFamily(int variant_id, int n) : opt_(one_set()), P_(opt_[0]->P), n_(n) {
        memset(&P_, 0, sizeof(P_));
        const int variant = variant_id >= 3 ? variant_id - 3 : variant_id;
        P_.variant = variant;
        P_.taskb = variant_id >= 3;
        agent_scale_ = 1.0 * SCALE;
        agent_speed_ = 12.0 * SCALE;
        P_.N = variant == V_ENDLESS ? 6 : 5;
        P_.allowed = variant == V_GRID ? 5 : 9;
        P_.visual_feedback = 1;
        P_.max_steps = -1;
        P_.initial_count = 1;
        P_.cmd_cap = variant == V_ENDLESS ? 512 : 32;
        if (variant == V_ENDLESS) {  // lab build only (tests/test_gpu_error_bits.py): a small capacity makes the overflow reachable
            const int cap = lab_int("MEMGYM_EMM_CMD_CAP", 0);
            if (cap >= 4 && cap <= 512) P_.cmd_cap = cap;
        }
        MortarOpt& O = *opt_[0];
        O.st_command_count.set(P_.command_count, {10});
        O.st_show_dur.set(P_.show_dur, {3});
        O.st_show_delay.set(P_.show_delay, {1});
        O.st_expl_dur.set(P_.expl_dur, {variant == V_GRID ? 2 : 6});
        O.st_expl_delay.set(P_.expl_delay, {variant == V_GRID ? 6 : 18});
        P_.r_fail = 0.0;
        P_.r_succ = 0.1;
        P_.r_ep_succ = 0.0;
        P_.r_new = 0.0;
        state_.alloc(n);
        cmds_.alloc((size_t)n * P_.cmd_cap);
        desc_.alloc(n);
        tdesc_.alloc(n);  // (terminal-frame descriptors of the FINAL one-launch step: 16 B per instance; allocated here so that no step allocates)
        rng_.alloc(n);
        err_.alloc();
        claims_.alloc((size_t)((n + 255) / 256) * 4);
        rescues_.alloc(1);
        sets_dev_.alloc(MG_MAX_OPTION_SETS);
        hipLaunchKernelGGL(mortar_init_kernel, dim3((n + 255) / 256), dim3(256), 0, 0, n, state_.p);
        MG_HIP(hipDeviceSynchronize());
        rebuild();
        defaults_ = P_;  // (short lists only: no device arrays behind them)
    }

    // include/memgym.h: mg_set_capacity.  "commands" (Endless-MortarMayhem-v0): entries of the command list per instance -- the reference's
    // list grows by one with every completed round (endless_mortar_mayhem.py:311-333); an episode that would need one more ends (capacity_dev)
    void set_capacity(const std::string& what, int64_t v) override {
        if (!(P_.variant == V_ENDLESS && what == "commands")) return Family::set_capacity(what, v);
        if (v < 4 || v > 32768) throw OptionError{-3, "commands: 4 .. 32,768"};
        if (seeded_) throw std::runtime_error("mg_set_capacity: before the first reset");
        if (2 * P_.initial_count > v) throw OptionError{-3, "commands: below twice the initial_command_count in force"};
        MG_HIP(hipDeviceSynchronize());
        P_.cmd_cap = (int)v;
        cmds_.alloc((size_t)n_ * P_.cmd_cap);
        for (size_t k = 1; k < opt_.size(); ++k) copy_geometry(opt_[k]->P, P_);
        copy_geometry(defaults_, P_);
        sets_dirty_ = true;
    }
    int64_t capacity(const std::string& what) const override {
        if (P_.variant == V_ENDLESS && what == "commands") return P_.cmd_cap;
        return Family::capacity(what);
    }
    // mg_single_step: the NEXT step's last kernel stores `ticket` to *flag (a word the host polls) once its results are out.  Only the
    // one-launch step of a one-instance handle without ground truth can promise that; everything else answers false and the caller puts a
    // stream memory operation behind the step instead.
    bool arm_done_flag(uint32_t* flag_dev, uint32_t ticket) override {
        if (!(n_ == 1 && obs_format == MG_OBS_U8_XYC && fuse_step() && !per_set() && gt_dim() == 0 && !dirty_)) return false;
        flag_dev_ = flag_dev;
        flag_ticket_ = ticket;
        flag_armed_ = true;
        return true;
    }
    int action_dim() const override { return P_.variant == V_GRID ? 1 : 2; }
    int gt_dim() const override { return P_.variant == V_ENDLESS ? 2 : 0; }
    int vec_dim() const override { return P_.taskb ? VEC_DIM : 0; }
    void bind_vector_obs(float* dev) override { vec_ = dev; }
    const char* info_name(int k) const override {
        if (P_.variant == V_ENDLESS) return k == 0 ? "commands_completed" : (k == 1 ? "max_command_sequence" : nullptr);
        return k == 0 ? "success" : (k == 1 ? "commands_completed" : nullptr);
    }

    // One key of the reset options, for option set `set` (0 = the handle-wide set of mg_set_option).  Sets > 0 hold everything
    // that does not change the geometry (atlases and templates are shared by the handle's instances).
    void set_option(const std::string& key, const double* v, int n) override { set_option_set(0, key, v, n); }
    void set_option_set(int set, const std::string& key, const double* v, int n) override {
        if (set < 0 || set >= MG_MAX_OPTION_SETS) throw OptionError{-3, "option set index out of range"};
        while ((int)opt_.size() <= set) {  // a new set starts from the constructor's defaults (= the reference's), geometry from set 0
            opt_.emplace_back(new MortarOpt());
            opt_.back()->P = defaults_;
            copy_geometry(opt_.back()->P, P_);
        }
        MortarOpt& O = *opt_[set];
        MortarParams& P = O.P;
        const bool endless = P_.variant == V_ENDLESS;
        auto geometry = [&]() {
            if (set != 0) throw OptionError{-3, "reset parameter " + key + " changes the geometry shared by the handle's instances: it can only be set for all of them (option set 0)"};
        };
        auto scalar_i = [&](int& dst) { dst = to_int_checked(v[0], key.c_str()); };
        // "sample one per episode" lists of any length (np_random.choice, e.g. mortar_mayhem_grid.py:181,253-254,268-269)
        auto list = [&](OptList& l, OptListStore& st, int lo, int hi) {
            if (n < 1) throw OptionError{-3, "option " + key + ": an empty list cannot be sampled"};
            std::vector<int> vals(n);
            for (int i = 0; i < n; ++i) {
                vals[i] = to_int_checked(v[i], key.c_str());
                if (vals[i] < lo || vals[i] > hi)
                    throw OptionError{-3, "option " + key + ": value out of the supported range " + std::to_string(lo) + ".." + std::to_string(hi)};
            }
            st.set(l, vals);
        };
        // (a geometry option in a set > 0 is accepted when it says what the handle's geometry already is)
        if (key == "agent_scale") { if (set != 0) { if (v[0] != agent_scale_) geometry(); } else { agent_scale_ = v[0]; dirty_ = true; } }
        else if (key == "allowed_commands") {
            int a = to_int_checked(v[0], key.c_str());
            if (a < 4 || a > 9) throw OptionError{-4, "assert 4 <= allowed_commands <= 9"};
            P.allowed = a;
        }
        // (entries are 16 bits in MortarState; an explosion entry of 0 is the reference's ZeroDivisionError in `% explosion_delay` (:304,343),
        // a show duration of 0 with a delay of 0 its IndexError at the reset's pop(0) (:257))
        else if (!P_.taskb && key == "command_show_duration") list(P.show_dur, O.st_show_dur, 1, 65535);
        else if (!P_.taskb && key == "command_show_delay") list(P.show_delay, O.st_show_delay, 0, 65535);
        else if (key == "explosion_duration") list(P.expl_dur, O.st_expl_dur, 1, 65535);
        else if (key == "explosion_delay") list(P.expl_delay, O.st_expl_delay, 1, 65535);
        else if (key == "visual_feedback") P.visual_feedback = v[0] != 0.0;
        else if (key == "reward_command_failure") P.r_fail = v[0];
        else if (key == "reward_command_success") P.r_succ = v[0];
        else if (endless && key == "max_steps") scalar_i(P.max_steps);
        else if (endless && key == "initial_command_count") {
            int c = to_int_checked(v[0], key.c_str());
            if (c < 1 || c > P_.cmd_cap / 2) throw OptionError{-3, "initial_command_count out of the supported range"};
            P.initial_count = c;
        }
        else if (endless && key == "reward_new_command_success") P.r_new = v[0];
        else if (!endless && key == "arena_size") {
            int a = to_int_checked(v[0], key.c_str());
            if (a < 2 || a > 6) throw OptionError{-4, "assert 2 <= arena_size <= 6"};
            if (set != 0 && a != P_.N) geometry();
            if (set == 0 && a != P_.N) {
                P_.N = a;
                dirty_ = true;
            }
        }
        else if (!endless && key == "command_count") list(P.command_count, O.st_command_count, 1, P_.taskb ? VEC_DIM / 9 : P_.cmd_cap);
        else if (!endless && key == "reward_episode_success") P.r_ep_succ = v[0];
        else if (P_.variant != V_GRID && key == "agent_speed") { if (set != 0) { if (v[0] != agent_speed_) geometry(); } else { agent_speed_ = v[0]; dirty_ = true; } }
        else throw OptionError{-2, "unknown reset parameter " + key};
        sets_dirty_ = true;
    }
    // instance i runs under option set set_of_dev[i] (device array [num_envs], caller-owned; NULL: every instance under set 0)
    void bind_option_sets(const int32_t* set_of_dev) override { set_of_ = set_of_dev; }

    void reset(const int64_t* seeds, const uint8_t* mask, void* obs, float* gt, hipStream_t s) override {
        if (dirty_) rebuild();
        if (!seeds && !seeded_) throw std::runtime_error("reset(seed=None) before any seeded reset");
        for (auto& O : opt_) {  // the display schedule (commands x (duration + delay) entries) is indexed with 16 bits
            const long long n_max = P_.variant == V_ENDLESS ? O->P.initial_count : O->st_command_count.max(O->P.command_count);
            if (!P_.taskb && n_max * ((long long)O->st_show_dur.max(O->P.show_dur) + O->st_show_delay.max(O->P.show_delay)) > 65535)
                throw OptionError{-3, "command_count x (command_show_duration + command_show_delay) exceeds the 65,535 entries of this build's display schedule"};
        }
        if (seeds) seeded_ = true;  // with a mask the caller is responsible for having seeded the other instances
        upload_sets(s);
        if (per_set())
            hipLaunchKernelGGL(mortar_reset_kernel<true>, dim3((n_ + 255) / 256), dim3(256), 0, s, P_, n_, io(), seeds, mask, gt_dim() ? gt : nullptr);
        else
            hipLaunchKernelGGL(mortar_reset_kernel<false>, dim3((n_ + 255) / 256), dim3(256), 0, s, P_, n_, io(), seeds, mask, gt_dim() ? gt : nullptr);
        if (mask && sparse_masked_raster()) {  // few frames of many: by the mask, not by a walk over every descriptor (mg_raster_v1.hpp)
            launch_raster_sparse<MortarComposer>(desc_.p, atlas_->dev(), obs, obs_format, n_, s, mask);
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
        const MortarStepArgs sa{P_, n_, io(), actions, reward, done, gt_dim() ? gt : nullptr, ib, autoreset, nullptr};
        // one launch: mortar_step_raster_kernel (handles with ONE option set: the per-set step code reads its parameters from memory)
        if (obs_format == MG_OBS_U8_XYC && fuse_step() && !per_set() && !capturing(s)) {
            epoch_ = epoch_ % 255u + 1u;  // 1 .. 255: never the 0 a reset's (or the two-launch step's) descriptors carry
            ++ticket_;                    // claim words hold the ticket of the last one-launch step: never this one
            const int logic_wgs = (n_ + 255) / 256;
            const int frames = n_ < raster_grid(n_) ? n_ : raster_grid(n_);
            // lab build, MEMGYM_LAB_LOGIC_LAST=1: the step workgroups at the END of the grid -- the dispatch order the design must survive
            static const bool logic_last = lab_int("MEMGYM_LAB_LOGIC_LAST", 0) != 0;
            prof.begin(1, s);
            if (flag_armed_ && n_ == 1) {
                hipLaunchKernelGGL(mortar_step_raster_kernel<true>, dim3(logic_wgs + frames), dim3(256), RASTER_LDS, s, sa, logic_wgs,
                                   logic_last ? frames : 0, epoch_, ticket_, claims_.p, rescues_.p, atlas_->dev(), obs, flag_dev_, flag_ticket_);
                flag_armed_ = false;
            } else if (ib.final_obs_dev && autoreset) {  // terminal observations kept by the launch itself (keeps_final_obs)
                MortarStepArgs fa = sa;
                fa.tdesc = tdesc_.p;
                hipLaunchKernelGGL((mortar_step_raster_kernel<false, true>), dim3(logic_wgs + frames), dim3(256), RASTER_LDS, s, fa, logic_wgs,
                                   logic_last ? frames : 0, epoch_, ticket_, claims_.p, rescues_.p, atlas_->dev(), obs, (uint32_t*)nullptr, 0u);
            } else {
                hipLaunchKernelGGL(mortar_step_raster_kernel<false>, dim3(logic_wgs + frames), dim3(256), RASTER_LDS, s, sa, logic_wgs,
                                   logic_last ? frames : 0, epoch_, ticket_, claims_.p, rescues_.p, atlas_->dev(), obs, (uint32_t*)nullptr, 0u);
            }
            MG_HIP(hipGetLastError());
            prof.end(1, s);
            return;
        }
        prof.begin(0, s);
        const int sb = step_block(256);
        if (per_set()) hipLaunchKernelGGL(mortar_step_kernel<true>, dim3((n_ + sb - 1) / sb), dim3(sb), 0, s, sa);
        else hipLaunchKernelGGL(mortar_step_kernel<false>, dim3((n_ + sb - 1) / sb), dim3(sb), 0, s, sa);
        end_logic(s);
        prof.begin(1, s);
        raster(obs, s);
        prof.end(1, s);
    }

    std::vector<std::pair<void*, size_t>> state_blobs() override {
        std::vector<std::pair<void*, size_t>> v = {{state_.p, state_.bytes()}, {cmds_.p, cmds_.bytes()}};
        rng_.blobs(v);
        return v;
    }

    void debug_rng(int i, uint64_t out[6]) override { rng_.debug(i, out); }
    void ground_truth64(double* out, hipStream_t s) override {
        if (!gt_dim() || !out) return;
        hipLaunchKernelGGL(mortar_gt64_kernel, dim3((n_ + 255) / 256), dim3(256), 0, s, n_, state_.p, out);
        MG_HIP(hipGetLastError());
    }
    int poll_errors() override {
        MG_HIP(hipDeviceSynchronize());
        return err_.take();
    }
    int peek_errors() override { return err_.peek(); }
    bool debug_counter(const std::string& name, int64_t* out) override {
        if (name == "cmd_list_max" || name == "cmd_list_ge12") {  // the instances' command lists as they stand (a scan of the state records)
            std::vector<MortarState> h(n_);
            MG_HIP(hipDeviceSynchronize());
            MG_HIP(hipMemcpy(h.data(), state_.p, sizeof(MortarState) * (size_t)n_, hipMemcpyDeviceToHost));
            int64_t mx = 0, ge = 0;
            for (const MortarState& s : h) {
                mx = s.num_cmds > mx ? s.num_cmds : mx;
                ge += s.num_cmds >= 12;
            }
            *out = name == "cmd_list_max" ? mx : ge;
            return true;
        }
        if (name != "one_launch_rescues") return false;  // 64-instance slots stepped by a frame wave since the handle was created
        uint32_t v = 0;
        MG_HIP(hipMemcpy(&v, rescues_.p, sizeof v, hipMemcpyDeviceToHost));
        *out = (int64_t)v;
        return true;
    }

   private:
    uint32_t* flag_dev_ = nullptr;  // arm_done_flag
    uint32_t flag_ticket_ = 0;
    bool flag_armed_ = false;
    uint32_t ticket_ = 0;  // one-launch step: number of the step, the value a slot's claim word takes when a wave claims it
    uint32_t epoch_ = 0;  // the one-launch step's descriptor epoch, 1 .. 255 (every step rewrites every descriptor, so the only stale
                          // values a frame workgroup can meet are the previous step's and the 0 of a reset / two-launch step)
    static bool fuse_step() {  // lab build: MEMGYM_MORTAR_FUSE=0 selects the two-launch form for A/B measurements
        static const bool on = lab_int("MEMGYM_MORTAR_FUSE", 1) != 0;
        return on;
    }
    static bool capturing(hipStream_t s) {
        hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
        return hipStreamIsCapturing(s, &st) == hipSuccess && st != hipStreamCaptureStatusNone;
    }
    MortarIO io() {
        MortarIO o;
        o.state = state_.p;
        o.cmds = cmds_.p;
        o.rng = rng_.view();
        o.desc = desc_.p;
        o.vec = vec_;
        o.err = err_.dev;
        o.sets = per_set() ? sets_dev_.p : nullptr;
        o.set_of = per_set() ? set_of_ : nullptr;
        return o;
    }

    // (re)build geometry-dependent constants, atlases and templates
    void rebuild() {
        int radius = 0;
        // (the grid variants accept agent_scale and never read it: GridCharacterController(SCALE, ...), mortar_mayhem_grid.py:249)
        std::vector<Stamp> sprites = build_agent_sprites(P_.variant == V_GRID ? 1.0 * SCALE : agent_scale_, &radius);
        std::vector<Stamp> glyphs = build_glyphs(SCALE);
        P_.tile = (int)(56 * SCALE);
        P_.arena_x0 = SCREEN / 2 - ((P_.tile * P_.N) >> 1);
        P_.radius = radius;
        P_.sprite_dim = sprites[0].w;
        double inv = 1.0 / std::sqrt(2.0);
        P_.v_axis = (1.0 / 1.0) * agent_speed_;
        P_.v_diag = inv * agent_speed_;
        P_.v_axis_i = (int)P_.v_axis;
        P_.v_diag_i = (int)P_.v_diag;
        P_.off_lo = (int)(-8 * SCALE);
        P_.off_hi = (int)(8 * SCALE);

        atlas_.reset(new Atlas());
        for (auto& sp : sprites) atlas_->add_stamp(sp);   // ids 0..7
        for (auto& g : glyphs) atlas_->add_stamp(g);      // ids 8..17
        {   // id 18: the debug view's ring around the target tile (box 2r x 2r, centre (r, r), like the coin)
            const int r = P_.tile / 2;
            Stamp ring_stamp(2 * r, 2 * r);
            circle(ring_stamp, r, r, r, (int)(8 * SCALE), 6);  // palette 6 = (0, 255, 0)
            atlas_->add_stamp(ring_stamp);
        }
        atlas_->set_templates(build_mortar_templates(P_.N, SCALE, SCREEN));
        atlas_->upload();
        P_.glyph_x0 = (int)((SCREEN / 2) - std::floor(88 * SCALE / 2));
        dirty_ = false;
        for (size_t k = 1; k < opt_.size(); ++k) copy_geometry(opt_[k]->P, P_);
        copy_geometry(defaults_, P_);
        sets_dirty_ = true;
    }

    // per-instance option sets
    struct MortarOpt {
        MortarParams P;
        OptListStore st_command_count, st_show_dur, st_show_delay, st_expl_dur, st_expl_delay;
    };
    static std::vector<std::unique_ptr<MortarOpt>> one_set() {
        std::vector<std::unique_ptr<MortarOpt>> v;
        v.emplace_back(new MortarOpt());
        return v;
    }
    // what the shared atlases and templates fix for every set of the handle
    static void copy_geometry(MortarParams& d, const MortarParams& s) {
        d.variant = s.variant; d.N = s.N; d.taskb = s.taskb; d.cmd_cap = s.cmd_cap; d.arena_x0 = s.arena_x0; d.tile = s.tile;
        d.radius = s.radius; d.sprite_dim = s.sprite_dim; d.glyph_x0 = s.glyph_x0; d.v_axis_i = s.v_axis_i; d.v_diag_i = s.v_diag_i;
        d.off_lo = s.off_lo; d.off_hi = s.off_hi; d.v_axis = s.v_axis; d.v_diag = s.v_diag;
    }
    bool per_set() const { return set_of_ != nullptr && opt_.size() > 1; }
    // the sets as the kernels read them, stream-ordered behind what the stream holds (pageable source: staged before the call returns)
    void upload_sets(hipStream_t s) {
        if (!per_set() || !sets_dirty_) return;
        MortarParams fresh = defaults_;  // a set that was never written: the reference's defaults under the handle's geometry (include/memgym.h)
        copy_geometry(fresh, P_);
        std::vector<MortarParams> host(MG_MAX_OPTION_SETS, fresh);
        for (size_t k = 0; k < opt_.size(); ++k) host[k] = opt_[k]->P;
        MG_HIP(hipMemcpyAsync(sets_dev_.p, host.data(), sizeof(MortarParams) * host.size(), hipMemcpyHostToDevice, s));
        MG_HIP(hipStreamSynchronize(s));  // (rare: only after an option of some set changed)
        sets_dirty_ = false;
    }

    void raster_only(void* obs, const uint8_t* only, hipStream_t s) override {
        launch_raster<MortarComposer>(desc_.p, atlas_->dev(), obs, obs_format, n_, s, only);
        MG_HIP(hipGetLastError());
    }
    // (the one-launch step keeps terminal observations itself: the conditions under which step() takes that launch; lab
    // MEMGYM_MORTAR_FINAL_FUSED=0: the generic path of mg_step)
    bool keeps_final_obs(hipStream_t s) override {
        static const bool wanted = lab_int("MEMGYM_MORTAR_FINAL_FUSED", 1) != 0;
        return wanted && obs_format == MG_OBS_U8_XYC && fuse_step() && !per_set() && !capturing(s) && !(flag_armed_ && n_ == 1);
    }

    void raster(void* obs, hipStream_t s) {
        launch_raster<MortarComposer>(desc_.p, atlas_->dev(), obs, obs_format, n_, s);
        MG_HIP(hipGetLastError());
    }

   public:
    void raster_debug(void* frames, hipStream_t s) override {
        if (dirty_) throw std::runtime_error("options that change geometry need a reset before the next render");
        DevArray<MortarDesc> dbg;
        dbg.alloc(n_, false);
        hipLaunchKernelGGL(mortar_debug_desc_kernel, dim3((n_ + 255) / 256), dim3(256), 0, s, P_, n_, io(), dbg.p);
        launch_raster<MortarDebugComposer>(dbg.p, atlas_->dev(), frames, MG_OBS_U8_XYC, n_, s);
        MG_HIP(hipGetLastError());
        MG_HIP(hipStreamSynchronize(s));  // dbg is released on return
    }

   private:

    std::vector<std::unique_ptr<MortarOpt>> opt_;  // [0] = the handle-wide set (P_ below is its parameter block)
    MortarParams& P_;
    MortarParams defaults_;
    const int32_t* set_of_ = nullptr;
    bool sets_dirty_ = true;
    DevArray<MortarParams> sets_dev_;
    int n_;
    std::unique_ptr<Atlas> atlas_;
    double agent_scale_, agent_speed_;
    bool dirty_ = true, seeded_ = false;

   public:
    void on_state_loaded() override { seeded_ = true; }

   private:
    float* vec_ = nullptr;
    DevArray<MortarState> state_;
    DevArray<uint8_t> cmds_;
    DevArray<MortarDesc> desc_, tdesc_;  // tdesc_: terminal-frame descriptors (FINAL form of the one-launch step)
    RngStore rng_;
    ErrorWord err_;
    DevArray<uint32_t> claims_, rescues_;  // one-launch step: one claim word per 64 instances; slots stepped by frame waves
};

Family* make_mortar(int variant, int num_envs) { return new MortarFamily(variant, num_envs); }

}  // namespace mg

#ifdef MG_LAB
extern "C" int mg_lab_set_window_offsets(const long long* off, int n) {
    long long v[16] = {0};
    for (int i = 0; i < n && i < 16; ++i) v[i] = off[i];
    return hipMemcpyToSymbol(HIP_SYMBOL(mg::v1::g_lab_win_off), v, sizeof v) == hipSuccess ? 0 : -1;
}
#endif
