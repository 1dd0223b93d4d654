/* include/memgym.h -- C ABI of libmemgym_hip.so (MI355X / gfx950 batched Memory Gym hot path).
 *
 * The reference (MarcoMeter/endless-memory-gym) has no FFI: the boundary trainers program against is
 * the Gymnasium Env protocol of its Python classes.  Each entry point below replaces the per-instance
 * Python method named next to it, batched over `num_envs` independent environment instances whose
 * state lives in HBM.  All `*_dev` pointers are device pointers owned by the caller (e.g. torch
 * tensors); `stream` is a hipStream_t (void* so that this header needs no HIP include); every call
 * only enqueues work on that stream (no host synchronisation) unless stated otherwise.
 *
 * Return value: 0 on success, negative on error (`mg_last_error()` holds the message).
 * A handle binds one device; it is not thread-safe, distinct handles are independent.
 *
 * Observation layout (identical to pygame.surfarray.array3d in the reference, e.g.
 * memory_gym/mortar_mayhem_grid.py:276,373): uint8 [num_envs][84 (x)][84 (y)][3 (rgb)].
 */
#ifndef MEMGYM_H
#define MEMGYM_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct mg_env mg_env;

#define MG_OBS_DIM 84
#define MG_OBS_BYTES (84 * 84 * 3) /* 21,168 */
#define MG_INFO_SLOTS 8

/* Per-env end-of-episode record, written by mg_step when an env reports done (the reference's
 * `info` dict returned on the terminal step, e.g. mortar_mayhem_grid.py:356-362).  Device SoA,
 * all arrays [num_envs]; any pointer may be NULL.  `aux[k]` meaning per env id: see mg_info_name(). */
typedef struct mg_info_buffers {
    /* Versioning: set to sizeof(mg_info_buffers) of the header the caller was built against.  The library reads that many
     * bytes (never more than it knows) and treats the fields a shorter, older struct lacks as NULL; 0, a size that is not a
     * multiple of the pointer size, or one larger than this build's struct plus 16 pointers (e.g. the device pointer a
     * caller built against the round-2 header -- which had no such member -- passes in this place) is refused (-1). */
    size_t struct_size;
    double* ep_reward_dev;         /* "reward": Python sum() of the step rewards, in double        */
    int32_t* ep_length_dev;        /* "length"                                                      */
    float* aux_dev[MG_INFO_SLOTS]; /* "success", "commands_completed", "num_fails", ... per env id */
    /* Optional (gymnasium 0.29 VectorEnv convention, info["final_observation"]): with autoreset != 0 and this pointer
     * set, mg_step also writes the TERMINAL observation of every instance that finished in this call to row i of this
     * buffer (same format and shape as obs_dev; rows of other instances are left untouched) while obs_dev row i holds
     * the first observation of the new episode.  uint8 observations, one option set: the step's own launches draw both frames (round 6;
     * Endless-MysteryPath: the terminal frames by one sparse raster launch behind them) -- 0-4 % of a step for eight env ids, 11-12 % for
     * SearingSpotlights-v0 and Endless-MysteryPath-v0.  Otherwise (float formats, per-instance option sets, the mortar family under graph
     * capture): a step without auto-reset, the terminal rows copied, a masked reset whose frames a sparse raster launch draws. */
    void* final_obs_dev;
    /* Optional: the step reward of every instance as the reference computes it -- a Python float, i.e. a double
     * (e.g. mortar_mayhem_grid.py:288-352) -- next to reward_dev's float32 rounding of it; [num_envs]. */
    double* reward64_dev;
    /* Optional (round 5): info["ground_truth"] as the reference returns it -- a float64 array (endless_mortar_mayhem.py:259,358,362,
     * endless_searing_spotlights.py:407,496, endless_mystery_path.py:92-97) -- [num_envs][mg_gt_dim]; gt_dev is its float32
     * rounding (0.6 -> 0.60000002).  Costs one small launch behind the step's; see also mg_ground_truth64. */
    double* gt64_dev;
    /* Optional (round 6): uint8 [num_envs], written by every mg_step: 1 where the instance's episode was ENDED IN THIS STEP BECAUSE IT
     * REACHED A CAPACITY OF THIS BUILD (done_dev is 1 for it as well), else 0.  The reference's lists grow without limit -- path
     * segments (pygame_assets.py:559, called from endless_mystery_path.py:333-335), fall-off cells (:385-393), the command list
     * (endless_mortar_mayhem.py:311-333), live spotlights (endless_searing_spotlights.py:191) -- here they hold 128 segments, 128
     * cells, 512 commands (mg_set_capacity raises the first and the third) and 16 spotlights per instance.  An instance that would need one more ends its episode like a truncation
     * (a reset follows under autoreset); the sticky error bit (mg_poll_errors: 4, 8, 32, 1) is raised as before, so a caller that
     * ignores this array still hears about it.  With it a trainer treats the instance as truncated and goes on with the batch. */
    uint8_t* capacity_dev;
} mg_info_buffers;

/* gymnasium.make(id) + Env.__init__  (memory_gym/__init__.py:13-61; e.g. mortar_mayhem_grid.py:55-90).
 * env_id: one of the reference's registered ids.  Allocates the SoA state for num_envs instances on
 * `device` and builds the stamp atlases / background templates.  Synchronous. */
int mg_create(const char* env_id, int32_t num_envs, int device, mg_env** out);
void mg_destroy(mg_env* env);
const char* mg_last_error(void);

/* Static properties (action_space / observation_space / ground_truth_space of the reference classes):
 * mg_action_dim: 1 = Discrete(4) (mortar_mayhem_grid.py:82, endless_mystery_path.py:83),
 *                2 = MultiDiscrete([3,3]) (e.g. mortar_mayhem.py:83).
 * mg_gt_dim:     0, or the size of info["ground_truth"] (endless_mortar_mayhem.py:91-96 -> 2,
 *                endless_mystery_path.py:92-97 -> 3, endless_searing_spotlights.py:112-117 -> 4). */
int32_t mg_num_envs(const mg_env* env);
int32_t mg_action_dim(const mg_env* env);
int32_t mg_gt_dim(const mg_env* env);
/* MortarMayhemB-Grid-v0 / MortarMayhemB-v0 return a Dict observation (mortar_mayhem_b_grid.py:83-96):
 * "visual_observation" is the usual frame, "vector_observation" the one-hot command list float32[20 * 9]
 * (_encode_commands_one_hot :100-129).  mg_vec_dim: 180 for those ids, else 0.  mg_bind_vector_obs registers the
 * caller's device buffer float32 [num_envs][mg_vec_dim]; the library writes row i whenever instance i is reset
 * (mg_reset and same-step auto-reset) -- the vector is constant within an episode.  NULL unbinds. */
int32_t mg_vec_dim(const mg_env* env);
int mg_bind_vector_obs(mg_env* env, float* vec_dev);
/* name of aux slot k of mg_info_buffers for this env id, or NULL */
const char* mg_info_name(const mg_env* env, int k);

/* One key of the reference's reset `options` dict (process_reset_params, e.g.
 * mortar_mayhem_grid.py:36-53).  `values`/`n`: scalars are n == 1, "sample one per episode" lists
 * have n >= 1.  Unknown key -> error -2 (the Python layer turns it into the reference's
 * AssertionError text); unsupported value -> error -3.  Takes effect at the next reset (also
 * auto-resets), for every instance of the handle that runs under option set 0 (all of them unless
 * mg_bind_option_sets says otherwise, below). */
int mg_set_option(mg_env* env, const char* key, const double* values, int n);

/* Per-instance reset options.  In the reference reset(seed, options) belongs to ONE environment instance (e.g.
 * mortar_mayhem_grid.py:213-236): a pool of workers runs different curricula side by side.  Here a handle holds up to
 * MG_MAX_OPTION_SETS option sets; set 0 is the one mg_set_option writes.  mg_set_option_set writes one key of set
 * `set_id` (a set that has never been written starts from the reference's defaults); mg_bind_option_sets registers the
 * caller's device array int32 [num_envs]: instance i runs under set set_of_dev[i] -- for its resets, auto-resets AND its steps
 * (rewards, limits and display options are read at step time, like the reference's self.reset_params) -- read by every
 * following mg_reset / mg_step; the caller changes an instance's entry when that instance is reset with other options
 * (stream-ordered, e.g. a tensor assignment on the launch stream).  NULL unbinds (every instance under set 0).
 * Options that change the geometry shared by the handle's instances (the *_scale, agent_speed, arena_size, radius and dim /
 * interval options that rebuild atlases, templates or tables) can only be set in set 0, i.e. for all instances: -3 otherwise.
 * A geometry option is accepted in a set > 0 when it says what the handle's geometry already is.
 * Every family (all ten ids).  With more than one set in use the per-set kernels read their parameters from memory and a handle
 * steps in its plain arrangement: a mortar-family handle with two launches, a SearingSpotlights handle resets inside its step
 * kernel, an Endless-MysteryPath handle serves its queue in a launch of its own (and generates every segment when it is due). */
#define MG_MAX_OPTION_SETS 8
int mg_set_option_set(mg_env* env, int set_id, const char* key, const double* values, int n);
int mg_bind_option_sets(mg_env* env, const int32_t* set_of_dev);

/* Capacities of the per-instance lists that the reference grows without limit.  mg_set_capacity(env, what, value), before the handle's
 * first mg_reset (it re-allocates the list's array; mg_state_size changes with it and a checkpoint only loads into a handle of the same
 * capacity); mg_capacity returns the value in force, -1 for a name the env id does not have.
 *   "path_segments"  Endless-MysteryPath-v0: segments of an episode's path (EndlessMysteryPath.add_path_segment, pygame_assets.py:559,
 *                    called from endless_mystery_path.py:333-335 whenever the agent enters the last but one); default 128 = 1,024 tiles,
 *                    4 .. 32,767, 52 bytes per segment and instance.  (Not a ring: an agent that falls off is put back to the START of
 *                    its path, endless_mystery_path.py:316-322, and walks every segment again.)
 *   "commands"       Endless-MortarMayhem-v0: entries of the command list (endless_mortar_mayhem.py:311-333); default 512, 4 .. 32,768,
 *                    one byte per entry and instance.
 *   "fall_off_cells" Endless-MysteryPath-v0 (read only): 128 distinct cells an episode may fall off at (:385-393).
 * An instance that would need one more entry than the capacity ends its episode in that step: done_dev, mg_info_buffers.capacity_dev and
 * the sticky error bit (mg_poll_errors) say so.  16 live spotlights per instance (the spotlight family) is fixed. */
int mg_set_capacity(mg_env* env, const char* what, int64_t value);
int64_t mg_capacity(mg_env* env, const char* what);

/* Observation format written to obs_dev by mg_reset / mg_step (default MG_OBS_U8_XYC).
 *   MG_OBS_U8_XYC   uint8   [num_envs][84 x][84 y][3]  -- the reference's observation: pygame.surfarray.array3d order
 *                                                         (e.g. mortar_mayhem_grid.py:277,372), 21,168 B per instance
 *   MG_OBS_F32_CYX  float32 [num_envs][3][84 y][84 x]  -- value / 255 in image (CHW) order, the tensor a trainer builds
 *                                                         from the observation before its CNN; 84,672 B per instance
 *   MG_OBS_F16_CYX  float16 [num_envs][3][84 y][84 x]  -- the float32 quotient rounded to nearest-even half
 *   MG_OBS_BF16_CYX bfloat16 [num_envs][3][84 y][84 x] -- the float32 quotient rounded to nearest-even bfloat16
 * The conversion is fused into the raster kernel's stream-out (no second pass over HBM).  mg_obs_bytes returns the
 * bytes per instance of the current format. */
#define MG_OBS_U8_XYC 0
#define MG_OBS_F32_CYX 1
#define MG_OBS_F16_CYX 2
#define MG_OBS_BF16_CYX 3
int mg_set_obs_format(mg_env* env, int format);
size_t mg_obs_bytes(const mg_env* env);

/* Env.reset(seed, options) (e.g. mortar_mayhem_grid.py:213-278) for all instances, or for those with
 * mask_dev[i] != 0.  seeds_dev: int64 [num_envs] -> instance i is re-seeded exactly like
 * gymnasium's reset(seed=s): Generator(PCG64(SeedSequence(s))); NULL -> keep each instance's stream
 * (reset(seed=None)).  Writes the first observation of every reset instance to obs_dev (others are
 * left untouched) and, if gt_dev != NULL and mg_gt_dim() > 0, info["ground_truth"] as float32
 * [num_envs][gt_dim]. */
int mg_reset(mg_env* env, const int64_t* seeds_dev, const uint8_t* mask_dev, void* obs_dev, float* gt_dev,
             void* stream);

/* Rasterise the CURRENT frame of every instance again into obs_dev (the frame the last mg_reset / mg_step produced; the
 * reference's _draw_surfaces + surfarray.array3d without stepping, e.g. mortar_mayhem_grid.py:92-102,373).  No state
 * changes; instances that the last call left untouched (a masked mg_reset) are skipped here as well.  Uses: observations into a second buffer, a changed observation format, and the placement probe of the
 * Python mirror (the store stream is 8-13 % faster into some allocations than into others: profiles/r01l_placement.md). */
int mg_render(mg_env* env, void* obs_dev, void* stream);

/* Env.render() with render_mode "debug_rgb_array" (e.g. mortar_mayhem_grid.py:403-405 -> _build_debug_surface :104-135)
 * for every instance: the ground-truth view (target tile ring / whole path and walls / everything the spotlight layer
 * hides drawn over it), stretched to 336 x 336 like pygame.transform.scale, in IMAGE order:
 * rgb_dev uint8 [num_envs][336 y][336 x][3].  Changes nothing the observations, rewards or RNG streams depend on; the
 * only state it touches is the mortar family's counter of entries popped from the reference's CLONE of the command
 * visualisation list (mortar_mayhem_grid.py:122: every debug render pops one), so that zero, one or several renders
 * between steps show what the reference shows.  Synchronous (allocates and frees a scratch buffer); not a hot path. */
int mg_render_debug(mg_env* env, uint8_t* rgb_dev, void* stream);

/* Env.step(action) (e.g. mortar_mayhem_grid.py:280-375) for all instances.
 * actions_dev: int32 [num_envs] (Discrete) or [num_envs][2] (MultiDiscrete).
 * reward_dev: float32 [num_envs] (the reference's Python float, rounded once to float32);
 * done_dev: uint8 [num_envs] (`truncation` is always False in the reference).
 * autoreset != 0: an instance that reports done is reset in the same call with seed=None (its RNG
 * stream continues, exactly what `env.reset()` right after a terminal step does) and obs_dev/gt_dev
 * hold the first observation of the new episode; the finished episode's info is in `info`. */
int mg_step(mg_env* env, const int32_t* actions_dev, void* obs_dev, float* reward_dev, uint8_t* done_dev,
            float* gt_dev, const mg_info_buffers* info, int autoreset, void* stream);

/* The float64 ground truth of every instance from the CURRENT state (what the last mg_reset / mg_step left), [num_envs][mg_gt_dim]
 * on the device: the doubles the reference puts into info["ground_truth"] after reset() as well as after step().  Enqueued on
 * `stream`; a no-op for env ids without ground truth. */
int mg_ground_truth64(mg_env* env, double* gt64_dev, void* stream);

/* The single-instance fast path (BASELINE config C1: gym.make(id), one instance, numpy in / numpy out -- the reference's own loop,
 * /root/reference/bench.py:12-30).  For a handle with num_envs == 1, mg_single_open allocates every per-call buffer in pinned,
 * device-mapped HOST memory and returns the host addresses: the kernels read the action from it and store observation, reward,
 * done, ground truth and the end-of-episode record straight into it (21 KB over PCIe), so that one call = store the action,
 * enqueue the step's launches, wait for the stream -- no copy operation, no allocation, no indexing kernel.
 *   mg_single_reset(env, seed, has_seed, stream)   Env.reset(seed) + wait; options through mg_set_option as ever
 *   mg_single_step(env, a0, a1, stream)            Env.step(action) WITHOUT auto-reset (the caller resets, like the reference's
 *                                                  loop) + wait; results are in the buffers of mg_single_io when it returns.
 *                                                  Returns 0, a negative error code, or -- POSITIVE -- the device error bits as they
 *                                                  stand (mg_peek_errors; sticky until mg_poll_errors).  The wait polls a word in the
 *                                                  pinned block that a stream memory operation writes behind the step's launches
 *                                                  (no hipStreamSynchronize; round 6)
 * `obs` has the handle's observation format; `gt` holds mg_gt_dim doubles; `vec` the MortarMayhemB vector observation or NULL.
 * The buffers live as long as the handle.  Same results as the batched path with one instance (tests/test_gpu_single_instance.py). */
typedef struct mg_single_io {
    size_t struct_size;   /* in: sizeof(mg_single_io) of the caller's header */
    void* obs;
    float* vec;
    double* reward;       /* the step reward as the reference's Python float */
    uint8_t* done;
    double* gt;
    double* ep_reward;    /* valid when *done: info["reward"], info["length"], aux[k] (mg_info_name) */
    int32_t* ep_length;
    float* aux[MG_INFO_SLOTS];
} mg_single_io;
int mg_single_open(mg_env* env, mg_single_io* io);
int mg_single_reset(mg_env* env, int64_t seed, int has_seed, void* stream);
int mg_single_step(mg_env* env, int32_t a0, int32_t a1, void* stream);

/* Checkpoint hooks (the reference cannot serialise an env; SoA state makes it free).  Synchronous.
 * mg_state_size: bytes needed.  The blob starts with a 64-byte header {magic "MGSTATE1", MG_STATE_VERSION, num_envs,
 * payload bytes, FNV-1a of the env id}; the layout behind it is private to one MG_STATE_VERSION.  mg_set_state refuses
 * (-1, message in mg_last_error) a blob whose magic, version, env id, num_envs or payload size differ from the handle's
 * instead of mis-assigning it. */
#define MG_STATE_VERSION 7u
size_t mg_state_size(const mg_env* env);
int mg_get_state(mg_env* env, void* host_buf, size_t size);
int mg_set_state(mg_env* env, const void* host_buf, size_t size);

/* Measurement hooks (bench.py's roofline leg): mg_set_profiling(env, N) makes every N-th mg_step (N = 1: every step,
 * 0: off) bracket its kernels with hipEvents recorded on the launch stream (a bracketed step costs ~15 us of stream
 * time, hence the sampling).  mg_get_profile(kind) synchronises, returns the summed elapsed milliseconds and the
 * number of bracketed launches since the last call, and clears.  kind 0 = logic kernel (one lane per instance),
 * kind 1 = raster kernel (the HBM-write-bound one). */
int mg_set_profiling(mg_env* env, int on);
int mg_get_profile(mg_env* env, int kind, double* total_ms, int64_t* launches);

/* Device-side error bits raised by the kernels since the last call (the reference's only failure paths are Python
 * exceptions, e.g. pygame_assets.py:723-724 "No valid path found"); 0 = none.  Synchronous.
 *   1  spotlight slots exhausted (more than 16 live spotlights)      2  path generation found no valid path
 *   4  endless path longer than 128 segments                         8  more than 128 distinct fall-off cells
 *  16  past-path window wider than 16 columns
 *  32  Endless Mortar Mayhem: the command list reached its capacity of 512 entries (the reference's list is unbounded,
 *      endless_mortar_mayhem.py:316-318); the episode of that instance was ended
 *  64  a deferred-reset queue was found over-full (an earlier fused launch failed before draining it); the excess
 *      entries were dropped
 * 256  SearingSpotlights-v0: use_exit = False at the reset of an instance that has never had an exit (the reference raises
 *      AttributeError there, searing_spotlights.py:431-435; with an earlier exit it keeps drawing that one, and so does this
 *      library); no exit was drawn
 * 128  reserved (rounds <= 3: a frame workgroup of the mortar family's one-launch step gave up waiting for its descriptor.
 *      Since round 4 that launch cannot time out -- a frame wave that waits too long steps the instances itself,
 *      mg_mortar.hip mortar_step_raster_kernel -- and the bit is never raised) */
int mg_poll_errors(mg_env* env, int* flags);

/* The same bits as they stand right now: no synchronisation, nothing cleared.  The error word lives in pinned host
 * memory that the kernels update with system-scope atomics, so a trainer can look after every mg_step for free and
 * sees a bit at the latest one step after the kernel that raised it finished.  (SearingSpotlights option sets that
 * overflow the 16 slots in every long enough episode are refused by mg_reset up front; the bit covers the rest.) */
int mg_peek_errors(mg_env* env, int* flags);

/* Observation-buffer allocator (optional; any device-accessible obs_dev works with mg_reset / mg_step).  The reference
 * has no counterpart: its observation is a host numpy array returned by pygame.surfarray.array3d
 * (mortar_mayhem_grid.py:277,372); here the batch of observations is one device buffer that the raster kernel streams
 * into, and WHERE that buffer lies in HBM decides 12-15 % of the kernel's speed: the MI355X's memory falls into three
 * zones of ~96 GB, a store stream confined to one zone runs at 5.1-5.5 TB/s, one split over two zones at 6.2-6.4 TB/s
 * (profiles/r02_zones.md), and an ordinary allocation lies in one zone.  mg_obs_alloc builds the buffer with the HIP
 * virtual-memory API from 304-MiB physical pieces, each classified against the first one with a 0.1-ms two-window store
 * probe, half of them from the first piece's zone and half from elsewhere, mapped alternately into one contiguous
 * virtual range (rounded up to whole pieces).  Pieces it does not need and spacer allocations of 8 GiB that are never
 * mapped or written keep the driver's allocator moving during the search and are released before the call returns (at
 * most `search_budget_bytes` in total; MG_OBS_SEARCH_DEFAULT = half of the free memory, at most 128 GiB: that much
 * VRAM is transiently unavailable to other processes on the GPU; 0 = no search.  A pristine VRAM can hand out 100-130 GiB
 * of ONE zone in a row: with the default budget such a process gets the plain allocation, info.zones == 1).  Pieces are
 * classified by thresholds that start from the figures measured on the development boxes and turn RELATIVE (geometric
 * mean of the slowest and the fastest pair seen, -/+ 3 %) once both kinds of pair have been observed.  Buffers of 304 MiB or
 * less, and runtimes without hipMemCreate, get a plain hipMalloc.  The range is accessible from the owning device and from
 * every device that reports peer access to it.  Virtual address ranges are never returned to the runtime (a reused range can
 * keep stale translations on ROCm 7.2): a process that allocates and frees buffers for ever uses address space, not memory.
 * Synchronous; 2-10 ms typically.  The search is bounded in time (mg_obs_set_search_ms, default 1,500 ms): the clock is read
 * after every handle the walk creates and the walk ends when the time spent plus the projected cost of handing everything
 * back would pass the bound, so that the call as a whole stays within 1.5 x the bound plus the assembly of the buffer
 * (info.search_ms; a search that runs out of time ends with what it has: two zones if it found them, else the plain
 * allocation).  The library reads no environment variable for any of this (the Python mirror forwards MEMGYM_OBS_SEARCH_MS /
 * MEMGYM_OBS_SEARCH_GB as arguments).  mg_obs_free releases a buffer obtained here (after synchronising the device); the
 * pieces go to a pool of at most ten for the next buffer. */
typedef struct mg_obs_alloc_info {
    int zones;               /* 2, 3 = pieces from that many zones; 1 = no second zone within the budget (plain
                              * allocation); 0 = plain allocation without a search (small buffer, no room)          */
    int pieces;              /* physical pieces mapped                                                                */
    size_t piece_bytes;
    size_t searched_bytes;   /* bytes of unused pieces and spacers walked over (transient, released)                  */
    double probe_same_tbps;  /* store probe, first piece + a piece classified "same zone" (largest seen; ~5.1)        */
    double probe_cross_tbps; /* ... + a piece classified "other zone" (smallest seen; ~6.3); 0 = none seen            */
    double search_ms;
} mg_obs_alloc_info;
#define MG_OBS_SEARCH_DEFAULT ((size_t)-1)
int mg_obs_alloc(int device, size_t bytes, size_t search_budget_bytes, void** out_dev, mg_obs_alloc_info* info);
/* The same for a buffer whose observations are `frame_bytes` each (= mg_obs_bytes of the handle: 21,168 for MG_OBS_U8_XYC -- what
 * mg_obs_alloc assumes --, 42,336 for the 16-bit formats, 84,672 for MG_OBS_F32_CYX).  A raster launch writes at fronts that are one
 * WINDOW = 14,336 observations apart, and the split that is fast is the one of the concurrently written fronts: pieces are dealt to the
 * zones window by window (neighbouring windows start in different zones, a window's pieces alternate).  For windows of about one
 * piece that is mg_obs_alloc's alternating order; for the 16-bit formats (1.9 pieces per window) the alternating order leaves all
 * fronts in one zone at any time (round 6: bfloat16 observations at 65,536 instances 0.72 -> 0.83 of peak). */
int mg_obs_alloc_for(int device, size_t bytes, size_t frame_bytes, size_t search_budget_bytes, void** out_dev, mg_obs_alloc_info* info);
/* The order mg_obs_alloc_for would use, without allocating anything (no GPU needed): returns the number of pieces k of a buffer of
 * `bytes` and, for the first max_pieces of them, the zone (0 .. zones - 1; zones = 2 or 3) each is taken from; *lead_bytes = where in the
 * assembled range the buffer starts (it sits in the middle of the k pieces).  tests/test_obs_plan.py walks the fronts of a launch over it. */
int mg_obs_plan(size_t bytes, size_t frame_bytes, int zones, size_t* piece_bytes, size_t* lead_bytes, int* zone_of_piece, int max_pieces);
int mg_obs_free(void* obs_dev);
int mg_obs_set_search_ms(double ms);  /* process-wide; >= 0 */
/* Measurement hook (bench.py's per-box control: roofline.box_probe_GBps).  DESTROYS THE BUFFER'S CONTENTS: the n_frames x 21,168
 * bytes at obs_dev are overwritten with zeros -- a caller that probes a live observation buffer steps or renders (mg_render) afterwards.
 * ONE launch of a pure store stream on `stream`; the caller times it with events.  pattern 0 = linear fill, one 16-byte store per
 * thread (the memory system's ceiling for stores at this size and placement); pattern 1 = the raster's store shape without compose
 * work (persistent 256-lane workgroups writing whole frames; the raster's grid and residency): a CONTROL for the raster, not a ceiling
 * (a raster with compose work can beat it); patterns 2 / 3 = other frame-shaped streams measured in round 5 (pairs of adjacent
 * vectors per lane; pairs + each wave a contiguous quarter of the frame: tools/store_shapes.py, profiles/r05_store_shapes.md); patterns
 * 4 / 5 (round 6, profiles/r06_store_counters.md): the frame walk with line-aligned ownership (every 128-byte line written by one
 * workgroup); one frame per workgroup without the persistent loop; 6 / 7: the frame walk with neighbouring frames on the same XCD;
 * eight consecutive frames per workgroup as 64-byte-aligned spans. */
int mg_store_probe(void* obs_dev, size_t n_frames, int pattern, void* stream);
/* Test hook: live buffers, pooled spare pieces, bytes of virtual address space reserved so far. */
int mg_obs_debug_stats(size_t* live_buffers, size_t* pooled_pieces, size_t* reserved_va_bytes);

/* Multi-GPU helper for caller-owned observation memory on ANOTHER GPU of the node (BASELINE config 5: rank r's raster
 * kernels store their frames straight into rank 0's buffer over xGMI; memory_gym_amd/dist.py PeerObsBuffer): checks
 * hipDeviceCanAccessPeer(device, peer_device) and switches peer access on for `device`.  0 = the kernels of a handle on
 * `device` may be given pointers into `peer_device`'s memory; -1 = no peer access (fall back to a gather). */
int mg_enable_peer_access(int device, int peer_device);

/* Test / telemetry hook: named counters of a handle.  "one_launch_rescues" (mortar family): 64-instance slots of the one-launch
 * step that a frame wave stepped itself because the step workgroup's wave had not claimed them in time (0 on a GPU that
 * dispatches the step workgroups first; tests/test_gpu_one_launch.py forces the other order).  "path_gen_ticks" / "path_gen_paths"
 * (finite Mystery Path: wave-ticks and paths of the A* generation inside the step's launches).  "emp_ahead_records"
 * (Endless-MysteryPath: first segments of NEXT episodes generated ahead of time, which a finishing instance's own step turns into its
 * reset -- EndlessMysteryPathEnv.reset, endless_mystery_path.py:195-280, without a queue entry); "emp_own_resets" (such resets; counted
 * by the lab build only).  Unknown name: -1.  Synchronous. */
int mg_debug_counter(mg_env* env, const char* name, int64_t* value);

/* Test hook: copy the numpy-compatible PCG64 words of instance i to host:
 * out[6] = {state_hi, state_lo, inc_hi, inc_lo, has_uint32, uinteger}.  Synchronous. */
int mg_debug_rng(mg_env* env, int32_t i, uint64_t* out);

#ifdef __cplusplus
}
#endif
#endif
