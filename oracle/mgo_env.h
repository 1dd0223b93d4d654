/* oracle/mgo_env.h -- TEST INFRASTRUCTURE (CPU oracle), not product code.
 *
 * Common pieces of the CPU restatement of the reference's single-instance environments:
 * pygame.Rect / Vector2 arithmetic, the three character controllers
 * (memory_gym/character_controller.py:6-283) and the instance vtable used by mgo_api.c.
 * SCALE-parametric on purpose: SCALE=1.0 renders are compared with the reference's own GIF
 * recordings (docs/assets/{emm,ess,emp}_0.gif), SCALE=0.25 is what the HIP path is compared with.
 */
#ifndef MGO_ENV_H
#define MGO_ENV_H
#include <stdio.h>

#include "mgo_raster.h"
#include "mgo_rng.h"

/* ---- pygame.Rect ------------------------------------------------------------------------- */
typedef struct {
    int x, y, w, h;
} mgo_rect;

static inline int mgo_floordiv2(int a) { return a >= 0 ? a / 2 : -((-a + 1) / 2); }
static inline int mgo_rect_cx(const mgo_rect* r) { return r->x + mgo_floordiv2(r->w); }
static inline int mgo_rect_cy(const mgo_rect* r) { return r->y + mgo_floordiv2(r->h); }
/* float -> int, half away from zero: what `Rect.center = (fx, fy)` did in the pygame build that
 * produced the reference's v1.0 GIFs (verified on emm_0.gif). */
static inline int mgo_round_haz(double v) { return v >= 0 ? (int)floor(v + 0.5) : -(int)floor(-v + 0.5); }
static inline void mgo_rect_set_center(mgo_rect* r, double cx, double cy) {
    r->x += mgo_round_haz(cx) - (r->x + (r->w >> 1));
    r->y += mgo_round_haz(cy) - (r->y + (r->h >> 1));
}

/* sin() and cos() evaluated by SEPARATE libm calls.  gcc would otherwise merge them into one sincos(), whose results
 * differ from sin()/cos() by 1 ulp for 6 of the 720 integer-degree angles with this glibc.  The golden fixtures were
 * produced through CPython's math.sin / math.cos (two calls), so that is the definition used here; what the real
 * pygame binary does depends on its compiler and the host libm (sub-ulp ambiguity of the reference itself). */
static __attribute__((noinline)) double mgo_sin(double x) { volatile double v = x; return sin(v); }
static __attribute__((noinline)) double mgo_cos(double x) { volatile double v = x; return cos(v); }

/* ---- pygame.math.Vector2.rotate (src_c/math.c:_vector2_rotate_helper) ----------------------- */
static inline void mgo_vec_rotate(double x, double y, double angle, double* ox, double* oy) {
    const double eps = 1e-6;
    angle = fmod(angle, 360.0);
    if (angle < 0) angle += 360.0;
    if (fmod(angle + eps, 90.0) < 2 * eps) {
        switch ((int)((angle + eps) / 90.0)) {
            case 0:
            case 4: *ox = x; *oy = y; break;
            case 1: *ox = -y; *oy = x; break;
            case 2: *ox = -x; *oy = -y; break;
            default: *ox = y; *oy = -x; break;
        }
    } else {
        double rad = angle * M_PI / 180.0, s = mgo_sin(rad), c = mgo_cos(rad);
        *ox = c * x - s * y;
        *oy = s * x + c * y;
    }
}

/* ---- CharacterController (character_controller.py:6-146) ------------------------------------ */
typedef struct {
    double speed, scale;
    int rotation, radius;
    mgo_surf* sprites[8];
    mgo_rect rect;
    double vx, vy;
    /* GridCharacterController (:155-216) */
    int gx, gy, grid_n;
    double grid_x0, grid_y0, grid_step; /* grid[i][j] = (grid_x0 + grid_step*i, grid_y0 + grid_step*j) */
} mgo_agent;

/* create_character_sprites (character_controller.py:29-75) */
static inline void mgo_agent_init(mgo_agent* a, double speed, double scale, int rotation) {
    for (int k = 0; k < 8; k++) mgo_surf_free(a->sprites[k]);
    memset(a, 0, sizeof(*a));
    a->speed = speed;
    a->scale = scale;
    a->rotation = rotation;
    a->radius = (int)(25 * scale);
    int hands_x = (int)(18 * scale), hand_y = (int)(12 * scale), hand_r = (int)(10 * scale), outline = (int)(3 * scale);
    int extension = 14, dim = a->radius * 2 + hand_r + extension;
    double cx = dim / 2, cy = dim / 2; /* Vector2(rect_dim // 2, rect_dim // 2) */
    double lx = dim / 2 - hands_x, ly = hand_y + extension / 2, rx = dim / 2 + hands_x, ry = ly;
    int k = 0;
    for (int i = 360; i >= 45; i -= 45, k++) { /* reversed(range(45, 405, 45)) */
        mgo_surf* s = mgo_surf_new(dim, dim);
        mgo_fill(s, 255);
        mgo_set_colorkey(s, 255);
        mgo_draw_circle(s, MGO_RGB(250, 204, 153), (int)cx, (int)cy, a->radius, 0);
        double lrx, lry, rrx, rry;
        mgo_vec_rotate(lx - cx, ly - cy, i, &lrx, &lry);
        mgo_vec_rotate(rx - cx, ry - cy, i, &rrx, &rry);
        lrx += cx; lry += cy; rrx += cx; rry += cy;
        mgo_draw_circle(s, MGO_RGB(250, 250, 250), (int)lrx, (int)lry, hand_r, 0);
        mgo_draw_circle(s, MGO_RGB(250, 250, 250), (int)rrx, (int)rry, hand_r, 0);
        mgo_draw_circle(s, MGO_RGB(50, 50, 50), (int)lrx, (int)lry, hand_r, outline);
        mgo_draw_circle(s, MGO_RGB(50, 50, 50), (int)rrx, (int)rry, hand_r, outline);
        a->sprites[k] = s;
    }
    a->rect.x = a->rect.y = 0;
    a->rect.w = a->rect.h = dim;
    mgo_rect_set_center(&a->rect, 0, 0);
}
static inline void mgo_agent_free(mgo_agent* a) {
    for (int k = 0; k < 8; k++) {
        mgo_surf_free(a->sprites[k]);
        a->sprites[k] = NULL;
    }
}

/* velocity/rotation part shared by CharacterController.step (:99-126) and ScreenWrap (:237-263) */
static inline void mgo_agent_velocity(mgo_agent* a, const int action[2], int truncate) {
    double vx = 0, vy = 0;
    if (action[0] == 1) { a->rotation = 90; vx = -1; }
    if (action[0] == 2) { a->rotation = 270; vx = 1; }
    if (action[1] == 1) { a->rotation = 0; vy = -1; }
    if (action[1] == 2) { a->rotation = 180; vy = 1; }
    if (vx < 0 && vy < 0) a->rotation = 45;
    if (vx < 0 && vy > 0) a->rotation = 135;
    if (vx > 0 && vy < 0) a->rotation = 315;
    if (vx > 0 && vy > 0) a->rotation = 225;
    double len = sqrt(vx * vx + vy * vy);
    if (len != 0.0) {
        vx = vx / len * a->speed;
        vy = vy / len * a->speed;
        if (truncate) { /* Vector2(int(velocity.x), int(velocity.y)) (:126) */
            vx = (double)(int)vx;
            vy = (double)(int)vy;
        }
    }
    a->vx = vx;
    a->vy = vy;
}

/* CharacterController.step (:89-146); boundary may be NULL */
static inline void mgo_agent_step(mgo_agent* a, const int action[2], const mgo_rect* b) {
    mgo_agent_velocity(a, action, 1);
    mgo_rect_set_center(&a->rect, mgo_rect_cx(&a->rect) + a->vx, mgo_rect_cy(&a->rect) + a->vy);
    if (b) {
        int x = mgo_rect_cx(&a->rect), y = mgo_rect_cy(&a->rect);
        if (x > b->x + b->w - a->radius) x = b->x + b->w - a->radius;
        if (x < b->x + a->radius) x = b->x + a->radius;
        if (y > b->y + b->h - a->radius) y = b->y + b->h - a->radius;
        if (y < b->y + a->radius) y = b->y + a->radius;
        mgo_rect_set_center(&a->rect, x, y);
    }
}

/* ScreenWrapCharacterController.step (:226-283): no int() on the velocity, Rect rounds */
static inline void mgo_agent_step_wrap(mgo_agent* a, const int action[2], const mgo_rect* b) {
    mgo_agent_velocity(a, action, 0);
    mgo_rect_set_center(&a->rect, mgo_rect_cx(&a->rect) + a->vx, mgo_rect_cy(&a->rect) + a->vy);
    const double offset = 0.5;
    if (b) {
        double x = mgo_rect_cx(&a->rect), y = mgo_rect_cy(&a->rect);
        double right = b->x + b->w, bottom = b->y + b->h, left = b->x, top = b->y;
        if (x > right + a->radius * offset) x = left - a->radius * offset;
        if (x < left - a->radius * offset) x = right + a->radius * offset;
        if (y > bottom + a->radius * offset) y = top - a->radius * offset;
        if (y < top - a->radius * offset) y = bottom + a->radius * offset;
        mgo_rect_set_center(&a->rect, x, y);
    }
}

/* GridCharacterController.step (:177-210) */
static inline void mgo_agent_step_grid(mgo_agent* a, int action) {
    if (action == 1) a->rotation = (a->rotation + 90) % 360;
    if (action == 2) a->rotation = ((a->rotation - 90) % 360 + 360) % 360;
    int face = a->rotation / 90; /* 0 N, 1 W, 2 S, 3 E */
    if (action == 3) {
        int x = a->gx, y = a->gy;
        if (face == 0) { if (y > 0) y -= 1; }
        else if (face == 3) { if (x < a->grid_n - 1) x += 1; }
        else if (face == 2) { if (y < a->grid_n - 1) y += 1; }
        else if (face == 1) { if (x > 0) x -= 1; }
        a->gx = x;
        a->gy = y;
        mgo_rect_set_center(&a->rect, a->grid_x0 + a->grid_step * x, a->grid_y0 + a->grid_step * y);
    }
}

/* ---- instance vtable ------------------------------------------------------------------------ */
struct mgo_env;
typedef struct mgo_vtbl {
    const char* id;
    int discrete; /* 1: Discrete(4) action, 0: MultiDiscrete([3,3]) */
    int gt_dim;
    int (*set_option)(struct mgo_env*, const char* key, const double* v, int n);
    void (*reset)(struct mgo_env*);
    void (*step)(struct mgo_env*, const int action[2]);
    double (*get)(struct mgo_env*, const char* field, int* ok);
    int (*get_list)(struct mgo_env*, const char* name, double* out, int cap);
    void (*destroy)(struct mgo_env*);
    /* render("debug_rgb_array"): _build_debug_surface before its final transform.scale; dst is screen_dim x screen_dim */
    void (*debug)(struct mgo_env*, mgo_surf* dst);
    /* Test hook (tests/test_oracle_old_gif_replay.py): put the instance into a given SCENE -- agent position / sprite, tile
     * states, cross, coins, exit, spotlight discs ... (family-specific vector, see each file) -- and draw the frame with the
     * family's own _draw_surfaces code.  Used to replay recordings of OTHER revisions of the reference frame by frame:
     * their dynamics differ, the drawing of a given scene does not.  NULL: not offered.  Requires one reset. */
    int (*scene)(struct mgo_env*, const double* v, int n);
    /* Test hook (tests/test_gpu_full_batch.py, policy axis): a COMPETENT action for the instance's current state -- follow the
     * command list / the path / go for the coins and the exit -- so that lock-step runs reach the states only a trained agent
     * sees (long command lists, appended path segments, opened exits).  Deterministic, reads nothing but the instance's state,
     * draws nothing from its stream.  Not part of the reference: it is the test's policy, computed where the state is. */
    void (*expert)(struct mgo_env*, int action[2]);
} mgo_vtbl;

typedef struct mgo_env {
    const mgo_vtbl* vt;
    double scale;
    int screen_dim;
    mgo_surf* screen;
    mgo_rng rng;
    int seeded;
    /* outputs of the last call */
    double reward;
    int done;
    double gt[4];
    /* episode accumulators: Python `sum(self.episode_rewards)` is a left-to-right double sum */
    double ep_sum;
    int ep_len;
    void* impl;
    char err[128];
} mgo_env;

static inline int mgo_opt_list(double* dst, int* n_dst, int cap, const double* v, int n) {
    if (n < 1 || n > cap) return -1;
    for (int i = 0; i < n; i++) dst[i] = v[i];
    *n_dst = n;
    return 0;
}

#endif
