/* oracle/mgo_spot.c -- TEST INFRASTRUCTURE (CPU oracle), not product code.
 *
 * Restatement of the Searing Spotlights family:
 *   Endless-SearingSpotlights-v0  memory_gym/endless_searing_spotlights.py  reset :294-407  step :409-506
 *                                 helpers _step_spotlight_task :179-231, _spawn_coin :256-269, _step_coin_task :271-292
 *   SearingSpotlights-v0          memory_gym/searing_spotlights.py          reset :332-450  step :452-562
 *                                 helpers :137-330
 * plus GridPositionSampler :7-59, Spotlight :61-131, Coin :133-167, Exit :169-220 and
 * get_tiled_background_surface :222-239 of memory_gym/pygame_assets.py.
 * Pinned by tests/golden/logic_{Endless_,}SearingSpotlights_v0.npz (logic), docs/assets/ess_0.gif (pixels of the endless
 * variant, SCALE 1.0, all 692 frames) and, for the finite variant, docs/assets/searing_spotlights_0.gif + _gt.gif (an older
 * revision's recording, every one of its 118 frames in both views: chessboards, dim ramp, filled discs, coin, the Exit's
 * rounded rectangle CLOSED (frames 0-52) and OPEN (frames 53-117, (48,141,70)): tests/test_oracle_old_gif_replay.py,
 * tests/test_oracle_old_gifs.py).
 * hide_chessboard / black_background repaint the two background surfaces an environment object keeps for its lifetime
 * (searing_spotlights.py:349-351, 234-235, 420-421; endless :313-315, 223-224, 376-377), so they stick to the INSTANCE
 * across episodes and option changes; black_background also gives every spotlight spawned under it a white 1-px border
 * (pygame_assets.py:62, 112-113).  Logic under these options is pinned by tests/golden/fuzz_*.npz; their pixels have no
 * reference artefact (no recording uses them, pygame is absent here): PARITY UNPINNED for the border ring
 * (mgo_raster.h: mgo_circle_thin) -- the fills and the blend are the routines the recordings do pin.
 */
#include "mgo_env.h"

#define SP_MAX 64
#define SP_MAXCOINS 16
#define SP_MAXLIST 256

typedef struct {
    double radius, speed, t;
    int done, has_border;
    double spawn_x, spawn_y, target_x, target_y, offset_x, offset_y, cur_x, cur_y;
} spot_t;

typedef struct {
    int endless;
    /* reset parameters */
    int max_steps, steps_per_coin, initial_spawns, spawn_interval, num_spawns;
    double initial_spawn_interval, spawn_interval_threshold, spawn_interval_decay;
    double spot_min_radius, spot_max_radius, spot_min_speed, spot_max_speed, spot_damage;
    int visual_feedback, light_dim_off_duration, light_threshold, black_background, hide_chessboard;
    int coin_enabled, coin_show_duration, coins_visible, use_exit, exit_visible, agent_visible, sample_agent_position;
    int show_last_action, show_last_positive_reward;
    double num_coins_list[SP_MAXLIST];
    int n_num_coins_list;
    double coin_scale, exit_scale, agent_speed, agent_health, agent_scale;
    double reward_inside, reward_outside, reward_death, reward_coin, reward_exit, reward_max_steps;
    /* surfaces created once in __init__ */
    mgo_surf *bg_blue, *bg_red, *spot_surf, *top_bar, *coin_surf, *exit_surf;
    mgo_rect walkable;
    uint8_t* spawn_mask; /* [y][x] */
    int dim;
    /* episode state */
    mgo_agent agent;
    int have_disp, disp_sprite, disp_x, disp_y; /* (rotated_agent_surface, rotated_agent_rect): informational only */
    double health;
    spot_t spots[SP_MAX];
    int n_spots, spawn_timer, t, coin_t;
    int intervals[128], n_intervals;
    int last_action[2];
    double last_reward;
    int bg_is_red;
    int coins_collected, num_coins, n_coins;
    int coin_x[SP_MAXCOINS], coin_y[SP_MAXCOINS], coin_radius;
    int has_coin; /* endless: self.coin is not None */
    int exit_x, exit_y, exit_open;
    double exit_radius;
    mgo_rect exit_rect;
    int quarter;
    mgo_rect act0, act1, coin_bar;
    /* info */
    int has_info, info_length, info_success;
    double info_reward, info_health, info_coins;
} sp_t;

/* get_tiled_background_surface (pygame_assets.py:222-239) */
static mgo_surf* sp_tiled_bg(int dim, uint32_t c2, double scale) {
    mgo_surf* s = mgo_surf_new(dim, dim);
    int ts = (int)(50 * scale), n = (dim + ts - 1) / ts;
    for (int x = 0; x < n; x++)
        for (int y = 0; y < n; y++)
            mgo_draw_rect(s, ((x + y) % 2 == 0) ? MGO_RGB(255, 255, 255) : c2, x * ts, y * ts, ts, ts, 0);
    return s;
}

/* GridPositionSampler.block_spawn_position (pygame_assets.py:46-59): strict x*x + y*y < r*r */
static void sp_block(sp_t* p, int px, int py, int r) {
    int n = p->dim;
    for (int y = 0; y < n; y++)
        for (int x = 0; x < n; x++) {
            int dx = x - px, dy = y - py;
            if (dx * dx + dy * dy < r * r) p->spawn_mask[y * n + x] = 1;
        }
}
/* GridPositionSampler.sample (pygame_assets.py:26-44): k-th unmasked cell in row-major order */
static void sp_sample(mgo_env* e, sp_t* p, int r, int* ox, int* oy) {
    int n = p->dim, free_cells = 0;
    for (int i = 0; i < n * n; i++) free_cells += !p->spawn_mask[i];
    int k = (int)mgo_integers(&e->rng, 0, free_cells);
    int v = 0;
    for (int i = 0; i < n * n; i++)
        if (!p->spawn_mask[i]) {
            if (k == 0) { v = i; break; }
            k--;
        }
    int y = (int)((double)v / n), x = v % n;
    sp_block(p, x, y, r);
    *ox = x;
    *oy = y;
}

/* Spotlight.__init__ (pygame_assets.py:62-101); numpy truncates float bounds of integers() toward zero */
static void sp_new_spot(mgo_env* e, sp_t* p) {
    if (p->n_spots >= SP_MAX) {
        snprintf(e->err, sizeof(e->err), "spotlight slots exhausted");
        return;
    }
    spot_t* s = &p->spots[p->n_spots++];
    s->radius = (double)mgo_integers(&e->rng, (int64_t)p->spot_min_radius, (int64_t)(p->spot_max_radius + 1));
    s->speed = mgo_uniform(&e->rng, p->spot_min_speed, p->spot_max_speed);
    s->t = 0;
    s->done = 0;
    s->has_border = p->black_background;
    int dim = p->dim;
    double cx = dim / 2, cy = dim / 2;
    double diagonal = sqrt(pow(dim, 2) + pow(dim, 2));
    double spawn_radius = diagonal / 2 + s->radius;
    int start = (int)mgo_integers(&e->rng, 0, 360);
    int inverted = start + 180;
    int target = inverted + (int)mgo_integers(&e->rng, -45, 45);
    int offset = target + (int)mgo_integers(&e->rng, -135, 135);
    double vx, vy;
    mgo_vec_rotate(spawn_radius, 0, start, &vx, &vy);
    s->spawn_x = cx + vx; s->spawn_y = cy + vy;
    s->cur_x = s->spawn_x; s->cur_y = s->spawn_y;
    mgo_vec_rotate(spawn_radius, 0, target, &vx, &vy);
    s->target_x = cx + vx; s->target_y = cy + vy;
    mgo_vec_rotate(spawn_radius, 0, offset, &vx, &vy);
    s->offset_x = cx + vx; s->offset_y = cy + vy;
}

/* _process_spawn_pos (endless_searing_spotlights.py:233-254) */
static void sp_process_spawn(mgo_env* e, sp_t* p, int* x, int* y) {
    int off = (int)(30 * e->scale), dim = p->dim;
    if (*x < off) *x = off; else if (*x > dim - off) *x = dim - off;
    if (*y < off) *y = off; else if (*y > dim - off) *y = dim - off;
}

/* Coin.draw (pygame_assets.py:145-152) */
static void sp_draw_coin(mgo_surf* s, double coin_scale, int x, int y) {
    int radius = (int)(10 * coin_scale);
    mgo_draw_circle(s, MGO_RGB(255, 255, 0), x, y, radius, 0);
    mgo_draw_circle(s, MGO_RGB(255, 165, 0), x, y, radius, (int)(2 * coin_scale));
}

/* endless: _spawn_coin (endless_searing_spotlights.py:256-269) */
static void sp_spawn_coin_endless(mgo_env* e, sp_t* p) {
    mgo_fill(p->coin_surf, 255);
    mgo_set_colorkey(p->coin_surf, 255);
    memset(p->spawn_mask, 0, (size_t)p->dim * p->dim); /* grid_sampler.reset */
    if (p->has_coin) sp_block(p, p->coin_x[0], p->coin_y[0], 28);
    int x, y;
    sp_sample(e, p, 28, &x, &y);
    x += (int)mgo_integers(&e->rng, 2, 4);
    y += (int)mgo_integers(&e->rng, 2, 4);
    sp_process_spawn(e, p, &x, &y);
    p->coin_x[0] = x;
    p->coin_y[0] = y;
    p->has_coin = 1;
    p->coin_radius = (int)(10 * p->coin_scale);
    sp_draw_coin(p->coin_surf, p->coin_scale, x, y);
}

/* Exit.draw (pygame_assets.py:191-205): rounded top corners.  pygame's draw_round_rect restated from its
 * published algorithm (filled: octagon + filled circle quadrants; outline: four thick lines + quadrant arcs).
 * Pinned pixel for pixel, closed and open, by the reference's searing_spotlights_0(_gt).gif
 * (tests/test_oracle_old_gif_replay.py). */
static void sp_quadrant(mgo_surf* s, int x0, int y0, int radius, int thickness, uint32_t c, int tr, int tl, int bl, int br) {
    /* pygame draw_circle_quadrant */
    int f = 1 - radius, ddx = 0, ddy = -2 * radius, x = 0, y = radius;
    int y1, i_y = radius - thickness, i_f = 1 - i_y, i_ddx = 0, i_ddy = -2 * i_y, i;
    if (radius == 1) {
        if (tr) mgo_hline(s, c, x0, y0 - 1, x0);
        if (tl) mgo_hline(s, c, x0 - 1, y0 - 1, x0 - 1);
        if (bl) mgo_hline(s, c, x0 - 1, y0, x0 - 1);
        if (br) mgo_hline(s, c, x0, y0, x0);
        return;
    }
    if (thickness != 0) {
        while (x < y) {
            if (f >= 0) { y--; ddy += 2; f += ddy; }
            if (i_f >= 0) { i_y--; i_ddy += 2; i_f += i_ddy; }
            x++; ddx += 2; f += ddx + 1;
            i_ddx += 2; i_f += i_ddx + 1;
            if (thickness > 1) thickness = y - i_y;
            if (tr) for (i = 0; i < thickness; i++) { y1 = y - i; if ((y0 - y1) < (y0 - x)) mgo_hline(s, c, x0 + x - 1, y0 - y1, x0 + x - 1); if ((x0 + y1 - 1) >= (x0 + x - 1)) mgo_hline(s, c, x0 + y1 - 1, y0 - x, x0 + y1 - 1); }
            if (tl) for (i = 0; i < thickness; i++) { y1 = y - i; if ((y0 - y1) <= (y0 - x)) mgo_hline(s, c, x0 - x, y0 - y1, x0 - x); if ((x0 - y1) < (x0 - x)) mgo_hline(s, c, x0 - y1, y0 - x, x0 - y1); }
            if (bl) for (i = 0; i < thickness; i++) { y1 = y - i; if ((x0 - y1) <= (x0 - x)) mgo_hline(s, c, x0 - y1, y0 + x - 1, x0 - y1); if ((y0 + y1 - 1) > (y0 + x - 1)) mgo_hline(s, c, x0 - x, y0 + y1 - 1, x0 - x); }
            if (br) for (i = 0; i < thickness; i++) { y1 = y - i; if ((y0 + y1 - 1) >= (y0 + x - 1)) mgo_hline(s, c, x0 + x - 1, y0 + y1 - 1, x0 + x - 1); if ((x0 + y1 - 1) > (x0 + x - 1)) mgo_hline(s, c, x0 + y1 - 1, y0 + x - 1, x0 + y1 - 1); }
        }
    } else {
        while (x < y) {
            if (f >= 0) { y--; ddy += 2; f += ddy; }
            x++; ddx += 2; f += ddx + 1;
            if (tr) { for (y1 = y0 - x; y1 <= y0; y1++) mgo_hline(s, c, x0 + y - 1, y1, x0 + y - 1); for (y1 = y0 - y; y1 <= y0; y1++) mgo_hline(s, c, x0 + x - 1, y1, x0 + x - 1); }
            if (tl) { for (y1 = y0 - x; y1 <= y0; y1++) mgo_hline(s, c, x0 - y, y1, x0 - y); for (y1 = y0 - y; y1 <= y0; y1++) mgo_hline(s, c, x0 - x, y1, x0 - x); }
            if (bl) { for (y1 = y0; y1 < y0 + x; y1++) mgo_hline(s, c, x0 - y, y1, x0 - y); for (y1 = y0; y1 < y0 + y; y1++) mgo_hline(s, c, x0 - x, y1, x0 - x); }
            if (br) { for (y1 = y0; y1 < y0 + x; y1++) mgo_hline(s, c, x0 + y - 1, y1, x0 + y - 1); for (y1 = y0; y1 < y0 + y; y1++) mgo_hline(s, c, x0 + x - 1, y1, x0 + x - 1); }
        }
    }
}
static void sp_round_rect(mgo_surf* s, uint32_t c, int x1, int y1, int x2, int y2, int width, int tl, int tr, int bl, int br) {
    int w = x2 - x1 + 1, h = y2 - y1 + 1;
    if ((tl + tr) > w || (tl + bl) > h || (tr + br) > h || (bl + br) > w) {
        float qt = w / (float)(tl + tr), ql = h / (float)(tl + bl), qb = w / (float)(bl + br), qr = h / (float)(tr + br);
        float f = fminf(fminf(fminf(qt, ql), qb), qr);
        tl = (int)(tl * f); tr = (int)(tr * f); bl = (int)(bl * f); br = (int)(br * f);
    }
    if (width == 0) {
        /* filled octagon (x1+tl,y1)-(x2-tr,y1)-(x2,y1+tr)-(x2,y2-br)-(x2-br,y2)-(x1+bl,y2)-(x1,y2-bl)-(x1,y1+tl) */
        for (int y = y1; y <= y2; y++) {
            int xa = x1, xb = x2;
            if (y < y1 + tl) xa = x1 + (tl - (y - y1));
            if (y < y1 + tr) xb = x2 - (tr - (y - y1));
            if (y > y2 - bl) xa = x1 + (bl - (y2 - y));
            if (y > y2 - br) xb = x2 - (br - (y2 - y));
            mgo_hline(s, c, xa, y, xb);
        }
        sp_quadrant(s, x2 - tr + 1, y1 + tr, tr, 0, c, 1, 0, 0, 0);
        sp_quadrant(s, x1 + tl, y1 + tl, tl, 0, c, 0, 1, 0, 0);
        sp_quadrant(s, x1 + bl, y2 - bl + 1, bl, 0, c, 0, 0, 1, 0);
        sp_quadrant(s, x2 - br + 1, y2 - br + 1, br, 0, c, 0, 0, 0, 1);
    } else {
        int o = (int)(width / 2) - 1 + width % 2, o2 = (int)(width / 2);
        if (x2 - tr == x1 + tl) { for (int i = 0; i < width; i++) mgo_hline(s, c, x1 + tl, y1 + i, x1 + tl); }
        else mgo_draw_line(s, c, x1 + tl, y1 + o, x2 - tr, y1 + o, width);
        if (y2 - bl == y1 + tl) { for (int i = 0; i < width; i++) mgo_hline(s, c, x1 + i, y1 + tl, x1 + i); }
        else mgo_draw_line(s, c, x1 + o, y1 + tl, x1 + o, y2 - bl, width);
        if (x2 - br == x1 + bl) { for (int i = 0; i < width; i++) mgo_hline(s, c, x1 + bl, y2 - i, x1 + bl); }
        else mgo_draw_line(s, c, x1 + bl, y2 - o2, x2 - br, y2 - o2, width);
        if (y2 - br == y1 + tr) { for (int i = 0; i < width; i++) mgo_hline(s, c, x2 - i, y1 + tr, x2 - i); }
        else mgo_draw_line(s, c, x2 - o2, y1 + tr, x2 - o2, y2 - br, width);
        sp_quadrant(s, x2 - tr + 1, y1 + tr, tr, width, c, 1, 0, 0, 0);
        sp_quadrant(s, x1 + tl, y1 + tl, tl, width, c, 0, 1, 0, 0);
        sp_quadrant(s, x1 + bl, y2 - bl + 1, bl, width, c, 0, 0, 1, 0);
        sp_quadrant(s, x2 - br + 1, y2 - br + 1, br, width, c, 0, 0, 0, 1);
    }
}
static void sp_exit_draw(mgo_env* e, sp_t* p, int open) {
    (void)e;
    if (open == p->exit_open) return;
    p->exit_open = open;
    int d = p->exit_surf->w, r = (int)(10 * p->exit_scale);
    uint32_t c = open ? MGO_RGB(48, 141, 70) : MGO_RGB(55, 55, 55);
    if (r <= 0 || d < 2) {
        mgo_draw_rect(p->exit_surf, c, 0, 0, d, d, 0);
        mgo_draw_rect(p->exit_surf, 0, 0, 0, d, d, (int)(2 * p->exit_scale));
    } else {
        sp_round_rect(p->exit_surf, c, 0, 0, d - 1, d - 1, 0, r, r, 0, 0);
        int w = (int)(2 * p->exit_scale);
        if (w > 0) sp_round_rect(p->exit_surf, 0, 0, 0, d - 1, d - 1, w, r, r, 0, 0);
        else sp_round_rect(p->exit_surf, 0, 0, 0, d - 1, d - 1, 0, r, r, 0, 0); /* width 0 == filled (black) */
    }
    p->exit_rect.x = p->exit_rect.y = 0;
    p->exit_rect.w = p->exit_rect.h = d;
    mgo_rect_set_center(&p->exit_rect, p->exit_x, p->exit_y);
}

static void sp_compose(mgo_env* e, sp_t* p, mgo_surf* agent_surf) {
    /* surface order built by the insert() calls of reset/step (endless :465-479, finite :526-543) */
    mgo_surf* bg = p->bg_is_red ? p->bg_red : p->bg_blue;
    int coin_above = p->endless ? (p->coins_visible || p->coin_t < p->coin_show_duration) : p->coins_visible;
    int have_coin_surf = p->endless ? p->coin_enabled : (p->num_coins > 0);
    mgo_blit(e->screen, bg, 0, 0);
    if (!coin_above && have_coin_surf) mgo_blit(e->screen, p->coin_surf, 0, 0);
    if (!p->endless && !p->exit_visible && p->exit_surf) mgo_blit(e->screen, p->exit_surf, p->exit_rect.x, p->exit_rect.y);
    if (!p->agent_visible) mgo_blit(e->screen, agent_surf, p->agent.rect.x, p->agent.rect.y);
    mgo_blit(e->screen, p->spot_surf, 0, 0);
    if (coin_above && have_coin_surf) mgo_blit(e->screen, p->coin_surf, 0, 0);
    if (!p->endless && p->exit_visible && p->exit_surf) mgo_blit(e->screen, p->exit_surf, p->exit_rect.x, p->exit_rect.y);
    /* NOTE: with agent_visible the reference's insert index (spot_surface_id + 3) can exceed the list length and
     * lands the agent on top of the top bar; list.insert clamps, so "append" is the faithful reading. */
    mgo_blit(e->screen, p->top_bar, 0, 0);
    if (p->agent_visible) mgo_blit(e->screen, agent_surf, p->agent.rect.x, p->agent.rect.y);
}

/* _build_debug_surface (searing_spotlights.py:157-185, endless_searing_spotlights.py:150-177): board, spotlight layer,
 * exit (finite), ALL coins on a fresh keyed surface, the agent (the stored (surface, rect) pair: stale after a reset
 * until the first step), top bar -- i.e. everything the dark layer hides in the observation is drawn over it. */
static void sp_debug(mgo_env* e, mgo_surf* dst) {
    sp_t* p = (sp_t*)e->impl;
    mgo_surf* coin = mgo_surf_new(p->dim, p->dim);
    mgo_fill(coin, 255);
    mgo_set_colorkey(coin, 255);
    if (p->endless) {
        if (p->coin_enabled && p->has_coin) sp_draw_coin(coin, p->coin_scale, p->coin_x[0], p->coin_y[0]);
    } else {
        for (int k = 0; k < p->n_coins; k++) sp_draw_coin(coin, p->coin_scale, p->coin_x[k], p->coin_y[k]);
    }
    mgo_fill(dst, 0);
    mgo_blit(dst, p->bg_is_red ? p->bg_red : p->bg_blue, 0, 0);
    mgo_blit(dst, p->spot_surf, 0, 0);
    if (!p->endless && p->exit_surf) mgo_blit(dst, p->exit_surf, p->exit_rect.x, p->exit_rect.y);
    mgo_blit(dst, coin, 0, 0);
    if (p->have_disp) {
        const mgo_surf* sp = p->agent.sprites[p->disp_sprite];
        mgo_blit(dst, sp, p->disp_x - sp->w / 2, p->disp_y - sp->h / 2);
    } else {
        mgo_blit(dst, p->agent.sprites[0], p->agent.rect.x, p->agent.rect.y);
    }
    mgo_blit(dst, p->top_bar, 0, 0);
    mgo_surf_free(coin);
}

/* scene hook: v = {bg_red, alpha, ax, ay, sprite, exit_x, exit_y, exit_open, n_coins, (x, y) * n_coins, n_spots,
 * (x, y, radius) * n_spots}.  Coins / exit / spotlight surfaces are painted like reset() / step() paint them; the top bar
 * is left as the last reset / step drew it. */
static int sp_scene(mgo_env* e, const double* v, int n) {
    sp_t* p = (sp_t*)e->impl;
    if (n < 10) return -1;
    int k = 0;
    p->bg_is_red = (int)v[k++];
    mgo_set_alpha(p->spot_surf, (int)v[k++]);
    const int ax = (int)v[k++], ay = (int)v[k++];
    mgo_rect_set_center(&p->agent.rect, ax, ay);
    p->have_disp = 1;
    p->disp_sprite = (int)v[k++] & 7;
    p->disp_x = ax;
    p->disp_y = ay;
    if (!p->endless) {
        p->exit_x = (int)v[k];
        p->exit_y = (int)v[k + 1];
        p->exit_open = !(int)v[k + 2]; /* force a repaint */
        sp_exit_draw(e, p, (int)v[k + 2]);
    }
    k += 3;
    const int nc = (int)v[k++];
    if (nc < 0 || nc > SP_MAXCOINS || k + 2 * nc + 1 > n) return -1;
    mgo_fill(p->coin_surf, 255);
    mgo_set_colorkey(p->coin_surf, 255);
    p->n_coins = nc;
    if (!p->endless) p->num_coins = nc > p->num_coins ? nc : p->num_coins;
    p->has_coin = nc > 0;
    for (int i = 0; i < nc; i++) {
        p->coin_x[i] = (int)v[k++];
        p->coin_y[i] = (int)v[k++];
        sp_draw_coin(p->coin_surf, p->coin_scale, p->coin_x[i], p->coin_y[i]);
    }
    const int ns = (int)v[k++];
    if (ns < 0 || ns > SP_MAX || k + 3 * ns > n) return -1;
    mgo_fill(p->spot_surf, 0);
    p->n_spots = ns;
    for (int i = 0; i < ns; i++) {
        spot_t* s = &p->spots[i];
        memset(s, 0, sizeof(*s));
        s->cur_x = v[k++];
        s->cur_y = v[k++];
        s->radius = v[k++];
        mgo_draw_circle(p->spot_surf, MGO_RGB(255, 0, 0), (int)s->cur_x, (int)s->cur_y, (int)s->radius, 0);
    }
    sp_compose(e, p, p->agent.sprites[p->disp_sprite]);
    return 0;
}

static void sp_reset(mgo_env* e) {
    sp_t* p = (sp_t*)e->impl;
    double S = e->scale;
    int dim = p->dim;
    p->has_info = 0;
    p->t = 0;
    p->coin_t = 0;
    if (p->hide_chessboard) { /* both surfaces turn white for the rest of the object's life */
        mgo_fill(p->bg_blue, MGO_RGB(255, 255, 255));
        mgo_fill(p->bg_red, MGO_RGB(255, 255, 255));
    }
    memset(p->spawn_mask, 0, (size_t)dim * dim);
    e->ep_sum = 0;
    e->ep_len = 0;
    p->last_action[0] = p->last_action[1] = 0;
    if (p->have_disp) { /* the stale (surface, rect) pair of the previous agent object lives on until the first step */
        p->disp_x = mgo_rect_cx(&p->agent.rect);
        p->disp_y = mgo_rect_cy(&p->agent.rect);
    }
    static const int ROT[8] = {0, 45, 90, 135, 180, 225, 270, 315};
    int rotation = ROT[mgo_choice_index(&e->rng, 8)];
    mgo_agent_init(&p->agent, p->agent_speed, p->agent_scale, rotation);
    int sx, sy;
    if (p->sample_agent_position) {
        sp_sample(e, p, 28, &sx, &sy);
        sx += (int)mgo_integers(&e->rng, 2, 4);
        sy += (int)mgo_integers(&e->rng, 2, 4);
    } else {
        sx = dim / 2;
        sy = dim / 2;
        sp_block(p, sx, sy, 21);
    }
    mgo_rect_set_center(&p->agent.rect, sx, sy);
    p->health = p->agent_health;
    /* top bar */
    p->quarter = (int)(dim / 4);
    int bar_h = (int)(16 * S);
    mgo_fill(p->top_bar, 0);
    if (!p->endless) mgo_draw_rect(p->top_bar, MGO_RGB(50, 50, 50), 0, 0, dim, bar_h, 0);
    mgo_draw_rect(p->top_bar, MGO_RGB(0, 255, 0), 0, 0, p->quarter * 2, bar_h, 0);
    static const uint32_t ACT[3] = {MGO_RGB(120, 120, 120), MGO_RGB(116, 1, 113), MGO_RGB(255, 94, 14)};
    if (p->show_last_action) {
        p->act0 = (mgo_rect){p->quarter * 2, 0, p->quarter, bar_h};
        p->act1 = (mgo_rect){p->quarter * 3, 0, p->quarter, bar_h};
        mgo_draw_rect(p->top_bar, ACT[0], p->act0.x, 0, p->act0.w, bar_h, 0);
        mgo_draw_rect(p->top_bar, ACT[0], p->act1.x, 0, p->act1.w, bar_h, 0);
    }
    if (p->show_last_positive_reward) {
        p->last_reward = 0.0;
        if (p->show_last_action) p->coin_bar = (mgo_rect){(int)(p->quarter * 2.75), 0, (int)(p->quarter * 0.5), bar_h};
        else p->coin_bar = (mgo_rect){(int)(p->quarter * 2), 0, (int)(p->quarter * 2), bar_h};
    }
    /* spotlights */
    if (!p->endless) { /* _compute_spawn_intervals (searing_spotlights.py:137-143) */
        double initial = p->initial_spawn_interval;
        p->n_intervals = 0;
        for (int i = 0; i < p->num_spawns && i < 128; i++) {
            p->intervals[p->n_intervals++] = (int)(initial + p->spawn_interval_threshold);
            initial = initial * pow(p->spawn_interval_decay, 1);
        }
    }
    if (p->light_dim_off_duration > 0) mgo_set_alpha(p->spot_surf, 0);
    else mgo_set_alpha(p->spot_surf, p->light_threshold);
    p->n_spots = 0;
    p->spawn_timer = 0;
    for (int i = 0; i < p->initial_spawns; i++) sp_new_spot(e, p);
    /* coins / exit */
    p->coins_collected = 0;
    p->n_coins = 0;
    if (p->endless) {
        p->has_coin = 0;
        if (p->coin_enabled) sp_spawn_coin_endless(e, p);
    } else {
        p->num_coins = p->n_num_coins_list > 0 ? (int)p->num_coins_list[mgo_choice_index(&e->rng, p->n_num_coins_list)] : 0;
        if (p->num_coins > 0) { /* _spawn_coins (:267-278) */
            mgo_fill(p->coin_surf, 255);
            mgo_set_colorkey(p->coin_surf, 255);
            p->coin_radius = (int)(10 * p->coin_scale);
            for (int i = 0; i < p->num_coins && i < SP_MAXCOINS; i++) {
                int x, y;
                sp_sample(e, p, 21, &x, &y);
                x += (int)mgo_integers(&e->rng, 2, 4);
                y += (int)mgo_integers(&e->rng, 2, 4);
                sp_process_spawn(e, p, &x, &y);
                sp_draw_coin(p->coin_surf, p->coin_scale, x, y);
                p->coin_x[p->n_coins] = x;
                p->coin_y[p->n_coins++] = y;
            }
        }
        /* use_exit == False (searing_spotlights.py:413-416): no exit is spawned -- and no position sampled, no number drawn -- but the
           frame still blits self.exit (:431-435): the Exit object of an EARLIER episode, at its place and in the state (open / closed)
           it was last drawn in.  Without an earlier exit the reference raises AttributeError there; here nothing is drawn. */
        if (p->use_exit) {
        /* _spawn_exit (:280-286) */
        int x, y;
        sp_sample(e, p, 21, &x, &y);
        x += (int)mgo_integers(&e->rng, 2, 4);
        y += (int)mgo_integers(&e->rng, 2, 4);
        sp_process_spawn(e, p, &x, &y);
        p->exit_x = x;
        p->exit_y = y;
        double rect_dim = 20 * p->exit_scale;
        mgo_surf_free(p->exit_surf);
        p->exit_surf = mgo_surf_new((int)rect_dim, (int)rect_dim);
        mgo_fill(p->exit_surf, 255);
        mgo_set_colorkey(p->exit_surf, 255);
        p->exit_radius = 20.0 / 2 * p->exit_scale;
        p->exit_open = 1;
        sp_exit_draw(e, p, 0);
        }
    }
    p->bg_is_red = 0;
    if (p->black_background) mgo_fill(p->bg_blue, 0);
    sp_compose(e, p, p->agent.sprites[0]); /* reset frame always shows sprite index 0 (App. D.19) */
    e->reward = 0;
    e->done = 0;
    if (p->endless) {
        e->gt[0] = (double)mgo_rect_cx(&p->agent.rect) / dim;
        e->gt[1] = (double)mgo_rect_cy(&p->agent.rect) / dim;
        e->gt[2] = p->coin_enabled ? (double)p->coin_x[0] / dim : 0.0;
        e->gt[3] = p->coin_enabled ? (double)p->coin_y[0] / dim : 0.0;
    }
}

static double sp_dist(double ax, double ay, double bx, double by) {
    double dx = bx - ax, dy = by - ay;
    return sqrt(dx * dx + dy * dy);
}

static void sp_step(mgo_env* e, const int action[2]) {
    sp_t* p = (sp_t*)e->impl;
    double S = e->scale;
    int dim = p->dim, bar_h = (int)(16 * S);
    static const uint32_t ACT[3] = {MGO_RGB(120, 120, 120), MGO_RGB(116, 1, 113), MGO_RGB(255, 94, 14)};
    mgo_agent_step(&p->agent, action, &p->walkable);
    p->have_disp = 1;
    p->disp_sprite = p->agent.rotation / 45;
    p->disp_x = mgo_rect_cx(&p->agent.rect);
    p->disp_y = mgo_rect_cy(&p->agent.rect);
    if (p->endless || p->show_last_action) {
        mgo_draw_rect(p->top_bar, ACT[p->last_action[0]], p->act0.x, 0, p->act0.w, bar_h, 0);
        mgo_draw_rect(p->top_bar, ACT[p->last_action[1]], p->act1.x, 0, p->act1.w, bar_h, 0);
        p->last_action[0] = action[0];
        p->last_action[1] = action[1];
    }
    if (p->spot_surf->alpha <= p->light_threshold) {
        if (p->light_dim_off_duration > 0) mgo_set_alpha(p->spot_surf, p->spot_surf->alpha + (int)(255.0 / p->light_dim_off_duration));
        else mgo_set_alpha(p->spot_surf, p->light_threshold);
    }
    double reward = 0.0;
    /* ---- _step_spotlight_task ---- */
    double r = 0.0;
    int spot_done = 0;
    p->spawn_timer += 1;
    if (p->endless) {
        if (p->spawn_timer >= p->spawn_interval) {
            sp_new_spot(e, p);
            p->spawn_timer = 0;
        }
    } else if (p->n_intervals > 0) {
        if (p->spawn_timer >= p->intervals[0]) {
            sp_new_spot(e, p);
            p->n_intervals--; /* list.pop() removes the LAST element while [0] is tested (App. D.8) */
            p->spawn_timer = 0;
        }
    }
    mgo_fill(p->spot_surf, 0);
    int hit = 0;
    int acx = mgo_rect_cx(&p->agent.rect), acy = mgo_rect_cy(&p->agent.rect);
    for (int i = 0; i < p->n_spots; i++) { /* list mutated while iterated: the element after a removed one is skipped */
        spot_t* s = &p->spots[i];
        if (s->done) {
            memmove(&p->spots[i], &p->spots[i + 1], sizeof(spot_t) * (size_t)(p->n_spots - i - 1));
            p->n_spots--;
        } else {
            double lx = s->target_x * (1 - s->t) + s->offset_x * s->t, ly = s->target_y * (1 - s->t) + s->offset_y * s->t;
            s->cur_x = s->spawn_x * (1 - s->t) + lx * s->t;
            s->cur_y = s->spawn_y * (1 - s->t) + ly * s->t;
            mgo_draw_circle(p->spot_surf, MGO_RGB(255, 0, 0), (int)s->cur_x, (int)s->cur_y, (int)s->radius, 0);
            if (s->has_border) mgo_draw_circle(p->spot_surf, MGO_RGB(255, 255, 255), (int)s->cur_x, (int)s->cur_y, (int)s->radius, 1);
            s->t += s->speed;
            if (s->t >= 1.0) {
                s->t = 1.0;
                s->done = 1;
            }
            if (sp_dist(s->cur_x, s->cur_y, acx, acy) <= s->radius + p->agent.radius) hit++;
        }
    }
    if (hit > 0) {
        p->health -= p->spot_damage;
        r += p->reward_inside;
        int width = (int)((dim / 2) * (1 - p->health / p->agent_health));
        mgo_draw_rect(p->top_bar, MGO_RGB(255, 0, 0), 0, 0, width, bar_h, 0);
        p->bg_is_red = p->visual_feedback ? 1 : 0;
    } else {
        p->bg_is_red = 0;
        r += p->reward_outside;
    }
    if (p->black_background) mgo_fill(p->bg_is_red ? p->bg_red : p->bg_blue, 0); /* bg.fill(0): the surface stays black */
    if (p->health <= 0) {
        spot_done = 1;
        r += p->reward_death;
    }
    reward += r;
    /* ---- coin task ---- */
    int done = 0, success = 0;
    if (p->endless) {
        if (p->coin_enabled) {
            double cr = 0.0;
            if (sp_dist(p->coin_x[0], p->coin_y[0], acx, acy) <= p->coin_radius + p->agent.radius) {
                cr += p->reward_coin;
                p->coins_collected += 1;
                p->coin_t = 0;
                sp_spawn_coin_endless(e, p);
            }
            reward += cr;
        }
        if (spot_done) done = 1;
        p->t += 1;
        p->coin_t += 1;
        if (p->coin_t == p->steps_per_coin && p->coin_enabled) done = 1;
        if (p->t == p->max_steps) done = 1;
    } else {
        int coins_done;
        if (p->num_coins > 0) { /* _step_coin_task (:288-311): remove-while-iterating skips the next coin */
            double cr = 0.0;
            int update = 0;
            for (int i = 0; i < p->n_coins; i++) {
                if (sp_dist(p->coin_x[i], p->coin_y[i], acx, acy) <= p->coin_radius + p->agent.radius) {
                    for (int j = i; j < p->n_coins - 1; j++) {
                        p->coin_x[j] = p->coin_x[j + 1];
                        p->coin_y[j] = p->coin_y[j + 1];
                    }
                    p->n_coins--;
                    cr += p->reward_coin;
                    update = 1;
                    p->coins_collected += 1;
                }
            }
            if (update) {
                mgo_fill(p->coin_surf, 255);
                for (int i = 0; i < p->n_coins; i++) sp_draw_coin(p->coin_surf, p->coin_scale, p->coin_x[i], p->coin_y[i]);
            }
            coins_done = p->n_coins == 0;
            reward += cr;
        } else {
            coins_done = 1;
        }
        int exit_done = 0;
        if (p->use_exit) { /* _step_exit_task (:313-330) */
            double er = 0.0;
            if (coins_done) {
                sp_exit_draw(e, p, 1);
                if (sp_dist(p->exit_x, p->exit_y, acx, acy) <= p->exit_radius + p->agent.radius) {
                    exit_done = 1;
                    er = p->reward_exit;
                }
            }
            reward += er;
        }
        if (spot_done) done = 1;
        else if (coins_done) { /* (:499-511) */
            if (p->use_exit) {
                if (exit_done) { done = 1; success = 1; }
            } else if (p->num_coins > 0) {
                done = 1;
                success = 1;
            }
        }
        p->t += 1;
        if (p->t == p->max_steps) done = 1;
    }
    if (p->show_last_positive_reward) {
        if (p->last_reward > 0) mgo_draw_rect(p->top_bar, MGO_RGB(255, 255, 0), p->coin_bar.x, 0, p->coin_bar.w, bar_h, 0);
        else mgo_draw_rect(p->top_bar, MGO_RGB(50, 50, 50), p->coin_bar.x, 0, p->coin_bar.w, bar_h, 0);
        p->last_reward = reward;
    }
    sp_compose(e, p, p->agent.sprites[p->disp_sprite]);
    e->ep_sum += reward;
    e->ep_len += 1;
    p->has_info = done;
    if (done) {
        p->info_reward = e->ep_sum;
        p->info_length = e->ep_len;
        p->info_health = p->health / p->agent_health;
        p->info_coins = p->endless ? (double)p->coins_collected : (double)p->coins_collected / (double)p->num_coins;
        p->info_success = success;
    }
    if (p->endless) {
        e->gt[0] = (double)mgo_rect_cx(&p->agent.rect) / dim;
        e->gt[1] = (double)mgo_rect_cy(&p->agent.rect) / dim;
        e->gt[2] = p->coin_enabled ? (double)p->coin_x[0] / dim : 0.0;
        e->gt[3] = p->coin_enabled ? (double)p->coin_y[0] / dim : 0.0;
    }
    e->reward = reward;
    e->done = done;
}

static int sp_set_option(mgo_env* e, const char* k, const double* v, int n) {
    sp_t* p = (sp_t*)e->impl;
#define D(name, field) if (!strcmp(k, name)) { p->field = v[0]; return 0; }
#define I(name, field) if (!strcmp(k, name)) { p->field = (int)v[0]; return 0; }
    I("max_steps", max_steps) I("initial_spawns", initial_spawns)
    D("spot_min_radius", spot_min_radius) D("spot_max_radius", spot_max_radius)
    D("spot_min_speed", spot_min_speed) D("spot_max_speed", spot_max_speed) D("spot_damage", spot_damage)
    I("visual_feedback", visual_feedback)
    I("black_background", black_background) I("hide_chessboard", hide_chessboard)
    I("light_dim_off_duration", light_dim_off_duration) I("light_threshold", light_threshold)
    D("coin_scale", coin_scale) I("coins_visible", coins_visible)
    D("agent_speed", agent_speed) D("agent_health", agent_health) D("agent_scale", agent_scale)
    I("agent_visible", agent_visible) I("sample_agent_position", sample_agent_position)
    I("show_last_action", show_last_action) I("show_last_positive_reward", show_last_positive_reward)
    D("reward_inside_spotlight", reward_inside) D("reward_outside_spotlight", reward_outside)
    D("reward_death", reward_death) D("reward_coin", reward_coin)
    if (p->endless) {
        I("steps_per_coin", steps_per_coin) I("spawn_interval", spawn_interval)
        I("coin_enabled", coin_enabled) I("coin_show_duration", coin_show_duration)
    } else {
        I("num_spawns", num_spawns) D("initial_spawn_interval", initial_spawn_interval)
        D("spawn_interval_threshold", spawn_interval_threshold) D("spawn_interval_decay", spawn_interval_decay)
        if (!strcmp(k, "num_coins")) return mgo_opt_list(p->num_coins_list, &p->n_num_coins_list, SP_MAXLIST, v, n);
        I("use_exit", use_exit) /* False: legal once the object has had an exit (its stale one is drawn), see sp_reset */
        D("exit_scale", exit_scale) I("exit_visible", exit_visible)
        D("reward_exit", reward_exit) D("reward_max_steps", reward_max_steps)
    }
#undef D
#undef I
    return -1;
}

static int sp_board_mode(const mgo_surf* b, int ts) { /* pixel (0,0) is a white tile, (ts,0) a coloured one */
    uint32_t c0 = b->px[0], c1 = b->px[ts];
    if (c0 == MGO_RGB(255, 255, 255) && c1 == MGO_RGB(255, 255, 255)) return 1;
    if (c0 == 0 && c1 == 0) return 2;
    return 0;
}

static double sp_get(mgo_env* e, const char* f, int* ok) {
    sp_t* p = (sp_t*)e->impl;
    *ok = 1;
#define F(name, expr) if (!strcmp(f, name)) return (double)(expr);
    F("ax", mgo_rect_cx(&p->agent.rect)) F("ay", mgo_rect_cy(&p->agent.rect)) F("arot", p->agent.rotation)
    F("disp_sprite", p->have_disp ? p->disp_sprite : -1)
    F("disp_x", p->have_disp ? p->disp_x : -1) F("disp_y", p->have_disp ? p->disp_y : -1)
    F("health", p->health) F("alpha", p->spot_surf->alpha) F("spawn_timer", p->spawn_timer) F("n_spots", p->n_spots)
    F("t", p->t) F("la0", p->last_action[0]) F("la1", p->last_action[1]) F("last_reward", p->last_reward)
    F("bg_red", p->bg_is_red) F("coins_collected", p->coins_collected)
    /* what hide_chessboard / black_background have left of the two boards: 0 chessboard, 1 white, 2 black */
    F("bg_blue_mode", sp_board_mode(p->bg_blue, (int)(50 * e->scale))) F("bg_red_mode", sp_board_mode(p->bg_red, (int)(50 * e->scale)))
    if (p->endless) {
        F("coin_t", p->coin_t) F("coin_x", p->coin_x[0]) F("coin_y", p->coin_y[0])
        F("gt0", e->gt[0]) F("gt1", e->gt[1]) F("gt2", e->gt[2]) F("gt3", e->gt[3])
    } else {
        F("num_coins", p->num_coins) F("n_coins_left", p->n_coins) F("exit_x", p->exit_x) F("exit_y", p->exit_y)
        F("exit_open", p->exit_open) F("n_intervals", p->n_intervals)
    }
    if (p->has_info) {
        F("info_reward", p->info_reward) F("info_length", p->info_length) F("info_agent_health", p->info_health)
        F("info_coins_collected", p->info_coins)
        if (!p->endless) { F("info_success", p->info_success) }
    }
#undef F
    *ok = 0;
    return NAN;
}

static int sp_get_list(mgo_env* e, const char* name, double* out, int cap) {
    sp_t* p = (sp_t*)e->impl;
    if (!strcmp(name, "spots")) {
        int n = 0;
        for (int i = 0; i < p->n_spots; i++) {
            spot_t* s = &p->spots[i];
            double v[12] = {s->radius, s->speed, s->t, (double)s->done, s->spawn_x, s->spawn_y, s->target_x, s->target_y,
                            s->offset_x, s->offset_y, s->cur_x, s->cur_y};
            for (int k = 0; k < 12; k++, n++)
                if (n < cap) out[n] = v[k];
        }
        return n;
    }
    if (!strcmp(name, "borders")) { /* Spotlight.has_border, list order */
        for (int i = 0; i < p->n_spots; i++)
            if (i < cap) out[i] = p->spots[i].has_border;
        return p->n_spots;
    }
    if (!strcmp(name, "coins") && !p->endless) {
        int n = 0;
        for (int i = 0; i < p->n_coins; i++) {
            if (n < cap) out[n] = p->coin_x[i];
            n++;
            if (n < cap) out[n] = p->coin_y[i];
            n++;
        }
        return n;
    }
    return -1;
}

static void sp_destroy(mgo_env* e) {
    sp_t* p = (sp_t*)e->impl;
    mgo_agent_free(&p->agent);
    mgo_surf_free(p->bg_blue); mgo_surf_free(p->bg_red); mgo_surf_free(p->spot_surf); mgo_surf_free(p->top_bar);
    mgo_surf_free(p->coin_surf); mgo_surf_free(p->exit_surf);
    free(p->spawn_mask);
    free(p);
}

/* expert hook (mgo_env.h): the (first remaining) coin, then the exit */
static void sp_expert(mgo_env* e, int a[2]) {
    sp_t* p = (sp_t*)e->impl;
    int tx, ty;
    a[0] = a[1] = 0;
    if (p->endless) {
        if (!p->coin_enabled) return;
        tx = p->coin_x[0];
        ty = p->coin_y[0];
    } else if (p->num_coins > 0 && p->n_coins > 0) {
        tx = p->coin_x[0];
        ty = p->coin_y[0];
    } else if (p->use_exit) {
        tx = p->exit_x;
        ty = p->exit_y;
    } else {
        return;
    }
    int dx = tx - mgo_rect_cx(&p->agent.rect), dy = ty - mgo_rect_cy(&p->agent.rect), slack = (int)p->agent.speed;
    a[0] = abs(dx) < slack ? 0 : (dx < 0 ? 1 : 2);
    a[1] = abs(dy) < slack ? 0 : (dy < 0 ? 1 : 2);
}

static const mgo_vtbl SP_VT[2] = {
    {"SearingSpotlights-v0", 0, 0, sp_set_option, sp_reset, sp_step, sp_get, sp_get_list, sp_destroy, sp_debug, sp_scene, sp_expert},
    {"Endless-SearingSpotlights-v0", 0, 4, sp_set_option, sp_reset, sp_step, sp_get, sp_get_list, sp_destroy, sp_debug, sp_scene, sp_expert},
};

int mgo_spot_create(mgo_env* e, int variant) {
    sp_t* p = (sp_t*)calloc(1, sizeof(sp_t));
    double S = e->scale;
    int dim = e->screen_dim;
    p->endless = variant;
    p->dim = dim;
    e->vt = &SP_VT[variant];
    e->impl = p;
    p->spot_min_radius = 30.0 * S; p->spot_max_radius = 55.0 * S;
    p->spot_min_speed = 0.0025; p->spot_max_speed = 0.0075; p->spot_damage = 1.0;
    p->visual_feedback = 1; p->light_dim_off_duration = 6; p->light_threshold = 255;
    p->coin_scale = 1.5 * S; p->coins_visible = 0;
    p->agent_speed = 12.0 * S; p->agent_scale = 1.0 * S; p->agent_visible = 0; p->sample_agent_position = 1;
    p->show_last_action = 1; p->show_last_positive_reward = 1;
    p->reward_coin = 0.25;
    if (variant) {
        p->max_steps = -1; p->steps_per_coin = 160; p->initial_spawns = 3; p->spawn_interval = 50;
        p->coin_enabled = 1; p->coin_show_duration = 6; p->agent_health = 10;
    } else {
        p->max_steps = 256; p->initial_spawns = 4; p->num_spawns = 30; p->initial_spawn_interval = 30;
        p->spawn_interval_threshold = 10; p->spawn_interval_decay = 0.95;
        p->num_coins_list[0] = 1; p->n_num_coins_list = 1; p->use_exit = 1; p->exit_scale = 2.0 * S; p->exit_visible = 0;
        p->agent_health = 5; p->reward_exit = 1.0;
    }
    p->bg_blue = sp_tiled_bg(dim, MGO_RGB(0, 0, 255), S);
    p->bg_red = sp_tiled_bg(dim, MGO_RGB(255, 0, 0), S);
    p->spot_surf = mgo_surf_new(dim, dim);
    mgo_fill(p->spot_surf, 0);
    mgo_set_colorkey(p->spot_surf, MGO_RGB(255, 0, 0));
    p->walkable = (mgo_rect){0, (int)(16 * S), dim, (int)(dim - 16 * S)};
    p->top_bar = mgo_surf_new(dim, (int)(16 * S));
    p->coin_surf = mgo_surf_new(dim, dim);
    p->spawn_mask = (uint8_t*)calloc((size_t)dim * dim, 1);
    return 0;
}
