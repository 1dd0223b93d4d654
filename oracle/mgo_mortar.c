/* oracle/mgo_mortar.c -- TEST INFRASTRUCTURE (CPU oracle), not product code.
 *
 * Restatement of the Mortar Mayhem family:
 *   MortarMayhem-Grid-v0     memory_gym/mortar_mayhem_grid.py   reset :213-278  step :280-375
 *   MortarMayhem-v0          memory_gym/mortar_mayhem.py        reset :206-272  step :274-369
 *   Endless-MortarMayhem-v0  memory_gym/endless_mortar_mayhem.py reset :194-259 step :261-373
 *   MortarMayhemB-Grid-v0    memory_gym/mortar_mayhem_b_grid.py reset :131-196  step :198-285  (no display phase; the
 *   MortarMayhemB-v0         memory_gym/mortar_mayhem_b.py      reset :132-198  step :200-287   commands are a vector obs)
 * plus Command / MortarTile / MortarArena / calc_max_episode_steps (memory_gym/pygame_assets.py:241-436).
 * Pinned by tests/golden/logic_{MortarMayhem_Grid_v0,MortarMayhem_v0,Endless_MortarMayhem_v0}.npz
 * (logic, captured from the reference) and docs/assets/emm_0.gif (pixels, SCALE 1.0).
 */
#include "mgo_env.h"

enum { MM_GRID = 0, MM_FREE = 1, MM_ENDLESS = 2, MM_B_GRID = 3, MM_B_FREE = 4 };
#define MM_IS_B(v) ((v) == MM_B_GRID || (v) == MM_B_FREE)
#define MM_IS_GRID(v) ((v) == MM_GRID || (v) == MM_B_GRID)
/* _encode_commands_one_hot (mortar_mayhem_b_grid.py:100-129): slot of each Command.COMMANDS id inside a block of 9 */
static const int CMD_ONE_HOT[9] = {1, 4, 2, 3, 0, 5, 6, 7, 8};
#define MM_MAXLIST 256 /* the reference samples from lists of any length; 256 covers every fixture */

static const int CMD_DX[9] = {1, 0, -1, 0, 0, 1, 1, -1, -1}; /* Command.COMMANDS (pygame_assets.py:242-252) */
static const int CMD_DY[9] = {0, 1, 0, -1, 0, 1, -1, 1, -1};
/* rotation of the right-pointing arrow per command (pygame_assets.py:286-301); "stay" (4) is special */
static const int CMD_ANGLE[9] = {0, 270, 180, 90, 0, 315, 45, 225, 135};

typedef struct {
    int variant;
    /* reset parameters (default_reset_parameters of the three classes) */
    double agent_scale, agent_speed;
    int arena_size, allowed_commands, visual_feedback, max_steps, initial_command_count;
    double command_count[MM_MAXLIST], show_duration[MM_MAXLIST], show_delay[MM_MAXLIST];
    double explosion_duration[MM_MAXLIST], explosion_delay[MM_MAXLIST];
    int n_command_count, n_show_duration, n_show_delay, n_explosion_duration, n_explosion_delay;
    double reward_command_failure, reward_command_success, reward_episode_success, reward_new_command_success;
    /* arena (MortarArena) */
    int N;
    double tile_dim;
    mgo_surf* arena_surf;
    mgo_rect arena_rect;
    int local_cx, local_cy;
    int tiles_on;
    /* agent */
    mgo_agent agent;
    int norm_x, norm_y;
    /* what the frame shows for the agent: (rotated_agent_surface, rotated_agent_rect) */
    mgo_surf* disp_surf; /* copy of a sprite (survives agent re-creation, as the Python object does) */
    int disp_sprite;
    mgo_rect disp_rect_store; /* a detached copy when the rect belongs to a previous agent */
    int disp_rect_is_agent;   /* 1: rotated_agent_rect IS self.agent.rect (same object) */
    int have_disp;
    /* commands */
    int* cmds;
    int num_commands, cmds_cap;
    int* vis; /* _command_visualization, entries 0..8 or 9 for "" */
    int vis_head, vis_len, vis_cap;
    int clone_pops; /* render(): entries popped so far from _command_visualization_clone (copied at reset and at an endless regeneration) */
    int glyph; /* glyph drawn in the last frame (-1 none) */
    mgo_surf* glyph_surf[10];
    int tx, ty, cur_cmd, cmd_steps, verify_step, total_completed, t;
    int expl_dur, expl_delay, show_dur, show_delay_v, max_episode_steps;
    /* info at done */
    double info_reward, info_commands_completed;
    int info_length, info_success, info_max_command_sequence, has_info;
} mm_t;

/* Command.__init__ (pygame_assets.py:254-304) */
static mgo_surf* mm_make_glyph(int cmd, double scale) {
    double rect_dim = 88 * scale;
    mgo_surf* s = mgo_surf_new((int)rect_dim, (int)rect_dim);
    mgo_fill(s, 0);
    mgo_set_colorkey(s, 0);
    int lw = (int)(8 * scale);
    uint32_t white = MGO_RGB(255, 255, 255);
    if (cmd == 4) { /* stay */
        double radius = floor(rect_dim / 2) - 4 * scale;
        double x = rect_dim - 12 * scale, y = floor(rect_dim / 2) - 8 * scale;
        mgo_draw_circle(s, white, (int)radius, (int)radius, (int)radius, lw);
        mgo_draw_line(s, white, 0, (int)y, (int)x, (int)y, lw);
    } else if (cmd >= 0 && cmd < 9) {
        double x1 = 2 * scale, x2 = 80 * scale, y1 = 40 * scale, y2 = 0;
        mgo_draw_line(s, white, (int)x1, (int)y1, (int)x2, (int)y1, lw);
        mgo_draw_line(s, white, (int)x2, (int)y1, (int)y1, (int)y2, lw);
        mgo_draw_line(s, white, (int)x2, (int)y1, (int)y1, (int)x2, lw);
        mgo_surf* r = mgo_rotate(s, CMD_ANGLE[cmd]);
        mgo_surf_free(s);
        s = r;
    }
    return s;
}

/* MortarTile.__init__/toggle_color drawn straight into the arena surface (pygame_assets.py:306-343,368,399,403) */
static void mm_draw_tile(mm_t* m, double scale, int i, int j, int red) {
    int x = (int)(m->tile_dim * i), y = (int)(m->tile_dim * j), d = (int)m->tile_dim;
    uint32_t c1 = red ? MGO_RGB(81, 18, 26) : MGO_RGB(21, 43, 77);
    uint32_t c2 = red ? MGO_RGB(112, 24, 36) : MGO_RGB(29, 60, 107);
    mgo_draw_rect(m->arena_surf, c1, x, y, d, d, 0);
    mgo_draw_rect(m->arena_surf, c2, x, y, d, d, (int)(4 * scale));
}

/* MortarArena.toggle_tiles (pygame_assets.py:385-403) */
static void mm_toggle_tiles(mm_t* m, double scale, int on, int tx, int ty, int change_color) {
    m->tiles_on = on;
    for (int i = 0; i < m->N; i++)
        for (int j = 0; j < m->N; j++) {
            if (on) {
                if (!(tx == i && ty == j) && change_color) mm_draw_tile(m, scale, i, j, 1);
            } else if (change_color) {
                mm_draw_tile(m, scale, i, j, 0);
            }
        }
}

static void mm_vis_clear(mm_t* m) { m->vis_head = m->vis_len = 0; }
static void mm_vis_push(mm_t* m, int g) {
    if (m->vis_len == m->vis_cap) {
        m->vis_cap = m->vis_cap ? m->vis_cap * 2 : 64;
        m->vis = (int*)realloc(m->vis, sizeof(int) * m->vis_cap);
    }
    m->vis[m->vis_len++] = g;
}
static int mm_vis_nonempty(const mm_t* m) { return m->vis_head < m->vis_len; }
static int mm_vis_pop(mm_t* m) { return m->vis[m->vis_head++]; }
/* _generate_command_visualization (mortar_mayhem_grid.py:192-211) */
static void mm_gen_vis(mm_t* m, const int* cmds, int n, int duration, int delay) {
    mm_vis_clear(m);
    for (int i = 0; i < n; i++) {
        for (int j = 0; j < duration; j++) mm_vis_push(m, cmds[i]);
        for (int k = 0; k < delay; k++) mm_vis_push(m, 9);
    }
}
static void mm_cmd_push(mm_t* m, int c) {
    if (m->num_commands == m->cmds_cap) {
        m->cmds_cap = m->cmds_cap ? m->cmds_cap * 2 : 32;
        m->cmds = (int*)realloc(m->cmds, sizeof(int) * m->cmds_cap);
    }
    m->cmds[m->num_commands++] = c;
}

static double mm_choice(mgo_env* e, const double* list, int n) { return list[mgo_choice_index(&e->rng, n)]; }
static double mm_max(const double* l, int n) {
    double m = l[0];
    for (int i = 1; i < n; i++) if (l[i] > m) m = l[i];
    return m;
}

/* _normalize_agent_position: (p - arena.rect[k]) // tile_dim  (float floor division) */
static void mm_normalize(mm_t* m, int px, int py, int* nx, int* ny) {
    *nx = (int)floor((double)(px - m->arena_rect.x) / m->tile_dim);
    *ny = (int)floor((double)(py - m->arena_rect.y) / m->tile_dim);
}

static void mm_draw_frame(mgo_env* e, mm_t* m, mgo_surf* agent_surf, const mgo_rect* agent_rect, int glyph) {
    mgo_fill(e->screen, 0); /* self.bg: a fresh black Surface blitted at (0,0) */
    mgo_blit(e->screen, m->arena_surf, m->arena_rect.x, m->arena_rect.y);
    if (agent_surf) mgo_blit(e->screen, agent_surf, agent_rect->x, agent_rect->y);
    if (glyph >= 0) {
        double rect_dim = 88 * e->scale;
        int p = (int)((e->screen_dim / 2) - floor(rect_dim / 2));
        mgo_blit(e->screen, m->glyph_surf[glyph], p, p);
    }
    m->glyph = glyph;
}

/* _build_debug_surface (mortar_mayhem_grid.py:104-135, mortar_mayhem.py:105-136, endless_mortar_mayhem.py:114-145): arena,
 * agent, a command glyph and a green ring around the target tile.  The glyph comes from a CLONE of the visualisation list
 * (copied when the list is made: reset :237/:257, endless regeneration :321) of which every debug render pops one entry
 * while the real list is not empty: clone_pops counts them, so any number of renders between steps (none, one as in a
 * recording loop, several) shows what the reference shows.  The reference raises IndexError once the clone is empty while
 * the real list is not (more renders than steps); here no glyph is drawn then. */
static void mm_debug(mgo_env* e, mgo_surf* dst) {
    mm_t* m = (mm_t*)e->impl;
    mgo_fill(dst, 0);
    mgo_blit(dst, m->arena_surf, m->arena_rect.x, m->arena_rect.y);
    if (m->have_disp) {
        const mgo_rect* r = m->disp_rect_is_agent ? &m->agent.rect : &m->disp_rect_store;
        mgo_blit(dst, m->disp_surf, r->x, r->y);
    } else {
        mgo_blit(dst, m->agent.sprites[0], m->agent.rect.x, m->agent.rect.y);
    }
    if (m->vis_head < m->vis_len) {
        int idx = m->clone_pops++, g = idx >= 0 && idx < m->vis_len ? m->vis[idx] : 9;
        if (g >= 0 && g < 9) { /* 9 = "" (delay frames) */
            double rect_dim = 88 * e->scale;
            int p = (int)((e->screen_dim / 2) - floor(rect_dim / 2));
            mgo_blit(dst, m->glyph_surf[g], p, p);
        }
    }
    /* translation = arena.rect.center[0] - arena.local_center[0] + tile_dim // 2 (floats); pos and radius truncate in draw.circle */
    double tr = mgo_rect_cx(&m->arena_rect) - m->local_cx + floor(m->tile_dim / 2);
    int px = (int)(m->tile_dim * m->tx + tr), py = (int)(m->tile_dim * m->ty + tr);
    mgo_draw_circle(dst, MGO_RGB(0, 255, 0), px, py, (int)floor(m->tile_dim / 2), (int)(8 * e->scale));
}

/* scene hook: v = {ax, ay, sprite (0..7, -1 none), tiles_on, tx, ty, glyph (0..8, 9 blank, -1 none)} */
static int mm_scene(mgo_env* e, const double* v, int n) {
    mm_t* m = (mm_t*)e->impl;
    if (n < 7) return -1;
    mgo_rect_set_center(&m->agent.rect, v[0], v[1]);
    const int sprite = (int)v[2], on = (int)v[3], glyph = (int)v[6];
    mm_toggle_tiles(m, e->scale, 0, 0, 0, 1);
    m->tx = (int)v[4];
    m->ty = (int)v[5];
    if (on) mm_toggle_tiles(m, e->scale, 1, m->tx, m->ty, 1);
    mm_draw_frame(e, m, sprite >= 0 ? m->agent.sprites[sprite & 7] : NULL, &m->agent.rect, glyph);
    return 0;
}

static void mm_set_disp(mm_t* m, int sprite, int rect_is_agent) {
    mgo_surf* src = m->agent.sprites[sprite];
    if (!m->disp_surf || m->disp_surf->w != src->w) {
        mgo_surf_free(m->disp_surf);
        m->disp_surf = mgo_surf_new(src->w, src->h);
    }
    memcpy(m->disp_surf->px, src->px, sizeof(uint32_t) * src->w * src->h);
    m->disp_surf->has_key = src->has_key;
    m->disp_surf->key = src->key;
    m->disp_sprite = sprite;
    m->disp_rect_is_agent = rect_is_agent;
    m->have_disp = 1;
}
static const mgo_rect* mm_disp_rect(const mm_t* m) { return m->disp_rect_is_agent ? &m->agent.rect : &m->disp_rect_store; }

static void mm_reset(mgo_env* e) {
    mm_t* m = (mm_t*)e->impl;
    double S = e->scale;
    m->has_info = 0;
    m->t = 0;
    if (m->variant == MM_ENDLESS) {
        m->max_episode_steps = m->max_steps;
    } else {
        /* calc_max_episode_steps with (delay, duration) passed into the (duration, delay) slots
         * (mortar_mayhem_grid.py:229-233 -> pygame_assets.py:420-436) */
        int cc = (int)mm_max(m->command_count, m->n_command_count);
        int sd = (int)mm_max(m->show_duration, m->n_show_duration), sl = (int)mm_max(m->show_delay, m->n_show_delay);
        if (MM_IS_B(m->variant)) sd = sl = 0; /* calc_max_episode_steps(cc, 0, 0, ...) (mortar_mayhem_b_grid.py:149-153) */
        int exec_duration = (int)mm_max(m->explosion_delay, m->n_explosion_delay);
        int exec_delay = (int)mm_max(m->explosion_duration, m->n_explosion_duration);
        int clue = (sd + sl) * cc, act = (exec_duration + exec_delay) * cc;
        act = act - exec_delay + 1;
        m->max_episode_steps = clue + act;
    }
    e->ep_sum = 0.0;
    e->ep_len = 0;

    /* MortarArena(SCALE, arena_size); arena.rect.center = (screen_dim // 2, screen_dim // 2) */
    m->N = m->variant == MM_ENDLESS ? 6 : m->arena_size;
    m->tile_dim = 56 * S;
    double rect_dim = m->tile_dim * m->N;
    mgo_surf_free(m->arena_surf);
    m->arena_surf = mgo_surf_new((int)rect_dim, (int)rect_dim);
    m->arena_rect.x = m->arena_rect.y = 0;
    m->arena_rect.w = m->arena_rect.h = (int)rect_dim;
    m->local_cx = mgo_rect_cx(&m->arena_rect);
    m->local_cy = mgo_rect_cy(&m->arena_rect);
    m->tiles_on = 0;
    for (int i = 0; i < m->N; i++)
        for (int j = 0; j < m->N; j++) mm_draw_tile(m, S, i, j, 0);
    mgo_rect_set_center(&m->arena_rect, e->screen_dim / 2, e->screen_dim / 2);

    /* the previous agent's rect lives on if the displayed rect referenced it (Endless quirk, App. D.5) */
    if (m->have_disp && m->disp_rect_is_agent) {
        m->disp_rect_store = m->agent.rect;
        m->disp_rect_is_agent = 0;
    }

    double translate_x = mgo_rect_cx(&m->arena_rect) - m->local_cx + floor(m->tile_dim / 2);
    double translate_y = mgo_rect_cy(&m->arena_rect) - m->local_cy + floor(m->tile_dim / 2);
    if (MM_IS_GRID(m->variant)) {
        int tile = (int)mgo_integers(&e->rng, 0, (int64_t)m->N * m->N);
        double sx = m->tile_dim * (tile / m->N), sy = m->tile_dim * (tile % m->N);
        double ax = sx + translate_x, ay = sy + translate_y;
        m->norm_x = (int)floor((ax - m->arena_rect.x) / m->tile_dim);
        m->norm_y = (int)floor((ay - m->arena_rect.y) / m->tile_dim);
        /* GridCharacterController(SCALE, normalized_position, arena.to_grid()) */
        mgo_agent_init(&m->agent, 0, S, 0);
        m->agent.grid_n = m->N;
        m->agent.grid_x0 = translate_x;
        m->agent.grid_y0 = translate_y;
        m->agent.grid_step = m->tile_dim;
        m->agent.gx = m->norm_x;
        m->agent.gy = m->norm_y;
        mgo_rect_set_center(&m->agent.rect, translate_x + m->tile_dim * m->norm_x, translate_y + m->tile_dim * m->norm_y);
    } else if (m->variant == MM_FREE) {
        mgo_agent_init(&m->agent, m->agent_speed, m->agent_scale, 0);
        int tile = (int)mgo_integers(&e->rng, 0, (int64_t)m->N * m->N);
        double sx = m->tile_dim * (tile / m->N), sy = m->tile_dim * (tile % m->N);
        mgo_rect_set_center(&m->agent.rect, sx + translate_x, sy + translate_y);
        mm_normalize(m, mgo_rect_cx(&m->agent.rect), mgo_rect_cy(&m->agent.rect), &m->norm_x, &m->norm_y);
    } else {
        mgo_agent_init(&m->agent, m->agent_speed, m->agent_scale, 0);
        int tile = (int)mgo_integers(&e->rng, 0, (int64_t)m->N * m->N);
        double sx = m->tile_dim * (tile / m->N), sy = m->tile_dim * (tile % m->N);
        /* offset = integers(-8*SCALE, 8*SCALE, 2): float bounds truncate toward zero */
        int64_t lo = (int64_t)(-8 * S), hi = (int64_t)(8 * S);
        int ox = (int)mgo_integers(&e->rng, lo, hi);
        int oy = (int)mgo_integers(&e->rng, lo, hi);
        mgo_rect_set_center(&m->agent.rect, sx + translate_x + ox, sy + translate_y + oy);
        mm_normalize(m, mgo_rect_cx(&m->agent.rect), mgo_rect_cy(&m->agent.rect), &m->norm_x, &m->norm_y);
    }

    /* command sequence */
    m->num_commands = 0;
    if (m->variant == MM_ENDLESS) {
        for (int i = 0; i < m->initial_command_count; i++) mm_cmd_push(m, (int)mgo_integers(&e->rng, 0, m->allowed_commands));
    } else {
        int n = (int)mm_choice(e, m->command_count, m->n_command_count);
        int sx = m->norm_x, sy = m->norm_y;
        for (int i = 0; i < n; i++) {
            int valid[9], nv = 0;
            for (int c = 0; c < m->allowed_commands; c++) { /* _get_valid_commands (:149-168) */
                int px = sx + CMD_DX[c], py = sy + CMD_DY[c];
                if (px >= 0 && px < m->arena_size && py >= 0 && py < m->arena_size) valid[nv++] = c;
            }
            int c = valid[mgo_integers(&e->rng, 0, nv)];
            mm_cmd_push(m, c);
            sx += CMD_DX[c];
            sy += CMD_DY[c];
        }
    }
    int glyph = -1;
    if (MM_IS_B(m->variant)) {
        mm_vis_clear(m); /* _command_visualization = None: every command is observable through the vector obs */
    } else {
        m->show_dur = (int)mm_choice(e, m->show_duration, m->n_show_duration);
        m->show_delay_v = (int)mm_choice(e, m->show_delay, m->n_show_delay);
        mm_gen_vis(m, m->cmds, m->num_commands, m->show_dur, m->show_delay_v);
        m->clone_pops = 0;
        glyph = mm_vis_pop(m);
    }

    if (m->variant == MM_ENDLESS) {
        m->tx = ((m->norm_x + CMD_DX[m->cmds[0]]) % 6 + 6) % 6;
        m->ty = ((m->norm_y + CMD_DY[m->cmds[0]]) % 6 + 6) % 6;
    } else {
        m->tx = m->norm_x + CMD_DX[m->cmds[0]];
        m->ty = m->norm_y + CMD_DY[m->cmds[0]];
    }
    m->cur_cmd = 0;
    m->cmd_steps = 0;
    m->verify_step = 0;
    m->total_completed = 0;
    m->expl_dur = (int)mm_choice(e, m->explosion_duration, m->n_explosion_duration);
    m->expl_delay = (int)mm_choice(e, m->explosion_delay, m->n_explosion_delay);

    /* reset frame always shows get_rotated_sprite(0) at the NEW agent's rect */
    mm_draw_frame(e, m, m->agent.sprites[0], &m->agent.rect, glyph);
    e->reward = 0;
    e->done = 0;
    e->gt[0] = m->tx / 5.0;
    e->gt[1] = m->ty / 5.0;
}

static void mm_step(mgo_env* e, const int action[2]) {
    mm_t* m = (mm_t*)e->impl;
    double S = e->scale;
    double reward = 0;
    int done = 0, success = 0, glyph = -1;

    if (mm_vis_nonempty(m)) {
        glyph = mm_vis_pop(m);
        if (m->variant == MM_ENDLESS) {
            if (!m->have_disp) mm_set_disp(m, 0, 1); /* only if (surface, rect) are still None (:277-278) */
        } else {
            mm_set_disp(m, 0, 1); /* get_rotated_sprite(0) every display step (:297) */
        }
    } else {
        if (MM_IS_GRID(m->variant)) {
            mgo_agent_step_grid(&m->agent, action[0]);
        } else if (m->variant == MM_FREE || m->variant == MM_B_FREE) {
            mgo_agent_step(&m->agent, action, &m->arena_rect);
        } else {
            mgo_agent_step_wrap(&m->agent, action, &m->arena_rect);
        }
        mm_set_disp(m, m->agent.rotation / 45, 1);
        mm_normalize(m, mgo_rect_cx(&m->agent.rect), mgo_rect_cy(&m->agent.rect), &m->norm_x, &m->norm_y);

        int verify = (m->cmd_steps % m->expl_delay == 0) && m->cmd_steps > 0;
        if (verify && !m->tiles_on) {
            if (m->cur_cmd < m->num_commands) {
                m->cur_cmd += 1;
                mm_toggle_tiles(m, S, 1, m->tx, m->ty, m->visual_feedback);
                if (m->norm_x == m->tx && m->norm_y == m->ty) {
                    reward += m->reward_command_success;
                    if (m->variant == MM_ENDLESS) {
                        m->total_completed += 1;
                        if (m->cur_cmd == m->num_commands) reward += m->reward_new_command_success;
                    }
                } else {
                    done = 1;
                    reward += m->reward_command_failure;
                }
            }
            if (m->cur_cmd >= m->num_commands) {
                if (m->variant == MM_ENDLESS) {
                    /* append one command and replay the display for the new command only (:311-321) */
                    int nc = (int)mgo_integers(&e->rng, 0, m->allowed_commands);
                    mm_cmd_push(m, nc);
                    m->cur_cmd = 0;
                    m->cmd_steps = 0;
                    m->verify_step = 0;
                    mm_gen_vis(m, &nc, 1, m->show_dur, m->show_delay_v);
                    m->clone_pops = 0;
                } else {
                    done = 1;
                    success = 1;
                    reward += m->reward_episode_success;
                }
            }
            m->cmd_steps = 1;
        }
        if (m->tiles_on) {
            if (m->verify_step % m->expl_dur == 0 && m->verify_step > 0) {
                mm_toggle_tiles(m, S, 0, 0, 0, m->visual_feedback);
                m->verify_step = 0;
                if (m->cur_cmd < m->num_commands) {
                    int c = m->cmds[m->cur_cmd];
                    if (m->variant == MM_ENDLESS) {
                        m->tx = ((m->tx + CMD_DX[c]) % 6 + 6) % 6;
                        m->ty = ((m->ty + CMD_DY[c]) % 6 + 6) % 6;
                    } else {
                        m->tx += CMD_DX[c];
                        m->ty += CMD_DY[c];
                    }
                }
            } else {
                if (!(m->norm_x == m->tx && m->norm_y == m->ty)) {
                    done = 1;
                    reward = m->reward_command_failure; /* overwrite, not += (:348) */
                }
                m->verify_step += 1;
            }
        } else {
            m->cmd_steps += 1;
        }
    }

    if (m->variant == MM_ENDLESS) {
        m->t += 1;
        if (m->t == m->max_episode_steps) done = 1;
    }
    e->ep_sum += reward;
    e->ep_len += 1;
    m->has_info = done;
    if (done) {
        m->info_reward = e->ep_sum;
        m->info_length = e->ep_len;
        m->info_success = success;
        if (m->variant == MM_ENDLESS) {
            m->info_commands_completed = m->total_completed;
            m->info_max_command_sequence = m->num_commands > 1 ? m->num_commands - 1 : 0;
        } else {
            m->info_commands_completed = (double)(m->cur_cmd - 1 + success) / (double)m->num_commands;
        }
    }
    e->gt[0] = m->tx / 5.0;
    e->gt[1] = m->ty / 5.0;
    mm_draw_frame(e, m, m->disp_surf, mm_disp_rect(m), glyph);
    e->reward = reward;
    e->done = done;
}

static int mm_set_option(mgo_env* e, const char* k, const double* v, int n) {
    mm_t* m = (mm_t*)e->impl;
#define SCALAR(name, field) if (!strcmp(k, name)) { m->field = v[0]; return 0; }
#define ISCALAR(name, field) if (!strcmp(k, name)) { m->field = (int)v[0]; return 0; }
#define LIST(name, field) if (!strcmp(k, name)) return mgo_opt_list(m->field, &m->n_##field, MM_MAXLIST, v, n);
    SCALAR("agent_scale", agent_scale)
    ISCALAR("allowed_commands", allowed_commands)
    LIST("command_show_duration", show_duration)
    LIST("command_show_delay", show_delay)
    LIST("explosion_duration", explosion_duration)
    LIST("explosion_delay", explosion_delay)
    ISCALAR("visual_feedback", visual_feedback)
    SCALAR("reward_command_failure", reward_command_failure)
    SCALAR("reward_command_success", reward_command_success)
    if (m->variant == MM_ENDLESS) {
        ISCALAR("max_steps", max_steps)
        SCALAR("agent_speed", agent_speed)
        ISCALAR("initial_command_count", initial_command_count)
        SCALAR("reward_new_command_success", reward_new_command_success)
    } else {
        ISCALAR("arena_size", arena_size)
        LIST("command_count", command_count)
        SCALAR("reward_episode_success", reward_episode_success)
        if (m->variant == MM_FREE || m->variant == MM_B_FREE) SCALAR("agent_speed", agent_speed)
    }
#undef SCALAR
#undef ISCALAR
#undef LIST
    return -1;
}

static double mm_get(mgo_env* e, const char* f, int* ok) {
    mm_t* m = (mm_t*)e->impl;
    *ok = 1;
#define F(name, expr) if (!strcmp(f, name)) return (double)(expr);
    F("ax", mgo_rect_cx(&m->agent.rect)) F("ay", mgo_rect_cy(&m->agent.rect)) F("arot", m->agent.rotation)
    F("disp_sprite", m->have_disp ? m->disp_sprite : -1)
    F("disp_x", m->have_disp ? mgo_rect_cx(mm_disp_rect(m)) : -1)
    F("disp_y", m->have_disp ? mgo_rect_cy(mm_disp_rect(m)) : -1)
    F("glyph", m->glyph) F("cur_cmd", m->cur_cmd) F("cmd_steps", m->cmd_steps) F("verify_step", m->verify_step)
    F("tiles_on", m->tiles_on) F("tx", m->tx) F("ty", m->ty) F("vis_len", m->vis_len - m->vis_head)
    F("num_commands", m->num_commands) F("nx", m->norm_x) F("ny", m->norm_y)
    F("expl_dur", m->expl_dur) F("expl_delay", m->expl_delay) F("max_episode_steps", m->max_episode_steps)
    F("total_completed", m->total_completed) F("t", m->t) F("show_dur", m->show_dur) F("show_delay", m->show_delay_v)
    F("gt0", e->gt[0]) F("gt1", e->gt[1])
    if (m->has_info) {
        F("info_reward", m->info_reward) F("info_length", m->info_length)
        if (m->variant != MM_ENDLESS) { F("info_success", m->info_success) }
        F("info_commands_completed", m->info_commands_completed)
        if (m->variant == MM_ENDLESS) { F("info_max_command_sequence", m->info_max_command_sequence) }
    }
#undef F
    *ok = 0;
    return NAN;
}

static int mm_get_list(mgo_env* e, const char* name, double* out, int cap) {
    mm_t* m = (mm_t*)e->impl;
    if (!strcmp(name, "cmds")) {
        int n = m->num_commands < cap ? m->num_commands : cap;
        for (int i = 0; i < n; i++) out[i] = m->cmds[i];
        return m->num_commands;
    }
    if (!strcmp(name, "vec") && MM_IS_B(m->variant)) { /* obs["vector_observation"], float32[20 * 9] */
        for (int i = 0; i < 180 && i < cap; i++) out[i] = 0.0;
        for (int c = 0; c < m->num_commands && c < 20; c++)
            if (9 * c + CMD_ONE_HOT[m->cmds[c]] < cap) out[9 * c + CMD_ONE_HOT[m->cmds[c]]] = 1.0;
        return 180;
    }
    return -1;
}

static void mm_destroy(mgo_env* e) {
    mm_t* m = (mm_t*)e->impl;
    mgo_agent_free(&m->agent);
    mgo_surf_free(m->arena_surf);
    mgo_surf_free(m->disp_surf);
    for (int i = 0; i < 10; i++) mgo_surf_free(m->glyph_surf[i]);
    free(m->cmds);
    free(m->vis);
    free(m);
}

/* expert hook (mgo_env.h): wait while commands are shown and while the tiles are on, else move to the target tile */
static void mm_expert(mgo_env* e, int a[2]) {
    mm_t* m = (mm_t*)e->impl;
    a[0] = a[1] = 0;
    if (mm_vis_nonempty(m) || m->tiles_on) return;
    if (MM_IS_GRID(m->variant)) {
        int nx = m->agent.gx, ny = m->agent.gy, rot = m->agent.rotation;
        if (nx == m->tx && ny == m->ty) return;
        int want = m->tx > nx ? 270 : (m->tx < nx ? 90 : (m->ty < ny ? 0 : 180));
        if (rot == want) { a[0] = 3; return; }
        int d = ((want - rot) % 360 + 360) % 360;
        a[0] = (d == 90 || d == 180) ? 1 : 2;
        return;
    }
    int cx = (int)(m->arena_rect.x + m->tile_dim * m->tx + floor(m->tile_dim / 2));
    int cy = (int)(m->arena_rect.y + m->tile_dim * m->ty + floor(m->tile_dim / 2));
    int dx = cx - mgo_rect_cx(&m->agent.rect), dy = cy - mgo_rect_cy(&m->agent.rect);
    if (m->variant == MM_ENDLESS) { /* the arena wraps: the shorter way round */
        int w = e->screen_dim, h = w / 2;
        dx = ((dx + h) % w + w) % w - h;
        dy = ((dy + h) % w + w) % w - h;
    }
    int slack = (int)m->agent.speed;
    a[0] = abs(dx) < slack ? 0 : (dx < 0 ? 1 : 2);
    a[1] = abs(dy) < slack ? 0 : (dy < 0 ? 1 : 2);
}

static const mgo_vtbl MM_VT[5] = {
    {"MortarMayhem-Grid-v0", 1, 0, mm_set_option, mm_reset, mm_step, mm_get, mm_get_list, mm_destroy, mm_debug, mm_scene, mm_expert},
    {"MortarMayhem-v0", 0, 0, mm_set_option, mm_reset, mm_step, mm_get, mm_get_list, mm_destroy, mm_debug, mm_scene, mm_expert},
    {"Endless-MortarMayhem-v0", 0, 2, mm_set_option, mm_reset, mm_step, mm_get, mm_get_list, mm_destroy, mm_debug, mm_scene, mm_expert},
    {"MortarMayhemB-Grid-v0", 1, 0, mm_set_option, mm_reset, mm_step, mm_get, mm_get_list, mm_destroy, mm_debug, mm_scene, mm_expert},
    {"MortarMayhemB-v0", 0, 0, mm_set_option, mm_reset, mm_step, mm_get, mm_get_list, mm_destroy, mm_debug, mm_scene, mm_expert},
};

int mgo_mortar_create(mgo_env* e, int variant) {
    mm_t* m = (mm_t*)calloc(1, sizeof(mm_t));
    double S = e->scale;
    m->variant = variant;
    e->vt = &MM_VT[variant];
    e->impl = m;
    /* default_reset_parameters, evaluated with the module constant SCALE */
    m->agent_scale = 1.0 * S;
    m->agent_speed = 12.0 * S;
    m->show_duration[0] = 3; m->n_show_duration = 1;
    m->show_delay[0] = 1; m->n_show_delay = 1;
    m->visual_feedback = 1;
    m->reward_command_failure = 0.0;
    m->reward_command_success = 0.1;
    if (MM_IS_GRID(variant)) {
        m->arena_size = 5; m->allowed_commands = 5;
        m->command_count[0] = 10; m->n_command_count = 1;
        m->explosion_duration[0] = 2; m->n_explosion_duration = 1;
        m->explosion_delay[0] = 6; m->n_explosion_delay = 1;
    } else {
        m->arena_size = 5; m->allowed_commands = 9;
        m->command_count[0] = 10; m->n_command_count = 1;
        m->explosion_duration[0] = 6; m->n_explosion_duration = 1;
        m->explosion_delay[0] = 18; m->n_explosion_delay = 1;
        m->max_steps = -1;
        m->initial_command_count = 1;
    }
    for (int g = 0; g < 10; g++) m->glyph_surf[g] = mm_make_glyph(g == 9 ? -1 : g, S);
    return 0;
}
