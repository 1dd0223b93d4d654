/* oracle/mgo_mystery.c -- TEST INFRASTRUCTURE (CPU oracle), not product code.
 *
 * Restatement of the Mystery Path family:
 *   MysteryPath-v0          memory_gym/mystery_path.py          reset :130-200  step :202-276
 *   MysteryPath-Grid-v0     memory_gym/mystery_path_grid.py     reset :129-199  step :201-277 (grid locomotion)
 *   Endless-MysteryPath-v0  memory_gym/endless_mystery_path.py  reset :195-280  step :282-444  drawing :111-160
 * plus Node :438-493, EndlessMysteryPath :495-604, MysteryPath (noisy A*) :606-736 and the icy-tile helpers
 * :780-817 of memory_gym/pygame_assets.py.
 * Pinned by tests/golden/logic_{MysteryPath_v0,Endless_MysteryPath_v0}.npz (logic) and docs/assets/emp_0.gif
 * (pixels, SCALE 1.0).
 */
#include "mgo_env.h"

#define G 7 /* grid_dim */
#define MP_MAXLIST 256

typedef struct {
    int x, y, rvis, svis;
} pnode;

typedef struct {
    int endless, grid;
    /* reset parameters */
    int max_steps, show_origin, show_goal, visual_feedback, show_past_path, show_background, show_stamina, stamina_level;
    double agent_scale, agent_speed, camera_offset_scale;
    double cardinal[MP_MAXLIST];
    int n_cardinal;
    double reward_goal, reward_fall_off, reward_path_progress, reward_path_progress_dense, reward_step;
    /* geometry */
    double tile_dim; /* MP: screen_dim / 7 (float); EMP: screen_dim // 7 (int) */
    mgo_surf *path_surf, *cross, *column_surf, *stamina_surf;
    mgo_rect cross_rect;
    /* agent */
    mgo_agent agent;
    int disp_sprite;
    int norm_x, norm_y;
    /* MP state */
    int sx, sy, ex, ey, off, fails, t;
    pnode path[G * G + 64]; /* MP: end-first, like MysteryPath.path */
    int path_len;
    int walls[G * G][2], n_walls;
    /* EMP state */
    pnode* epath; /* flat, start-first */
    int epath_len, epath_cap;
    int seg_start[4096], num_segments;
    int have_start, start_y, end_x, end_y;
    int cur_node, cur_seg, stamina, max_x, tiles_visited;
    double camera_offset, camera_x, bg_scroll;
    int agent_draw_x;
    int falloff[256][2], n_falloff;
    int td[3];
    /* info */
    int has_info, info_length, info_success;
    double info_reward;
} mp_t;

/* ---- MysteryPath.__init__ (pygame_assets.py:606-724): walls + noisy A* ------------------------------------ */
typedef struct {
    int wall;
    double f, g, h;
    int prev; /* flat index or -1 */
} anode;

static int mp_nb(int idx, int k) { /* Node.add_neighbors order: x+1, x-1, y+1, y-1 (:465-472) */
    int x = idx / G, y = idx % G;
    switch (k) {
        case 0: return x < G - 1 ? (x + 1) * G + y : -1;
        case 1: return x > 0 ? (x - 1) * G + y : -1;
        case 2: return y < G - 1 ? x * G + (y + 1) : -1;
        default: return y > 0 ? x * G + (y - 1) : -1;
    }
}
static int mp_diag(int idx, int k) { /* (:474-481) */
    int x = idx / G, y = idx % G;
    switch (k) {
        case 0: return (x < G - 1 && y < G - 1) ? (x + 1) * G + y + 1 : -1;
        case 1: return (x > 0 && y > 0) ? (x - 1) * G + y - 1 : -1;
        case 2: return (x < G - 1 && y > 0) ? (x + 1) * G + y - 1 : -1;
        default: return (x > 0 && y < G - 1) ? (x - 1) * G + y + 1 : -1;
    }
}
static int mp_is_nb(int a, int b) {
    for (int k = 0; k < 4; k++)
        if (mp_nb(a, k) == b) return 1;
    return 0;
}

/* returns path length (end-first flat indices in out[]), or -1 ("No valid path found") */
static int mp_generate(mgo_rng* rng, int sx, int sy, int ex, int ey, int* out, int walls[][2], int* n_walls) {
    anode n[G * G];
    int nw = 0;
    for (int i = 0; i < G; i++)
        for (int j = 0; j < G; j++) {
            anode* a = &n[i * G + j];
            a->wall = 0;
            a->f = a->g = a->h = 0;
            a->prev = -1;
            if (i > 0 && i < G - 2 && j > 0 && j < G - 2) {
                if (mgo_integers(rng, 0, 100) < 33) {
                    a->wall = 1;
                    walls[nw][0] = i;
                    walls[nw++][1] = j;
                }
            }
        }
    int start = sx * G + sy, end = ex * G + ey;
    int outer[G * G], n_outer = 0;
    for (int i = 0; i < G; i++)
        for (int j = 0; j < G; j++) {
            if (!(i == 0 || i == G - 1 || j == 0 || j == G - 1)) continue;
            int idx = i * G + j;
            if (idx == start || idx == end) continue;
            if (mp_is_nb(start, idx) || mp_is_nb(end, idx)) continue;
            int adj = 0;
            for (int k = 0; k < 4 && !adj; k++) {
                int q = mp_nb(idx, k);
                if (q >= 0 && n[q].wall) adj = 1;
            }
            for (int k = 0; k < 4 && !adj; k++) {
                int q = mp_diag(idx, k);
                if (q >= 0 && n[q].wall) adj = 1;
            }
            if (!adj) outer[n_outer++] = idx;
        }
    int n_iter = mgo_choice_index(rng, 2) == 0 ? 4 : 8; /* rng.choice([4, 8]) */
    for (int it = 0; it < n_iter; it++) {
        if (n_outer > 0) {
            int k = mgo_choice_index(rng, n_outer);
            int idx = outer[k];
            n[idx].wall = 1;
            walls[nw][0] = idx / G;
            walls[nw++][1] = idx % G;
            for (int q = k; q < n_outer - 1; q++) outer[q] = outer[q + 1];
            n_outer--;
        }
    }
    *n_walls = nw;
    int open[G * G * 4], n_open = 0, closed[G * G], n_closed = 0;
    open[n_open++] = start;
    for (;;) {
        if (n_open == 0) return -1;
        int w = 0;
        for (int i = 0; i < n_open; i++)
            if (n[open[i]].f < n[open[w]].f) { /* w stays 0 until the first strictly better node, then break (:686-689) */
                w = i;
                break;
            }
        int cur = open[w];
        if (cur == end) {
            int len = 0, t = cur;
            out[len++] = end;
            while (n[t].prev >= 0) {
                out[len++] = n[t].prev;
                t = n[t].prev;
            }
            return len;
        }
        for (int q = w; q < n_open - 1; q++) open[q] = open[q + 1]; /* open_set.remove(current) */
        n_open--;
        closed[n_closed++] = cur;
        for (int k = 0; k < 4; k++) {
            int nb = mp_nb(cur, k);
            if (nb < 0) continue;
            int in_closed = 0;
            for (int q = 0; q < n_closed; q++)
                if (closed[q] == nb) in_closed = 1;
            if (in_closed || n[nb].wall) continue;
            double g = n[cur].g + (double)mgo_integers(rng, 1, 9);
            int new_path = 0, in_open = 0;
            for (int q = 0; q < n_open; q++)
                if (open[q] == nb) in_open = 1;
            if (in_open) {
                if (g < n[nb].g) new_path = 1; /* `neighbor.g = g` typo: g_cost is NOT updated (:711-713) */
            } else {
                n[nb].g = g;
                new_path = 1;
                open[n_open++] = nb;
            }
            if (new_path) {
                int ax = nb / G, ay = nb % G;
                n[nb].h = sqrt((double)((ax - ex) * (ax - ex)) + (double)(abs(ay - ey) * abs(ay - ey)));
                n[nb].f = n[nb].g + n[nb].h;
                n[nb].prev = cur;
            }
        }
    }
}

/* fall-off cross (mystery_path.py:175-182) */
static mgo_surf* mp_make_cross(double S) {
    double dim = 40 * S;
    mgo_surf* s = mgo_surf_new((int)dim, (int)dim);
    mgo_fill(s, 0);
    mgo_set_colorkey(s, 0);
    mgo_draw_line(s, MGO_RGB(255, 0, 0), 0, 0, (int)(dim - 1), (int)(dim - 1), (int)(12 * S));
    mgo_draw_line(s, MGO_RGB(255, 0, 0), (int)(dim - 1), 0, 0, (int)(dim - 1), (int)(12 * S));
    mgo_set_alpha(s, 0);
    return s;
}

/* ============================================ MysteryPath-v0 ============================================ */
static void mpf_reset(mgo_env* e) {
    mp_t* m = (mp_t*)e->impl;
    double S = e->scale;
    m->has_info = 0;
    m->t = 0;
    e->ep_sum = 0;
    e->ep_len = 0;
    int cardinal = (int)m->cardinal[mgo_choice_index(&e->rng, m->n_cardinal)];
    if (cardinal == 0) {
        m->sx = 0; m->sy = (int)mgo_integers(&e->rng, 0, G);
        m->ex = G - 1; m->ey = (int)mgo_integers(&e->rng, 0, G);
    } else if (cardinal == 1) {
        m->sx = G - 1; m->sy = (int)mgo_integers(&e->rng, 0, G);
        m->ex = 0; m->ey = (int)mgo_integers(&e->rng, 0, G);
    } else if (cardinal == 2) {
        m->sx = (int)mgo_integers(&e->rng, 0, G); m->sy = 0;
        m->ex = (int)mgo_integers(&e->rng, 0, G); m->ey = G - 1;
    } else {
        m->sx = (int)mgo_integers(&e->rng, 0, G); m->sy = G - 1;
        m->ex = (int)mgo_integers(&e->rng, 0, G); m->ey = 0;
    }
    int idx[G * G];
    int len = mp_generate(&e->rng, m->sx, m->sy, m->ex, m->ey, idx, m->walls, &m->n_walls);
    if (len < 0) {
        snprintf(e->err, sizeof(e->err), "No valid path found");
        len = 0;
    }
    m->path_len = len;
    for (int i = 0; i < len; i++) m->path[i] = (pnode){idx[i] / G, idx[i] % G, 0, 0};
    /* path surface: black unless show_goal / show_origin (MysteryPath.draw_to_surface :738-762) */
    mgo_fill(m->path_surf, 0);
    for (int i = 0; i < len; i++) {
        int px = (int)(m->path[i].x * m->tile_dim), py = (int)(m->path[i].y * m->tile_dim), d = (int)m->tile_dim;
        if (i == 0 && m->show_goal) mgo_draw_rect(m->path_surf, MGO_RGB(0, 255, 0), px, py, d, d, 0);
        else if (i == len - 1 && m->show_origin) mgo_draw_rect(m->path_surf, MGO_RGB(0, 0, 255), px, py, d, d, 0);
    }
    mgo_surf_free(m->cross);
    m->cross = mp_make_cross(S);
    m->cross_rect = (mgo_rect){0, 0, m->cross->w, m->cross->h};
    if (m->grid) { /* GridCharacterController(SCALE, start, mystery_path.to_grid(tile_dim), 0) (mystery_path_grid.py:184-189) */
        mgo_agent_init(&m->agent, 0, S, 0);
        m->agent.grid_n = G;
        m->agent.grid_x0 = m->agent.grid_y0 = floor(m->tile_dim / 2);
        m->agent.grid_step = m->tile_dim;
        m->agent.gx = m->sx;
        m->agent.gy = m->sy;
        mgo_rect_set_center(&m->agent.rect, m->agent.grid_x0 + m->tile_dim * m->sx, m->agent.grid_y0 + m->tile_dim * m->sy);
    } else {
        mgo_agent_init(&m->agent, m->agent_speed, m->agent_scale, 0);
        mgo_rect_set_center(&m->agent.rect, m->sx * m->tile_dim + m->agent.radius, m->sy * m->tile_dim + m->agent.radius);
    }
    m->disp_sprite = 0;
    m->norm_x = (int)floor(mgo_rect_cx(&m->agent.rect) / m->tile_dim);
    m->norm_y = (int)floor(mgo_rect_cy(&m->agent.rect) / m->tile_dim);
    m->off = 0;
    m->fails = 0;
    mgo_blit(e->screen, m->path_surf, 0, 0);
    mgo_blit(e->screen, m->agent.sprites[0], m->agent.rect.x, m->agent.rect.y);
    e->reward = 0;
    e->done = 0;
}

static void mpf_step(mgo_env* e, const int action[2]) {
    mp_t* m = (mp_t*)e->impl;
    double reward = 0;
    int done = 0, success = 0;
    mgo_rect screen = {0, 0, e->screen_dim, e->screen_dim};
    if (m->grid) {
        if (!m->off) {
            mgo_agent_step_grid(&m->agent, action[0]);
        } else { /* agent.reset_position(start); agent.step(0) (mystery_path_grid.py:218-221) */
            m->agent.gx = m->sx;
            m->agent.gy = m->sy;
            mgo_rect_set_center(&m->agent.rect, m->agent.grid_x0 + m->tile_dim * m->sx, m->agent.grid_y0 + m->tile_dim * m->sy);
            mgo_agent_step_grid(&m->agent, 0);
        }
    } else if (!m->off) {
        mgo_agent_step(&m->agent, action, &screen);
    } else {
        static const int noop[2] = {0, 0};
        mgo_rect_set_center(&m->agent.rect, m->sx * m->tile_dim + m->agent.radius, m->sy * m->tile_dim + m->agent.radius);
        mgo_agent_step(&m->agent, noop, &screen);
    }
    m->disp_sprite = m->agent.rotation / 45;
    m->norm_x = (int)floor(mgo_rect_cx(&m->agent.rect) / m->tile_dim);
    m->norm_y = (int)floor(mgo_rect_cy(&m->agent.rect) / m->tile_dim);
    if (m->norm_x == m->ex && m->norm_y == m->ey) {
        reward += m->reward_goal;
        done = 1;
        success = 1;
    } else {
        int on_path = 0;
        for (int i = 0; i < m->path_len; i++) {
            pnode* nd = &m->path[i];
            if (m->norm_x == nd->x && m->norm_y == nd->y) {
                on_path = 1;
                if (!nd->rvis && !(nd->x == m->sx && nd->y == m->sy) && !(nd->x == m->ex && nd->y == m->ey)) {
                    reward += m->reward_path_progress;
                    nd->rvis = 1;
                }
                break;
            }
        }
        if (!on_path) {
            reward += m->reward_fall_off;
            m->fails += 1;
            if (m->visual_feedback) mgo_set_alpha(m->cross, 255);
            m->off = 1;
        } else {
            mgo_set_alpha(m->cross, 0);
            m->off = 0;
        }
        mgo_rect_set_center(&m->cross_rect, mgo_rect_cx(&m->agent.rect), mgo_rect_cy(&m->agent.rect));
    }
    reward += m->reward_step;
    m->t += 1;
    if (m->t == m->max_steps) done = 1;
    e->ep_sum += reward;
    e->ep_len += 1;
    m->has_info = done;
    if (done) {
        m->info_reward = e->ep_sum;
        m->info_length = e->ep_len;
        m->info_success = success;
    }
    mgo_blit(e->screen, m->path_surf, 0, 0);
    mgo_blit(e->screen, m->agent.sprites[m->disp_sprite], m->agent.rect.x, m->agent.rect.y);
    mgo_blit(e->screen, m->cross, m->cross_rect.x, m->cross_rect.y);
    e->reward = reward;
    e->done = done;
}

/* scene hook (finite variants): v = {ax, ay, sprite, cross_on, cross_cx, cross_cy, sx, sy, ex, ey, show_origin, show_goal} */
static int mpf_scene(mgo_env* e, const double* v, int n) {
    mp_t* m = (mp_t*)e->impl;
    if (n < 12) return -1;
    const int d = (int)m->tile_dim;
    mgo_fill(m->path_surf, 0);
    if ((int)v[11]) mgo_draw_rect(m->path_surf, MGO_RGB(0, 255, 0), (int)((int)v[8] * m->tile_dim), (int)((int)v[9] * m->tile_dim), d, d, 0);
    if ((int)v[10]) mgo_draw_rect(m->path_surf, MGO_RGB(0, 0, 255), (int)((int)v[6] * m->tile_dim), (int)((int)v[7] * m->tile_dim), d, d, 0);
    mgo_rect_set_center(&m->agent.rect, v[0], v[1]);
    m->disp_sprite = (int)v[2] & 7;
    mgo_set_alpha(m->cross, (int)v[3] ? 255 : 0);
    mgo_rect_set_center(&m->cross_rect, v[4], v[5]);
    mgo_blit(e->screen, m->path_surf, 0, 0);
    mgo_blit(e->screen, m->agent.sprites[m->disp_sprite], m->agent.rect.x, m->agent.rect.y);
    mgo_blit(e->screen, m->cross, m->cross_rect.x, m->cross_rect.y);
    return 0;
}

/* ======================================== Endless-MysteryPath-v0 ======================================== */
static void emp_push(mp_t* m, pnode nd) {
    if (m->epath_len == m->epath_cap) {
        m->epath_cap = m->epath_cap ? m->epath_cap * 2 : 256;
        m->epath = (pnode*)realloc(m->epath, sizeof(pnode) * m->epath_cap);
    }
    m->epath[m->epath_len++] = nd;
}

/* EndlessMysteryPath.add_path_segment (pygame_assets.py:559-604) */
static void emp_add_segment(mgo_env* e, mp_t* m) {
    int sy;
    if (!m->have_start) {
        sy = (int)mgo_integers(&e->rng, 0, G);
        m->have_start = 1;
    } else {
        sy = m->end_y;
    }
    m->start_y = sy;
    m->end_x = G - 1;
    m->end_y = (int)mgo_integers(&e->rng, 0, G);
    int idx[G * G], walls[G * G][2], nw;
    int len = mp_generate(&e->rng, 0, sy, m->end_x, m->end_y, idx, walls, &nw);
    if (len < 0) {
        snprintf(e->err, sizeof(e->err), "No valid path found");
        len = 0;
    }
    int k = m->num_segments;
    int shift = k == 0 ? 0 : k * G + k;
    m->seg_start[k] = m->epath_len;
    for (int i = len - 1; i >= 0; i--) emp_push(m, (pnode){idx[i] / G + shift, idx[i] % G, 0, 0});
    m->num_segments += 1;
    emp_push(m, (pnode){m->end_x + m->num_segments + (m->num_segments - 1) * G, m->end_y, 0, 0}); /* transition node */
    m->seg_start[m->num_segments] = m->epath_len;
}

static void emp_set_direction(mgo_env* e, mp_t* m) {
    if (m->cur_node + 1 < m->epath_len) {
        int x = m->epath[m->cur_node + 1].x - m->epath[m->cur_node].x;
        int y = m->epath[m->cur_node + 1].y - m->epath[m->cur_node].y;
        if (x == 1) { m->td[0] = 1; m->td[1] = 0; m->td[2] = 0; }
        else if (y == -1) { m->td[0] = 0; m->td[1] = 1; m->td[2] = 0; }
        else if (y == 1) { m->td[0] = 0; m->td[1] = 0; m->td[2] = 1; }
    }
    for (int i = 0; i < 3; i++) e->gt[i] = m->td[i];
}

/* _draw_surfaces (endless_mystery_path.py:134-160) incl. _draw_past_path (:111-132) */
static void emp_draw(mgo_env* e, mp_t* m) {
    double S = e->scale;
    int td = (int)m->tile_dim, dim = e->screen_dim;
    if (m->show_background) {
        int ncol = (int)ceil((double)dim / td) + 2;
        for (int i = 0; i < ncol; i++) mgo_blit(e->screen, m->column_surf, (int)(i * td + m->bg_scroll - td), 0);
    } else {
        mgo_fill(e->screen, 0);
    }
    if (m->show_past_path) {
        int x = m->norm_x - 1;
        if (x >= 0) {
            int depth = (int)m->camera_offset_scale;
            int past_x = x - depth > 0 ? x - depth : 0;
            int node = m->cur_node - 1;
            while (x >= past_x && x >= 0) {
                if (node < 0) break;
                x = m->epath[node].x;
                int y = m->epath[node].y;
                double draw_x = x * td - m->camera_x;
                mgo_draw_rect(e->screen, MGO_RGB(255, 255, 255), (int)draw_x, y * td, td, td, 0);
                mgo_draw_rect(e->screen, MGO_RGB(210, 210, 210), (int)draw_x, y * td, td, td, 1);
                if (x == past_x) break;
                node -= 1;
            }
        }
    }
    mgo_blit(e->screen, m->agent.sprites[m->disp_sprite], m->agent_draw_x, m->agent.rect.y);
    if (m->show_stamina) mgo_blit(e->screen, m->stamina_surf, (int)(dim - 16 * S), 0);
    if (m->visual_feedback) mgo_blit(e->screen, m->cross, m->cross_rect.x, m->cross_rect.y);
}

static void emp_stamina_bar(mgo_env* e, mp_t* m, int use_max) {
    double S = e->scale;
    int dim = e->screen_dim, maxv = m->stamina_level;
    int st = use_max ? (m->stamina > maxv ? m->stamina : maxv) : (m->stamina < maxv ? m->stamina : maxv);
    mgo_fill(m->stamina_surf, MGO_RGB(0, 255, 0));
    int height = (int)(dim * (1 - ((double)st / maxv)));
    mgo_draw_rect(m->stamina_surf, MGO_RGB(255, 0, 0), 0, 0, (int)(16 * S), height, 0);
}

static void emp_reset(mgo_env* e) {
    mp_t* m = (mp_t*)e->impl;
    double S = e->scale;
    int td = (int)m->tile_dim;
    m->has_info = 0;
    m->t = 0;
    e->ep_sum = 0;
    e->ep_len = 0;
    m->epath_len = 0;
    m->num_segments = 0;
    m->have_start = 0;
    for (int i = 0; i < 3; i++) emp_add_segment(e, m);
    m->epath[0].rvis = 1;
    mgo_surf_free(m->cross);
    m->cross = mp_make_cross(S);
    m->cross_rect = (mgo_rect){0, 0, m->cross->w, m->cross->h};
    double cos_ = m->camera_offset_scale < 0 ? 0 : (m->camera_offset_scale > 5.5 ? 5.5 : m->camera_offset_scale);
    m->camera_offset = -td * cos_;
    m->camera_x = m->camera_offset;
    m->bg_scroll = 0;
    mgo_agent_init(&m->agent, m->agent_speed, m->agent_scale, 270);
    m->disp_sprite = 270 / 45;
    mgo_rect_set_center(&m->agent.rect, m->epath[0].x * td + m->agent.radius, m->epath[0].y * td + m->agent.radius);
    m->agent_draw_x = (int)(m->agent.rect.x - m->camera_offset);
    m->norm_x = (int)floor((double)mgo_rect_cx(&m->agent.rect) / td);
    m->norm_y = (int)floor((double)mgo_rect_cy(&m->agent.rect) / td);
    m->cur_node = 0;
    emp_set_direction(e, m);
    m->off = 0;
    m->cur_seg = 0;
    m->fails = 0;
    m->n_falloff = 0;
    m->stamina = m->stamina_level;
    m->max_x = 0;
    m->tiles_visited = 0;
    emp_stamina_bar(e, m, 1);
    emp_draw(e, m);
    e->reward = 0;
    e->done = 0;
}

static void emp_step(mgo_env* e, const int action_in[2]) {
    mp_t* m = (mp_t*)e->impl;
    int td = (int)m->tile_dim;
    int action[2] = {0, 0};
    if (action_in[0] == 1) action[0] = 2;
    else if (action_in[0] == 2) action[1] = 1;
    else if (action_in[0] == 3) action[1] = 2;
    double reward = 0;
    int done = 0;
    if (!m->off) {
        mgo_agent_step(&m->agent, action, NULL);
        m->camera_x += m->agent.vx;
        m->bg_scroll -= m->agent.vx;
        if (fabs(m->bg_scroll) >= td) {
            double remainder = fmod(fabs(m->bg_scroll), fabs(m->agent.vx));
            double sign = m->bg_scroll / fabs(m->bg_scroll);
            m->bg_scroll = remainder * sign;
        }
    } else {
        static const int noop[2] = {0, 0};
        mgo_rect_set_center(&m->agent.rect, m->epath[0].x * td + m->agent.radius, m->epath[0].y * td + m->agent.radius);
        mgo_agent_step(&m->agent, noop, NULL);
        m->camera_x = m->camera_offset;
        m->bg_scroll = 0;
    }
    m->disp_sprite = m->agent.rotation / 45;
    int cx = mgo_rect_cx(&m->agent.rect), cy = mgo_rect_cy(&m->agent.rect);
    m->norm_x = (int)floor((double)cx / td);
    m->norm_y = (int)floor((double)cy / td);
    m->cur_seg = m->norm_x / (G + 1);
    int seg = m->cur_seg;
    int s0 = m->seg_start[seg], s1 = m->seg_start[seg + 1];
    if (m->cur_seg > m->num_segments - 2) emp_add_segment(e, m);
    int on_path = 0;
    for (int i = s0; i < s1; i++) {
        pnode* nd = &m->epath[i];
        if (m->norm_x == nd->x && m->norm_y == nd->y) {
            on_path = 1;
            m->cur_node = i;
            int is_start = nd->x == m->epath[0].x && nd->y == m->epath[0].y;
            if (!nd->rvis && !is_start) {
                reward += m->reward_path_progress;
                m->tiles_visited += 1;
                nd->rvis = 1;
            }
            if (!nd->svis && !is_start) {
                reward += m->reward_path_progress_dense;
                m->stamina = m->stamina_level;
                nd->svis = 1;
            }
            break;
        }
    }
    if (!on_path) {
        reward += m->reward_fall_off;
        m->fails += 1;
        if (m->visual_feedback) mgo_set_alpha(m->cross, 255);
        m->off = 1;
        if (m->norm_x < m->max_x) {
            done = 1;
        } else {
            int found = 0;
            for (int i = 0; i < m->n_falloff; i++)
                if (m->falloff[i][0] == m->norm_x && m->falloff[i][1] == m->norm_y) {
                    done = 1;
                    found = 1;
                    break;
                }
            if (!found && m->n_falloff < 256) {
                m->falloff[m->n_falloff][0] = m->norm_x;
                m->falloff[m->n_falloff++][1] = m->norm_y;
            }
        }
        for (int i = 0; i < m->epath_len; i++) m->epath[i].svis = 0;
        m->stamina = m->stamina_level;
    } else {
        mgo_set_alpha(m->cross, 0);
        m->off = 0;
    }
    mgo_rect_set_center(&m->cross_rect, cx - m->camera_x, cy);
    reward += m->reward_step;
    m->stamina -= 1;
    if (m->stamina == 0) done = 1;
    m->t += 1;
    if (m->t == m->max_steps) done = 1;
    emp_set_direction(e, m);
    if (m->norm_x > m->max_x && on_path) m->max_x = m->norm_x;
    e->ep_sum += reward;
    e->ep_len += 1;
    m->has_info = done;
    if (done) {
        m->info_reward = e->ep_sum;
        m->info_length = e->ep_len;
    }
    emp_stamina_bar(e, m, 0);
    emp_draw(e, m);
    e->reward = reward;
    e->done = done;
}

/* _build_debug_surface, finite (mystery_path.py:103-117, mystery_path_grid.py:102-116): MysteryPath.draw_to_surface with
 * origin, goal, path AND walls shown (pygame_assets.py:738-762: path[0] green, path[-1] blue, the nodes between white,
 * wall nodes red), the agent, the fall-off cross. */
static void mpf_debug(mgo_env* e, mgo_surf* dst) {
    mp_t* m = (mp_t*)e->impl;
    int d = (int)m->tile_dim;
    mgo_fill(dst, 0);
    for (int i = 0; i < m->path_len; i++) {
        int px = (int)(m->path[i].x * m->tile_dim), py = (int)(m->path[i].y * m->tile_dim);
        uint32_t c = i == 0 ? MGO_RGB(0, 255, 0) : (i == m->path_len - 1 ? MGO_RGB(0, 0, 255) : MGO_RGB(255, 255, 255));
        mgo_draw_rect(dst, c, px, py, d, d, 0);
    }
    for (int i = 0; i < m->n_walls; i++)
        mgo_draw_rect(dst, MGO_RGB(255, 0, 0), (int)(m->walls[i][0] * m->tile_dim), (int)(m->walls[i][1] * m->tile_dim), d, d, 0);
    mgo_blit(dst, m->agent.sprites[m->disp_sprite], m->agent.rect.x, m->agent.rect.y);
    mgo_blit(dst, m->cross, m->cross_rect.x, m->cross_rect.y);
}

/* _build_debug_surface, endless (endless_mystery_path.py:162-182): scrolling background, the WHOLE path (EndlessMysteryPath.
 * surface, regenerated whenever a segment is added: white tiles, colour key black, surface alpha 200, blitted at
 * (-camera_x, 0)), the agent, the fall-off cross, the stamina bar (always, whatever show_stamina says). */
static void emp_debug(mgo_env* e, mgo_surf* dst) {
    mp_t* m = (mp_t*)e->impl;
    double S = e->scale;
    int td = (int)m->tile_dim, dim = e->screen_dim;
    mgo_fill(dst, 0);
    if (m->show_background) {
        int ncol = (int)ceil((double)dim / td) + 2;
        for (int i = 0; i < ncol; i++) mgo_blit(dst, m->column_surf, (int)(i * td + m->bg_scroll - td), 0);
    }
    mgo_surf* tile = mgo_surf_new(td, td);
    mgo_fill(tile, MGO_RGB(255, 255, 255));
    mgo_set_alpha(tile, 200);
    int ox = (int)(-m->camera_x);
    /* a tile may appear twice in the path list only at segment joints (the transition node is its own tile): draw each cell once */
    for (int i = 0; i < m->epath_len; i++) {
        int dup = 0;
        for (int j = i - 1; j >= 0 && j >= i - 2 * (G + 1); j--)
            if (m->epath[j].x == m->epath[i].x && m->epath[j].y == m->epath[i].y) dup = 1;
        int px = m->epath[i].x * td + ox;
        if (!dup && px > -td && px < dim) mgo_blit(dst, tile, px, m->epath[i].y * td);
    }
    mgo_surf_free(tile);
    mgo_blit(dst, m->agent.sprites[m->disp_sprite], m->agent_draw_x, m->agent.rect.y);
    mgo_blit(dst, m->cross, m->cross_rect.x, m->cross_rect.y);
    mgo_blit(dst, m->stamina_surf, (int)(dim - 16 * S), 0);
}

static int mp_set_option(mgo_env* e, const char* k, const double* v, int n) {
    mp_t* m = (mp_t*)e->impl;
#define D(name, field) if (!strcmp(k, name)) { m->field = v[0]; return 0; }
#define I(name, field) if (!strcmp(k, name)) { m->field = (int)v[0]; return 0; }
    I("max_steps", max_steps) D("agent_scale", agent_scale)
    if (!m->grid) { D("agent_speed", agent_speed) }
    I("show_origin", show_origin) I("visual_feedback", visual_feedback)
    D("reward_fall_off", reward_fall_off) D("reward_path_progress", reward_path_progress) D("reward_step", reward_step)
    if (m->endless) {
        I("show_past_path", show_past_path) I("show_background", show_background) I("show_stamina", show_stamina)
        D("camera_offset_scale", camera_offset_scale) I("stamina_level", stamina_level)
        D("reward_path_progress_dense", reward_path_progress_dense)
    } else {
        if (!strcmp(k, "cardinal_origin_choice")) return mgo_opt_list(m->cardinal, &m->n_cardinal, MP_MAXLIST, v, n);
        I("show_goal", show_goal) D("reward_goal", reward_goal)
    }
#undef D
#undef I
    return -1;
}

static double mp_get(mgo_env* e, const char* f, int* ok) {
    mp_t* m = (mp_t*)e->impl;
    *ok = 1;
#define F(name, expr) if (!strcmp(f, name)) return (double)(expr);
    F("ax", mgo_rect_cx(&m->agent.rect)) F("ay", mgo_rect_cy(&m->agent.rect)) F("arot", m->agent.rotation)
    F("disp_sprite", m->disp_sprite) F("off", m->off) F("fails", m->fails) F("t", m->t)
    F("cross_alpha", m->cross->alpha) F("cross_x", mgo_rect_cx(&m->cross_rect)) F("cross_y", mgo_rect_cy(&m->cross_rect))
    F("nx", m->norm_x) F("ny", m->norm_y)
    if (!m->endless) {
        F("disp_x", mgo_rect_cx(&m->agent.rect)) F("disp_y", mgo_rect_cy(&m->agent.rect))
        F("sx", m->sx) F("sy", m->sy) F("ex", m->ex) F("ey", m->ey)
        if (m->has_info) {
            F("info_reward", m->info_reward) F("info_length", m->info_length) F("info_success", m->info_success)
            F("info_num_fails", m->fails)
        }
    } else {
        F("rect_y", m->agent.rect.y) F("agent_draw_x", m->agent_draw_x) F("camera_x", m->camera_x) F("bg_scroll", m->bg_scroll)
        F("stamina", m->stamina) F("max_x", m->max_x) F("tiles_visited", m->tiles_visited) F("cur_seg", m->cur_seg)
        F("num_seg", m->num_segments) F("cur_nx", m->epath[m->cur_node].x) F("cur_ny", m->epath[m->cur_node].y)
        F("n_falloff", m->n_falloff) F("gt0", m->td[0]) F("gt1", m->td[1]) F("gt2", m->td[2])
        if (m->has_info) {
            F("info_reward", m->info_reward) F("info_length", m->info_length) F("info_num_fails", m->fails)
            F("info_max_x", m->max_x) F("info_tiles_visited", m->tiles_visited)
        }
    }
#undef F
    *ok = 0;
    return NAN;
}

static int mp_get_list(mgo_env* e, const char* name, double* out, int cap) {
    mp_t* m = (mp_t*)e->impl;
    int n = 0;
#define PUT(v) do { if (n < cap) out[n] = (v); n++; } while (0)
    if (!m->endless) {
        if (!strcmp(name, "path")) { for (int i = 0; i < m->path_len; i++) { PUT(m->path[i].x); PUT(m->path[i].y); } return n; }
        if (!strcmp(name, "visited")) { for (int i = 0; i < m->path_len; i++) PUT(m->path[i].rvis); return n; }
        if (!strcmp(name, "walls")) { for (int i = 0; i < m->n_walls; i++) { PUT(m->walls[i][0]); PUT(m->walls[i][1]); } return n; }
    } else {
        if (!strcmp(name, "path")) { for (int i = 0; i < m->epath_len; i++) { PUT(m->epath[i].x); PUT(m->epath[i].y); } return n; }
        if (!strcmp(name, "seglen")) { for (int i = 0; i < m->num_segments; i++) PUT(m->seg_start[i + 1] - m->seg_start[i]); return n; }
        if (!strcmp(name, "rvis")) { for (int i = 0; i < m->epath_len; i++) PUT(m->epath[i].rvis); return n; }
        if (!strcmp(name, "svis")) { for (int i = 0; i < m->epath_len; i++) PUT(m->epath[i].svis); return n; }
    }
#undef PUT
    return -1;
}

static void mp_destroy(mgo_env* e) {
    mp_t* m = (mp_t*)e->impl;
    mgo_agent_free(&m->agent);
    mgo_surf_free(m->path_surf); mgo_surf_free(m->cross); mgo_surf_free(m->column_surf); mgo_surf_free(m->stamina_surf);
    free(m->epath);
    free(m);
}

/* expert hooks (mgo_env.h): follow the path */
static int mp_toward(int d) { return d == 0 ? 0 : (d < 0 ? 1 : 2); }
static void mpf_expert(mgo_env* e, int a[2]) {
    mp_t* m = (mp_t*)e->impl;
    a[0] = a[1] = 0;
    int idx = -1;
    for (int i = 0; i < m->path_len; i++)
        if (m->path[i].x == m->norm_x && m->path[i].y == m->norm_y) { idx = i; break; }
    if (idx <= 0) return; /* off the path (the next step puts the agent back) or on the goal */
    const pnode* nx = &m->path[idx - 1]; /* end-first list */
    if (m->grid) {
        int dx = nx->x - m->norm_x, dy = nx->y - m->norm_y, rot = m->agent.rotation;
        int want = dx > 0 ? 270 : (dx < 0 ? 90 : (dy < 0 ? 0 : 180));
        if (rot == want) { a[0] = 3; return; }
        int d = ((want - rot) % 360 + 360) % 360;
        a[0] = (d == 90 || d == 180) ? 1 : 2;
        return;
    }
    int half = (int)floor(m->tile_dim / 2);
    a[0] = mp_toward((int)(nx->x * m->tile_dim) + half - mgo_rect_cx(&m->agent.rect));
    a[1] = mp_toward((int)(nx->y * m->tile_dim) + half - mgo_rect_cy(&m->agent.rect));
}
static void emp_expert(mgo_env* e, int a[2]) {
    mp_t* m = (mp_t*)e->impl;
    a[0] = a[1] = 0;
    int k = m->cur_node, td = (int)m->tile_dim;
    if (m->off || k + 1 >= m->epath_len) return;
    int dx = m->epath[k + 1].x * td + td / 2 - mgo_rect_cx(&m->agent.rect);
    int dy = m->epath[k + 1].y * td + td / 2 - mgo_rect_cy(&m->agent.rect);
    if (dy < 0) a[0] = 2;
    else if (dy > 0) a[0] = 3;
    else if (dx > 0) a[0] = 1;
}

static const mgo_vtbl MP_VT[3] = {
    {"MysteryPath-v0", 0, 0, mp_set_option, mpf_reset, mpf_step, mp_get, mp_get_list, mp_destroy, mpf_debug, mpf_scene, mpf_expert},
    {"Endless-MysteryPath-v0", 1, 3, mp_set_option, emp_reset, emp_step, mp_get, mp_get_list, mp_destroy, emp_debug, NULL, emp_expert},
    {"MysteryPath-Grid-v0", 1, 0, mp_set_option, mpf_reset, mpf_step, mp_get, mp_get_list, mp_destroy, mpf_debug, mpf_scene, mpf_expert},
};

int mgo_mystery_create(mgo_env* e, int variant) {
    mp_t* m = (mp_t*)calloc(1, sizeof(mp_t));
    double S = e->scale;
    int dim = e->screen_dim;
    m->endless = variant == 1;
    m->grid = variant == 2;
    e->vt = &MP_VT[variant];
    e->impl = m;
    m->agent_scale = 1.0 * S;
    m->agent_speed = 12.0 * S;
    m->show_origin = 0;
    m->visual_feedback = 1;
    m->reward_fall_off = 0.0;
    m->reward_path_progress = 0.1;
    m->reward_step = 0.0;
    if (variant != 1) {
        m->max_steps = m->grid ? 128 : 512;
        if (m->grid) m->reward_path_progress = 0.0;
        for (int i = 0; i < 4; i++) m->cardinal[i] = i;
        m->n_cardinal = 4;
        m->show_goal = 0;
        m->reward_goal = 1.0;
        m->tile_dim = (double)dim / G;
        m->path_surf = mgo_surf_new(dim, dim);
    } else {
        m->max_steps = -1;
        m->show_past_path = 1;
        m->camera_offset_scale = 5.0;
        m->stamina_level = 20;
        m->tile_dim = dim / G;
        int td = dim / G;
        /* draw_column_tile_surface / draw_icy_surface (pygame_assets.py:780-817) */
        m->column_surf = mgo_surf_new(td, td * G);
        for (int i = 0; i < G; i++) {
            mgo_draw_rect(m->column_surf, MGO_RGB(125, 177, 250), 0, i * td, td, td, 0);
            mgo_draw_rect(m->column_surf, MGO_RGB(210, 210, 210), 0, i * td, td, td, 1);
        }
        m->stamina_surf = mgo_surf_new((int)(16 * S), dim);
        m->td[0] = m->td[1] = m->td[2] = 0;
    }
    return 0;
}
