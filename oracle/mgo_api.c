/* oracle/mgo_api.c -- TEST INFRASTRUCTURE (CPU oracle), not product code.
 *
 * C entry points (loaded with ctypes by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg ONLY).  One handle == one single-instance environment, like the reference's gym.Env objects
 * (memory_gym/__init__.py:13-61); `mgo_batch_*` loops a set of handles for batched parity checks and
 * for the CPU baseline timing.
 */
#include "mgo_env.h"

int mgo_mortar_create(mgo_env* e, int variant);
int mgo_mystery_create(mgo_env* e, int variant);
int mgo_spot_create(mgo_env* e, int variant);

mgo_env* mgo_create(const char* env_id, double scale) {
    mgo_env* e = (mgo_env*)calloc(1, sizeof(mgo_env));
    e->scale = scale;
    e->screen_dim = (int)(336 * scale);
    e->screen = mgo_surf_new(e->screen_dim, e->screen_dim);
    int rc = -1;
    if (!strcmp(env_id, "MortarMayhem-Grid-v0")) rc = mgo_mortar_create(e, 0);
    else if (!strcmp(env_id, "MortarMayhem-v0")) rc = mgo_mortar_create(e, 1);
    else if (!strcmp(env_id, "Endless-MortarMayhem-v0")) rc = mgo_mortar_create(e, 2);
    else if (!strcmp(env_id, "MortarMayhemB-Grid-v0")) rc = mgo_mortar_create(e, 3);
    else if (!strcmp(env_id, "MortarMayhemB-v0")) rc = mgo_mortar_create(e, 4);
    else if (!strcmp(env_id, "MysteryPath-v0")) rc = mgo_mystery_create(e, 0);
    else if (!strcmp(env_id, "Endless-MysteryPath-v0")) rc = mgo_mystery_create(e, 1);
    else if (!strcmp(env_id, "MysteryPath-Grid-v0")) rc = mgo_mystery_create(e, 2);
    else if (!strcmp(env_id, "SearingSpotlights-v0")) rc = mgo_spot_create(e, 0);
    else if (!strcmp(env_id, "Endless-SearingSpotlights-v0")) rc = mgo_spot_create(e, 1);
    if (rc != 0) {
        mgo_surf_free(e->screen);
        free(e);
        return NULL;
    }
    return e;
}

void mgo_destroy(mgo_env* e) {
    if (!e) return;
    e->vt->destroy(e);
    mgo_surf_free(e->screen);
    free(e);
}

int mgo_set_option(mgo_env* e, const char* key, const double* v, int n) { return e->vt->set_option(e, key, v, n); }
int mgo_is_discrete(mgo_env* e) { return e->vt->discrete; }
int mgo_gt_dim(mgo_env* e) { return e->vt->gt_dim; }
int mgo_screen_dim(mgo_env* e) { return e->screen_dim; }

/* gym.Env.reset(seed): seed >= 0 -> Generator(PCG64(SeedSequence(seed))); seed < 0 -> keep the stream */
int mgo_reset(mgo_env* e, int64_t seed, uint8_t* obs) {
    if (seed >= 0) {
        mgo_rng_seed(&e->rng, (uint64_t)seed);
        e->seeded = 1;
    }
    if (!e->seeded) return -1;
    e->vt->reset(e);
    if (obs) mgo_array3d(e->screen, obs);
    return 0;
}

int mgo_step(mgo_env* e, const int* action, uint8_t* obs, double* reward, int* done) {
    int a[2] = {action[0], e->vt->discrete ? 0 : action[1]};
    e->vt->step(e, a);
    if (obs) mgo_array3d(e->screen, obs);
    if (reward) *reward = e->reward;
    if (done) *done = e->done;
    return 0;
}

/* Env.render() with render_mode "debug_rgb_array" (e.g. mortar_mayhem_grid.py:403-405): the debug surface, stretched to
 * 336 x 336 like pygame.transform.scale (transform.c stretch(): an error accumulator per axis -- an integer factor
 * replicates every pixel), then fliplr(rot90(array3d, 3)) = image order [y][x][c]. */
int mgo_render_debug(mgo_env* e, uint8_t* out) {
    if (!e->vt->debug) return -1;
    const int sd = e->screen_dim, dd = 336;
    mgo_surf* s = mgo_surf_new(sd, sd);
    e->vt->debug(e, s);
    int h_err = 2 * sd - 2 * dd, sy = 0;
    for (int y = 0; y < dd; y++) {
        int w_err = 2 * sd - 2 * dd, sx = 0;
        for (int x = 0; x < dd; x++) {
            uint32_t p = s->px[sy * sd + sx];
            uint8_t* o = out + ((size_t)y * dd + x) * 3;
            o[0] = (uint8_t)(p >> 16);
            o[1] = (uint8_t)(p >> 8);
            o[2] = (uint8_t)p;
            while (w_err >= 0) {
                sx++;
                w_err -= 2 * dd;
            }
            w_err += 2 * sd;
        }
        while (h_err >= 0) {
            sy++;
            h_err -= 2 * dd;
        }
        h_err += 2 * sd;
    }
    mgo_surf_free(s);
    return 0;
}

/* test hook: see mgo_vtbl.scene */
int mgo_scene(mgo_env* e, const double* v, int n, uint8_t* obs) {
    if (!e->vt->scene || !e->seeded) return -1;
    int rc = e->vt->scene(e, v, n);
    if (rc == 0 && obs) mgo_array3d(e->screen, obs);
    return rc;
}

double mgo_get(mgo_env* e, const char* field, int* ok) {
    int k = 0;
    double v = e->vt->get(e, field, &k);
    if (ok) *ok = k;
    return v;
}
int mgo_get_list(mgo_env* e, const char* name, double* out, int cap) { return e->vt->get_list(e, name, out, cap); }
void mgo_get_gt(mgo_env* e, double* out) {
    for (int i = 0; i < e->vt->gt_dim; i++) out[i] = e->gt[i];
}
void mgo_rng_words(mgo_env* e, uint64_t* out) {
    out[0] = (uint64_t)(e->rng.state >> 64);
    out[1] = (uint64_t)e->rng.state;
    out[2] = (uint64_t)(e->rng.inc >> 64);
    out[3] = (uint64_t)e->rng.inc;
    out[4] = (uint64_t)e->rng.has_u32;
    out[5] = (uint64_t)e->rng.buf;
}

/* ---- raw RNG access for tests/test_oracle_rng.py ------------------------------------------------ */
void mgo_test_rng(uint64_t seed, const int32_t* ops, const int64_t* lo, const int64_t* hi, int n, double* out) {
    mgo_rng r;
    mgo_rng_seed(&r, seed);
    for (int i = 0; i < n; i++) {
        switch (ops[i]) {
            case 0: out[i] = (double)mgo_integers(&r, lo[i], hi[i]); break;
            case 1: out[i] = mgo_next_double(&r); break;
            case 2: out[i] = (double)(mgo_next_u64(&r) >> 11); break;
            default: out[i] = mgo_uniform(&r, (double)lo[i] / 1e6, (double)hi[i] / 1e6); break;
        }
    }
}

/* ---- batched helpers: N independent instances, env i seeded seed0+i ------------------------------ */
typedef struct {
    int n;
    mgo_env** envs;
} mgo_batch;

mgo_batch* mgo_batch_create(const char* env_id, int n, double scale) {
    mgo_batch* b = (mgo_batch*)calloc(1, sizeof(mgo_batch));
    b->n = n;
    b->envs = (mgo_env**)calloc(n, sizeof(mgo_env*));
    for (int i = 0; i < n; i++) {
        b->envs[i] = mgo_create(env_id, scale);
        if (!b->envs[i]) return NULL;
    }
    return b;
}
void mgo_batch_destroy(mgo_batch* b) {
    for (int i = 0; i < b->n; i++) mgo_destroy(b->envs[i]);
    free(b->envs);
    free(b);
}
mgo_env* mgo_batch_env(mgo_batch* b, int i) { return b->envs[i]; }
int mgo_batch_set_option(mgo_batch* b, const char* key, const double* v, int n) {
    int rc = 0;
    for (int i = 0; i < b->n; i++) rc |= mgo_set_option(b->envs[i], key, v, n);
    return rc;
}
void mgo_batch_reset(mgo_batch* b, const int64_t* seeds, uint8_t* obs) {
    size_t fs = (size_t)b->envs[0]->screen_dim * b->envs[0]->screen_dim * 3;
#pragma omp parallel for schedule(static)
    for (int i = 0; i < b->n; i++) mgo_reset(b->envs[i], seeds ? seeds[i] : -1, obs ? obs + fs * i : NULL);
}
/* step every env; if autoreset, an env that reports done is reset immediately (seed=None: the RNG
 * stream continues) and obs holds the first frame of the new episode, as a trainer loop would do. */
void mgo_batch_step(mgo_batch* b, const int32_t* actions, int autoreset, uint8_t* obs, double* reward, uint8_t* done) {
    size_t fs = (size_t)b->envs[0]->screen_dim * b->envs[0]->screen_dim * 3;
    int disc = b->envs[0]->vt->discrete;
#pragma omp parallel for schedule(static)
    for (int i = 0; i < b->n; i++) {
        int a[2] = {actions[disc ? i : 2 * i], disc ? 0 : actions[2 * i + 1]};
        double r;
        int d;
        mgo_step(b->envs[i], a, NULL, &r, &d);
        if (reward) reward[i] = r;
        if (done) done[i] = (uint8_t)d;
        if (d && autoreset) {
            mgo_reset(b->envs[i], -1, obs ? obs + fs * i : NULL);
        } else if (obs) {
            mgo_array3d(b->envs[i]->screen, obs + fs * i);
        }
    }
}

/* The test's POLICY for lock-step runs under competent play (mgo_vtbl.expert): instance i plays its expert action, or -- with
 * probability eps -- a uniformly random one.  The random numbers are a counter-based hash of (seed, step, i): they belong to
 * the test, not to any instance's stream.  actions: int32 [n] (Discrete) or [n][2]. */
static uint64_t mgo_mix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
int mgo_batch_expert(mgo_batch* b, double eps, uint64_t seed, uint64_t step, int32_t* actions) {
    int disc = b->envs[0]->vt->discrete;
    if (!b->envs[0]->vt->expert) return -1;
#pragma omp parallel for schedule(static)
    for (int i = 0; i < b->n; i++) {
        int a[2] = {0, 0};
        uint64_t h = mgo_mix64(mgo_mix64(seed ^ (step * 0xD1B54A32D192ED03ull)) + (uint64_t)i);
        if ((double)(h >> 11) * (1.0 / 9007199254740992.0) < eps) {
            uint64_t h2 = mgo_mix64(h);
            a[0] = (int)(h2 % (disc ? 4 : 3));
            a[1] = disc ? 0 : (int)((h2 >> 20) % 3);
        } else {
            b->envs[i]->vt->expert(b->envs[i], a);
        }
        if (disc) actions[i] = a[0];
        else { actions[2 * i] = a[0]; actions[2 * i + 1] = a[1]; }
    }
    return 0;
}
/* one named state field of every instance (mgo_get), for the run's statistics; NaN where the field does not apply */
void mgo_batch_get(mgo_batch* b, const char* field, double* out) {
#pragma omp parallel for schedule(static)
    for (int i = 0; i < b->n; i++) {
        int ok = 0;
        double v = b->envs[i]->vt->get(b->envs[i], field, &ok);
        out[i] = ok ? v : NAN;
    }
}

/* Test hook (tests/test_oracle_properties.py): pygame.draw.circle(surface, white, (cx, cy), radius, width) on a black dim x dim
 * surface; out[y * dim + x] = 1 where a pixel was drawn. */
int mgo_test_circle(int dim, int cx, int cy, int radius, int width, uint8_t* out) {
    mgo_surf* s = mgo_surf_new(dim, dim);
    if (!s) return -1;
    mgo_fill(s, 0);
    mgo_draw_circle(s, MGO_RGB(255, 255, 255), cx, cy, radius, width);
    for (int i = 0; i < dim * dim; i++) out[i] = s->px[i] != 0;
    mgo_surf_free(s);
    return 0;
}

