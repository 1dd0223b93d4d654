/* oracle/mgo_raster.h -- TEST INFRASTRUCTURE (CPU oracle), not product code.
 *
 * Software surfaces + the handful of pygame 2.4 / SDL2 drawing routines the reference calls.
 * Their source is third-party (pygame==2.4.0 pinned at /root/reference/setup.py:36, bundling SDL2)
 * and is NOT under /root/reference, so the published algorithms are restated here
 *  and pinned against the reference's own recordings
 * docs/assets/{emm,ess,emp}_0.gif (tests/test_oracle_gif.py, fixtures tests/golden/gif_*.npz).
 *
 * Call sites in the reference (every pixel the envs produce goes through these):
 *   draw.circle : character_controller.py:58,67-70  pygame_assets.py:111,151,152,275
 *   draw.line   : pygame_assets.py:276,283-285      mystery_path.py:180,181
 *   draw.rect   : pygame_assets.py:238,328,329,342,343  endless_searing_spotlights.py:213,340,346,347,422,423,459,461
 *   transform.rotate : pygame_assets.py:303
 *   Surface.blit (colour key, surface alpha) : every _draw_surfaces(), e.g. mortar_mayhem_grid.py:92-102
 *   surfarray.array3d : mortar_mayhem_grid.py:276,373
 * Pixel format: 0x00RRGGBB (so fill(255)/set_colorkey(255) == blue, as in SDL XRGB8888).
 */
#ifndef MGO_RASTER_H
#define MGO_RASTER_H
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
    int w, h;
    uint32_t* px;
    int has_key;
    uint32_t key;
    int alpha; /* 0..255, 255 = opaque */
} mgo_surf;

#define MGO_RGB(r, g, b) ((((uint32_t)(r)) << 16) | (((uint32_t)(g)) << 8) | ((uint32_t)(b)))

static inline mgo_surf* mgo_surf_new(int w, int h) {
    mgo_surf* s = (mgo_surf*)calloc(1, sizeof(mgo_surf));
    s->w = w;
    s->h = h;
    s->px = (uint32_t*)calloc((size_t)(w > 0 ? w : 1) * (size_t)(h > 0 ? h : 1), sizeof(uint32_t));
    s->alpha = 255;
    return s;
}
static inline void mgo_surf_free(mgo_surf* s) {
    if (s) {
        free(s->px);
        free(s);
    }
}
static inline void mgo_fill(mgo_surf* s, uint32_t c) {
    for (int i = 0; i < s->w * s->h; i++) s->px[i] = c;
}
static inline void mgo_set_colorkey(mgo_surf* s, uint32_t c) {
    s->has_key = 1;
    s->key = c;
}
/* Surface.set_alpha clamps to 0..255 (read back by endless_searing_spotlights.py:427-431) */
static inline void mgo_set_alpha(mgo_surf* s, int a) { s->alpha = a < 0 ? 0 : (a > 255 ? 255 : a); }

/* inclusive, clipped horizontal line; swaps ends like pygame's drawhorzlineclipbounding */
static inline void mgo_hline(mgo_surf* s, uint32_t c, int x1, int y, int x2) {
    if (y < 0 || y >= s->h) return;
    if (x2 < x1) {
        int t = x1;
        x1 = x2;
        x2 = t;
    }
    if (x1 < 0) x1 = 0;
    if (x2 > s->w - 1) x2 = s->w - 1;
    for (int x = x1; x <= x2; x++) s->px[y * s->w + x] = c;
}
static inline void mgo_vline(mgo_surf* s, uint32_t c, int y1, int x, int y2) {
    if (x < 0 || x >= s->w) return;
    if (y1 < 0) y1 = 0;
    if (y2 > s->h - 1) y2 = s->h - 1;
    for (int y = y1; y <= y2; y++) s->px[y * s->w + x] = c;
}

/* pygame draw.c draw_circle_filled (even-diameter disc) */
static inline void mgo_circle_filled(mgo_surf* s, uint32_t c, int x0, int y0, int radius) {
    int f = 1 - radius, ddx = 0, ddy = -2 * radius, x = 0, y = radius;
    while (x < y) {
        if (f >= 0) {
            y--;
            ddy += 2;
            f += ddy;
        }
        x++;
        ddx += 2;
        f += ddx + 1;
        if (f >= 0) {
            mgo_hline(s, c, x0 - x, y0 + y - 1, x0 + x - 1);
            mgo_hline(s, c, x0 - x, y0 - y, x0 + x - 1);
        }
        mgo_hline(s, c, x0 - y, y0 + x - 1, x0 + y - 1);
        mgo_hline(s, c, x0 - y, y0 - x, x0 + y - 1);
    }
}

/* pygame 2.4.0 draw.c draw_circle_bresenham_thin (width == 1): the end points of the spans draw_circle_filled walks, for
 * every x step.  Only reachable through Spotlight.draw's border (black_background = True, pygame_assets.py:112-113); no
 * recording or fixture of the reference shows such a frame: PARITY UNPINNED for this routine (restated from the
 * published pygame source). */
static inline void mgo_put(mgo_surf* s, uint32_t c, int x, int y) {
    if (x >= 0 && x < s->w && y >= 0 && y < s->h) s->px[y * s->w + x] = c;
}
static inline void mgo_circle_thin(mgo_surf* s, uint32_t c, int x0, int y0, int radius) {
    int f = 1 - radius, ddx = 0, ddy = -2 * radius, x = 0, y = radius;
    while (x < y) {
        if (f >= 0) {
            y--;
            ddy += 2;
            f += ddy;
        }
        x++;
        ddx += 2;
        f += ddx + 1;
        mgo_put(s, c, x0 + x - 1, y0 + y - 1);
        mgo_put(s, c, x0 - x, y0 + y - 1);
        mgo_put(s, c, x0 + x - 1, y0 - y);
        mgo_put(s, c, x0 - x, y0 - y);
        mgo_put(s, c, x0 + y - 1, y0 + x - 1);
        mgo_put(s, c, x0 + y - 1, y0 - x);
        mgo_put(s, c, x0 - y, y0 + x - 1);
        mgo_put(s, c, x0 - y, y0 - x);
    }
}

/* pygame draw.c draw_circle_bresenham (thick ring, 1 < thickness < radius) */
static inline void mgo_circle_thick(mgo_surf* s, uint32_t c, int x0, int y0, int radius, int thickness) {
    long long x = 0, y = radius, r2 = (long long)radius * radius, D = 2 * r2;
    double d1 = r2 * (1.25 - radius);
    long long dx = 0, dy = D * y;
    int line = 1;
    long long ri = radius - thickness + 1, xi = 0, yi = ri, ri2 = ri * ri, Di = 2 * ri2;
    double d1i = ri2 * (1.25 - ri), d2i = 0;
    long long dxi = 0, dyi = Di * yi;
#define MGO_EMIT()                                                              \
    do {                                                                        \
        if (line) {                                                             \
            mgo_hline(s, c, x0 - (int)x, y0 - (int)y, x0 + (int)x - 1);         \
            mgo_hline(s, c, x0 - (int)x, y0 + (int)y - 1, x0 + (int)x - 1);     \
        } else {                                                                \
            mgo_hline(s, c, x0 - (int)x, y0 - (int)y, x0 - (int)xi);            \
            mgo_hline(s, c, x0 - (int)x, y0 + (int)y - 1, x0 - (int)xi);        \
            mgo_hline(s, c, x0 + (int)xi - 1, y0 - (int)y, x0 + (int)x - 1);    \
            mgo_hline(s, c, x0 + (int)xi - 1, y0 + (int)y - 1, x0 + (int)x - 1);\
        }                                                                       \
    } while (0)
#define MGO_INNER_A()                 \
    do {                              \
        while (d1i < 0) {             \
            xi += 1;                  \
            dxi += Di;                \
            d1i += dxi + ri2;         \
        }                             \
        xi++;                         \
        yi--;                         \
        dxi += Di;                    \
        dyi -= Di;                    \
        d1i += dxi - dyi + ri2;       \
    } while (0)
    while (dx < dy) {
        while (d1 < 0) {
            x++;
            dx += D;
            d1 += dx + r2;
        }
        MGO_EMIT();
        x++;
        y--;
        dx += D;
        dy -= D;
        d1 += dx - dy + r2;
        if (line && y < ri) line = 0;
        if (!line) MGO_INNER_A();
    }
    d1 = r2 * ((x + 0.5) * (x + 0.5) + (y - 1) * (y - 1) - r2);
    while (y >= 0) {
        MGO_EMIT();
        if (d1 > 0) {
            y--;
            dy -= D;
            d1 += r2 - dy;
        } else {
            y--;
            x++;
            dx += D;
            dy -= D;
            d1 += dx - dy + r2;
        }
        if (line && y < ri) line = 0;
        if (!line) {
            if (dxi < dyi) {
                MGO_INNER_A();
            } else {
                if (!d2i) d2i = ri2 * ((xi + 0.5) * (xi + 0.5) + (yi - 1) * (yi - 1) - ri2);
                if (d2i > 0) {
                    yi--;
                    dyi -= Di;
                    d2i += ri2 - dyi;
                } else {
                    yi--;
                    xi++;
                    dxi += Di;
                    dyi -= Di;
                    d2i += dxi - dyi + ri2;
                }
            }
        }
    }
#undef MGO_EMIT
#undef MGO_INNER_A
}

/* plain Bresenham (pygame draw_line, width 1) -- used by the thin circle only via callers; kept for width==1 lines */
static inline void mgo_line1(mgo_surf* s, uint32_t c, int x1, int y1, int x2, int y2) {
    int dx = abs(x2 - x1), sx = x1 < x2 ? 1 : -1, dy = abs(y2 - y1), sy = y1 < y2 ? 1 : -1;
    int err = (dx > dy ? dx : -dy) / 2, e2;
    while (x1 != x2 || y1 != y2) {
        if (x1 >= 0 && x1 < s->w && y1 >= 0 && y1 < s->h) s->px[y1 * s->w + x1] = c;
        e2 = err;
        if (e2 > -dx) {
            err -= dy;
            x1 += sx;
        }
        if (e2 < dy) {
            err += dx;
            y1 += sy;
        }
    }
    if (x2 >= 0 && x2 < s->w && y2 >= 0 && y2 < s->h) s->px[y2 * s->w + x2] = c;
}

/* pygame.draw.circle(surface, color, center, radius, width): center/radius truncated by the caller */
static inline void mgo_draw_circle(mgo_surf* s, uint32_t c, int x0, int y0, int radius, int width) {
    if (radius < 1 || width < 0) return;
    if (width > radius) width = radius;
    if (width == 0 || width == radius) {
        mgo_circle_filled(s, c, x0, y0, radius);
    } else if (width == 1) {
        mgo_circle_thin(s, c, x0, y0, radius);
    } else {
        mgo_circle_thick(s, c, x0, y0, radius, width);
    }
}

/* pygame draw.c draw_line_width: thickness grows along one axis only, flat ends */
static inline void mgo_draw_line(mgo_surf* s, uint32_t c, int x1, int y1, int x2, int y2, int width) {
    if (width < 1) return;
    if (width == 1) {
        mgo_line1(s, c, x1, y1, x2, y2);
        return;
    }
    int extra = 1 - (width % 2), h = width / 2;
    int xinc = abs(x1 - x2) <= abs(y1 - y2);
    int dx = abs(x2 - x1), sx = x1 < x2 ? 1 : -1, dy = abs(y2 - y1), sy = y1 < y2 ? 1 : -1;
    int err = (dx > dy ? dx : -dy) / 2, e2;
    if (xinc) {
        while (y1 != y2 + sy) {
            if (y1 >= 0 && y1 < s->h) {
                int a = x1 - h + extra, b = x1 + h;
                if (a < 0) a = 0;
                if (b > s->w - 1) b = s->w - 1;
                if (a <= b) mgo_hline(s, c, a, y1, b);
            }
            e2 = err;
            if (e2 > -dx) {
                err -= dy;
                x1 += sx;
            }
            if (e2 < dy) {
                err += dx;
                y1 += sy;
            }
        }
    } else {
        while (x1 != x2 + sx) {
            if (x1 >= 0 && x1 < s->w) {
                int a = y1 - h + extra, b = y1 + h;
                if (a < 0) a = 0;
                if (b > s->h - 1) b = s->h - 1;
                if (a <= b) mgo_vline(s, c, a, x1, b);
            }
            e2 = err;
            if (e2 > -dx) {
                err -= dy;
                x1 += sx;
            }
            if (e2 < dy) {
                err += dx;
                y1 += sy;
            }
        }
    }
}

/* pygame.draw.rect(surface, color, (x,y,w,h), width): rect already truncated to ints.
 * width==0 -> solid; width>0 -> inset ring of `width` pixels. */
static inline void mgo_draw_rect(mgo_surf* s, uint32_t c, int x, int y, int w, int h, int width) {
    if (w <= 0 || h <= 0) return;
    /* pygame 2.4 draw.c rect(): a ring only if it leaves an interior, otherwise SDL_FillRect */
    if (!(width > 0 && width * 2 < w && width * 2 < h)) {
        for (int yy = y; yy < y + h; yy++) mgo_hline(s, c, x, yy, x + w - 1);
        return;
    }
    for (int k = 0; k < width; k++) {
        mgo_hline(s, c, x, y + k, x + w - 1);
        mgo_hline(s, c, x, y + h - 1 - k, x + w - 1);
        mgo_vline(s, c, y, x + k, y + h - 1);
        mgo_vline(s, c, y, x + w - 1 - k, y + h - 1);
    }
}

/* pygame.transform.rotate (counter-clockwise; exact quarter turns, else 16.16 fixed-point nearest neighbour) */
static inline mgo_surf* mgo_rotate(const mgo_surf* src, int angle) {
    if (angle % 90 == 0) {
        int k = ((angle / 90) % 4 + 4) % 4;
        int w = src->w, h = src->h;
        mgo_surf* d = mgo_surf_new(k % 2 ? h : w, k % 2 ? w : h);
        d->has_key = src->has_key;
        d->key = src->key;
        d->alpha = src->alpha;
        for (int y = 0; y < d->h; y++)
            for (int x = 0; x < d->w; x++) {
                int sx, sy;
                switch (k) {
                    case 0: sx = x; sy = y; break;
                    case 1: sx = w - 1 - y; sy = x; break;             /* 90 deg CCW  */
                    case 2: sx = w - 1 - x; sy = h - 1 - y; break;     /* 180         */
                    default: sx = y; sy = h - 1 - x; break;            /* 270 deg CCW */
                }
                d->px[y * d->w + x] = src->px[sy * w + sx];
            }
        return d;
    }
    double rad = angle * .01745329251994329, sa = sin(rad), ca = cos(rad);
    double x = src->w, y = src->h, cx = ca * x, cy_ = ca * y, sx_ = sa * x, sy_ = sa * y;
    double m1 = fmax(fmax(fmax(fabs(cx + sy_), fabs(cx - sy_)), fabs(-cx + sy_)), fabs(-cx - sy_));
    double m2 = fmax(fmax(fmax(fabs(sx_ + cy_), fabs(sx_ - cy_)), fabs(-sx_ + cy_)), fabs(-sx_ - cy_));
    int nx = (int)m1, ny = (int)m2;
    mgo_surf* d = mgo_surf_new(nx, ny);
    d->has_key = src->has_key;
    d->key = src->key;
    d->alpha = src->alpha;
    uint32_t bg = src->has_key ? src->key : src->px[0];
    int cy = ny / 2, xd = (src->w - nx) << 15, yd = (src->h - ny) << 15;
    int isin = (int)(sa * 65536), icos = (int)(ca * 65536);
    int ax = (nx << 15) - (int)(ca * ((nx - 1) << 15));
    int ay = (ny << 15) - (int)(sa * ((nx - 1) << 15));
    int xmax = (src->w << 16) - 1, ymax = (src->h << 16) - 1;
    for (int yy = 0; yy < ny; yy++) {
        int dx = (ax + (isin * (cy - yy))) + xd;
        int dy = (ay - (icos * (cy - yy))) + yd;
        for (int xx = 0; xx < nx; xx++) {
            if (dx < 0 || dy < 0 || dx > xmax || dy > ymax)
                d->px[yy * nx + xx] = bg;
            else
                d->px[yy * nx + xx] = src->px[(dy >> 16) * src->w + (dx >> 16)];
            dx += icos;
            dy += isin;
        }
    }
    return d;
}

/* Surface.blit(src, (dx,dy)) with colour key and per-surface alpha
 * (SDL2 ALPHA_BLEND_RGB: d += (s-d)*A/255 with C integer division) */
static inline void mgo_blit(mgo_surf* dst, const mgo_surf* src, int dx, int dy) {
    if (!src) return;
    int a = src->alpha;
    for (int y = 0; y < src->h; y++) {
        int ty = dy + y;
        if (ty < 0 || ty >= dst->h) continue;
        for (int x = 0; x < src->w; x++) {
            int tx = dx + x;
            if (tx < 0 || tx >= dst->w) continue;
            uint32_t sp = src->px[y * src->w + x];
            if (src->has_key && sp == src->key) continue;
            if (a >= 255) {
                dst->px[ty * dst->w + tx] = sp;
            } else {
                uint32_t dp = dst->px[ty * dst->w + tx];
                int sr = (sp >> 16) & 255, sg = (sp >> 8) & 255, sb = sp & 255;
                int dr = (dp >> 16) & 255, dg = (dp >> 8) & 255, db = dp & 255;
                dr = (uint8_t)((((sr - dr) * a) / 255) + dr);
                dg = (uint8_t)((((sg - dg) * a) / 255) + dg);
                db = (uint8_t)((((sb - db) * a) / 255) + db);
                dst->px[ty * dst->w + tx] = MGO_RGB(dr, dg, db);
            }
        }
    }
}

/* pygame.surfarray.array3d(display) -> uint8 [x][y][c] */
static inline void mgo_array3d(const mgo_surf* s, uint8_t* out) {
    for (int x = 0; x < s->w; x++)
        for (int y = 0; y < s->h; y++) {
            uint32_t p = s->px[y * s->w + x];
            uint8_t* o = out + ((size_t)x * s->h + y) * 3;
            o[0] = (uint8_t)(p >> 16);
            o[1] = (uint8_t)(p >> 8);
            o[2] = (uint8_t)p;
        }
}

#endif
