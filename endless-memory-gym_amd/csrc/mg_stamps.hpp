// mg_stamps.hpp -- host-side builder of the palette-indexed stamp atlases and background templates
// that the raster kernels sample.  Runs once in mg_create(); nothing here is on the per-step path.
//
// What is produced (all geometry at the reference's module constant SCALE = 0.25, 84x84 screen):
//   * agent sprites: 8 rotations (character_controller.py:29-75 create_character_sprites)
//   * command glyphs: 9 commands + blank (pygame_assets.py:254-304 Command)
//   * mortar arena templates (pygame_assets.py:306-418 MortarTile/MortarArena) as full frames in the
//     observation layout [x][y][c]
// The integer rasterisation rules (even-diameter discs, one-axis thick lines, 16.16 fixed-point rotation)
// are those of pygame 2.4 / SDL2 which the reference renders with; parity is enforced by the tests
// against the CPU oracle and, through it, against the reference's GIF recordings.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <vector>

namespace mg {

// A small indexed image; 0 = transparent (colour key).
struct Stamp {
    int w = 0, h = 0;
    std::vector<uint8_t> px;
    Stamp() {}
    Stamp(int w_, int h_) : w(w_), h(h_), px((size_t)w_ * h_, 0) {}
    uint8_t& at(int x, int y) { return px[(size_t)y * w + x]; }
    uint8_t get(int x, int y) const { return px[(size_t)y * w + x]; }
    void span(int xa, int y, int xb, uint8_t v) {  // inclusive, clipped, order-free
        if (y < 0 || y >= h) return;
        if (xb < xa) std::swap(xa, xb);
        if (xa < 0) xa = 0;
        if (xb > w - 1) xb = w - 1;
        for (int x = xa; x <= xb; ++x) at(x, y) = v;
    }
    void vspan(int ya, int x, int yb, uint8_t v) {
        if (x < 0 || x >= w) return;
        if (ya < 0) ya = 0;
        if (yb > h - 1) yb = h - 1;
        for (int y = ya; y <= yb; ++y) at(x, y) = v;
    }
};

// Filled disc with pygame's even-diameter midpoint rule: covers columns x0-r .. x0+r-1.
inline void disc(Stamp& s, int x0, int y0, int r, uint8_t v) {
    if (r < 1) return;
    int f = 1 - r, ddx = 0, ddy = -2 * r, x = 0, y = r;
    while (x < y) {
        if (f >= 0) { --y; ddy += 2; f += ddy; }
        ++x; ddx += 2; f += ddx + 1;
        if (f >= 0) {
            s.span(x0 - x, y0 + y - 1, x0 + x - 1, v);
            s.span(x0 - x, y0 - y, x0 + x - 1, v);
        }
        s.span(x0 - y, y0 + x - 1, x0 + y - 1, v);
        s.span(x0 - y, y0 - x, x0 + y - 1, v);
    }
}

// Ring of thickness t (1 < t < r): outer and inner midpoint ellipses walked together.
inline void ring(Stamp& s, int x0, int y0, int r, int t, uint8_t v) {
    using ll = long long;
    ll x = 0, y = r, r2 = (ll)r * r, D = 2 * r2, dx = 0, dy = D * y;
    double d1 = r2 * (1.25 - r);
    bool solid = true;
    ll ri = r - t + 1, xi = 0, yi = ri, ri2 = ri * ri, Di = 2 * ri2, dxi = 0, dyi = Di * yi;
    double d1i = ri2 * (1.25 - ri), d2i = 0;
    auto emit = [&]() {
        if (solid) {
            s.span(x0 - (int)x, y0 - (int)y, x0 + (int)x - 1, v);
            s.span(x0 - (int)x, y0 + (int)y - 1, x0 + (int)x - 1, v);
        } else {
            s.span(x0 - (int)x, y0 - (int)y, x0 - (int)xi, v);
            s.span(x0 - (int)x, y0 + (int)y - 1, x0 - (int)xi, v);
            s.span(x0 + (int)xi - 1, y0 - (int)y, x0 + (int)x - 1, v);
            s.span(x0 + (int)xi - 1, y0 + (int)y - 1, x0 + (int)x - 1, v);
        }
    };
    auto inner_region1 = [&]() {
        while (d1i < 0) { ++xi; dxi += Di; d1i += dxi + ri2; }
        ++xi; --yi; dxi += Di; dyi -= Di; d1i += dxi - dyi + ri2;
    };
    while (dx < dy) {
        while (d1 < 0) { ++x; dx += D; d1 += dx + r2; }
        emit();
        ++x; --y; dx += D; dy -= D; d1 += dx - dy + r2;
        if (solid && y < ri) solid = false;
        if (!solid) inner_region1();
    }
    d1 = r2 * ((x + 0.5) * (x + 0.5) + (y - 1) * (y - 1) - r2);
    while (y >= 0) {
        emit();
        if (d1 > 0) { --y; dy -= D; d1 += r2 - dy; }
        else { --y; ++x; dx += D; dy -= D; d1 += dx - dy + r2; }
        if (solid && y < ri) solid = false;
        if (!solid) {
            if (dxi < dyi) inner_region1();
            else {
                if (!d2i) d2i = ri2 * ((xi + 0.5) * (xi + 0.5) + (yi - 1) * (yi - 1) - ri2);
                if (d2i > 0) { --yi; dyi -= Di; d2i += ri2 - dyi; }
                else { --yi; ++xi; dxi += Di; dyi -= Di; d2i += dxi - dyi + ri2; }
            }
        }
    }
}

// 1-px circle (pygame 2.4 draw.c draw_circle_bresenham_thin, what draw.circle does for width == 1: the hand outline for
// agent_scale in [1/3, 2/3), the coin's ring for coin_scale in [0.5, 1)): the end points of the spans disc() fills.
inline void thin_circle(Stamp& s, int x0, int y0, int r, uint8_t v) {
    int f = 1 - r, ddx = 0, ddy = -2 * r, x = 0, y = r;
    auto put = [&](int px, int py) { s.span(px, py, px, v); };
    while (x < y) {
        if (f >= 0) { --y; ddy += 2; f += ddy; }
        ++x; ddx += 2; f += ddx + 1;
        put(x0 + x - 1, y0 + y - 1); put(x0 - x, y0 + y - 1); put(x0 + x - 1, y0 - y); put(x0 - x, y0 - y);
        put(x0 + y - 1, y0 + x - 1); put(x0 + y - 1, y0 - x); put(x0 - y, y0 + x - 1); put(x0 - y, y0 - x);
    }
}

// pygame.draw.circle(surface, colour, centre, radius, width)
inline void circle(Stamp& s, int x0, int y0, int r, int width, uint8_t v) {
    if (r < 1 || width < 0) return;
    if (width > r) width = r;
    if (width == 0 || width == r) disc(s, x0, y0, r, v);
    else if (width == 1) thin_circle(s, x0, y0, r, v);
    else ring(s, x0, y0, r, width, v);
}

// Thick line (width >= 2; the hot path never draws 1-px lines): Bresenham centre line, thickness grown
// along the minor axis only, flat ends.
inline void thick_line(Stamp& s, int x1, int y1, int x2, int y2, int width, uint8_t v) {
    if (width < 1) return;
    int extra = 1 - (width % 2), half = width / 2;
    bool grow_x = std::abs(x1 - x2) <= std::abs(y1 - y2);
    int dx = std::abs(x2 - x1), sx = x1 < x2 ? 1 : -1, dy = std::abs(y2 - y1), sy = y1 < y2 ? 1 : -1;
    int err = (dx > dy ? dx : -dy) / 2;
    auto advance = [&]() {
        int e2 = err;
        if (e2 > -dx) { err -= dy; x1 += sx; }
        if (e2 < dy) { err += dx; y1 += sy; }
    };
    if (grow_x) {
        while (y1 != y2 + sy) {
            int a = x1 - half + extra, b = x1 + half;
            if (a <= b && y1 >= 0 && y1 < s.h) s.span(std::max(a, 0), y1, std::min(b, s.w - 1), v);
            advance();
        }
    } else {
        while (x1 != x2 + sx) {
            int a = y1 - half + extra, b = y1 + half;
            if (a <= b && x1 >= 0 && x1 < s.w) s.vspan(a, x1, b, v);
            advance();
        }
    }
}

// Counter-clockwise rotation: exact quarter turns, otherwise inverse nearest-neighbour in 16.16 fixed point.
inline Stamp rotate_ccw(const Stamp& src, int angle) {
    if (angle % 90 == 0) {
        int k = ((angle / 90) % 4 + 4) % 4;
        Stamp d(k % 2 ? src.h : src.w, k % 2 ? src.w : src.h);
        for (int y = 0; y < d.h; ++y)
            for (int x = 0; x < d.w; ++x) {
                int sx, sy;
                if (k == 0) { sx = x; sy = y; }
                else if (k == 1) { sx = src.w - 1 - y; sy = x; }
                else if (k == 2) { sx = src.w - 1 - x; sy = src.h - 1 - y; }
                else { sx = y; sy = src.h - 1 - x; }
                d.at(x, y) = src.get(sx, sy);
            }
        return d;
    }
    double rad = angle * .01745329251994329, sa = std::sin(rad), ca = std::cos(rad);
    double cx = ca * src.w, cy = ca * src.h, sx = sa * src.w, sy = sa * src.h;
    int nx = (int)std::fmax(std::fmax(std::fmax(std::fabs(cx + sy), std::fabs(cx - sy)), std::fabs(-cx + sy)), std::fabs(-cx - sy));
    int ny = (int)std::fmax(std::fmax(std::fmax(std::fabs(sx + cy), std::fabs(sx - cy)), std::fabs(-sx + cy)), std::fabs(-sx - cy));
    Stamp d(nx, ny);
    int mid = ny / 2, xd = (src.w - nx) << 15, yd = (src.h - ny) << 15;
    int isin = (int)(sa * 65536), icos = (int)(ca * 65536);
    int ax = (nx << 15) - (int)(ca * ((nx - 1) << 15));
    int ay = (ny << 15) - (int)(sa * ((nx - 1) << 15));
    int xmax = (src.w << 16) - 1, ymax = (src.h << 16) - 1;
    for (int y = 0; y < ny; ++y) {
        int fx = ax + isin * (mid - y) + xd, fy = ay - icos * (mid - y) + yd;
        for (int x = 0; x < nx; ++x, fx += icos, fy += isin)
            d.at(x, y) = (fx < 0 || fy < 0 || fx > xmax || fy > ymax) ? 0 : src.get(fx >> 16, fy >> 16);
    }
    return d;
}

// pygame Vector2.rotate: multiples of 90 degrees are exact, else plain cos/sin in double.
inline void rotate_vec(double x, double y, double deg, double& ox, double& oy) {
    const double eps = 1e-6;
    deg = std::fmod(deg, 360.0);
    if (deg < 0) deg += 360.0;
    if (std::fmod(deg + eps, 90.0) < 2 * eps) {
        switch ((int)((deg + eps) / 90.0)) {
            case 1: ox = -y; oy = x; break;
            case 2: ox = -x; oy = -y; break;
            case 3: ox = y; oy = -x; break;
            default: ox = x; oy = y; break;
        }
    } else {
        double rad = deg * M_PI / 180.0, s = std::sin(rad), c = std::cos(rad);
        ox = c * x - s * y;
        oy = s * x + c * y;
    }
}

// stamp pixel values are ids of the shared palette (mg_raster.hpp): 0 key, 1 body, 2 hand, 3 hand outline, 4 white, 5 red
enum : uint8_t { PAL_KEY = 0, PAL_BODY = 1, PAL_HAND = 2, PAL_OUTLINE = 3, PAL_WHITE = 4, PAL_RED = 5, PAL_YELLOW = 8, PAL_ORANGE = 9,
                 PAL_BLACK = 15, PAL_EXIT_OPEN = 16, PAL_EXIT_CLOSED = 17 };

// 8 agent sprites; sprite k shows rotation 45k degrees (the hands are rotated by 360-45k about the centre).
inline std::vector<Stamp> build_agent_sprites(double agent_scale, int* radius_out) {
    int radius = (int)(25 * agent_scale);
    int hands_x = (int)(18 * agent_scale), hand_y = (int)(12 * agent_scale);
    int hand_r = (int)(10 * agent_scale), outline = (int)(3 * agent_scale);
    const int extension = 14;
    int dim = radius * 2 + hand_r + extension, c = dim / 2;
    std::vector<Stamp> out;
    for (int k = 0; k < 8; ++k) {
        Stamp s(dim, dim);
        disc(s, c, c, radius, PAL_BODY);
        double lx, ly, rx, ry;
        rotate_vec(-hands_x, hand_y + extension / 2 - c, 360 - 45 * k, lx, ly);
        rotate_vec(hands_x, hand_y + extension / 2 - c, 360 - 45 * k, rx, ry);
        int lxi = (int)(lx + c), lyi = (int)(ly + c), rxi = (int)(rx + c), ryi = (int)(ry + c);
        circle(s, lxi, lyi, hand_r, 0, PAL_HAND);
        circle(s, rxi, ryi, hand_r, 0, PAL_HAND);
        circle(s, lxi, lyi, hand_r, outline, PAL_OUTLINE);
        circle(s, rxi, ryi, hand_r, outline, PAL_OUTLINE);
        out.push_back(s);
    }
    if (radius_out) *radius_out = radius;
    return out;
}

// The sprite surfaces are mostly colour key (28x28 for a 12-px body): crop the transparent margin common to all
// stamps of a set from the four sides; the caller moves the blit position by the returned margin.
inline int crop_common_margin(std::vector<Stamp>& v) {
    int m = v.empty() ? 0 : v[0].w;
    for (const Stamp& s : v)
        for (int y = 0; y < s.h; ++y)
            for (int x = 0; x < s.w; ++x)
                if (s.get(x, y)) m = std::min(m, std::min(std::min(x, y), std::min(s.w - 1 - x, s.h - 1 - y)));
    if (m <= 0) return 0;
    for (Stamp& s : v) {
        Stamp c(s.w - 2 * m, s.h - 2 * m);
        for (int y = 0; y < c.h; ++y)
            for (int x = 0; x < c.w; ++x) c.at(x, y) = s.get(x + m, y + m);
        s = c;
    }
    return m;
}

// 10 glyphs: Command.COMMANDS order right, down, left, up, stay, right_down, right_up, left_down, left_up; 9 = blank.
inline std::vector<Stamp> build_glyphs(double scale) {
    static const int ANGLE[9] = {0, 270, 180, 90, 0, 315, 45, 225, 135};
    double rect_dim = 88 * scale;
    int dim = (int)rect_dim, lw = (int)(8 * scale);
    std::vector<Stamp> out;
    for (int g = 0; g < 10; ++g) {
        Stamp s(dim, dim);
        if (g == 4) {
            double radius = std::floor(rect_dim / 2) - 4 * scale;
            double x = rect_dim - 12 * scale, y = std::floor(rect_dim / 2) - 8 * scale;
            circle(s, (int)radius, (int)radius, (int)radius, lw, PAL_WHITE);
            thick_line(s, 0, (int)y, (int)x, (int)y, lw, PAL_WHITE);
        } else if (g < 9) {
            int x1 = (int)(2 * scale), x2 = (int)(80 * scale), y1 = (int)(40 * scale);
            thick_line(s, x1, y1, x2, y1, lw, PAL_WHITE);
            thick_line(s, x2, y1, y1, 0, lw, PAL_WHITE);
            thick_line(s, x2, y1, y1, x2, lw, PAL_WHITE);
            s = rotate_ccw(s, ANGLE[g]);
        }
        out.push_back(s);
    }
    return out;
}

// Mortar arena frame templates in observation layout [x][y][c]; template 0 = all tiles blue,
// template 1 + tx*N + ty = every tile red except (tx,ty).  Returns (1+N*N) * 84*84*3 bytes.
inline std::vector<uint8_t> build_mortar_templates(int N, double scale, int screen) {
    const uint8_t BLUE[3] = {21, 43, 77}, LBLUE[3] = {29, 60, 107}, RED[3] = {81, 18, 26}, LRED[3] = {112, 24, 36};
    int tile = (int)(56 * scale), border = (int)(4 * scale), arena = tile * N;
    int x0 = screen / 2 - (arena >> 1), y0 = x0;
    size_t frame = (size_t)screen * screen * 3;
    std::vector<uint8_t> out((size_t)(1 + N * N) * frame, 0);
    for (int t = 0; t < 1 + N * N; ++t) {
        int tx = t == 0 ? -1 : (t - 1) / N, ty = t == 0 ? -1 : (t - 1) % N;
        uint8_t* f = out.data() + (size_t)t * frame;
        for (int x = 0; x < screen; ++x)
            for (int y = 0; y < screen; ++y) {
                int ax = x - x0, ay = y - y0;
                if (ax < 0 || ay < 0 || ax >= arena || ay >= arena) continue;
                int i = ax / tile, j = ay / tile, u = ax % tile, v = ay % tile;
                bool edge = (border * 2 < tile) ? (u < border || v < border || u >= tile - border || v >= tile - border) : true;
                bool red = t != 0 && !(i == tx && j == ty);
                const uint8_t* c = red ? (edge ? LRED : RED) : (edge ? LBLUE : BLUE);
                uint8_t* p = f + ((size_t)x * screen + y) * 3;
                p[0] = c[0]; p[1] = c[1]; p[2] = c[2];
            }
    }
    return out;
}

// Fall-off cross (mystery_path.py:175-182): 40*scale square, two diagonal lines of width int(12*scale), red on colour key.
inline Stamp build_cross(double scale) {
    double dim = 40 * scale;
    Stamp s((int)dim, (int)dim);
    thick_line(s, 0, 0, (int)(dim - 1), (int)(dim - 1), (int)(12 * scale), PAL_RED);
    thick_line(s, (int)(dim - 1), 0, 0, (int)(dim - 1), (int)(12 * scale), PAL_RED);
    return s;
}

// Coin (pygame_assets.py:133-152): yellow disc, then an orange circle of width int(2*scale) (0 = filled) on top.
// Box 2r x 2r, disc centre at (r, r): blit with the top-left at (x - r, y - r).
inline Stamp build_coin(double coin_scale) {
    int r = (int)(10 * coin_scale);
    Stamp s(2 * r > 0 ? 2 * r : 1, 2 * r > 0 ? 2 * r : 1);
    circle(s, r, r, r, 0, PAL_YELLOW);
    circle(s, r, r, r, (int)(2 * coin_scale), PAL_ORANGE);
    return s;
}

// Chessboard backgrounds (pygame_assets.py:222-239) as frame templates [x][y][c]: 0 = white/blue, 1 = white/red; then what
// hide_chessboard / black_background leave of them: 2 = all white, 3 = all black.
inline std::vector<uint8_t> build_chessboards(double scale, int screen) {
    int ts = (int)(50 * scale);
    size_t frame = (size_t)screen * screen * 3;
    std::vector<uint8_t> out(4 * frame, 0);
    for (int t = 0; t < 2; ++t)
        for (int x = 0; x < screen; ++x)
            for (int y = 0; y < screen; ++y) {
                bool white = ((x / ts) + (y / ts)) % 2 == 0;
                uint8_t* p = out.data() + t * frame + ((size_t)x * screen + y) * 3;
                p[0] = white ? 255 : (t == 1 ? 255 : 0);
                p[1] = white ? 255 : 0;
                p[2] = white ? 255 : (t == 0 ? 255 : 0);
            }
    std::fill(out.begin() + 2 * frame, out.begin() + 3 * frame, (uint8_t)255);
    return out;
}

// Rounded rectangle pieces of pygame's draw.rect(..., border_*_radius): filled circle quadrants / quadrant arcs.
inline void quadrant(Stamp& s, int x0, int y0, int radius, int thickness, uint8_t v, bool tr, bool tl, bool bl, bool br) {
    int f = 1 - radius, ddx = 0, ddy = -2 * radius, x = 0, y = radius;
    int i_y = radius - thickness, i_f = 1 - i_y, i_ddx = 0, i_ddy = -2 * i_y;
    if (radius == 1) {
        if (tr) s.span(x0, y0 - 1, x0, v);
        if (tl) s.span(x0 - 1, y0 - 1, x0 - 1, v);
        if (bl) s.span(x0 - 1, y0, x0 - 1, v);
        if (br) s.span(x0, y0, x0, v);
        return;
    }
    if (thickness != 0) {
        while (x < y) {
            if (f >= 0) { --y; ddy += 2; f += ddy; }
            if (i_f >= 0) { --i_y; i_ddy += 2; i_f += i_ddy; }
            ++x; ddx += 2; f += ddx + 1;
            i_ddx += 2; i_f += i_ddx + 1;
            if (thickness > 1) thickness = y - i_y;
            for (int i = 0; i < thickness; ++i) {
                int y1 = y - i;
                if (tr) { if ((y0 - y1) < (y0 - x)) s.span(x0 + x - 1, y0 - y1, x0 + x - 1, v); if ((x0 + y1 - 1) >= (x0 + x - 1)) s.span(x0 + y1 - 1, y0 - x, x0 + y1 - 1, v); }
                if (tl) { if ((y0 - y1) <= (y0 - x)) s.span(x0 - x, y0 - y1, x0 - x, v); if ((x0 - y1) < (x0 - x)) s.span(x0 - y1, y0 - x, x0 - y1, v); }
                if (bl) { if ((x0 - y1) <= (x0 - x)) s.span(x0 - y1, y0 + x - 1, x0 - y1, v); if ((y0 + y1 - 1) > (y0 + x - 1)) s.span(x0 - x, y0 + y1 - 1, x0 - x, v); }
                if (br) { if ((y0 + y1 - 1) >= (y0 + x - 1)) s.span(x0 + x - 1, y0 + y1 - 1, x0 + x - 1, v); if ((x0 + y1 - 1) > (x0 + x - 1)) s.span(x0 + y1 - 1, y0 + x - 1, x0 + y1 - 1, v); }
            }
        }
    } else {
        while (x < y) {
            if (f >= 0) { --y; ddy += 2; f += ddy; }
            ++x; ddx += 2; f += ddx + 1;
            if (tr) { for (int y1 = y0 - x; y1 <= y0; ++y1) s.span(x0 + y - 1, y1, x0 + y - 1, v); for (int y1 = y0 - y; y1 <= y0; ++y1) s.span(x0 + x - 1, y1, x0 + x - 1, v); }
            if (tl) { for (int y1 = y0 - x; y1 <= y0; ++y1) s.span(x0 - y, y1, x0 - y, v); for (int y1 = y0 - y; y1 <= y0; ++y1) s.span(x0 - x, y1, x0 - x, v); }
            if (bl) { for (int y1 = y0; y1 < y0 + x; ++y1) s.span(x0 - y, y1, x0 - y, v); for (int y1 = y0; y1 < y0 + y; ++y1) s.span(x0 - x, y1, x0 - x, v); }
            if (br) { for (int y1 = y0; y1 < y0 + x; ++y1) s.span(x0 + y - 1, y1, x0 + y - 1, v); for (int y1 = y0; y1 < y0 + y; ++y1) s.span(x0 + x - 1, y1, x0 + x - 1, v); }
        }
    }
}

inline void round_rect(Stamp& s, uint8_t v, int x1, int y1, int x2, int y2, int width, int tl, int tr, int bl, int br) {
    int w = x2 - x1 + 1, h = y2 - y1 + 1;
    if ((tl + tr) > w || (tl + bl) > h || (tr + br) > h || (bl + br) > w) {
        float qt = w / (float)(tl + tr), ql = h / (float)(tl + bl), qb = w / (float)(bl + br), qr = h / (float)(tr + br);
        float f = std::fmin(std::fmin(std::fmin(qt, ql), qb), qr);
        tl = (int)(tl * f); tr = (int)(tr * f); bl = (int)(bl * f); br = (int)(br * f);
    }
    if (width == 0) {
        for (int y = y1; y <= y2; ++y) {  // octagon between the corner cut-offs
            int xa = x1, xb = x2;
            if (y < y1 + tl) xa = x1 + (tl - (y - y1));
            if (y < y1 + tr) xb = x2 - (tr - (y - y1));
            if (y > y2 - bl) xa = x1 + (bl - (y2 - y));
            if (y > y2 - br) xb = x2 - (br - (y2 - y));
            s.span(xa, y, xb, v);
        }
        quadrant(s, x2 - tr + 1, y1 + tr, tr, 0, v, true, false, false, false);
        quadrant(s, x1 + tl, y1 + tl, tl, 0, v, false, true, false, false);
        quadrant(s, x1 + bl, y2 - bl + 1, bl, 0, v, false, false, true, false);
        quadrant(s, x2 - br + 1, y2 - br + 1, br, 0, v, false, false, false, true);
    } else {
        int o = width / 2 - 1 + width % 2, o2 = width / 2;
        auto hline_or_line = [&](int ax, int ay, int bx, int by) {
            if (width == 1) {
                if (ay == by) s.span(ax, ay, bx, v);
                else s.vspan(ay < by ? ay : by, ax, ay < by ? by : ay, v);
            } else thick_line(s, ax, ay, bx, by, width, v);
        };
        if (x2 - tr == x1 + tl) { for (int i = 0; i < width; ++i) s.span(x1 + tl, y1 + i, x1 + tl, v); }
        else hline_or_line(x1 + tl, y1 + o, x2 - tr, y1 + o);
        if (y2 - bl == y1 + tl) { for (int i = 0; i < width; ++i) s.span(x1 + i, y1 + tl, x1 + i, v); }
        else hline_or_line(x1 + o, y1 + tl, x1 + o, y2 - bl);
        if (x2 - br == x1 + bl) { for (int i = 0; i < width; ++i) s.span(x1 + bl, y2 - i, x1 + bl, v); }
        else hline_or_line(x1 + bl, y2 - o2, x2 - br, y2 - o2);
        if (y2 - br == y1 + tr) { for (int i = 0; i < width; ++i) s.span(x2 - i, y1 + tr, x2 - i, v); }
        else hline_or_line(x2 - o2, y1 + tr, x2 - o2, y2 - br);
        quadrant(s, x2 - tr + 1, y1 + tr, tr, width, v, true, false, false, false);
        quadrant(s, x1 + tl, y1 + tl, tl, width, v, false, true, false, false);
        quadrant(s, x1 + bl, y2 - bl + 1, bl, width, v, false, false, true, false);
        quadrant(s, x2 - br + 1, y2 - br + 1, br, width, v, false, false, false, true);
    }
}

// Exit (pygame_assets.py:169-205): rect_dim = 20*scale square, top corners rounded with radius int(10*scale),
// filled open/closed colour, then a black outline of width int(2*scale).
inline Stamp build_exit(double exit_scale, bool open) {
    int d = (int)(20 * exit_scale), r = (int)(10 * exit_scale), w = (int)(2 * exit_scale);
    Stamp s(d, d);
    uint8_t c = open ? PAL_EXIT_OPEN : PAL_EXIT_CLOSED;
    round_rect(s, c, 0, 0, d - 1, d - 1, 0, r, r, 0, 0);
    round_rect(s, PAL_BLACK, 0, 0, d - 1, d - 1, w, r, r, 0, 0);
    return s;
}

}  // namespace mg
