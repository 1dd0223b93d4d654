// tools/microbench/raster_bench.hip -- standalone A/B harness for raster-kernel STRUCTURES (not part of the library):
// write-only bounds, one-workgroup-per-frame vs persistent LDS composition, LDS-size (occupancy) sensitivity.
// Results of round 1 are recorded in profiles/r01_raster_microbench.md.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I../../endless-memory-gym_amd/csrc -o raster_bench raster_bench.hip
#include <cstdio>
#include <random>

#include "mg_atlas_v1.hpp"
#include "mg_raster_v1.hpp"

namespace mg {
using namespace v1;  // the structures compared here are those of raster generation 1 (mg_raster_v1.hpp)
void set_error(const std::string&) {}

// ---- variant: specialised 3-layer kernel of the first milestone (descriptor instead of display list) ----
struct OldDesc { int16_t sx, sy; uint16_t tmpl; uint8_t sprite, glyph; uint32_t pad[2]; };
struct OldAtlas {
    const uint8_t* templates; const uint8_t* sprites; const uint8_t* glyphs;
    int sprite_dim, glyph_box; int glyph_dim[10]; int glyph_x0; uint32_t palette[8];
};
__global__ __launch_bounds__(256) void old_kernel(const OldDesc* __restrict__ descs, OldAtlas A, uint8_t* __restrict__ obs) {
    extern __shared__ __attribute__((aligned(16))) uint8_t frame[];
    const int env = blockIdx.x, tid = threadIdx.x;
    const OldDesc d = descs[env];
    const uint4* src = reinterpret_cast<const uint4*>(A.templates + (size_t)d.tmpl * FRAME_BYTES);
    uint4* lds = reinterpret_cast<uint4*>(frame);
#pragma unroll
    for (int k = 0; k < 6; ++k) { int c = tid + k * 256; if (c < FRAME_VEC16) lds[c] = src[c]; }
    __syncthreads();
    if (d.sprite != 0xFF) {
        const int D = A.sprite_dim;
        const uint8_t* sp = A.sprites + (int)d.sprite * D * D;
        for (int p = tid; p < D * D; p += 256) {
            int px = p / D, py = p - px * D;
            uint8_t idx = sp[p];
            int X = d.sx + px, Y = d.sy + py;
            if (idx && (unsigned)X < 84u && (unsigned)Y < 84u) put_rgb(frame, X, Y, A.palette[idx]);
        }
    }
    if (d.glyph < 9) {
        __syncthreads();
        const int G = A.glyph_dim[d.glyph], B = A.glyph_box;
        const uint8_t* gp = A.glyphs + (int)d.glyph * B * B;
        for (int p = tid; p < G * G; p += 256) {
            int px = p / G, py = p - px * G;
            if (gp[px * B + py]) put_rgb(frame, A.glyph_x0 + px, A.glyph_x0 + py, 0x00FFFFFFu);
        }
    }
    __syncthreads();
    uint4* dst = reinterpret_cast<uint4*>(obs + (size_t)env * FRAME_BYTES);
#pragma unroll
    for (int k = 0; k < 6; ++k) { int c = tid + k * 256; if (c < FRAME_VEC16) dst[c] = lds[c]; }
}

// ---- bounds: pure fill and template copy without LDS ----
__global__ __launch_bounds__(256) void fill_kernel(uint8_t* __restrict__ obs) {
    uint4* dst = reinterpret_cast<uint4*>(obs + (size_t)blockIdx.x * FRAME_BYTES);
    uint4 v = make_uint4(blockIdx.x, 1, 2, 3);
#pragma unroll
    for (int k = 0; k < 6; ++k) { int c = threadIdx.x + k * 256; if (c < FRAME_VEC16) dst[c] = v; }
}
__global__ __launch_bounds__(256) void copy_kernel(const OldDesc* __restrict__ descs, const uint8_t* templates, uint8_t* __restrict__ obs) {
    const OldDesc d = descs[blockIdx.x];
    const uint4* src = reinterpret_cast<const uint4*>(templates + (size_t)d.tmpl * FRAME_BYTES);
    uint4* dst = reinterpret_cast<uint4*>(obs + (size_t)blockIdx.x * FRAME_BYTES);
#pragma unroll
    for (int k = 0; k < 6; ++k) { int c = threadIdx.x + k * 256; if (c < FRAME_VEC16) dst[c] = src[c]; }
}
// grid-stride persistent fill: 2048 blocks
__global__ __launch_bounds__(256) void fill_persistent(uint8_t* __restrict__ obs, size_t nvec) {
    uint4* dst = reinterpret_cast<uint4*>(obs);
    uint4 v = make_uint4(1, 1, 2, 3);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (size_t)gridDim.x * 256) dst[i] = v;
}
// persistent frame-contiguous fill: WG loops over frames
__global__ __launch_bounds__(256) void fill_persistent_frames(uint8_t* __restrict__ obs, int n) {
    uint4 v = make_uint4(1, 1, 2, 3);
    for (int f = blockIdx.x; f < n; f += gridDim.x) {
        uint4* dst = reinterpret_cast<uint4*>(obs + (size_t)f * FRAME_BYTES);
#pragma unroll
        for (int k = 0; k < 6; ++k) { int c = threadIdx.x + k * 256; if (c < FRAME_VEC16) dst[c] = v; }
    }
}
// persistent version of the specialised LDS kernel; optional prefetch of the next descriptor
template <bool PREFETCH>
__global__ __launch_bounds__(256) void old_persistent(const OldDesc* __restrict__ descs, OldAtlas A, uint8_t* __restrict__ obs, int n) {
    extern __shared__ __attribute__((aligned(16))) uint8_t frame[];
    const int tid = threadIdx.x;
    uint4* lds = reinterpret_cast<uint4*>(frame);
    OldDesc d = descs[blockIdx.x];
    for (int env = blockIdx.x; env < n; env += gridDim.x) {
        OldDesc dn = d;
        int nxt = env + gridDim.x;
        if (PREFETCH) { if (nxt < n) dn = descs[nxt]; } 
        const uint4* src = reinterpret_cast<const uint4*>(A.templates + (size_t)d.tmpl * FRAME_BYTES);
#pragma unroll
        for (int k = 0; k < 6; ++k) { int c = tid + k * 256; if (c < FRAME_VEC16) lds[c] = src[c]; }
        __syncthreads();
        if (d.sprite != 0xFF) {
            const int D = A.sprite_dim;
            const uint8_t* sp = A.sprites + (int)d.sprite * D * D;
            for (int p = tid; p < D * D; p += 256) {
                int px = p / D, py = p - px * D;
                uint8_t idx = sp[p];
                int X = d.sx + px, Y = d.sy + py;
                if (idx && (unsigned)X < 84u && (unsigned)Y < 84u) put_rgb(frame, X, Y, A.palette[idx]);
            }
        }
        if (d.glyph < 9) {
            __syncthreads();
            const int G = A.glyph_dim[d.glyph], B = A.glyph_box;
            const uint8_t* gp = A.glyphs + (int)d.glyph * B * B;
            for (int p = tid; p < G * G; p += 256) {
                int px = p / G, py = p - px * G;
                if (gp[px * B + py]) put_rgb(frame, A.glyph_x0 + px, A.glyph_x0 + py, 0x00FFFFFFu);
            }
        }
        __syncthreads();
        uint4* dst = reinterpret_cast<uint4*>(obs + (size_t)env * FRAME_BYTES);
#pragma unroll
        for (int k = 0; k < 6; ++k) { int c = tid + k * 256; if (c < FRAME_VEC16) dst[c] = lds[c]; }
        __syncthreads();  // LDS is reused by the next frame
        if (PREFETCH) d = dn; else if (nxt < n) d = descs[nxt];
    }
}
}  // namespace mg

using namespace mg;
using namespace mg::v1;  // the structures compared here are those of raster generation 1

template <typename F>
static double time_it(const char* name, int n, F&& launch, int iters = 30) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 5; ++i) launch();
    hipDeviceSynchronize();
    hipEventRecord(a, 0);
    for (int i = 0; i < iters; ++i) launch();
    hipEventRecord(b, 0);
    hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    double us = ms * 1e3 / iters, gbps = (double)n * FRAME_BYTES / (us * 1e-6) / 1e9;
    printf("%-28s %8.1f us/launch  %7.1f GB/s  (%.1f%% of 8 TB/s)\n", name, us, gbps, gbps / 80.0);
    return us;
}

int main(int argc, char** argv) {
    int n = argc > 1 ? atoi(argv[1]) : 65536;
    int radius;
    auto sprites = build_agent_sprites(0.25, &radius);
    auto glyphs = build_glyphs(0.25);
    auto templ = build_mortar_templates(5, 0.25, 84);
    std::mt19937 rng(1);
    std::vector<OldDesc> descs(n);
    for (int i = 0; i < n; ++i) {
        int t = rng() % 26, sp = (rng() % 4) * 2, gx = rng() % 5, gy = rng() % 5, gl = (rng() % 10 < 6) ? (rng() % 5) : 0xFF;
        int sx = 7 + 14 * gx + 7 - 14, sy = 7 + 14 * gy + 7 - 14;
        descs[i] = OldDesc{(int16_t)sx, (int16_t)sy, (uint16_t)t, (uint8_t)sp, (uint8_t)gl, {0, 0}};
    }
    DevArray<OldDesc> dd; dd.upload(descs);
    DevArray<uint8_t> obs; obs.alloc((size_t)n * FRAME_BYTES, false);

    // old atlas
    OldAtlas OA;
    int D = sprites[0].w, B = 31;
    std::vector<uint8_t> sp((size_t)8 * D * D), gl((size_t)10 * B * B, 0);
    for (int k = 0; k < 8; ++k) for (int x = 0; x < D; ++x) for (int y = 0; y < D; ++y) sp[(size_t)k * D * D + x * D + y] = sprites[k].get(x, y);
    for (int k = 0; k < 10; ++k) { OA.glyph_dim[k] = glyphs[k].w; for (int x = 0; x < glyphs[k].w; ++x) for (int y = 0; y < glyphs[k].h; ++y) gl[(size_t)k * B * B + x * B + y] = glyphs[k].get(x, y) ? 1 : 0; }
    DevArray<uint8_t> spd, gld, tpd; spd.upload(sp); gld.upload(gl); tpd.upload(templ);
    OA.templates = tpd.p; OA.sprites = spd.p; OA.glyphs = gld.p; OA.sprite_dim = D; OA.glyph_box = B; OA.glyph_x0 = 31;
    uint32_t pal[8] = {0, 250 | (204 << 8) | (153 << 16), 0xFAFAFA, 0x323232, 0xFFFFFF, 0xFF, 0, 0};
    for (int k = 0; k < 8; ++k) OA.palette[k] = pal[k];

    printf("n = %d instances, %d bytes each\n", n, FRAME_BYTES);
    time_it("fill (1 WG/frame)", n, [&] { hipLaunchKernelGGL(fill_kernel, dim3(n), dim3(256), 0, 0, obs.p); });
    time_it("fill persistent 2048 WG", n, [&] { hipLaunchKernelGGL(fill_persistent, dim3(2048), dim3(256), 0, 0, obs.p, (size_t)n * FRAME_VEC16); });
    time_it("template copy (no LDS)", n, [&] { hipLaunchKernelGGL(copy_kernel, dim3(n), dim3(256), 0, 0, dd.p, tpd.p, obs.p); });
    time_it("old specialised kernel", n, [&] { hipLaunchKernelGGL(old_kernel, dim3(n), dim3(256), FRAME_BYTES, 0, dd.p, OA, obs.p); });
    for (int lds : {21168, 22976, 23552, 24576, 27000, 32768, 40000})  {
        char nm[64]; snprintf(nm, 64, "old kernel, LDS %d", lds);
        time_it(nm, n, [&] { hipLaunchKernelGGL(old_kernel, dim3(n), dim3(256), lds, 0, dd.p, OA, obs.p); });
    }
    for (int g : {1024, 1792, 2048, 4096})  {
        char nm[64]; snprintf(nm, 64, "persistent frame fill, %d WG", g);
        time_it(nm, n, [&] { hipLaunchKernelGGL(fill_persistent_frames, dim3(g), dim3(256), 0, 0, obs.p, n); });
    }
    for (int g : {1024, 1792, 3584})  {
        char nm[64]; snprintf(nm, 64, "persistent old, %d WG", g);
        time_it(nm, n, [&] { hipLaunchKernelGGL(old_persistent<false>, dim3(g), dim3(256), FRAME_BYTES, 0, dd.p, OA, obs.p, n); });
        snprintf(nm, 64, "persistent old+prefetch, %d WG", g);
        time_it(nm, n, [&] { hipLaunchKernelGGL(old_persistent<true>, dim3(g), dim3(256), FRAME_BYTES, 0, dd.p, OA, obs.p, n); });
    }
    time_it("fill (1 WG/frame) again", n, [&] { hipLaunchKernelGGL(fill_kernel, dim3(n), dim3(256), 0, 0, obs.p); });
    time_it("old specialised again", n, [&] { hipLaunchKernelGGL(old_kernel, dim3(n), dim3(256), FRAME_BYTES, 0, dd.p, OA, obs.p); });
    return 0;
}
