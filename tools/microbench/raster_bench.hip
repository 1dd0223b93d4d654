// tools/microbench/raster_bench.hip -- standalone A/B harness for raster-kernel variants (not part of the library).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I../../endless-memory-gym_amd/csrc -o raster_bench raster_bench.hip
#include <cstdio>
#include <random>

#include "mg_atlas.hpp"
#include "mg_raster.hip"  // kernel under test (generic display-list raster)

namespace mg {
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
// ---- candidate: templated persistent skeleton + inlined helpers, atlas tables in LDS ----
struct MMDesc { int16_t sx, sy; uint16_t tmpl; uint8_t sprite, glyph; uint32_t pad[2]; };
__device__ __forceinline__ void h_template(uint4* lds16, const uint8_t* templates, int t, int tid) {
    const uint4* src = reinterpret_cast<const uint4*>(templates + (size_t)t * FRAME_BYTES);
    uint4 v0 = src[tid], v1 = src[tid + 256], v2 = src[tid + 512], v3 = src[tid + 768], v4 = src[tid + 1024];
    uint4 v5 = make_uint4(0, 0, 0, 0);
    if (tid < TAIL) v5 = src[tid + 1280];
    lds16[tid] = v0; lds16[tid + 256] = v1; lds16[tid + 512] = v2; lds16[tid + 768] = v3; lds16[tid + 1024] = v4;
    if (tid < TAIL) lds16[tid + 1280] = v5;
}
template <int TABMODE>
__device__ __forceinline__ void h_stamp(uint8_t* frame, const RasterAtlas& A, const AtlasTables* T, int id, int x, int y, int tid) {
    const StampInfo si = TABMODE == 0 ? T->stamps[id] : A.tables->stamps[id];
    const uint8_t* sp = A.stamp_data + si.off;
    const int w = si.w, h = si.h, npx = w * h;
    for (int p = tid; p < npx; p += 256) {
        int px = p / h, py = p - px * h;
        uint8_t idx = sp[p];
        int X = x + px, Y = y + py;
        if (idx && (unsigned)X < 84u && (unsigned)Y < 84u) put_rgb(frame, X, Y, TABMODE == 0 ? T->palette[idx] : A.tables->palette[idx]);
    }
}
template <int TABMODE>
__global__ __launch_bounds__(256) void cand_kernel(const MMDesc* __restrict__ descs, RasterAtlas A, uint8_t* __restrict__ obs, int n) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    uint8_t* frame = smem;
    AtlasTables* T = reinterpret_cast<AtlasTables*>(smem + FRAME_BYTES);
    const int tid = threadIdx.x;
    uint4* lds16 = reinterpret_cast<uint4*>(frame);
    if (TABMODE == 0) {
        if (tid < (int)(sizeof(AtlasTables) / 4)) reinterpret_cast<uint32_t*>(T)[tid] = reinterpret_cast<const uint32_t*>(A.tables)[tid];
        __syncthreads();
    }
    for (int env = blockIdx.x; env < n; env += gridDim.x) {
        const MMDesc d = descs[env];
        h_template(lds16, A.templates, d.tmpl, tid);
        __syncthreads();
        if (d.sprite != 0xFF) h_stamp<TABMODE>(frame, A, T, d.sprite, d.sx, d.sy, tid);
        if (d.glyph < 9) {
            __syncthreads();
            h_stamp<TABMODE>(frame, A, T, 8 + d.glyph, 31, 31, tid);
        }
        __syncthreads();
        uint4* dst = reinterpret_cast<uint4*>(obs + (size_t)env * FRAME_BYTES);
        uint4 v0 = lds16[tid], v1 = lds16[tid + 256], v2 = lds16[tid + 512], v3 = lds16[tid + 768], v4 = lds16[tid + 1024];
        uint4 v5 = make_uint4(0, 0, 0, 0);
        if (tid < TAIL) v5 = lds16[tid + 1280];
        dst[tid] = v0; dst[tid + 256] = v1; dst[tid + 512] = v2; dst[tid + 768] = v3; dst[tid + 1024] = v4;
        if (tid < TAIL) dst[tid + 1280] = v5;
        __syncthreads();
    }
}
}  // namespace mg

using namespace mg;

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
    Atlas atlas;
    for (auto& s : sprites) atlas.add_stamp(s);
    for (auto& g : glyphs) atlas.add_stamp(g);
    auto templ = build_mortar_templates(5, 0.25, 84);
    atlas.set_templates(templ);
    atlas.upload();

    std::mt19937 rng(1);
    std::vector<DrawList> lists(n);
    std::vector<OldDesc> descs(n);
    for (int i = 0; i < n; ++i) {
        int t = rng() % 26, sp = (rng() % 4) * 2, gx = rng() % 5, gy = rng() % 5, gl = (rng() % 10 < 6) ? (rng() % 5) : 0xFF;
        int sx = 7 + 14 * gx + 7 - 14, sy = 7 + 14 * gy + 7 - 14;
        DrawList& L = lists[i];
        int k = 0;
        L.c[k++] = DrawCmd{OP_TEMPLATE, 0, 0, 0, (uint16_t)t};
        L.c[k++] = DrawCmd{(uint8_t)(OP_STAMP | SYNC_BEFORE), (uint8_t)sp, (int16_t)sx, (int16_t)sy, 0};
        if (gl != 0xFF) L.c[k++] = DrawCmd{(uint8_t)(OP_STAMP | SYNC_BEFORE), (uint8_t)(8 + gl), 31, 31, 0};
        L.c[k++] = DrawCmd{OP_END, 0, 0, 0, 0};
        descs[i] = OldDesc{(int16_t)sx, (int16_t)sy, (uint16_t)t, (uint8_t)sp, (uint8_t)gl, {0, 0}};
    }
    DevArray<DrawList> dl; dl.upload(lists);
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
    for (int mode = 0; mode < 3; ++mode) for (int g : {1792, 3584, 65536}) {
        g_raster_listmode = mode; g_raster_grid = g;
        char nm[64]; snprintf(nm, 64, "generic mode %d grid %d", mode, g);
        time_it(nm, n, [&] { launch_raster(dl.p, atlas.dev(), obs.p, n, 0); });
    }
    g_raster_listmode = 0; g_raster_grid = 3584;
    {
        std::vector<DrawList> l2 = lists;
        for (auto& L : l2) L.c[1] = DrawCmd{OP_END, 0, 0, 0, 0};
        DevArray<DrawList> d2; d2.upload(l2);
        time_it("generic: template only", n, [&] { launch_raster(d2.p, atlas.dev(), obs.p, n, 0); });
        l2 = lists;
        for (auto& L : l2) L.c[2] = DrawCmd{OP_END, 0, 0, 0, 0};
        d2.upload(l2);
        time_it("generic: template+sprite", n, [&] { launch_raster(d2.p, atlas.dev(), obs.p, n, 0); });
        l2 = lists;
        for (auto& L : l2) { L.c[0] = DrawCmd{OP_CLEAR, 0, 0, 0, 0}; L.c[1] = DrawCmd{OP_END, 0, 0, 0, 0}; }
        d2.upload(l2);
        time_it("generic: clear only", n, [&] { launch_raster(d2.p, atlas.dev(), obs.p, n, 0); });
    }
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
    {
        DevArray<MMDesc> md; std::vector<MMDesc> mdv(n);
        for (int i = 0; i < n; ++i) mdv[i] = MMDesc{descs[i].sx, descs[i].sy, descs[i].tmpl, descs[i].sprite, descs[i].glyph, {0, 0}};
        md.upload(mdv);
        time_it("candidate (LDS tables) 3584", n, [&] { hipLaunchKernelGGL(cand_kernel<0>, dim3(3584), dim3(256), FRAME_BYTES + sizeof(AtlasTables), 0, md.p, atlas.dev(), obs.p, n); });
        time_it("candidate (global tables) 3584", n, [&] { hipLaunchKernelGGL(cand_kernel<1>, dim3(3584), dim3(256), FRAME_BYTES + sizeof(AtlasTables), 0, md.p, atlas.dev(), obs.p, n); });
        time_it("candidate (LDS tables) 1792", n, [&] { hipLaunchKernelGGL(cand_kernel<0>, dim3(1792), dim3(256), FRAME_BYTES + sizeof(AtlasTables), 0, md.p, atlas.dev(), obs.p, n); });
    }
    time_it("fill (1 WG/frame) again", n, [&] { hipLaunchKernelGGL(fill_kernel, dim3(n), dim3(256), 0, 0, obs.p); });
    time_it("old specialised again", n, [&] { hipLaunchKernelGGL(old_kernel, dim3(n), dim3(256), FRAME_BYTES, 0, dd.p, OA, obs.p); });
    time_it("generic again", n, [&] { launch_raster(dl.p, atlas.dev(), obs.p, n, 0); });
    return 0;
}
