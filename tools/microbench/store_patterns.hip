// tools/microbench/store_patterns.hip -- which ORDER of 16-byte stores does the MI355X memory system digest best when
// 65,536 frames of 21,168 B (1.387 GB) are written?  Pure store streams, no LDS, constant data.
// Build: hipcc --offload-arch=gfx950 -O3 -o store_patterns store_patterns.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
constexpr int FRAME_VEC = 1323;  // 21,168 / 16
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// V1: linear, one vector per thread (what torch's fill does)
__global__ void v_linear(u32x4* out, size_t nvec) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nvec) out[i] = (u32x4)(0x01020304u);
}
// V1b: linear, grid-stride
__global__ void v_linear_gs(u32x4* out, size_t nvec) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += (size_t)gridDim.x * blockDim.x) out[i] = (u32x4)(0x01020304u);
}
// frame walkers: workgroup (256 lanes) writes one frame as 6 strided vectors per lane; `order` decides which frame next
template <int MODE>
__global__ __launch_bounds__(256) void v_frames(u32x4* out, int n, int per_wg) {
    const int tid = threadIdx.x;
    for (int k = 0;; ++k) {
        int f;
        if (MODE == 0) f = blockIdx.x + k * gridDim.x;       // persistent, stride = grid
        else f = blockIdx.x * per_wg + k;                    // contiguous block of frames per workgroup
        if (MODE == 0 ? f >= n : (k >= per_wg || f >= n)) break;
        u32x4* dst = out + (size_t)f * FRAME_VEC;
#pragma unroll
        for (int j = 0; j < 5; ++j) dst[tid + 256 * j] = (u32x4)(0x01020304u);
        if (tid < FRAME_VEC - 1280) dst[tid + 1280] = (u32x4)(0x01020304u);
    }
}
// the same frame walk with bigger workgroups: fewer frames in flight per CU (2,048 lanes / BLOCK), each written faster
template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void v_frames_big(u32x4* out, int n) {
    const int tid = threadIdx.x;
    for (int f = blockIdx.x; f < n; f += gridDim.x) {
        u32x4* dst = out + (size_t)f * FRAME_VEC;
        for (int v = tid; v < FRAME_VEC; v += BLOCK) dst[v] = (u32x4)(0x01020304u);
    }
}
// frame walk that limits the stores a wave has in flight: wait for all of them after every WAIT-th store
template <int WAIT>
__global__ __launch_bounds__(256) void v_frames_throttled(u32x4* out, int n) {
    extern __shared__ unsigned char lds_pad[];  // dynamic LDS only limits how many workgroups share a CU
    const int tid = threadIdx.x;
    for (int f = blockIdx.x; f < n; f += gridDim.x) {
        u32x4* dst = out + (size_t)f * FRAME_VEC;
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            dst[tid + 256 * j] = (u32x4)(0x01020304u);
            if (WAIT && (j + 1) % WAIT == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        if (tid < FRAME_VEC - 1280) dst[tid + 1280] = (u32x4)(0x01020304u);
        if (WAIT) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
}
// linear, K vectors per thread at a stride of 256 vectors: workgroup b writes the contiguous K * 4 KB block b
template <int K>
__global__ __launch_bounds__(256) void v_linear_k(u32x4* out, size_t nvec) {
    const size_t base = (size_t)blockIdx.x * 256 * K + threadIdx.x;
#pragma unroll
    for (int j = 0; j < K; ++j)
        if (base + 256 * j < nvec) out[base + 256 * j] = (u32x4)(0x01020304u);
}
// output-centric template copy: thread = one 16-byte vector of the whole batch, reads its frame's descriptor word and
// the matching vector of one of 26 L2-resident templates, stores once (no LDS, no barrier)
__global__ __launch_bounds__(256) void v_gather_copy(u32x4* out, size_t nvec, const unsigned* desc, const u32x4* templ) {
    const size_t v = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (v >= nvec) return;
    const unsigned f = (unsigned)(v / FRAME_VEC), k = (unsigned)(v - (size_t)f * FRAME_VEC);
    const unsigned t = desc[4 * f + 1] & 31u;
    out[v] = templ[(size_t)t * FRAME_VEC + k];
}
// the same without the descriptor: every frame copies template 0 (one L2-resident 21 KB block, no dependent load chain)
__global__ __launch_bounds__(256) void v_gather_copy_t0(u32x4* out, size_t nvec, const u32x4* templ) {
    const size_t v = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (v >= nvec) return;
    const unsigned k = (unsigned)(v % FRAME_VEC);
    out[v] = templ[k];
}
// descriptor and a speculative template-0 vector requested together; the real template only if it differs (25 %)
__global__ __launch_bounds__(256) void v_gather_copy_spec(u32x4* out, size_t nvec, const unsigned* desc, const u32x4* templ) {
    const size_t v = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (v >= nvec) return;
    const unsigned f = (unsigned)(v / FRAME_VEC), k = (unsigned)(v - (size_t)f * FRAME_VEC);
    const unsigned t = desc[4 * f + 1] & 31u;
    u32x4 val = templ[k];
    if (t >= 19) val = templ[(size_t)t * FRAME_VEC + k];
    out[v] = val;
}
// descriptor through the scalar cache: the wave's first frame index is uniform, lanes past a frame boundary use f0 + 1
typedef const unsigned __attribute__((address_space(4))) * cuptr;
__global__ __launch_bounds__(256) void v_gather_copy_scalar(u32x4* out, size_t nvec, const unsigned* desc, const u32x4* templ, int nframes) {
    const size_t v = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (v >= nvec) return;
    const unsigned f = (unsigned)(v / FRAME_VEC), k = (unsigned)(v - (size_t)f * FRAME_VEC);
    const unsigned f0 = __builtin_amdgcn_readfirstlane(f);
    cuptr d = (cuptr)desc;
    const unsigned t0 = d[4 * f0 + 1] & 31u, t1 = d[4 * (f0 + 1 < (unsigned)nframes ? f0 + 1 : f0) + 1] & 31u;
    const unsigned t = f == f0 ? t0 : t1;
    out[v] = templ[(size_t)t * FRAME_VEC + k];
}
// linear, K ADJACENT vectors per thread (thread writes 16 K contiguous bytes, workgroup K x 4 KB)
template <int K>
__global__ __launch_bounds__(256) void v_linear_adj(u32x4* out, size_t nvec) {
    const size_t base = ((size_t)blockIdx.x * 256 + threadIdx.x) * K;
#pragma unroll
    for (int j = 0; j < K; ++j)
        if (base + j < nvec) out[base + j] = (u32x4)(0x01020304u);
}
// gather copy with K adjacent vectors per thread and the descriptor through scalar loads
template <int K>
__global__ __launch_bounds__(256) void v_gather_adj(u32x4* out, size_t nvec, const unsigned* desc, const u32x4* templ, int nframes) {
    const size_t base = ((size_t)blockIdx.x * 256 + threadIdx.x) * K;
    if (base >= nvec) return;
    const unsigned f = (unsigned)(base / FRAME_VEC);
    const unsigned f0 = __builtin_amdgcn_readfirstlane(f);
    cuptr d = (cuptr)desc;
    unsigned tt[3];
#pragma unroll
    for (int q = 0; q < 3; ++q) tt[q] = d[4 * (f0 + q < (unsigned)nframes ? f0 + q : f0) + 1] & 31u;
    u32x4 val[K];
#pragma unroll
    for (int j = 0; j < K; ++j) {
        const size_t v = base + j < nvec ? base + j : nvec - 1;
        const unsigned fj = (unsigned)(v / FRAME_VEC), k = (unsigned)(v - (size_t)fj * FRAME_VEC);
        const unsigned t = fj == f0 ? tt[0] : (fj == f0 + 1 ? tt[1] : tt[2]);
        val[j] = templ[(size_t)t * FRAME_VEC + k];
    }
#pragma unroll
    for (int j = 0; j < K; ++j)
        if (base + j < nvec) out[base + j] = val[j];
}
// alignment probe: the same 1,323 vectors per frame, frames placed at a stride of STRIDE_VEC vectors
template <int STRIDE_VEC>
__global__ __launch_bounds__(256) void v_frames_stride(u32x4* out, int n) {
    const int tid = threadIdx.x;
    for (int f = blockIdx.x; f < n; f += gridDim.x) {
        u32x4* dst = out + (size_t)f * STRIDE_VEC;
#pragma unroll
        for (int j = 0; j < 5; ++j) dst[tid + 256 * j] = (u32x4)(0x01020304u);
        if (tid < FRAME_VEC - 1280) dst[tid + 1280] = (u32x4)(0x01020304u);
    }
}
// each WAVE writes a contiguous quarter of the frame (5.3 KB) instead of interleaved 1-KB chunks
__global__ __launch_bounds__(256) void v_frames_wavecontig(u32x4* out, int n) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int f = blockIdx.x; f < n; f += gridDim.x) {
        u32x4* dst = out + (size_t)f * FRAME_VEC;
        const int lo = w * 331, hi = w == 3 ? FRAME_VEC : lo + 331;  // 331 * 4 = 1324 >= 1323
        for (int v = lo + lane; v < hi; v += 64) dst[v] = (u32x4)(0x01020304u);
    }
}

template <class F>
static void run(const char* name, F launch, size_t bytes) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 3; ++i) launch();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int i = 0; i < 10; ++i) launch();
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms;
    CK(hipEventElapsedTime(&ms, a, b));
    ms /= 10;
    printf("%-58s %7.1f us  %5.2f TB/s\n", name, ms * 1e3, bytes / ms / 1e9);
}

int main(int argc, char** argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 65536;
    const size_t nvec = (size_t)n * FRAME_VEC, bytes = nvec * 16;
    u32x4* out;
    CK(hipMalloc(&out, bytes));
    CK(hipFuncSetAttribute((const void*)v_frames_throttled<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160000));
    CK(hipFuncSetAttribute((const void*)v_frames_throttled<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160000));
    CK(hipFuncSetAttribute((const void*)v_frames_throttled<8>, hipFuncAttributeMaxDynamicSharedMemorySize, 160000));
    printf("%d frames, %.1f MB\n", n, bytes / 1e6);
    run("linear, 1 vector/thread (torch fill)", [&] { hipLaunchKernelGGL(v_linear, dim3((nvec + 255) / 256), dim3(256), 0, 0, out, nvec); }, bytes);
    for (int g : {2048, 16384}) {
        char nm[96]; snprintf(nm, sizeof nm, "linear grid-stride, grid %d", g);
        run(nm, [&] { hipLaunchKernelGGL(v_linear_gs, dim3(g), dim3(256), 0, 0, out, nvec); }, bytes);
    }
    for (int g : {1792, 2048, 3584, 7168, 14336, 28672, 65536}) {
        if (g > n) continue;
        char nm[96]; snprintf(nm, sizeof nm, "frame walker, persistent stride, grid %d", g);
        run(nm, [&] { hipLaunchKernelGGL(v_frames<0>, dim3(g), dim3(256), 0, 0, out, n, 0); }, bytes);
    }
    for (int per : {2, 4, 8, 37}) {
        char nm[96]; snprintf(nm, sizeof nm, "frame walker, %d consecutive frames per workgroup", per);
        run(nm, [&] { hipLaunchKernelGGL(v_frames<1>, dim3((n + per - 1) / per), dim3(256), 0, 0, out, n, per); }, bytes);
    }
    {
        unsigned* desc;
        u32x4* templ;
        CK(hipMalloc(&desc, (size_t)n * 16));
        CK(hipMalloc(&templ, (size_t)26 * FRAME_VEC * 16));
        std::vector<unsigned> h((size_t)n * 4);
        for (int i = 0; i < n; ++i) { h[4 * i] = i; h[4 * i + 1] = (unsigned)((i * 2654435761u) >> 20) % 26; h[4 * i + 2] = h[4 * i + 3] = 0; }
        CK(hipMemcpy(desc, h.data(), h.size() * 4, hipMemcpyHostToDevice));
        CK(hipMemset(templ, 7, (size_t)26 * FRAME_VEC * 16));
        run("gather copy of template 0, no descriptor", [&] { hipLaunchKernelGGL(v_gather_copy_t0, dim3((nvec + 255) / 256), dim3(256), 0, 0, out, nvec, templ); }, bytes);
        run("gather copy, speculative template 0 + descriptor in parallel", [&] { hipLaunchKernelGGL(v_gather_copy_spec, dim3((nvec + 255) / 256), dim3(256), 0, 0, out, nvec, desc, templ); }, bytes);
        run("gather copy, scalar descriptor, 2 adjacent vectors/thread", [&] { hipLaunchKernelGGL(v_gather_adj<2>, dim3((nvec / 2 + 255) / 256), dim3(256), 0, 0, out, nvec, desc, templ, n); }, bytes);
        run("gather copy, scalar descriptor, 4 adjacent vectors/thread", [&] { hipLaunchKernelGGL(v_gather_adj<4>, dim3((nvec / 4 + 255) / 256), dim3(256), 0, 0, out, nvec, desc, templ, n); }, bytes);
        run("gather copy, descriptor via scalar loads (constant AS)", [&] { hipLaunchKernelGGL(v_gather_copy_scalar, dim3((nvec + 255) / 256), dim3(256), 0, 0, out, nvec, desc, templ, n); }, bytes);
        run("gather copy: descriptor + template vector -> 1 store/thread", [&] { hipLaunchKernelGGL(v_gather_copy, dim3((nvec + 255) / 256), dim3(256), 0, 0, out, nvec, desc, templ); }, bytes);
    }
    run("linear, 2 ADJACENT vectors/thread", [&] { hipLaunchKernelGGL(v_linear_adj<2>, dim3((nvec / 2 + 255) / 256), dim3(256), 0, 0, out, nvec); }, bytes);
    run("linear, 4 ADJACENT vectors/thread", [&] { hipLaunchKernelGGL(v_linear_adj<4>, dim3((nvec / 4 + 255) / 256), dim3(256), 0, 0, out, nvec); }, bytes);
    run("linear, 8 ADJACENT vectors/thread", [&] { hipLaunchKernelGGL(v_linear_adj<8>, dim3((nvec / 8 + 255) / 256), dim3(256), 0, 0, out, nvec); }, bytes);
    run("linear, 2 vectors/thread", [&] { hipLaunchKernelGGL(v_linear_k<2>, dim3((nvec + 511) / 512), dim3(256), 0, 0, out, nvec); }, bytes);
    run("linear, 3 vectors/thread", [&] { hipLaunchKernelGGL(v_linear_k<3>, dim3((nvec + 767) / 768), dim3(256), 0, 0, out, nvec); }, bytes);
    run("linear, 4 vectors/thread", [&] { hipLaunchKernelGGL(v_linear_k<4>, dim3((nvec + 1023) / 1024), dim3(256), 0, 0, out, nvec); }, bytes);
    run("linear, 6 vectors/thread", [&] { hipLaunchKernelGGL(v_linear_k<6>, dim3((nvec + 1535) / 1536), dim3(256), 0, 0, out, nvec); }, bytes);
    run("linear, 8 vectors/thread", [&] { hipLaunchKernelGGL(v_linear_k<8>, dim3((nvec + 2047) / 2048), dim3(256), 0, 0, out, nvec); }, bytes);
    run("linear, 16 vectors/thread", [&] { hipLaunchKernelGGL(v_linear_k<16>, dim3((nvec + 4095) / 4096), dim3(256), 0, 0, out, nvec); }, bytes);
    {
        u32x4* big;
        CK(hipMalloc(&big, (size_t)n * 1536 * 16));
        for (int g : {14336, 65536}) {
            char nm[96];
            snprintf(nm, sizeof nm, "frame stride 21,168 B (1323 vec), grid %d", g);
            run(nm, [&] { hipLaunchKernelGGL(v_frames_stride<1323>, dim3(g), dim3(256), 0, 0, big, n); }, bytes);
            snprintf(nm, sizeof nm, "frame stride 21,248 B (128-B aligned), grid %d", g);
            run(nm, [&] { hipLaunchKernelGGL(v_frames_stride<1328>, dim3(g), dim3(256), 0, 0, big, n); }, bytes);
            snprintf(nm, sizeof nm, "frame stride 21,504 B (512-B aligned), grid %d", g);
            run(nm, [&] { hipLaunchKernelGGL(v_frames_stride<1344>, dim3(g), dim3(256), 0, 0, big, n); }, bytes);
            snprintf(nm, sizeof nm, "frame stride 24,576 B (4-KB aligned), grid %d", g);
            run(nm, [&] { hipLaunchKernelGGL(v_frames_stride<1536>, dim3(g), dim3(256), 0, 0, big, n); }, bytes);
        }
        CK(hipFree(big));
    }
    for (int lds : {0, 54000}) {
        for (int g : {14336, 65536}) {
            char nm[96];
            snprintf(nm, sizeof nm, "frame walker, LDS %d B/WG, grid %d, wait after each store", lds, g);
            run(nm, [&] { hipLaunchKernelGGL(v_frames_throttled<1>, dim3(g), dim3(256), lds, 0, out, n); }, bytes);
            snprintf(nm, sizeof nm, "frame walker, LDS %d B/WG, grid %d, wait per frame", lds, g);
            run(nm, [&] { hipLaunchKernelGGL(v_frames_throttled<8>, dim3(g), dim3(256), lds, 0, out, n); }, bytes);
            snprintf(nm, sizeof nm, "frame walker, LDS %d B/WG, grid %d, no wait", lds, g);
            run(nm, [&] { hipLaunchKernelGGL(v_frames_throttled<0>, dim3(g), dim3(256), lds, 0, out, n); }, bytes);
        }
    }
    for (int g : {512, 1024, 2048, 4096, 16384, 65536}) {
        if (g > n) continue;
        char nm[96]; snprintf(nm, sizeof nm, "frame walker, 1024-lane workgroups, grid %d", g);
        run(nm, [&] { hipLaunchKernelGGL(v_frames_big<1024>, dim3(g), dim3(1024), 0, 0, out, n); }, bytes);
    }
    for (int g : {1024, 2048, 4096, 16384, 65536}) {
        if (g > n) continue;
        char nm[96]; snprintf(nm, sizeof nm, "frame walker, 512-lane workgroups, grid %d", g);
        run(nm, [&] { hipLaunchKernelGGL(v_frames_big<512>, dim3(g), dim3(512), 0, 0, out, n); }, bytes);
    }
    for (int g : {1792, 14336}) {
        char nm[96]; snprintf(nm, sizeof nm, "frame walker, wave-contiguous quarters, grid %d", g);
        run(nm, [&] { hipLaunchKernelGGL(v_frames_wavecontig, dim3(g), dim3(256), 0, 0, out, n); }, bytes);
    }
    return 0;
}
