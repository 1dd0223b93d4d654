// tools/vmm/placement_lab.hip -- measurement helper (not product code): device allocations of every kind the HIP runtime
// offers, so that tools/placement_lab.py can time the raster kernel's store stream into each.  Built on the spot:
//   hipcc --offload-arch=gfx950 -O3 -fPIC -shared -o tools/vmm/libplacement_lab.so tools/vmm/placement_lab.hip
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <map>
#include <random>
#include <vector>

namespace {
struct Vmm {
    void* va = nullptr;
    size_t va_size = 0;
    std::vector<hipMemGenericAllocationHandle_t> handles;
};
std::map<void*, Vmm> g_vmm;
char g_err[256] = "";
#define LAB(x)                                                                       \
    do {                                                                             \
        hipError_t e_ = (x);                                                         \
        if (e_ != hipSuccess) {                                                      \
            snprintf(g_err, sizeof g_err, "%s: %s", #x, hipGetErrorString(e_));      \
            return -1;                                                               \
        }                                                                            \
    } while (0)

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__global__ void fill_linear(u32x4* out, size_t nvec) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nvec) out[i] = (u32x4)(0x01020304u);
}
// frame walker over two bases: persistent workgroup b writes frame (b, k) for k = 0..nwin-1; windows alternate between the bases
__global__ __launch_bounds__(256) void fw2(u32x4* base0, u32x4* base1, int G, int nwin, int wrap_frames) {
    extern __shared__ unsigned char pad[];
    const int tid = threadIdx.x;
    for (int k = 0; k < nwin; ++k) {
        long f = (long)(k / 2) * G + blockIdx.x;
        if (wrap_frames) f %= wrap_frames;
        u32x4* dst = ((k & 1) ? base1 : base0) + f * 1323;
#pragma unroll
        for (int j = 0; j < 5; ++j) dst[tid + 256 * j] = (u32x4)(0x01020304u + k);
        if (tid < 43) dst[tid + 1280] = (u32x4)(0x01020304u + k);
    }
}
__global__ void dummy_rw(unsigned* p, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = p[i] * 3 + 1;
}
}  // namespace

extern "C" {
// average microseconds of `reps` fw2 launches (a small read-modify-write kernel on `scratch` runs between them if scratch != 0)
double lab_fw2_us(void* base0, void* base1, int G, int nwin, int wrap_frames, int lds_bytes, void* scratch, size_t scratch_bytes, int reps) {
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    double total = 0;
    for (int r = -2; r < reps; ++r) {
        if (scratch) dummy_rw<<<1024, 256>>>((unsigned*)scratch, scratch_bytes / 4);
        hipEventRecord(a, 0);
        fw2<<<G, 256, lds_bytes>>>((u32x4*)base0, (u32x4*)base1, G, nwin, wrap_frames);
        hipEventRecord(b, 0);
        hipEventSynchronize(b);
        float ms = 0;
        hipEventElapsedTime(&ms, a, b);
        if (r >= 0) total += ms;
    }
    hipEventDestroy(a);
    hipEventDestroy(b);
    return total / reps * 1e3;
}
const char* lab_error() { return g_err; }

long lab_granularity(int device, int recommended) {
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = device;
    size_t g = 0;
    if (hipMemGetAllocationGranularity(&g, &prop, recommended ? hipMemAllocationGranularityRecommended : hipMemAllocationGranularityMinimum) != hipSuccess) return -1;
    return (long)g;
}

// kind 0: hipMalloc; 1: VMM (physical chunks of `chunk` bytes, mapped into one reserved range; order: 0 = as created,
// 1 = shuffled with `seed`, 2 = every second of twice as many chunks (the others are released));
// 2: hipDeviceMallocContiguous; 3: hipDeviceMallocUncached; 4: hipDeviceMallocFinegrained
int lab_alloc(size_t bytes, int kind, size_t chunk, int order, int seed, int device, void** out) {
    *out = nullptr;
    if (kind == 0) {
        LAB(hipMalloc(out, bytes));
        return 0;
    }
    if (kind >= 2) {
        unsigned flag = kind == 2 ? hipDeviceMallocContiguous : kind == 3 ? hipDeviceMallocUncached : hipDeviceMallocFinegrained;
        LAB(hipExtMallocWithFlags(out, bytes, flag));
        return 0;
    }
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = device;
    size_t gran = 0;
    LAB(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityMinimum));
    if (chunk == 0) chunk = (bytes + gran - 1) / gran * gran;
    chunk = (chunk + gran - 1) / gran * gran;
    const size_t nchunk = (bytes + chunk - 1) / chunk;
    Vmm v;
    v.va_size = nchunk * chunk;
    LAB(hipMemAddressReserve(&v.va, v.va_size, 2u << 20, nullptr, 0));
    const size_t ncreate = order == 2 ? 2 * nchunk : nchunk;
    std::vector<hipMemGenericAllocationHandle_t> hs(ncreate);
    for (size_t i = 0; i < ncreate; ++i) LAB(hipMemCreate(&hs[i], chunk, &prop, 0));
    if (order == 1) {
        std::mt19937 rng(seed);
        std::shuffle(hs.begin(), hs.end(), rng);
    } else if (order == 2) {
        std::vector<hipMemGenericAllocationHandle_t> keep;
        for (size_t i = 0; i < ncreate; ++i) {
            if (i % 2 == 0) keep.push_back(hs[i]);
            else LAB(hipMemRelease(hs[i]));
        }
        hs.swap(keep);
    }
    for (size_t i = 0; i < nchunk; ++i) LAB(hipMemMap((char*)v.va + i * chunk, chunk, 0, hs[i], 0));
    hipMemAccessDesc acc = {};
    acc.location.type = hipMemLocationTypeDevice;
    acc.location.id = device;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    LAB(hipMemSetAccess(v.va, v.va_size, &acc, 1));
    v.handles = hs;
    g_vmm[v.va] = v;
    *out = v.va;
    return 0;
}

int lab_free(void* p) {
    auto it = g_vmm.find(p);
    if (it == g_vmm.end()) {
        LAB(hipFree(p));
        return 0;
    }
    Vmm& v = it->second;
    LAB(hipMemUnmap(v.va, v.va_size));
    for (auto h : v.handles) LAB(hipMemRelease(h));
    LAB(hipMemAddressFree(v.va, v.va_size));
    g_vmm.erase(it);
    return 0;
}

// primitives: reserve a virtual range, create one physical handle, map / unmap
int lab_reserve(size_t bytes, void** out) {
    LAB(hipMemAddressReserve(out, bytes, 2u << 20, nullptr, 0));
    return 0;
}
int lab_create(size_t bytes, int device, void** handle_out) {
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = device;
    hipMemGenericAllocationHandle_t h;
    LAB(hipMemCreate(&h, bytes, &prop, 0));
    *handle_out = (void*)h;
    return 0;
}
int lab_create_exportable(size_t bytes, int device, void** handle_out) {
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.requestedHandleType = hipMemHandleTypePosixFileDescriptor;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = device;
    hipMemGenericAllocationHandle_t h;
    LAB(hipMemCreate(&h, bytes, &prop, 0));
    *handle_out = (void*)h;
    return 0;
}
int lab_release(void* handle) {
    LAB(hipMemRelease((hipMemGenericAllocationHandle_t)handle));
    return 0;
}
int lab_map(void* va, size_t bytes, void* handle, int device) {
    LAB(hipMemMap(va, bytes, 0, (hipMemGenericAllocationHandle_t)handle, 0));
    hipMemAccessDesc acc = {};
    acc.location.type = hipMemLocationTypeDevice;
    acc.location.id = device;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    LAB(hipMemSetAccess(va, bytes, &acc, 1));
    return 0;
}
int lab_unmap(void* va, size_t bytes) {
    LAB(hipDeviceSynchronize());
    LAB(hipMemUnmap(va, bytes));
    return 0;
}

// best-of-`reps` time in microseconds of a linear 16-byte-per-thread fill of [p, p + bytes)
double lab_fill_us(void* p, size_t bytes, int reps) {
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    const size_t nvec = bytes / 16;
    const unsigned grid = (unsigned)((nvec + 255) / 256);
    float best = 1e30f;
    fill_linear<<<grid, 256>>>((u32x4*)p, nvec);
    for (int r = 0; r < reps; ++r) {
        hipEventRecord(a, 0);
        fill_linear<<<grid, 256>>>((u32x4*)p, nvec);
        hipEventRecord(b, 0);
        hipEventSynchronize(b);
        float ms = 0;
        hipEventElapsedTime(&ms, a, b);
        best = std::min(best, ms);
    }
    hipEventDestroy(a);
    hipEventDestroy(b);
    return best * 1e3;
}
}
