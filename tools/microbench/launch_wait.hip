// tools/microbench/launch_wait.hip -- what one "launch a small kernel, wait for it on the host" costs with three ways of waiting
// (round 6, the single-instance path mg_single_step): hipStreamSynchronize; a stream memory operation behind the kernel
// (hipStreamWriteValue32) + host polling; a flag the kernel's last lane stores itself (system scope) + host polling.
//   hipcc --offload-arch=gfx950 -O3 -o launch_wait launch_wait.hip && ./launch_wait
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void work(uint32_t* out, uint32_t* flag, uint32_t ticket) {
    // ~a frame: 21 KB of stores into pinned host memory by 256 lanes
    uint4* o = reinterpret_cast<uint4*>(out);
    for (int j = threadIdx.x; j < 1323; j += 256) o[j] = make_uint4(ticket, j, 0, 0);
    if (flag) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");  // this wave's stores are performed system-wide
        __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_store(flag, ticket, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}
int main() {
    hipStream_t s;
    CK(hipStreamCreate(&s));
    char* host;
    CK(hipHostMalloc((void**)&host, 1 << 16, hipHostMallocMapped | hipHostMallocCoherent));
    char* dev;
    CK(hipHostGetDevicePointer((void**)&dev, host, 0));
    uint32_t* out = (uint32_t*)dev;
    volatile uint32_t* hflag = (volatile uint32_t*)(host + 32768);
    uint32_t* dflag = (uint32_t*)(dev + 32768);
    const int N = 20000;
    for (int mode = 0; mode < 3; ++mode) {
        for (int rep = 0; rep < 2; ++rep) {
            auto t0 = std::chrono::steady_clock::now();
            for (int k = 1; k <= N; ++k) {
                const uint32_t ticket = (uint32_t)(mode * 1000000 + rep * 100000 + k);
                hipLaunchKernelGGL(work, dim3(1), dim3(256), 0, s, out, mode == 2 ? dflag : nullptr, ticket);
                if (mode == 0) { CK(hipStreamSynchronize(s)); }
                else {
                    if (mode == 1) CK(hipStreamWriteValue32(s, dflag, ticket, 0));
                    while (__atomic_load_n(hflag, __ATOMIC_ACQUIRE) != ticket) __builtin_ia32_pause();
                }
            }
            const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / N;
            printf("%-58s %6.2f us per launch + wait\n", mode == 0 ? "hipStreamSynchronize" : (mode == 1 ? "hipStreamWriteValue32 behind the kernel + host poll" : "flag stored by the kernel (system scope) + host poll"), us);
        }
    }
    return 0;
}
