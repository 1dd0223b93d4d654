// any_order.hip -- can two kernels enqueued on ONE stream run at the same time on this ROCm?
//
//   hipcc --offload-arch=gfx950 -O2 -o any_order any_order.hip && ./any_order
//
// A (one workgroup) spins for ~300 us and stamps its start / end with the shader's real-time clock (100 MHz), B (one workgroup)
// stamps its start.  B is enqueued right behind A on the same stream
//   1. with hipLaunchKernelGGL                              -> expected: B starts after A ends (barrier bit in the AQL packet),
//   2. with hipExtLaunchKernelGGL(..., hipExtAnyOrderLaunch) -> if the flag is honoured on gfx950, B starts while A spins,
//   3. on a second stream (non-blocking), no events           -> the concurrency two queues give, for reference.
// Printed: B.start - A.start and A.end - A.start in us for each case, and how long the host took per pair of launches.
#include <hip/hip_ext.h>
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>

#define CK(e)                                                                        \
    do {                                                                             \
        hipError_t r_ = (e);                                                         \
        if (r_ != hipSuccess) {                                                      \
            printf("%s failed: %s (line %d)\n", #e, hipGetErrorString(r_), __LINE__); \
            return 1;                                                                \
        }                                                                            \
    } while (0)

__global__ void spin_kernel(unsigned long long* t, unsigned long long ticks) {
    const unsigned long long t0 = wall_clock64();
    if (threadIdx.x == 0) t[0] = t0;
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
    if (threadIdx.x == 0) t[1] = wall_clock64();
}
__global__ void stamp_kernel(unsigned long long* t) {
    if (threadIdx.x == 0) t[2] = wall_clock64();
}

int main() {
    unsigned long long *d, h[3];
    CK(hipMalloc((void**)&d, 64));
    hipStream_t s, s2;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    int rate_khz = 0;
    CK(hipDeviceGetAttribute(&rate_khz, hipDeviceAttributeWallClockRate, 0));
    const double us_per_tick = 1e3 / (double)rate_khz;
    const unsigned long long ticks = (unsigned long long)(300.0 / us_per_tick);
    printf("wall clock %d kHz\n", rate_khz);
    for (int mode = 0; mode < 3; ++mode) {
        double dB = 0, dA = 0, host = 0;
        const int reps = 20;
        for (int r = 0; r < reps + 2; ++r) {
            CK(hipMemsetAsync(d, 0, 64, s));
            CK(hipStreamSynchronize(s));
            auto w0 = std::chrono::steady_clock::now();
            hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, s, d, ticks);
            if (mode == 0) hipLaunchKernelGGL(stamp_kernel, dim3(1), dim3(64), 0, s, d);
            else if (mode == 1) hipExtLaunchKernelGGL(stamp_kernel, dim3(1), dim3(64), 0, s, nullptr, nullptr, hipExtAnyOrderLaunch, d);
            else hipLaunchKernelGGL(stamp_kernel, dim3(1), dim3(64), 0, s2, d);
            auto w1 = std::chrono::steady_clock::now();
            CK(hipGetLastError());
            CK(hipStreamSynchronize(s));
            CK(hipStreamSynchronize(s2));
            CK(hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost));
            if (r < 2) continue;
            dB += ((double)h[2] - (double)h[0]) * us_per_tick;
            dA += ((double)h[1] - (double)h[0]) * us_per_tick;
            host += std::chrono::duration<double, std::micro>(w1 - w0).count();
        }
        static const char* names[3] = {"same stream, plain launch", "same stream, hipExtAnyOrderLaunch", "second stream, no events"};
        printf("%-36s B.start - A.start = %8.1f us   A.end - A.start = %8.1f us   host %.1f us per pair\n", names[mode], dB / reps, dA / reps, host / reps);
    }
    return 0;
}
