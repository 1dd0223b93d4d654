// side_stream.hip -- what does it cost to run a small kernel BESIDE the main stream's big kernel, once per step, with events?
//
//   hipcc --offload-arch=gfx950 -O2 -o side_stream side_stream.hip && ./side_stream
//
// One "step" on the main stream s: A (spins a us) -> B (spins b us).  Variants:
//   0  s: A, B                                                     (what the step costs without any side work)
//   1  s: A, record e1, B, wait e2 ;  t: wait e1, C (c us), record e2   (C beside B: fork after A, join before the next A)
//   2  like 1 without the join (t free-running behind e1)           (cost of the fork alone)
//   3  s: A, C, B                                                   (C serial)
// All kernels are one workgroup that spins on the real-time clock, so the numbers are pure scheduling cost.
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>

#define CK(e)                                                                        \
    do {                                                                             \
        hipError_t r_ = (e);                                                         \
        if (r_ != hipSuccess) {                                                      \
            printf("%s failed: %s (line %d)\n", #e, hipGetErrorString(r_), __LINE__); \
            return 1;                                                                \
        }                                                                            \
    } while (0)

__global__ void spin_kernel(unsigned long long ticks) {
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(2);
}

int main(int argc, char** argv) {
    const double a_us = argc > 1 ? atof(argv[1]) : 10, b_us = argc > 2 ? atof(argv[2]) : 60, c_us = argc > 3 ? atof(argv[3]) : 40;
    hipStream_t s, t;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&t, hipStreamNonBlocking));
    hipEvent_t e1, e2, t0, t1;
    CK(hipEventCreateWithFlags(&e1, hipEventDisableTiming));
    CK(hipEventCreateWithFlags(&e2, hipEventDisableTiming));
    CK(hipEventCreate(&t0));
    CK(hipEventCreate(&t1));
    const unsigned long long ta = (unsigned long long)(a_us * 100), tb = (unsigned long long)(b_us * 100), tc = (unsigned long long)(c_us * 100);
    printf("A %.0f us, B %.0f us, C %.0f us\n", a_us, b_us, c_us);
    for (int mode = 0; mode < 4; ++mode) {
        const int steps = 300;
        float ms = 0;
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(t0, s));
            for (int k = 0; k < steps; ++k) {
                hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, s, ta);
                if (mode == 1 || mode == 2) {
                    CK(hipEventRecord(e1, s));
                    CK(hipStreamWaitEvent(t, e1, 0));
                    hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, t, tc);
                    if (mode == 1) CK(hipEventRecord(e2, t));
                }
                if (mode == 3) hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, s, tc);
                hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, s, tb);
                if (mode == 1) CK(hipStreamWaitEvent(s, e2, 0));
            }
            CK(hipEventRecord(t1, s));
            CK(hipEventSynchronize(t1));
            CK(hipDeviceSynchronize());
            CK(hipEventElapsedTime(&ms, t0, t1));
        }
        static const char* names[4] = {"s: A B", "s: A [e1] B [wait e2] | t: [wait e1] C [e2]", "s: A [e1] B | t: [wait e1] C", "s: A C B"};
        printf("%-48s %7.1f us per step\n", names[mode], ms * 1e3 / steps);
    }
    return 0;
}
