// Test fixture (not product code): a kernel that holds a chosen number of workgroups -- i.e. CUs -- busy for a chosen WALL
// time, the way a persistent collective kernel (an RCCL ring) looks to the hardware scheduler.  Every wave leaves when the
// 100 MHz wall clock passes its deadline, so the grid always drains.  tests/test_hip_callers.py runs the GroupNorm cluster
// kernels next to it.
#include <hip/hip_runtime.h>

__global__ void __launch_bounds__(256) spin_for(unsigned long long ticks, unsigned* sink) {
    const unsigned long long t0 = wall_clock64();
    unsigned acc = 0;
    while (wall_clock64() - t0 < ticks) {
        acc += 1;
        __builtin_amdgcn_s_sleep(8);
    }
    if (acc == 0xFFFFFFFFu) *sink = acc;                       // keeps the loop alive for the compiler
}

extern "C" int spin_launch(int workgroups, int lds_bytes, double milliseconds, void* sink, void* stream) {
    if (workgroups <= 0 || milliseconds <= 0 || milliseconds > 2000.0 || lds_bytes < 0 || lds_bytes > 160 * 1024) return -1;
    if (lds_bytes > 64 * 1024 &&
        hipFuncSetAttribute((const void*)spin_for, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes) != hipSuccess)
        return -2;
    const unsigned long long ticks = (unsigned long long)(milliseconds * 1e5);         // wall_clock64: 100 MHz
    hipLaunchKernelGGL(spin_for, dim3((unsigned)workgroups), dim3(256), (size_t)lds_bytes, (hipStream_t)stream, ticks,
                       (unsigned*)sink);
    return (int)hipGetLastError();
}
