// clockprobe: effective shader clock seen by a single short wave, in three load regimes
// (development aid).  clock = d(s_memtime) / d(s_memrealtime) * 100 MHz.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <unistd.h>
__global__ void probe(unsigned long long* out, int n)
{
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    float a = threadIdx.x;
    for (int i = 0; i < n; i++) a = a * 1.0001f + 0.5f;
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = r1 - r0; out[2] = (unsigned long long)a; }
}
__global__ void burn(float* out, int n)
{
    float a = threadIdx.x;
    for (int i = 0; i < n; i++) a = a * 1.0001f + 0.5f;
    out[blockIdx.x * blockDim.x + threadIdx.x] = a;
}
int main()
{
    unsigned long long* d; hipMalloc(&d, 64); float* fb; hipMalloc(&fb, 4 * 1024 * 1024 * 4);
    unsigned long long h[3];
    auto one = [&](const char* tag) {
        hipLaunchKernelGGL(probe, 1, 64, 0, 0, d, 20000);
        hipMemcpy(h, d, 24, hipMemcpyDeviceToHost);
        printf("%-28s cycles %llu  real(10ns ticks) %llu  -> %.0f MHz, %.1f us, %.2f cyc/iter\n", tag, h[0], h[1],
               (double)h[0] / h[1] * 100.0, h[1] * 0.01, (double)h[0] / 20000);
    };
    one("cold start");
    usleep(200000); one("after 200 ms idle");
    for (int i = 0; i < 5; i++) one("back-to-back single wave");
    for (int i = 0; i < 2000; i++) hipLaunchKernelGGL(probe, 1, 64, 0, 0, d, 2000);
    one("after 2000 tiny launches");
    hipLaunchKernelGGL(burn, 4096, 256, 0, 0, fb, 200000);   // ~ tens of ms of full-chip load
    one("right after full-chip burn");
    usleep(5000); one("5 ms after burn");
    usleep(100000); one("100 ms after burn");
    return 0;
}
