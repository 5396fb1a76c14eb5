#include <hip/hip_runtime.h>
#include <cstdio>
struct Big { unsigned long long v[400]; };   // 3200 B each, two of them = 6400 B
__global__ void k(const Big a, const Big b, unsigned long long* out) { out[0] = a.v[399] + b.v[399] + a.v[threadIdx.x & 3]; }
int main() {
    Big a, b; for (int i = 0; i < 400; i++) { a.v[i] = i; b.v[i] = 1000 + i; }
    unsigned long long* d; hipMalloc(&d, 8);
    hipLaunchKernelGGL(k, dim3(1), dim3(1), 0, 0, a, b, d);
    hipError_t e = hipGetLastError(); hipError_t e2 = hipDeviceSynchronize();
    unsigned long long h = 0; hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
    printf("launch %d sync %d value %llu (expect %d)\n", (int)e, (int)e2, h, 399 + 1399 + 0);
}
