// ldsprobe.hip -- does a byte-misaligned ds_read_b64 return the eight bytes at its address on this device, and what does it cost?
// (round 6: the one-wave LK iteration reads its two 8-byte tap rows at arbitrary byte offsets of the search tile)
//   hipcc -O3 --offload-arch=gfx950 tools/ldsprobe.hip -o tools/ldsprobe && tools/ldsprobe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
typedef uint64_t __attribute__((aligned(1), may_alias)) u64u;

__global__ void probe(const uint8_t* in, uint64_t* out, int off, int stride)
{
    __shared__ __attribute__((aligned(16))) uint8_t s[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) s[i] = in[i];
    __syncthreads();
    out[threadIdx.x] = *(const u64u*)(s + off + threadIdx.x * stride);
}

// REPS dependent reads per lane: misaligned b64 against three aligned dwords + two v_alignbyte (what the kernel did before)
template <bool UNALIGNED>
__global__ void timing(const uint8_t* in, unsigned long long* out, int off, int reps)
{
    __shared__ __attribute__((aligned(16))) uint8_t s[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) s[i] = in[i];
    __syncthreads();
    unsigned a = off + (threadIdx.x / 3) * 44 + (threadIdx.x % 3) * 7;
    unsigned long long acc = 0;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int r = 0; r < reps; r++) {
        unsigned long long v;
        if (UNALIGNED) v = *(const u64u*)(s + a);
        else {
            const uint32_t* p = (const uint32_t*)(s + (a & ~3u));
            const uint32_t d0 = p[0], d1 = p[1], d2 = p[2];
            const uint32_t lo = __builtin_amdgcn_alignbyte(d1, d0, a & 3), hi = __builtin_amdgcn_alignbyte(d2, d1, a & 3);
            v = ((unsigned long long)hi << 32) | lo;
        }
        acc += v;
        a = (a + (unsigned)(v & 3) + 1) & 2047;          // dependent address
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[threadIdx.x] = acc;
    if (threadIdx.x == 0) out[64] = t1 - t0;
}

int main()
{
    uint8_t h[4096];
    for (int i = 0; i < 4096; i++) h[i] = (uint8_t)(i * 37 + (i >> 8) * 11 + 5);
    uint8_t* d; uint64_t* o; unsigned long long* t;
    hipMalloc(&d, 4096); hipMalloc(&o, 64 * 8); hipMalloc(&t, 65 * 8);
    hipMemcpy(d, h, 4096, hipMemcpyHostToDevice);
    int bad = 0;
    for (int stride = 1; stride <= 44; stride += (stride < 9 ? 1 : 7))
        for (int off = 0; off < 9; off++) {
            probe<<<1, 64>>>(d, o, off, stride);
            uint64_t r[64]; hipMemcpy(r, o, sizeof(r), hipMemcpyDeviceToHost);
            for (int l = 0; l < 64; l++) { uint64_t w; memcpy(&w, h + off + l * stride, 8); if (w != r[l]) bad++; }
        }
    printf("misaligned ds_read_b64: %s (%d mismatches)\n", bad ? "WRONG" : "correct", bad);
    for (int u = 0; u < 2; u++) {
        unsigned long long r[65];
        for (int rep = 0; rep < 3; rep++) {
            if (u) timing<true><<<1, 64>>>(d, t, 3, 4096); else timing<false><<<1, 64>>>(d, t, 3, 4096);
            hipMemcpy(r, t, sizeof(r), hipMemcpyDeviceToHost);
        }
        printf("%s: %.1f s_memtime ticks per dependent read (one wave)\n", u ? "misaligned b64" : "3 dwords + 2 alignbyte", (double)r[64] / 4096);
    }
    return bad != 0;
}
