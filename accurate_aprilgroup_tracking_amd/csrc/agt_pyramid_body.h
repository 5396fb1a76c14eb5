// agt_pyramid_body.h -- device body of cv::pyrDown (u8, 5x5 [1 4 6 4 1]^2 / 256, BORDER_REFLECT_101).
// Included by agt_pyramid.hip (stand-alone kernel) and agt_step.hip (fused per-frame launch).
// Semantics: OpenCV modules/imgproc/src/pyramids.cpp pyrDown_, restated in oracle/cv_lk.c cvo_pyr_down_u8.
//
// HBM-bound integer work, written to keep the VALU out of the way (the first version spent 657
// VALU instructions per wave and was VALU-bound at 1.9 TB/s; this one spends ~180):
//   * source tile (2*TH+3 rows x 288 B) fetched with 16-byte loads, every load of a thread in
//     flight before the first is consumed, written to LDS with ds_write_b128;
//   * horizontal [1 4 6 4 1]: one v_dot4_u32_u8 per output (weights packed 0x04060401 over the
//     four leading taps, fifth tap as the accumulator input), even outputs via v_alignbyte_b32;
//     eight outputs per thread, stored as eight u16 (one ds_write_b128);
//   * vertical [1 4 6 4 1]: packed 16-bit math (v_pk_add_u16 / v_pk_mad_u16 / v_pk_lshrrev_b16,
//     the largest intermediate is 65,408 < 2^16), eight outputs per thread, one 8-byte store.
#pragma once
#include "agt_device.h"
#include "agt_kernels.h"

namespace agt_pyr {

constexpr int TW = 128;             // output tile width  (pixels)
constexpr int TH = 16;              // output tile height
constexpr int NCH = 18;             // 16-byte chunks staged per source row: x in [2*ox0-16, 2*ox0+272)
constexpr int SW = NCH * 16;        // staged source bytes per row (288)
constexpr int SH = 2 * TH + 3;      // staged source rows: y in [2*oy0-2, 2*oy0+2*TH+1)
constexpr int NT = 256;
constexpr int HP = TW * 2;          // bytes per row of the u16 horizontal plane (256)

constexpr int PYR_LDS_BYTES = SH * SW + SH * HP;     // 10,080 + 8,960

typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ uint32_t pk_add(uint32_t a, uint32_t b)
{
    return __builtin_bit_cast(uint32_t, __builtin_bit_cast(u16x2, a) + __builtin_bit_cast(u16x2, b));
}
__device__ __forceinline__ uint32_t pk_mad(uint32_t a, uint32_t b, uint32_t c)
{
    return __builtin_bit_cast(uint32_t, __builtin_bit_cast(u16x2, a) * __builtin_bit_cast(u16x2, b) + __builtin_bit_cast(u16x2, c));
}
__device__ __forceinline__ uint32_t pk_shr8(uint32_t a)
{
    return __builtin_bit_cast(uint32_t, __builtin_bit_cast(u16x2, a) >> (unsigned short)8);
}

// image-edge path of the tile fetch: 16 bytes assembled with reflect-101 (rare, out of line)
__device__ __noinline__ uint4 load_chunk_reflect(const uint8_t* __restrict__ row, int gx, int w)
{
    uint32_t v[4];
    for (int d = 0; d < 4; d++) {
        uint32_t t = 0;
        for (int k = 0; k < 4; k++) t |= (uint32_t)row[agt_reflect101(gx + 4 * d + k, w)] << (8 * k);
        v[d] = t;
    }
    return make_uint4(v[0], v[1], v[2], v[3]);
}

// eight horizontal [1 4 6 4 1] sums (u16) from 16 source bytes d (output o centred at byte 2 o), the 4 bytes before (pm)
// and the 4 bytes after (nx): one v_dot4_u32_u8 per output
__device__ __forceinline__ uint4 hgroup8(uint4 d, uint32_t pm, uint32_t nx)
{
    const uint32_t W4 = 0x04060401u;                                      // taps c-2, c-1, c, c+1
    // even outputs start two bytes before an aligned dword, odd outputs on one
    const uint32_t e0 = __builtin_amdgcn_alignbyte(d.x, pm, 2), e2 = __builtin_amdgcn_alignbyte(d.y, d.x, 2);
    const uint32_t e4 = __builtin_amdgcn_alignbyte(d.z, d.y, 2), e6 = __builtin_amdgcn_alignbyte(d.w, d.z, 2);
    const uint32_t h0 = __builtin_amdgcn_udot4(e0, W4, (d.x >> 16) & 0xff, false);     // fifth tap: byte 2
    const uint32_t h1 = __builtin_amdgcn_udot4(d.x, W4, d.y & 0xff, false);            // byte 4
    const uint32_t h2 = __builtin_amdgcn_udot4(e2, W4, (d.y >> 16) & 0xff, false);     // byte 6
    const uint32_t h3 = __builtin_amdgcn_udot4(d.y, W4, d.z & 0xff, false);            // byte 8
    const uint32_t h4 = __builtin_amdgcn_udot4(e4, W4, (d.z >> 16) & 0xff, false);     // byte 10
    const uint32_t h5 = __builtin_amdgcn_udot4(d.z, W4, d.w & 0xff, false);            // byte 12
    const uint32_t h6 = __builtin_amdgcn_udot4(e6, W4, (d.w >> 16) & 0xff, false);     // byte 14
    const uint32_t h7 = __builtin_amdgcn_udot4(d.w, W4, nx & 0xff, false);             // byte 16
    return make_uint4(h0 | (h1 << 16), h2 | (h3 << 16), h4 | (h5 << 16), h6 | (h7 << 16));
}

// eight vertical [1 4 6 4 1] sums + (v + 128) >> 8 from five rows of eight u16 each -> eight bytes (packed 16-bit math)
__device__ __forceinline__ uint2 vgroup8(uint4 r0, uint4 r1, uint4 r2, uint4 r3, uint4 r4)
{
    const uint32_t K4 = 0x00040004u, K6 = 0x00060006u, K128 = 0x00800080u;
    auto col = [&](uint32_t a0, uint32_t a1, uint32_t a2, uint32_t a3, uint32_t a4) -> uint32_t {
        uint32_t v = pk_mad(pk_add(a1, a3), K4, pk_add(a0, a4));
        v = pk_mad(a2, K6, v);
        return pk_shr8(pk_add(v, K128));                                 // two results, one per 16-bit half
    };
    const uint32_t p0 = col(r0.x, r1.x, r2.x, r3.x, r4.x), p1 = col(r0.y, r1.y, r2.y, r3.y, r4.y);
    const uint32_t p2 = col(r0.z, r1.z, r2.z, r3.z, r4.z), p3 = col(r0.w, r1.w, r2.w, r3.w, r4.w);
    // gather the low byte of each half: (p.lo, p.hi, q.lo, q.hi) -> one dword
    return make_uint2(__builtin_amdgcn_perm(p1, p0, 0x06040200u), __builtin_amdgcn_perm(p3, p2, 0x06040200u));
}

// ---- round 5: the same two passes with fewer instructions (the register-rolling kernels are VALU-bound once their re-reads are gone).
// hgroup8b: every output as a CHAIN of two dot4 on aligned dwords -- no v_alignbyte, no tap extraction:
//   even output of dword j (centre byte 4 j):     (x[-2], x[-1]) . (1, 4) of the dword before  +  (x0, x1, x2) . (6, 4, 1)
//   odd  output of dword j (centre byte 4 j + 2): (x0 .. x3) . (1, 4, 6, 4)                    +  x0 of the dword after
// 16 dot4 + 4 packs per 16 source bytes (before: 8 dot4 + 4 alignbyte + 8 extractions + 4 packs).  BIAS: 16 is added to each of the
// eight sums (free: it is the accumulator input of the chain's first dot4).  The vertical pass takes the rows at ODD offsets of its
// window biased: 4 * (16 + 16) = the +128 of the rounding, so vgroup8b needs no rounding add; and it takes the HIGH byte of every
// 16-bit sum with the byte gather itself (no shift): 4 instructions per register pair of the window + 1 gather per two (before 7 + 1).
// The largest intermediate is unchanged: 16 * 4080 + 128 = 65,408 < 2^16.  Same bits as hgroup8 / vgroup8 (tests/test_gpu_parity.py).
template <bool BIAS>
__device__ __forceinline__ uint4 hgroup8b(uint4 d, uint32_t pm, uint32_t nx)
{
    const uint32_t WB = 0x04010000u, WA = 0x00010406u, WC = 0x04060401u, WD = 0x00000001u;
    const uint32_t c0 = BIAS ? 16u : 0u;
    const uint32_t e0 = __builtin_amdgcn_udot4(d.x, WA, __builtin_amdgcn_udot4(pm, WB, c0, false), false);
    const uint32_t o0 = __builtin_amdgcn_udot4(d.x, WC, __builtin_amdgcn_udot4(d.y, WD, c0, false), false);
    const uint32_t e1 = __builtin_amdgcn_udot4(d.y, WA, __builtin_amdgcn_udot4(d.x, WB, c0, false), false);
    const uint32_t o1 = __builtin_amdgcn_udot4(d.y, WC, __builtin_amdgcn_udot4(d.z, WD, c0, false), false);
    const uint32_t e2 = __builtin_amdgcn_udot4(d.z, WA, __builtin_amdgcn_udot4(d.y, WB, c0, false), false);
    const uint32_t o2 = __builtin_amdgcn_udot4(d.z, WC, __builtin_amdgcn_udot4(d.w, WD, c0, false), false);
    const uint32_t e3 = __builtin_amdgcn_udot4(d.w, WA, __builtin_amdgcn_udot4(d.z, WB, c0, false), false);
    const uint32_t o3 = __builtin_amdgcn_udot4(d.w, WC, __builtin_amdgcn_udot4(nx, WD, c0, false), false);
    return make_uint4(e0 | (o0 << 16), e1 | (o1 << 16), e2 | (o2 << 16), e3 | (o3 << 16));
}

// t * 4 + c on both 16-bit halves, as ONE instruction (the compiler turns the multiplication by 4 into a shift and an add)
__device__ __forceinline__ uint32_t pk_mad4(uint32_t t, uint32_t c)
{
    uint32_t r;
    asm("v_pk_mad_u16 %0, %1, 4, %2 op_sel_hi:[1,0,1]" : "=v"(r) : "v"(t), "v"(c));
    return r;
}

// one register (two columns) of the vertical pass on a window whose rows 1 and 3 carry the +16 bias: the two sums, UNSHIFTED
__device__ __forceinline__ uint32_t vcol_b(uint32_t a0, uint32_t a1, uint32_t a2, uint32_t a3, uint32_t a4)
{
    const uint32_t K6 = 0x00060006u;
    uint32_t v = pk_mad4(pk_add(a1, a3), pk_add(a0, a4));
    return pk_mad(a2, K6, v);
}

__device__ __forceinline__ uint2 vgroup8b(uint4 r0, uint4 r1, uint4 r2, uint4 r3, uint4 r4)
{
    const uint32_t p0 = vcol_b(r0.x, r1.x, r2.x, r3.x, r4.x), p1 = vcol_b(r0.y, r1.y, r2.y, r3.y, r4.y);
    const uint32_t p2 = vcol_b(r0.z, r1.z, r2.z, r3.z, r4.z), p3 = vcol_b(r0.w, r1.w, r2.w, r3.w, r4.w);
    // the HIGH byte of each half = (sum + 128) >> 8: (p.b1, p.b3, q.b1, q.b3) -> one dword
    return make_uint2(__builtin_amdgcn_perm(p1, p0, 0x07050301u), __builtin_amdgcn_perm(p3, p2, 0x07050301u));
}

// one 128x16 output tile (bx, by) of the image at `img` -> `out`.  lds: PYR_LDS_BYTES, 16-B aligned
__device__ __forceinline__ void pyr_down_body(const AgtPyrArgs& A, int bx, int by, const uint8_t* __restrict__ img,
                                              uint8_t* __restrict__ out, uint8_t* lds)
{
    uint8_t* s_src = lds;
    uint8_t* s_h = lds + SH * SW;
    const int sw = A.sw, sh = A.sh, dw = A.dw, dh = A.dh;
    const int tid = threadIdx.x;
    const int ox0 = bx * TW, oy0 = by * TH;
    const int sx0 = 2 * ox0 - 16, sy0 = 2 * oy0 - 2;
    const bool aligned16 = ((reinterpret_cast<uintptr_t>(img) | (uintptr_t)A.spitch) & 15) == 0;

    // ---- fetch: thread -> (row r0 + 14 k, chunk c), 14 rows x 18 chunks per round, 3 rounds
    // A tile whose staged ROWS lie inside an aligned image of a width that is a multiple of 16 (block-uniform test: all but
    // the top and bottom tile rows) needs no reflection, and a chunk is either wholly inside or wholly outside: one 24-bit
    // multiply-add per thread, a scalar row step per round, the image base in SGPRs.  (The general path below costs ~30 VALU
    // per load -- reflect-101 of the row, a 64-bit multiply for the row address, the chunk classification -- which was a
    // third of the kernel's vector instructions.)
    if (aligned16 && (sw & 15) == 0 && sy0 >= 0 && sy0 + SH <= sh && A.spitch < (1L << 23)) {
        typedef const __attribute__((address_space(1))) uint8_t* G8;
        typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
        typedef const __attribute__((address_space(1))) u32x4_t* G128;
        const int r0 = tid / NCH, c = tid - r0 * NCH;
        const int gx = sx0 + 16 * c;
        const bool lane_on = tid < 14 * NCH && gx >= 0 && gx + 16 <= sw;       // (outside chunks: patched from their reflections below)
        const G8 base = (G8)img + ((long)sy0 * A.spitch + sx0);
        const int p32 = (int)A.spitch;
        const int off = __mul24(r0, p32) + 16 * c;                 // (r0 < 15; a pitch beyond 2^23 takes the general path)
        uint4 v[3];
#pragma unroll
        for (int k = 0; k < 3; k++) {
            v[k] = make_uint4(0, 0, 0, 0);
            if (lane_on && r0 + 14 * k < SH) { const u32x4_t t = *(G128)(base + (off + 14 * k * p32)); v[k] = make_uint4(t.x, t.y, t.z, t.w); }
        }
#pragma unroll
        for (int k = 0; k < 3; k++)
            if (tid < 14 * NCH && r0 + 14 * k < SH) *reinterpret_cast<uint4*>(s_src + (r0 + 14 * k) * SW + 16 * c) = v[k];
    } else {
        const int r0 = tid / NCH, c = tid - r0 * NCH;
        const int gx = sx0 + 16 * c;
        const bool lane_on = tid < 14 * NCH;
        const bool inside = aligned16 && gx >= 0 && gx + 15 < sw;
        // chunks wholly outside the image are not fetched: the <= 2 halo bytes per side that the
        // filter reads there are patched from their reflections below.  Only a chunk that straddles
        // the right edge of an image whose width is not a multiple of 16 (or an unaligned image)
        // takes the byte-wise path.
        const bool outside = gx + 15 < 0 || gx >= sw;
        uint4 v[3];
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const int r = r0 + 14 * k;
            v[k] = make_uint4(0, 0, 0, 0);
            if (lane_on && r < SH && !outside) {
                const uint8_t* row = img + (long)agt_reflect101(sy0 + r, sh) * A.spitch;
                v[k] = inside ? *reinterpret_cast<const uint4*>(row + gx) : load_chunk_reflect(row, gx, sw);
            }
        }
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const int r = r0 + 14 * k;
            if (lane_on && r < SH) *reinterpret_cast<uint4*>(s_src + r * SW + 16 * c) = v[k];
        }
    }
    __syncthreads();
    // reflect-101 halo columns of edge tiles: x = -2, -1 and x = sw, sw + 1 (block-uniform branch)
    if (sx0 + 14 < 0 || sx0 + SW > sw) {
        if (tid < SH) {
            uint8_t* row = s_src + tid * SW;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int x = k < 2 ? k - 2 : sw + k - 2;            // -2, -1, sw, sw + 1
                const int j = x - sx0, jr = agt_reflect101(x, sw) - sx0;
                if (j >= 0 && j < SW && jr >= 0 && jr < SW) row[j] = row[jr];
            }
        }
        __syncthreads();
    }

    // ---- horizontal: thread -> (row r, group q of 8 outputs); output o of the group has its centre
    // at staged byte 16 (q + 1) + 2 o
    for (int i = tid; i < SH * (TW / 8); i += NT) {
        const int r = i / (TW / 8), q = i - r * (TW / 8);
        const uint8_t* s = s_src + r * SW + 16 * (q + 1);
        const uint4 d = *reinterpret_cast<const uint4*>(s);                   // bytes 0..15
        // bytes -4..-1 and 16..19 are the neighbouring groups' last / first dword: the 16 groups of a row sit in one
        // DPP row, so they come from the neighbour lanes; only the two edge groups read LDS.  (As LDS reads of every lane
        // the two stride-16-byte dwords were 8-way bank conflicts: half of the kernel's LDS cycles.)
        uint32_t pm = 0, nx = 0;
        if (q == 0) pm = *reinterpret_cast<const uint32_t*>(s - 4);
        if (q == TW / 8 - 1) nx = *reinterpret_cast<const uint32_t*>(s + 16);
        pm = (uint32_t)__builtin_amdgcn_update_dpp((int)pm, (int)d.w, 0x111, 0xf, 0xf, false);    // row_shr:1
        nx = (uint32_t)__builtin_amdgcn_update_dpp((int)nx, (int)d.x, 0x101, 0xf, 0xf, false);    // row_shl:1
        const uint32_t W4 = 0x04060401u;                                      // taps c-2, c-1, c, c+1
        // even outputs start two bytes before an aligned dword, odd outputs on one
        const uint32_t e0 = __builtin_amdgcn_alignbyte(d.x, pm, 2), e2 = __builtin_amdgcn_alignbyte(d.y, d.x, 2);
        const uint32_t e4 = __builtin_amdgcn_alignbyte(d.z, d.y, 2), e6 = __builtin_amdgcn_alignbyte(d.w, d.z, 2);
        const uint32_t h0 = __builtin_amdgcn_udot4(e0, W4, (d.x >> 16) & 0xff, false);     // fifth tap: byte 2
        const uint32_t h1 = __builtin_amdgcn_udot4(d.x, W4, d.y & 0xff, false);            // byte 4
        const uint32_t h2 = __builtin_amdgcn_udot4(e2, W4, (d.y >> 16) & 0xff, false);     // byte 6
        const uint32_t h3 = __builtin_amdgcn_udot4(d.y, W4, d.z & 0xff, false);            // byte 8
        const uint32_t h4 = __builtin_amdgcn_udot4(e4, W4, (d.z >> 16) & 0xff, false);     // byte 10
        const uint32_t h5 = __builtin_amdgcn_udot4(d.z, W4, d.w & 0xff, false);            // byte 12
        const uint32_t h6 = __builtin_amdgcn_udot4(e6, W4, (d.w >> 16) & 0xff, false);     // byte 14
        const uint32_t h7 = __builtin_amdgcn_udot4(d.w, W4, nx & 0xff, false);             // byte 16
        *reinterpret_cast<uint4*>(s_h + r * HP + 16 * q) =
            make_uint4(h0 | (h1 << 16), h2 | (h3 << 16), h4 | (h5 << 16), h6 | (h7 << 16));
    }
    __syncthreads();

    // ---- vertical + (v + 128) >> 8: thread -> (output row oy, group q of 8 outputs)
    {
        const int oy = tid / (TW / 8), q = tid - oy * (TW / 8);
        const int gy = oy0 + oy, gx = ox0 + 8 * q;
        if (gy < dh && gx < dw) {
            const uint8_t* h = s_h + (2 * oy) * HP + 16 * q;
            const uint4 r0 = *reinterpret_cast<const uint4*>(h), r1 = *reinterpret_cast<const uint4*>(h + HP);
            const uint4 r2 = *reinterpret_cast<const uint4*>(h + 2 * HP), r3 = *reinterpret_cast<const uint4*>(h + 3 * HP);
            const uint4 r4 = *reinterpret_cast<const uint4*>(h + 4 * HP);
            const uint32_t K4 = 0x00040004u, K6 = 0x00060006u, K128 = 0x00800080u;
            auto col = [&](uint32_t a0, uint32_t a1, uint32_t a2, uint32_t a3, uint32_t a4) -> uint32_t {
                uint32_t v = pk_mad(pk_add(a1, a3), K4, pk_add(a0, a4));
                v = pk_mad(a2, K6, v);
                return pk_shr8(pk_add(v, K128));                                 // two results, one per 16-bit half
            };
            const uint32_t p0 = col(r0.x, r1.x, r2.x, r3.x, r4.x), p1 = col(r0.y, r1.y, r2.y, r3.y, r4.y);
            const uint32_t p2 = col(r0.z, r1.z, r2.z, r3.z, r4.z), p3 = col(r0.w, r1.w, r2.w, r3.w, r4.w);
            // gather the low byte of each half: (p.lo, p.hi, q.lo, q.hi) -> one dword
            const uint32_t lo = __builtin_amdgcn_perm(p1, p0, 0x06040200u), hi = __builtin_amdgcn_perm(p3, p2, 0x06040200u);
            uint8_t* o = out + (long)gy * A.dpitch + gx;
            if (gx + 7 < dw && (((uintptr_t)o) & 7) == 0) {
                *reinterpret_cast<uint2*>(o) = make_uint2(lo, hi);
            } else {
                for (int k = 0; k < 8 && gx + k < dw; k++) o[k] = (uint8_t)((k < 4 ? lo : hi) >> (8 * (k & 3)));
            }
        }
    }
}

}  // namespace agt_pyr
