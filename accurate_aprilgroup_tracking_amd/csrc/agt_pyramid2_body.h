// agt_pyramid2_body.h -- TWO pyrDown levels in one pass: L0 -> L1 -> L2 (cv::pyrDown twice, bit-identical to applying
// agt_pyramid_body.h twice; oracle: oracle/cv_lk.c cvo_pyr_down_u8).  SURVEY.md section 7 step 4 / section 8d: the
// streamed-frame byte count W*H*1.3125 assumes exactly this -- L0 read once, every coarser level written once, L1 never
// re-read from HBM (two launches moved 1.5625 W*H and paid a second launch boundary / pipeline stage).
//
// One 256-thread workgroup = one 64 x 16 tile of L2 = the 128 x 32 tile of L1 above it.  The L2 tile needs the L1 tile plus
// a halo (2 columns left / 1 right, 2 rows above / 1 below: 131 x 35), which in turn needs 265 x 73 pixels of L0; with tile
// origins on multiples of 256 L0 columns that is the same 18 chunks of 16 bytes per row the single-level kernel stages.
//   1  fetch 73 rows x 288 B of L0 with 16-byte loads (all of a thread's loads in flight), reflect-101 rows; LDS
//   2  horizontal [1 4 6 4 1] of the 73 rows -> u16 plane (131 columns: 16 groups of 8 by v_dot4_u32_u8 + 3 halo columns)
//   3  vertical pass -> 35 x 131 L1 pixels: the 32 x 128 interior goes to HBM (8-byte stores) AND the whole extended tile to
//      LDS; L1 positions outside the image are then rewritten by reflect-101 OF L1 (pyrDown(L1) reflects L1, which differs
//      from pushing reflected L0 through the filter at the right / bottom edges)
//   4  horizontal + vertical pass on the L1 tile -> 16 x 64 L2 pixels, 8-byte stores
// Redundant work: the halo (73 instead of 64 + 3 source rows: 14 % more row reads, served by the XCD's L2 for
// neighbouring tiles).  LDS: 40.9 KB per workgroup.
#pragma once
#include "agt_pyramid_body.h"

namespace agt_pyr2 {

using namespace agt_pyr;

#ifndef AGT_PYR2_TH2
#define AGT_PYR2_TH2 16
#endif
constexpr int TW2 = 64, TH2 = AGT_PYR2_TH2;  // L2 tile
constexpr int R1 = 2 * TH2 + 3;              // 35 extended L1 rows
constexpr int SHF = 2 * R1 + 3;              // 73 staged L0 rows
constexpr int HP1 = 272;                     // plane 1 pitch: 128 u16 (groups) + 3 u16 (halo columns) + pad
constexpr int L1P = 160;                     // L1 tile pitch: column 2X at byte 16, columns 2X-2 .. 2X+128 at bytes 14 .. 144
constexpr int HP2 = 2 * TW2;                 // plane 2 pitch: 64 u16
constexpr int SRC_BYTES = SHF * SW;          // 21,024 (re-used for the L1 tile and plane 2 once the first horizontal pass is done)
constexpr int PYR2_LDS_BYTES = SRC_BYTES + SHF * HP1;     // 40,880
static_assert(R1 * L1P + R1 * HP2 <= SRC_BYTES, "L1 tile + plane 2 alias the source tile");

// A0: level 0 -> 1 geometry, A1: level 1 -> 2 geometry (A1.sw == A0.dw ...).  (bx, by): tile of L2.
__device__ __forceinline__ void pyr_down2_body(const AgtPyrArgs& A0, const AgtPyrArgs& A1, int bx, int by,
                                               const uint8_t* __restrict__ img, uint8_t* __restrict__ out1, uint8_t* __restrict__ out2,
                                               uint8_t* lds)
{
    uint8_t* s_src = lds;
    uint8_t* s_h1 = lds + SRC_BYTES;
    uint8_t* s_l1 = lds;                       // aliases s_src (dead after pass 2)
    uint8_t* s_h2 = lds + R1 * L1P;
    const int sw = A0.sw, sh = A0.sh, w1 = A0.dw, h1 = A0.dh, w2 = A1.dw, h2 = A1.dh;
    const int tid = threadIdx.x;
    const int X2 = bx * TW2, Y2 = by * TH2;    // L2 origin
    const int X1 = 2 * X2, Y1 = 2 * Y2;        // L1 interior origin
    const int sx0 = 4 * X2 - 16, sy0 = 4 * Y2 - 6;
    const bool aligned16 = ((reinterpret_cast<uintptr_t>(img) | (uintptr_t)A0.spitch) & 15) == 0;

    // ---- 1. fetch: thread -> (row r0 + 14 k, chunk c), 14 rows x 18 chunks per round, 6 rounds
    // (row-interior tiles of an aligned image: the fast path of agt_pyramid_body.h -- no reflection, no 64-bit row multiply)
    if (aligned16 && (sw & 15) == 0 && sy0 >= 0 && sy0 + SHF <= sh && A0.spitch < (1L << 23)) {
        typedef const __attribute__((address_space(1))) uint8_t* G8;
        typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
        typedef const __attribute__((address_space(1))) u32x4_t* G128;
        const int r0 = tid / NCH, c = tid - r0 * NCH;
        const int gx = sx0 + 16 * c;
        const bool lane_on = tid < 14 * NCH && gx >= 0 && gx + 16 <= sw;
        const G8 base = (G8)img + ((long)sy0 * A0.spitch + sx0);
        const int p32 = (int)A0.spitch;
        const int off = __mul24(r0, p32) + 16 * c;
        constexpr int NR = (SHF + 13) / 14;
        uint4 v[NR];
#pragma unroll
        for (int k = 0; k < NR; k++) {
            v[k] = make_uint4(0, 0, 0, 0);
            if (lane_on && r0 + 14 * k < SHF) { const u32x4_t t = *(G128)(base + (off + 14 * k * p32)); v[k] = make_uint4(t.x, t.y, t.z, t.w); }
        }
#pragma unroll
        for (int k = 0; k < NR; k++)
            if (tid < 14 * NCH && r0 + 14 * k < SHF) *reinterpret_cast<uint4*>(s_src + (r0 + 14 * k) * SW + 16 * c) = v[k];
    } else {
        const int r0 = tid / NCH, c = tid - r0 * NCH;
        const int gx = sx0 + 16 * c;
        const bool lane_on = tid < 14 * NCH;
        const bool inside = aligned16 && gx >= 0 && gx + 15 < sw;
        const bool outside = gx + 15 < 0 || gx >= sw;      // not fetched: the <= 2 halo bytes the filter reads there are patched below
        constexpr int NR = (SHF + 13) / 14;
        uint4 v[NR];
#pragma unroll
        for (int k = 0; k < NR; k++) {
            const int r = r0 + 14 * k;
            v[k] = make_uint4(0, 0, 0, 0);
            if (lane_on && r < SHF && !outside) {
                const uint8_t* row = img + (long)agt_reflect101(sy0 + r, sh) * A0.spitch;
                v[k] = inside ? *reinterpret_cast<const uint4*>(row + gx) : load_chunk_reflect(row, gx, sw);
            }
        }
#pragma unroll
        for (int k = 0; k < NR; k++) {
            const int r = r0 + 14 * k;
            if (lane_on && r < SHF) *reinterpret_cast<uint4*>(s_src + r * SW + 16 * c) = v[k];
        }
    }
    __syncthreads();
    // reflect-101 halo columns of edge tiles: x = -2, -1 and x = sw, sw + 1 (block-uniform branch)
    if (sx0 + 14 < 0 || sx0 + SW > sw) {
        if (tid < SHF) {
            uint8_t* row = s_src + tid * SW;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int x = k < 2 ? k - 2 : sw + k - 2;
                const int j = x - sx0, jr = agt_reflect101(x, sw) - sx0;
                if (j >= 0 && j < SW && jr >= 0 && jr < SW) row[j] = row[jr];
            }
        }
        __syncthreads();
    }

    // ---- 2. horizontal pass of the 73 rows: 16 groups of 8 outputs (L1 columns X1 .. X1 + 127) ...
    for (int i = tid; i < SHF * (TW / 8); i += NT) {
        const int r = i / (TW / 8), q = i - r * (TW / 8);
        const uint8_t* s = s_src + r * SW + 16 * (q + 1);
        const uint4 d = *reinterpret_cast<const uint4*>(s);
        uint32_t pm = 0, nx = 0;                       // neighbour dwords from the neighbour lanes (one DPP row = one image row)
        if (q == 0) pm = *reinterpret_cast<const uint32_t*>(s - 4);
        if (q == TW / 8 - 1) nx = *reinterpret_cast<const uint32_t*>(s + 16);
        pm = (uint32_t)__builtin_amdgcn_update_dpp((int)pm, (int)d.w, 0x111, 0xf, 0xf, false);    // row_shr:1
        nx = (uint32_t)__builtin_amdgcn_update_dpp((int)nx, (int)d.x, 0x101, 0xf, 0xf, false);    // row_shl:1
        *reinterpret_cast<uint4*>(s_h1 + r * HP1 + 16 * q) = hgroup8(d, pm, nx);
    }
    // ... and the three halo columns X1 - 2, X1 - 1, X1 + 128 (centres at staged bytes 12, 14, 272)
    for (int i = tid; i < SHF * 3; i += NT) {
        const int r = i / 3, e = i - r * 3;
        const uint8_t* s = s_src + r * SW + (e == 0 ? 12 : e == 1 ? 14 : 272);
        const uint32_t hv = (uint32_t)s[-2] + 4u * s[-1] + 6u * s[0] + 4u * s[1] + (uint32_t)s[2];
        *reinterpret_cast<uint16_t*>(s_h1 + r * HP1 + 256 + 2 * e) = (uint16_t)hv;
    }
    __syncthreads();

    // ---- 3. vertical pass -> extended L1 tile (rows Y1 - 2 .. Y1 + 32): interior to HBM, everything to LDS
    for (int i = tid; i < R1 * (TW / 8); i += NT) {
        const int oy = i / (TW / 8), q = i - oy * (TW / 8);
        const uint8_t* h = s_h1 + (2 * oy) * HP1 + 16 * q;
        const uint2 px = vgroup8(*reinterpret_cast<const uint4*>(h), *reinterpret_cast<const uint4*>(h + HP1),
                                 *reinterpret_cast<const uint4*>(h + 2 * HP1), *reinterpret_cast<const uint4*>(h + 3 * HP1),
                                 *reinterpret_cast<const uint4*>(h + 4 * HP1));
        *reinterpret_cast<uint2*>(s_l1 + oy * L1P + 16 + 8 * q) = px;
        const int gy = Y1 + oy - 2, gx = X1 + 8 * q;
        if (oy >= 2 && oy < R1 - 1 && gy < h1 && gx < w1) {
            uint8_t* o = out1 + (long)gy * A0.dpitch + gx;
            if (gx + 7 < w1 && (((uintptr_t)o) & 7) == 0) *reinterpret_cast<uint2*>(o) = px;
            else for (int k = 0; k < 8 && gx + k < w1; k++) o[k] = (uint8_t)((k < 4 ? px.x : px.y) >> (8 * (k & 3)));
        }
    }
    for (int i = tid; i < R1 * 3; i += NT) {
        const int oy = i / 3, e = i - oy * 3;
        const uint16_t* h = reinterpret_cast<const uint16_t*>(s_h1 + (2 * oy) * HP1 + 256 + 2 * e);
        const uint32_t v = (uint32_t)h[0] + 4u * h[HP1 / 2] + 6u * h[HP1] + 4u * h[3 * HP1 / 2] + (uint32_t)h[2 * HP1];
        s_l1[oy * L1P + (e == 0 ? 14 : e == 1 ? 15 : 144)] = (uint8_t)((v + 128) >> 8);
    }
    __syncthreads();
    // L1 positions outside the image: reflect-101 of L1 itself, columns first, then rows (block-uniform branches).
    // Tile byte B of a row <-> L1 column X1 - 16 + B; tile row oy <-> L1 row Y1 - 2 + oy.
    if (X1 - 2 < 0 || X1 + TW >= w1) {
        if (tid < R1) {
            uint8_t* row = s_l1 + tid * L1P;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int x = k < 2 ? k - 2 : w1 + k - 2;                   // -2, -1, w1, w1 + 1
                const int j = x - X1 + 16, jr = agt_reflect101(x, w1) - X1 + 16;
                if (j >= 14 && j <= 144 && jr >= 14 && jr <= 144) row[j] = row[jr];
            }
        }
        __syncthreads();
    }
    if (Y1 - 2 < 0 || Y1 + 2 * TH2 >= h1) {
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int y = k < 2 ? k - 2 : h1 + k - 2;                       // -2, -1, h1, h1 + 1
            const int j = y - Y1 + 2, jr = agt_reflect101(y, h1) - Y1 + 2;
            if (j >= 0 && j < R1 && jr >= 0 && jr < R1 && tid < 131) s_l1[j * L1P + 14 + tid] = s_l1[jr * L1P + 14 + tid];
        }
        __syncthreads();
    }

    // ---- 4a. horizontal pass of the L1 tile: 8 groups of 8 outputs per row (L2 columns X2 .. X2 + 63)
    for (int i = tid; i < R1 * (TW2 / 8); i += NT) {
        const int r = i / (TW2 / 8), q = i - r * (TW2 / 8);
        const uint8_t* s = s_l1 + r * L1P + 16 * (q + 1);
        const uint4 d = *reinterpret_cast<const uint4*>(s);
        const uint32_t pm = *reinterpret_cast<const uint32_t*>(s - 4), nx = *reinterpret_cast<const uint32_t*>(s + 16);
        *reinterpret_cast<uint4*>(s_h2 + r * HP2 + 16 * q) = hgroup8(d, pm, nx);
    }
    __syncthreads();
    // ---- 4b. vertical pass -> L2
    if (tid < TH2 * (TW2 / 8)) {
        const int oy = tid / (TW2 / 8), q = tid - oy * (TW2 / 8);
        const int gy = Y2 + oy, gx = X2 + 8 * q;
        if (gy < h2 && gx < w2) {
            const uint8_t* h = s_h2 + (2 * oy) * HP2 + 16 * q;
            const uint2 px = vgroup8(*reinterpret_cast<const uint4*>(h), *reinterpret_cast<const uint4*>(h + HP2),
                                     *reinterpret_cast<const uint4*>(h + 2 * HP2), *reinterpret_cast<const uint4*>(h + 3 * HP2),
                                     *reinterpret_cast<const uint4*>(h + 4 * HP2));
            uint8_t* o = out2 + (long)gy * A1.dpitch + gx;
            if (gx + 7 < w2 && (((uintptr_t)o) & 7) == 0) *reinterpret_cast<uint2*>(o) = px;
            else for (int k = 0; k < 8 && gx + k < w2; k++) o[k] = (uint8_t)((k < 4 ? px.x : px.y) >> (8 * (k & 3)));
        }
    }
}

}  // namespace agt_pyr2
