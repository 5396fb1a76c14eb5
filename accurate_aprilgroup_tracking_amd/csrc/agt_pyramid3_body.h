// agt_pyramid3_body.h -- cv::pyrDown (u8, 5x5 [1 4 6 4 1]^2 / 256, BORDER_REFLECT_101), REGISTER-ROLLING form: no LDS, no
// barriers.  Same arithmetic, bit for bit, as agt_pyramid_body.h (whose hgroup8 / vgroup8 it uses); oracle: oracle/cv_lk.c
// cvo_pyr_down_u8.  Round 4: the tiled kernel (stage a 35 x 288 B tile in LDS, horizontal pass into a u16 plane in LDS, vertical
// pass, two barriers, 24 VALU per output pixel) streamed 64 x 720p at 4.0 TB/s L0 -> L1 and 1.9 TB/s L1 -> L2 and was the
// longest link of the cold-pair form of BASELINE configs[2].
//
// Mapping: a 16-lane DPP row is one UNIT = (strip s of `oh` output rows, column tile c of 256 source bytes = 128 outputs); lane q
// of the row owns the 16 source bytes [256 c + 16 q, +16) of every source row of the strip and the 8 outputs above them.
//   * per source row a lane issues ONE 16-byte buffer load (row offset in the scalar offset, no address arithmetic); the four
//     bytes before / after its 16 come from the neighbour lanes by DPP row shifts; only lanes 0 / 15 of a row fetch one extra
//     dword across the tile boundary (image edges: reflected from the lane's own bytes);
//   * the horizontal [1 4 6 4 1] sums of a row (8 x u16, four registers) are formed at once (one v_dot4_u32_u8 per output) and
//     kept in a FIVE-ROW ROLLING WINDOW in registers; every second source row one output row is completed by the packed
//     16-bit vertical pass and stored with one 8-byte buffer store;
//   * RING = 8 source rows are in flight per lane at any time (a ring of load destinations with static slots: the loop body
//     handles 8 source rows = 4 output rows), so a wave keeps 8 KB of reads outstanding without any other wave's help.
// Top / bottom image edges (reflect-101 rows) and partial last strips take the EDGE form of the row loop (per-lane reflected
// row index, one multiply-add per load); the choice is wave-uniform.  Work per output pixel: ~12 VALU, no LDS traffic.
// Eligibility (agt_pyr3_plan): 16-byte aligned source with pitch and width multiples of 16, 8-byte aligned destination;
// everything else (odd widths, tiny images, unaligned crops) keeps the tiled kernel.
#pragma once
#include "agt_pyramid_body.h"

#ifndef AGT_PYR3_RING
#define AGT_PYR3_RING 8
#endif

namespace agt_pyr3 {

using namespace agt_pyr;

constexpr int RING = AGT_PYR3_RING;             // source rows in flight per lane
constexpr int UO = RING / 2;        // output rows per loop trip
constexpr int UNITS_PER_BLOCK = NT / 16;

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int v4u __attribute__((__vector_size__(16)));
typedef unsigned int v2u __attribute__((__vector_size__(8)));

// reflect-101 of a row index that is at most one period outside, clamped into the image (rows past the reflected band only
// feed outputs that are never stored)
__device__ __forceinline__ int roll_reflect_row(int y, int sh)
{
    y = y < 0 ? -y : y;
    const int m = 2 * (sh - 1) - y;
    y = y < m ? y : m;
    return y < 0 ? 0 : y;
}

// horizontal sums of one source row of the lane (d: its 16 bytes, e: the dword across the tile boundary for lanes 0 / 15);
// ODD: the row sits at an odd offset of the vertical windows it enters (rows 1 / 3 of five): its sums carry the +16 bias of
// agt_pyramid_body.h hgroup8b / vgroup8b
template <bool ODD>
__device__ __forceinline__ uint4 roll_hrow(u32x4 d, uint32_t e, bool left_edge, bool right_edge)
{
    // bytes -2, -1 of the image = bytes 2, 1 (reflect-101): as the high half of the dword "before"
    const uint32_t pm_old = left_edge ? __builtin_amdgcn_perm(d.x, d.x, 0x01020000u) : e;
    uint32_t pm = (uint32_t)__builtin_amdgcn_update_dpp((int)pm_old, (int)d.w, 0x111, 0xf, 0xf, false);    // row_shr:1
    uint32_t nx = (uint32_t)__builtin_amdgcn_update_dpp((int)e, (int)d.x, 0x101, 0xf, 0xf, false);         // row_shl:1
    nx = right_edge ? (d.w >> 16) : nx;                                   // byte 16 = byte sw of the image = byte sw - 2
    return hgroup8b<ODD>(make_uint4(d.x, d.y, d.z, d.w), pm, nx);
}

// One 256-thread workgroup = 16 units of the image at `img` (units blk * 16 ..).  A.pad = output rows per strip (multiple of UO).
template <bool EDGE, typename RS>
__device__ __forceinline__ void pyr_roll_rows(const AgtPyrArgs& A, const RS rs, const RS rd, int y0, int oy0, int g, int q, int G, bool lane_on)
{
    const int sh = A.sh, dh = A.dh, oh = A.pad;
    const int pitch = (int)A.spitch, dpitch = (int)A.dpitch;
    const int NR = 2 * oh + 3;
    const bool left_edge = g == 0, right_edge = g == G - 1;
    const bool eon = lane_on && ((q == 0 && g > 0) || (q == 15 && g + 1 < G));
    const int eoff = q == 0 ? -4 : 16;
    const int xoff = g * 16;
    // Lanes that own no group (and lanes without an edge dword) load from an offset past the buffer: the load is issued
    // unconditionally and returns 0 for them (buffer range check) -- no exec-mask changes and no copies of the old slot
    // contents around every load (round 5; 0x80000000 + any in-image offset stays >= 2^31 > num_records)
    const int vbase = lane_on ? (EDGE ? xoff : y0 * pitch + xoff) : (int)0x80000000u;           // (non-edge waves: every row of the strip is inside the image)
    const int ebase = eon ? (EDGE ? xoff : y0 * pitch + xoff) + eoff : (int)0x80000000u;
    u32x4 d[RING];
    uint32_t e[RING];
    auto issue = [&](int slot, int r) {                          // r: row of the strip (wave-uniform)
        int ro = 0, so = 0;
        if constexpr (EDGE) ro = roll_reflect_row(y0 + r, sh) * pitch;
        else so = r * pitch;
        const v4u t = __builtin_amdgcn_raw_buffer_load_b128(rs, vbase + ro, so, 0); d[slot] = __builtin_bit_cast(u32x4, t);
        e[slot] = __builtin_amdgcn_raw_buffer_load_b32(rs, ebase + ro, so, 0);
    };
#pragma unroll
    for (int k = 0; k < RING; k++) issue(k, k);
    uint4 H0, H1, H2;
    H0 = roll_hrow<false>(d[0], e[0], left_edge, right_edge); if (RING + 0 < NR) issue(0, RING + 0);
    H1 = roll_hrow<true>(d[1], e[1], left_edge, right_edge); if (RING + 1 < NR) issue(1, RING + 1);
    H2 = roll_hrow<false>(d[2], e[2], left_edge, right_edge); if (RING + 2 < NR) issue(2, RING + 2);
    const int obase = oy0 * dpitch + g * 8;
    for (int t = 0; t < oh / UO; t++) {
#pragma unroll
        for (int u = 0; u < UO; u++) {
            const int r = 3 + RING * t + 2 * u;                  // rows r, r + 1 complete output row j
            const int s0 = (3 + 2 * u) % RING, s1 = (4 + 2 * u) % RING;
            const uint4 H3 = roll_hrow<true>(d[s0], e[s0], left_edge, right_edge);
            if (r + RING < NR) issue(s0, r + RING);
            const uint4 H4 = roll_hrow<false>(d[s1], e[s1], left_edge, right_edge);
            if (r + 1 + RING < NR) issue(s1, r + 1 + RING);
            const uint2 o = vgroup8b(H0, H1, H2, H3, H4);
            const int j = UO * t + u;
            if (lane_on && oy0 + j < dh)
                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(v2u, o), rd, obase, j * dpitch, 0);
            H0 = H2; H1 = H3; H2 = H4;
        }
    }
}

__device__ __forceinline__ void pyr_roll_body(const AgtPyrArgs& A, int blk, const uint8_t* __restrict__ img, uint8_t* __restrict__ out)
{
    const int tid = threadIdx.x, q = tid & 15;
    const int G = A.sw >> 4, ncol = (G + 15) >> 4, oh = A.pad;
    const int nstrip = (A.dh + oh - 1) / oh, units = nstrip * ncol;
    const int u = blk * UNITS_PER_BLOCK + (tid >> 4);
    const bool uvalid = u < units;
    const int s = uvalid ? u / ncol : 0, c = uvalid ? u - s * ncol : 0;
    const int g = c * 16 + q;
    const bool lane_on = uvalid && g < G;
    const int oy0 = s * oh, y0 = 2 * oy0 - 2;
    const bool edge = uvalid && (y0 < 0 || y0 + 2 * oh + 2 > A.sh - 1);
    const auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(img), 0, A.sh * (int)A.spitch, 0x00020000);
    const auto rd = __builtin_amdgcn_make_buffer_rsrc(out, 0, A.dh * (int)A.dpitch, 0x00020000);
    if (__builtin_amdgcn_ballot_w64(uvalid) == 0) return;
    if (__builtin_amdgcn_ballot_w64(edge) != 0) pyr_roll_rows<true>(A, rs, rd, y0, oy0, g, q, G, lane_on);
    else pyr_roll_rows<false>(A, rs, rd, y0, oy0, g, q, G, lane_on);
}

// blocks per image of the rolling form for a level of sw x sh -> (dw, dh), strips of oh output rows
__host__ __device__ inline int roll_blocks(int sw, int dh, int oh)
{
    const int G = sw >> 4, ncol = (G + 15) >> 4, nstrip = (dh + oh - 1) / oh;
    return (nstrip * ncol + UNITS_PER_BLOCK - 1) / UNITS_PER_BLOCK;
}

}  // namespace agt_pyr3
