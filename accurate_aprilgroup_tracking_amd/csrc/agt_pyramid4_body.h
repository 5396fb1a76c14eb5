// agt_pyramid4_body.h -- TWO pyrDown levels in one pass, register-rolling form: L0 -> L1 -> L2 with level 0 read once, levels 1
// and 2 written once (the W * H * 1.3125 bytes of SURVEY.md section 8d), no LDS, no barriers.  Bit-identical to cv::pyrDown
// applied twice (oracle: oracle/cv_lk.c cvo_pyr_down_u8); same arithmetic as agt_pyramid_body.h / agt_pyramid3_body.h.
//
// A 16-lane DPP row is one UNIT = (strip of `oh2` level-2 rows, column tile of 14 groups): lane q owns level-0 group
// g = 14 c - 1 + q (16 source bytes per row) -- lanes 1..14 are the tile's interior, lanes 0 and 15 a one-group halo on either side
// (every group is interior in exactly one tile).  Per lane and level-0 row: one 16-byte buffer load; horizontal sums (8 x u16) by
// v_dot4_u32_u8 with the neighbours' bytes through DPP row shifts; a five-row rolling window -> one level-1 row (8 pixels) per two
// level-0 rows, stored by the interior lanes AND fed (again through DPP: two level-1 pixels from the left neighbour, one from the
// right -- which is all the halo lanes are for; their own missing neighbours never reach a value that is used) into a second
// horizontal pass (4 x u16) and a second five-row window -> one level-2 row (4 pixels per lane) per two level-1 rows.
// Per level-2 row a lane reads 64 bytes and spends ~200 VALU for 16 + 4 output pixels.
// Redundancy: the halo lanes (16 / 14) and per strip 9 extra level-0 rows (level-1 rows 2 y2 - 2 .. 2 y2 + 2 of the strip's first and
// last level-2 row need level-0 rows 4 y2 - 6 .. 4 y2 + 6): re-read through the XCD's L2, not from HBM.
// Reflection: level-0 rows and the image's left / right edge as in agt_pyramid3_body.h; level-1 rows / columns outside the image
// are reflections OF LEVEL 1 (pyrDown(L1) reflects L1, which differs from filtering reflected level-0 pixels): columns by the same
// in-register trick on the level-1 bytes, rows by picking the reflected entry of the level-1 window (all of them are in it).
// Both exist only in the EDGE form of the row loop (wave-uniform choice).
// Shipped uses: (round 4) the FUSED UPLOAD of agt_track_host_frame -- the pass reads a gray frame straight from pinned host memory
// (one frame: 22.9 us for both levels, PCIe-bound, against a 19 us copy-engine transfer + ~8 us of its submission + a 6 us pyramid
// launch) and stores the frame's level 0 to HBM on the way (COPY); (round 5) the plain HBM -> HBM pyramid pass of every launch of >= 16
// images (agt_pyramid.hip agt_pyr2_plan; pyr_roll2_kernel and the pyramid role of the split pipeline, pyr_group_kernel): with strips
// walked in alternating directions and the lean horizontal / vertical passes below it moves 1.004 x SURVEY 8d's bytes and streams 64
// frames of 1280x720 in 18.6 us alone (profiles/r05_c3_kernel_stats.csv); smaller launches keep the tiled two-level pass.
#pragma once
#include "agt_pyramid3_body.h"

namespace agt_pyr4 {

using namespace agt_pyr;
using agt_pyr3::UNITS_PER_BLOCK;
using agt_pyr3::u32x4;
using agt_pyr3::v4u;
using agt_pyr3::roll_reflect_row;

constexpr int TILE_GROUPS = 14;     // interior groups per column tile
// cache-policy bits of the pass's buffer loads / stores (aux operand: 1 = sc0, 2 = nt, 16 = sc1): experiment builds only
#ifndef AGT_PYR4_LOAD_AUX
#define AGT_PYR4_LOAD_AUX 0
#endif
#ifndef AGT_PYR4_STORE_AUX
#define AGT_PYR4_STORE_AUX 0
#endif
#ifndef AGT_PYR4_RING
#define AGT_PYR4_RING 8
#endif
constexpr int RING = AGT_PYR4_RING; // level-0 rows in flight per lane (a strip is a serial chain of 4 oh2 + 9 rows per lane: the deeper
                                    // the ring, the fewer memory round trips it takes); a loop trip consumes RING rows = RING / 4 level-2 rows
constexpr int L2_PER_TRIP = RING / 4;
static_assert(RING == 8 || RING == 16, "slot arithmetic below: the prologue consumes rows 0..8");

// four horizontal [1 4 6 4 1] sums (u16) from the lane's 8 level-1 bytes (lo: pixels 0..3, hi: 4..7), the dword before (pm: its two
// top bytes are pixels -2, -1) and the byte after (nx & 0xff: pixel 8) -> two registers
__device__ __forceinline__ uint2 hgroup4(uint32_t lo, uint32_t hi, uint32_t pm, uint32_t nx)
{
    const uint32_t W4 = 0x04060401u;
    const uint32_t e0 = __builtin_amdgcn_alignbyte(lo, pm, 2), e2 = __builtin_amdgcn_alignbyte(hi, lo, 2);
    const uint32_t h0 = __builtin_amdgcn_udot4(e0, W4, (lo >> 16) & 0xff, false);      // centre pixel 0, fifth tap: pixel 2
    const uint32_t h1 = __builtin_amdgcn_udot4(lo, W4, hi & 0xff, false);              // centre 2, fifth tap 4
    const uint32_t h2 = __builtin_amdgcn_udot4(e2, W4, (hi >> 16) & 0xff, false);      // centre 4, fifth tap 6
    const uint32_t h3 = __builtin_amdgcn_udot4(hi, W4, nx & 0xff, false);              // centre 6, fifth tap 8
    return make_uint2(h0 | (h1 << 16), h2 | (h3 << 16));
}

// four vertical sums + (v + 128) >> 8 from five rows of four u16 -> four bytes
__device__ __forceinline__ uint32_t vgroup4(uint2 r0, uint2 r1, uint2 r2, uint2 r3, uint2 r4)
{
    const uint32_t K4 = 0x00040004u, K6 = 0x00060006u, K128 = 0x00800080u;
    auto col = [&](uint32_t a0, uint32_t a1, uint32_t a2, uint32_t a3, uint32_t a4) -> uint32_t {
        uint32_t v = pk_mad(pk_add(a1, a3), K4, pk_add(a0, a4));
        v = pk_mad(a2, K6, v);
        return pk_shr8(pk_add(v, K128));
    };
    const uint32_t p0 = col(r0.x, r1.x, r2.x, r3.x, r4.x), p1 = col(r0.y, r1.y, r2.y, r3.y, r4.y);
    return __builtin_amdgcn_perm(p1, p0, 0x06040200u);
}

// round 5: the chain form of agt_pyramid_body.h hgroup8b for the level-1 row of a lane (8 bytes: lo, hi), BIAS = +16 on every sum
template <bool BIAS>
__device__ __forceinline__ uint2 hgroup4b(uint32_t lo, uint32_t hi, uint32_t pm, uint32_t nx)
{
    const uint32_t WB = 0x04010000u, WA = 0x00010406u, WC = 0x04060401u, WD = 0x00000001u;
    const uint32_t c0 = BIAS ? 16u : 0u;
    const uint32_t e0 = __builtin_amdgcn_udot4(lo, WA, __builtin_amdgcn_udot4(pm, WB, c0, false), false);
    const uint32_t o0 = __builtin_amdgcn_udot4(lo, WC, __builtin_amdgcn_udot4(hi, WD, c0, false), false);
    const uint32_t e1 = __builtin_amdgcn_udot4(hi, WA, __builtin_amdgcn_udot4(lo, WB, c0, false), false);
    const uint32_t o1 = __builtin_amdgcn_udot4(hi, WC, __builtin_amdgcn_udot4(nx, WD, c0, false), false);
    return make_uint2(e0 | (o0 << 16), e1 | (o1 << 16));
}

// ... and of vgroup8b: rows 1 and 3 of the window carry the bias
__device__ __forceinline__ uint32_t vgroup4b(uint2 r0, uint2 r1, uint2 r2, uint2 r3, uint2 r4)
{
    const uint32_t p0 = vcol_b(r0.x, r1.x, r2.x, r3.x, r4.x), p1 = vcol_b(r0.y, r1.y, r2.y, r3.y, r4.y);
    return __builtin_amdgcn_perm(p1, p0, 0x07050301u);
}

// level-0 horizontal sums of one row (no edge dword: the halo lanes stand in for the tile's neighbours)
template <bool ODD, bool LR>
__device__ __forceinline__ uint4 roll2_hrow(u32x4 d, bool left_edge, bool right_edge)
{
    // (the image's first group sits on lane 1 of tile 0, behind an idle halo lane: selected AFTER the shift, unlike agt_pyramid3_body.h)
    // (bound_ctrl: the row's first / last lane reads 0 -- no register has to be zeroed for the shift's "old" operand)
    uint32_t pm = (uint32_t)__builtin_amdgcn_mov_dpp((int)d.w, 0x111, 0xf, 0xf, true);                            // row_shr:1
    uint32_t nx = (uint32_t)__builtin_amdgcn_mov_dpp((int)d.x, 0x101, 0xf, 0xf, true);                            // row_shl:1
    if constexpr (LR) {       // (waves whose column tiles touch neither image edge skip the reflection selects: wave-uniform choice)
        pm = left_edge ? __builtin_amdgcn_perm(d.x, d.x, 0x01020000u) : pm;
        nx = right_edge ? (d.w >> 16) : nx;
    }
    return hgroup8b<ODD>(make_uint4(d.x, d.y, d.z, d.w), pm, nx);
}

template <bool V> struct Odd { static constexpr bool value = V; };

// COPY: the level-0 rows the strip owns (4 oy2 .. 4 oy2 + 4 oh2 - 1; interior lanes) are also stored to `rc` (pitch cpitch): the
// source then is the caller's frame in pinned HOST memory, read over PCIe exactly once per byte that matters, and the copy is the
// frame's level 0 in HBM -- upload and pyramid in one pass (agt_track_host_frame).
// REV (round 5): the unit may walk its strip BOTTOM-UP (`rev`, per lane: the units of a wave belong to different strips).  The filters
// are symmetric, so the same rolling windows work on the reversed row sequence; only the row a stream index stands for changes
// (level 0: ylast - r, level 1: y1last - i, level 2: oy2 + oh2 - 1 - j).  Alternating directions (strip s top-down, s + 1 bottom-up)
// make neighbouring strips read their shared 9 halo rows AT THE SAME TIME -- both start at, or both end at, their common border --
// so the second reader finds the lines in the XCD's L2 instead of fetching them again from memory (measured before: FETCH_SIZE of
// the pass = (4 oh2 + 9) / (4 oh2) of the image, every halo row came from memory a second time).
template <bool EDGE, bool COPY, bool REV, bool LR, typename RS>
__device__ __forceinline__ void pyr_roll2_rows(const AgtPyrArgs& A0, const AgtPyrArgs& A1, const RS rs, const RS r1, const RS r2, const RS rc, int cpitch,
                                               int oy2, int g, int q, int G, bool lane_on, bool rev)
{
    const int sh = A0.sh, h1 = A0.dh, h2 = A1.dh, oh2 = A0.pad;
    const int pitch = (int)A0.spitch, pitch1 = (int)A0.dpitch, pitch2 = (int)A1.dpitch;
    const int NR = 4 * oh2 + 9;                                  // level-0 rows of the strip
    const int y1a = 2 * oy2 - 2;                                 // first level-1 row the strip computes (two halo rows above its own)
    const int y0 = 2 * y1a - 2;                                  // first level-0 row
    const bool left_edge = g == 0, right_edge = g == G - 1;
    const bool writer = lane_on && q >= 1 && q <= TILE_GROUPS;   // interior lane: stores its level-1 and level-2 pixels
    const int xoff = g * 16;
    const int ylast = y0 + NR - 1;
    const int ybase = REV && rev ? ylast : y0, dir = REV && rev ? -1 : 1;      // level-0 row of stream index r: ybase + dir * r
    // (lanes without a group load from past the buffer's end: issued unconditionally, 0 comes back -- agt_pyramid3_body.h)
    const int vbase = lane_on ? (EDGE ? xoff : ybase * pitch + xoff) : (int)0x80000000u;
    const int dpitch_lane = lane_on ? dir * pitch : 0;
    u32x4 d[RING];
    // (a row past the strip's last one is "loaded" from past the buffer's end as well: no branch around the load, no copy of
    // the slot's old contents for the path that skips it)
    auto issue = [&](int slot, int r) {
        int vo = vbase, so = 0;
        if constexpr (EDGE) vo = vbase + roll_reflect_row(ybase + dir * r, sh) * pitch;
        else if constexpr (REV) vo = vbase + r * dpitch_lane;
        else so = r * pitch;
        if (r >= NR) vo = (int)0x80000000u;
        const v4u t = __builtin_amdgcn_raw_buffer_load_b128(rs, vo, so, AGT_PYR4_LOAD_AUX); d[slot] = __builtin_bit_cast(u32x4, t);
    };
    auto take = [&](int slot, int r, auto odd) {                 // horizontal sums of strip row r (in `slot`), then refill the slot
        if constexpr (COPY) {
            const int y = ybase + dir * r;
            if (writer && y >= 4 * oy2 && y < 4 * oy2 + 4 * oh2 && y < sh)
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u, d[slot]), rc, y * cpitch + xoff, 0, 0);
        }
        const uint4 h = roll2_hrow<decltype(odd)::value, LR>(d[slot], left_edge, right_edge);
        issue(slot, r + RING);
        return h;
    };
    constexpr Odd<false> EV{}; constexpr Odd<true> OD{};
    // level-1 row `i` of the strip (image row y1a + i) from the level-0 window: the pixels (two dwords), stored by the interior
    // lanes if the row is the strip's own, and their horizontal sums for level 2
    const int o1base = 2 * oy2 * pitch1 + g * 8;                   // (level-1 row i = 2: the strip's first own row; offsets stay non-negative)
    const int y1last = y1a + 2 * oh2 + 2;
    // Stores, REV form: the strip's own rows as a RANGE OF STREAM INDICES per lane (empty for lanes that store nothing) and the
    // address as base + index * step -- two compares against the wave-uniform index and one 24-bit multiply-add per row.
    //   level 1, top-down:  row y1a + i,    own rows 2 <= i < 2 oh2 + 2,  inside the image while i < h1 - y1a
    //            bottom-up: row y1last - i, own rows 1 <= i <= 2 oh2,     inside the image while i > y1last - h1
    //   level 2, top-down:  row oy2 + j (j < h2 - oy2);  bottom-up: row oy2 + oh2 - 1 - j (j >= oy2 + oh2 - h2)
    const bool rv = REV && rev;
    int i_lo = rv ? (y1last - h1 + 1 > 1 ? y1last - h1 + 1 : 1) : 2, i_hi = rv ? 2 * oh2 + 1 : (h1 - y1a < 2 * oh2 + 2 ? h1 - y1a : 2 * oh2 + 2);
    int j_lo = rv ? (oy2 + oh2 - h2 > 0 ? oy2 + oh2 - h2 : 0) : 0, j_hi = rv ? oh2 : (h2 - oy2 < oh2 ? h2 - oy2 : oh2);
    if (!writer) { i_lo = i_hi = 0; j_lo = j_hi = 0; }
    const int a1_lane = (rv ? y1last : y1a) * pitch1 + g * 8, p1_lane = rv ? -pitch1 : pitch1;
    const int a2_lane = (rv ? oy2 + oh2 - 1 : oy2) * pitch2 + g * 4, p2_lane = rv ? -pitch2 : pitch2;
    auto level1 = [&](const uint4& a0, const uint4& a1, const uint4& a2, const uint4& a3, const uint4& a4, int i, auto odd) {
        const uint2 px = vgroup8b(a0, a1, a2, a3, a4);
        if constexpr (REV) {
            if (i >= i_lo && i < i_hi)
                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(agt_pyr3::v2u, px), r1, a1_lane + __mul24(i, p1_lane), 0, AGT_PYR4_STORE_AUX);
        } else {
            const int y1 = y1a + i;
            if (writer && i >= 2 && i < 2 * oh2 + 2 && y1 < h1)
                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(agt_pyr3::v2u, px), r1, o1base, (i - 2) * pitch1, AGT_PYR4_STORE_AUX);
        }
        // neighbours' level-1 pixels: -2, -1 from the lane on the left, 8 from the lane on the right; the image's own edges
        // reflect level 1 (pixel -2 = 2, -1 = 1; pixel w1 = w1 - 2)
        uint32_t pm = (uint32_t)__builtin_amdgcn_mov_dpp((int)px.y, 0x111, 0xf, 0xf, true);
        uint32_t nx = (uint32_t)__builtin_amdgcn_mov_dpp((int)px.x, 0x101, 0xf, 0xf, true);
        if constexpr (LR) {
            pm = left_edge ? __builtin_amdgcn_perm(px.x, px.x, 0x01020000u) : pm;
            nx = right_edge ? (px.y >> 16) : nx;
        }
        return hgroup4b<decltype(odd)::value>(px.x, px.y, pm, nx);
    };
#pragma unroll
    for (int k = 0; k < RING; k++) issue(k, k);
    // prologue: level-0 rows 0..8 -> level-1 rows 0, 1, 2 (row r lives in slot r % RING)
    // (stream rows at odd indices and level-1 rows at odd indices are rows 1 / 3 of the vertical windows they enter: biased sums)
    uint4 H0 = take(0, 0, EV), H1 = take(1, 1, OD), H2 = take(2, 2, EV);
    uint2 K0, K1, K2;
    {
        uint4 H3 = take(3, 3, OD), H4 = take(4, 4, EV);
        K0 = level1(H0, H1, H2, H3, H4, 0, EV); H0 = H2; H1 = H3; H2 = H4;
        H3 = take(5, 5, OD); H4 = take(6, 6, EV);
        K1 = level1(H0, H1, H2, H3, H4, 1, OD); H0 = H2; H1 = H3; H2 = H4;
        H3 = take(7, 7, OD); H4 = take(8 % RING, 8, EV);
        K2 = level1(H0, H1, H2, H3, H4, 2, EV); H0 = H2; H1 = H3; H2 = H4;
    }
    const int o2base = oy2 * pitch2 + g * 4;
    for (int T = 0; T < oh2 / L2_PER_TRIP; T++) {
#pragma unroll
        for (int u = 0; u < L2_PER_TRIP; u++) {
            const int j = L2_PER_TRIP * T + u;                   // level-2 row of the strip
            const int r = 9 + 4 * j;                             // its four new level-0 rows r .. r + 3: slots (9 + 4 u + k) % RING
            uint4 H3 = take((9 + 4 * u) % RING, r, OD), H4 = take((10 + 4 * u) % RING, r + 1, EV);
            const uint2 K3 = level1(H0, H1, H2, H3, H4, 3 + 2 * j, OD); H0 = H2; H1 = H3; H2 = H4;
            H3 = take((11 + 4 * u) % RING, r + 2, OD); H4 = take((12 + 4 * u) % RING, r + 3, EV);
            const uint2 K4 = level1(H0, H1, H2, H3, H4, 4 + 2 * j, EV); H0 = H2; H1 = H3; H2 = H4;
            uint2 E0 = K0, E1 = K1, E3 = K3, E4 = K4;
            const int row2 = REV && rev ? oy2 + oh2 - 1 - j : oy2 + j;           // the level-2 row this trip completes
            if constexpr (EDGE) {
                // level-1 rows outside the image: their reflections are in the window (centre row yc = 2 row2).  In space the window is
                // (up2, up1, c, dn1, dn2) = (K0 .. K4) top-down and (K4 .. K0) bottom-up; the rules: above the first row up2 <- dn2,
                // up1 <- dn1; below the last row dn1 <- up1, dn2 <- c (one row out) or up2 (two rows out)
                const int yc = 2 * row2;
                const bool top = yc == 0, b1 = yc + 2 == h1, b2 = yc + 1 == h1;
                const uint2 up2 = rv ? K4 : K0, up1 = rv ? K3 : K1, dn1 = rv ? K1 : K3, dn2 = rv ? K0 : K4;
                E0.x = top ? dn2.x : up2.x; E0.y = top ? dn2.y : up2.y;
                E1.x = top ? dn1.x : up1.x; E1.y = top ? dn1.y : up1.y;
                E3.x = b2 ? up1.x : dn1.x; E3.y = b2 ? up1.y : dn1.y;
                E4.x = b1 ? K2.x : (b2 ? up2.x : dn2.x); E4.y = b1 ? K2.y : (b2 ? up2.y : dn2.y);
            }
            const uint32_t o = vgroup4b(E0, E1, K2, E3, E4);
            if constexpr (REV) { if (j >= j_lo && j < j_hi) __builtin_amdgcn_raw_buffer_store_b32(o, r2, a2_lane + __mul24(j, p2_lane), 0, AGT_PYR4_STORE_AUX); }
            else if (writer && oy2 + j < h2) __builtin_amdgcn_raw_buffer_store_b32(o, r2, o2base, j * pitch2, AGT_PYR4_STORE_AUX);
            K0 = K2; K1 = K3; K2 = K4;
        }
    }
}

// One 256-thread workgroup = 16 units of the image at `img`.  A0: level 0 -> 1 geometry (A0.pad = level-2 rows per strip, even),
// A1: level 1 -> 2 geometry.
template <bool COPY = false, bool REV = true>
__device__ __forceinline__ void pyr_roll2_body(const AgtPyrArgs& A0, const AgtPyrArgs& A1, int blk, const uint8_t* __restrict__ img,
                                               uint8_t* __restrict__ out1, uint8_t* __restrict__ out2, uint8_t* copy = nullptr, int cpitch = 0)
{
    const int tid = threadIdx.x, q = tid & 15;
    const int G = A0.sw >> 4, ncol = (G + TILE_GROUPS - 1) / TILE_GROUPS, oh2 = A0.pad;
    const int nstrip = (A1.dh + oh2 - 1) / oh2, units = nstrip * ncol;
    // Unit order: column tile fastest -- the four units of a wave are (mostly) four ADJACENT COLUMN TILES of one strip, 896 contiguous
    // bytes of every row.  Round 5 measured the other order (a wave = four consecutive strips of one column tile: strips that share
    // halo rows in lockstep, and only the waves of the first / last column tile carrying the image-edge selects -- 43 instead of 50
    // VALU per row): the pass alone 19.4-20.8 -> 21.6-22.1 us, the pipelined cold-pair step 46.5-47.5 -> 55.1-55.9 us.  Four
    // 256-byte row segments 51 KB apart per wave instruction are the worse memory pattern by far (profiles/r05_experiments.md).
    const int u = blk * UNITS_PER_BLOCK + (tid >> 4);
    const bool uvalid = u < units;
    const int s = uvalid ? u / ncol : 0, c = uvalid ? u - s * ncol : 0;
    const int g = c * TILE_GROUPS - 1 + q;
    const bool lane_on = uvalid && g >= 0 && g < G;
    const int oy2 = (uvalid ? s : 0) * oh2;
    const int y0 = 4 * oy2 - 6;
    // EDGE: some level-0 row of the strip is outside the image, or some level-1 row its level-2 rows use is
    const bool edge = uvalid && (y0 < 0 || y0 + 4 * oh2 + 8 > A0.sh - 1 || 2 * (oy2 + oh2 - 1) + 2 > A0.dh - 1);
    const auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(img), 0, A0.sh * (int)A0.spitch, 0x00020000);
    const auto r1 = __builtin_amdgcn_make_buffer_rsrc(out1, 0, A0.dh * (int)A0.dpitch, 0x00020000);
    const auto r2 = __builtin_amdgcn_make_buffer_rsrc(out2, 0, A1.dh * (int)A1.dpitch, 0x00020000);
    const auto rc = __builtin_amdgcn_make_buffer_rsrc(COPY ? copy : out2, 0, COPY ? A0.sh * cpitch : 0, 0x00020000);
    if (__builtin_amdgcn_ballot_w64(uvalid) == 0) return;
    const bool rev = REV && (s & 1) && A0.rsv_ == 0;              // odd strips bottom-up
    const bool lr = __builtin_amdgcn_ballot_w64(lane_on && (g == 0 || g == G - 1)) != 0;
    if (__builtin_amdgcn_ballot_w64(edge) != 0) pyr_roll2_rows<true, COPY, REV, true>(A0, A1, rs, r1, r2, rc, cpitch, oy2, g, q, G, lane_on, rev);
    else if (lr) pyr_roll2_rows<false, COPY, REV, true>(A0, A1, rs, r1, r2, rc, cpitch, oy2, g, q, G, lane_on, rev);
    else pyr_roll2_rows<false, COPY, REV, false>(A0, A1, rs, r1, r2, rc, cpitch, oy2, g, q, G, lane_on, rev);
}

__host__ __device__ inline int roll2_blocks(int sw, int h2, int oh2)
{
    const int G = sw >> 4, ncol = (G + TILE_GROUPS - 1) / TILE_GROUPS, nstrip = (h2 + oh2 - 1) / oh2;
    return (nstrip * ncol + UNITS_PER_BLOCK - 1) / UNITS_PER_BLOCK;
}

}  // namespace agt_pyr4
