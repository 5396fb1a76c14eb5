// agt_lk_rs_body.h -- cv::calcOpticalFlowPyrLK per-point tracker, ONE WAVE PER CORNER, 21x21 window:
// the throughput kernel of big batches (BASELINE configs[2]: 64 x 1280x720, 3072 corners per step).
// Same semantics and the same bits as agt_lk_body.h (oracle: oracle/cv_lk.c, CVO_ACC_EXACT); different mapping.
//
// Round 1's one-wave kernel gave lane l the window pixels l, l + 64, ... (7 per lane): 28 single-byte LDS reads
// and ~20 VALU per pixel per iteration, a Scharr tile written to and re-read from LDS per level -- 4,184 VALU and
// 633 LDS instructions per corner, VALU-issue bound at three to four waves per SIMD.  Here:
//   * lane l = (row r, segment s) = (l / 3, l % 3) owns the 7 CONSECUTIVE window pixels [7s, 7s + 7) of window row r
//     (63 lanes; lane 63 idles).  Its two source rows come as three aligned dwords each (6 LDS dword reads per
//     iteration instead of 28 byte reads), aligned with v_alignbyte_b32;
//   * the fixed-point bilinear tap  v00*iw00 + v01*iw01 + v10*iw10 + v11*iw11  is two v_dot4_u32_u8: the four bytes
//     are packed into one register with v_perm_b32 and the 15-bit weights are split into high and low bytes
//     (sum = 256 * dot4(P, WH) + dot4(P, WL), exact); the rounding constant rides in the accumulator input
//     [round 6: two v_dot2_i32_i16 with int16 weights instead -- rs_weights_i16 / rs_tap2 below];
//   * the I side of a level lives in registers only.  By linearity  sum_ij w_ij * Scharr(I)(x+j, y+i)  =
//     Scharr(B)(x, y)  with  B = sum_ij w_ij * I(.+j, .+i)  taken WITHOUT rounding (B < 2^22, Scharr(B) < 2^27):
//     each lane interpolates its 3 x 9 patch of B (27 dot4 pairs) and applies the Scharr taps to it -- no derivative
//     tile in LDS, no barrier.  Valid while every derivative position of the window lies inside the image (the
//     derivative image has a ZERO border); corners whose window touches the image border at some level take the
//     general kernel body (agt_lk_body.h) instead, chosen per corner at entry;
//   * the exact sums: per-lane partials (< 2^28) are split into 16-bit halves, each half is reduced over the wave in
//     int32 without overflow (two sums share one DPP chain through v_permlane32_swap), and  hi * 65536 + lo  is formed
//     in FP64 (exact) and rounded to float once -- the same value as (float)(double)(int64 sum).
#pragma once
#include "agt_lk_body.h"

namespace agt_lk {

// Round 6 (the iteration's instruction diet): the fixed-point bilinear tap
//     v00 * iw00 + v01 * iw01 + v10 * iw10 + v11 * iw11 + c
// is TWO v_dot2_i32_i16 -- one per image row: the row's two neighbouring bytes zero-extended into the halves of a register (one
// v_perm_b32), the two weights of the row as int16 halves of a scalar register, the first dot product's result as the second
// one's accumulator.  Rounds 2-5 ran two v_dot4_u32_u8 per tap (the 15-bit weights split into byte planes) plus a shift-add to
// join the planes, packed the weights into byte planes on the scalar unit (~25 instructions per iteration) and needed a
// correction path for iw11 = -1 (a byte plane cannot hold it): the signed 16-bit form has none of the three.
// byte selectors of v_perm_b32(S0 = bytes 4..7, S1 = bytes 0..3): { b[j], 0, b[j+1], 0 } for j = 0..3 (0x0c selects the constant 0)
constexpr uint32_t RS_PAIR0 = 0x0c010c00u, RS_PAIR1 = 0x0c020c01u, RS_PAIR2 = 0x0c030c02u, RS_PAIR3 = 0x0c040c03u;

typedef short rs_i16x2 __attribute__((ext_vector_type(2)));

// bilinear_weights (agt_lk_body.h) for the dot2 form: a, b = the position's fractions; W01 = { iw00, iw01 }, W23 = { iw10, iw11 } as
// int16 halves.  Same integers as OpenCV's cvRound((1 - a) * (1 - b) * 2^14) etc., in fewer instructions:
//   * the factor 2^14 is applied to (1 - b) and b first: a product scaled by a power of two rounds to the scaled rounded product
//     (no overflow, no denormals: the fractions are multiples of 2^-13 at image coordinates below 2^10 .. 2^11);
//   * cvRound (nearest, ties to even) of 0 <= t <= 2^14 is the float addition t + 1.5 * 2^23: the sum's low mantissa bits ARE the
//     rounded integer -- the conversion, and everything after it, happens on the scalar unit (the weights are wave-uniform).
// a wave-uniform value moved to a scalar register, opaquely on both sides: what is computed FROM it stays on the scalar unit, and the
// read-first-lane is not hoisted through the float addition that produced it (the compiler otherwise adds on the vector unit after the move)
__device__ __forceinline__ int rs_scalar(int v)
{
    asm("" : "+v"(v));
    v = __builtin_amdgcn_readfirstlane(v);
    asm("" : "+s"(v));
    return v;
}
__device__ __forceinline__ float rs_scalar(float v) { return __int_as_float(rs_scalar(__float_as_int(v))); }
// a scalar value pinned to a scalar register (the read-first-lane is free when the compiler already holds it in one)
__device__ __forceinline__ int rs_pin(int v) { v = __builtin_amdgcn_readfirstlane(v); asm("" : "+s"(v)); return v; }

__device__ __forceinline__ void rs_weights_i16(float a, float b, uint32_t& W01, uint32_t& W23)
{
    const float MAGIC = 12582912.f;                      // 1.5 * 2^23 = 0x4B400000
    const float na = 1.f - a, nb = 1.f - b;
    const float s1 = nb * (float)(1 << W_BITS), s2 = b * (float)(1 << W_BITS);
    const float y00 = na * s1 + MAGIC, y01 = a * s1 + MAGIC, y10 = na * s2 + MAGIC;
    const int iw00 = rs_scalar(__float_as_int(y00)) - 0x4B400000, iw01 = rs_scalar(__float_as_int(y01)) - 0x4B400000;
    const int iw10 = rs_scalar(__float_as_int(y10)) - 0x4B400000;
    const int iw11 = (1 << W_BITS) - iw00 - iw01 - iw10;              // (may be -1: a signed half holds it)
    W01 = ((uint32_t)iw00 & 0xffffu) | ((uint32_t)iw01 << 16);
    W23 = ((uint32_t)iw10 & 0xffffu) | ((uint32_t)iw11 << 16);
}

// one tap: A = { upper row's bytes k, k + 1 }, B = { lower row's }, both zero-extended to 16-bit halves; acc: rounding constant etc.
// (the first dot product is written out: the compiler selects the two-address v_dot2c_i32_i16 for the builtin and copies the
// accumulator -- a loop-carried constant -- in front of it, one move per tap; the three-address form reads it in place.  The second one's
// accumulator is the first one's result: dead afterwards, the two-address form costs nothing there.)
template <bool ACC = true>
__device__ __forceinline__ int rs_tap2(uint32_t A, uint32_t B, uint32_t W01, uint32_t W23, int acc = 0)
{
    int t;
    if (ACC) asm("v_dot2_i32_i16 %0, %1, %2, %3" : "=v"(t) : "v"(A), "s"(W01), "v"(acc));
    else asm("v_dot2_i32_i16 %0, %1, %2, 0" : "=v"(t) : "v"(A), "s"(W01));
    return __builtin_amdgcn_sdot2(__builtin_bit_cast(rs_i16x2, B), __builtin_bit_cast(rs_i16x2, W23), t, false);
}

// the byte pairs (k, k + 1), k = 0 .. N - 1, of one row given as aligned dwords r[0], r[1], .. (bytes 0..3, 4..7, ..): pair k sits in
// dwords k / 4 and k / 4 + 1 at byte k % 4
template <int N, int ND>
__device__ __forceinline__ void rs_row_pairs(const uint32_t (&r)[ND], uint32_t (&Q)[N])
{
    static_assert(N + 1 <= 4 * ND, "pair N - 1 reads byte N");
#pragma unroll
    for (int k = 0; k < N; k++) {
        const uint32_t lo = r[k / 4], hi = (k % 4 == 3) ? r[k / 4 + 1 < ND ? k / 4 + 1 : ND - 1] : 0u;
        Q[k] = (k % 4 == 0) ? __builtin_amdgcn_perm(hi, lo, RS_PAIR0) : (k % 4 == 1) ? __builtin_amdgcn_perm(hi, lo, RS_PAIR1)
             : (k % 4 == 2) ? __builtin_amdgcn_perm(hi, lo, RS_PAIR2) : __builtin_amdgcn_perm(hi, lo, RS_PAIR3);
    }
}

// a * b + c for operands within 24 bits, as ONE full-rate instruction (the compiler prefers v_mul_i32_i24 + v_add3_u32 trees:
// shorter dependency chains, more instructions -- the one-wave kernel is bound by instruction issue, not by latency)
__device__ __forceinline__ int rs_mad24(int a, int b, int c)
{
    int r;
    asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

// three aligned dwords of an LDS row from byte address `a` on; `sh` = a & 3 is applied by the caller
__device__ __forceinline__ void rs_row3(const uint8_t* s, int a, uint32_t& d0, uint32_t& d1, uint32_t& d2)
{
    const uint32_t* p = reinterpret_cast<const uint32_t*>(s + (a & ~3));
    d0 = p[0]; d1 = p[1]; d2 = p[2];
}

// Exact wave sums of two int32 per lane (|v| < 2^28), identical in every lane, ROUNDED ONCE to float.
// v_permlane32_swap folds { v0 | v1 } into one register (lanes 0-31: pair sums of v0, lanes 32-63: of v1); the pair sums (< 2^29)
// and two DPP steps (x 4: < 2^31) still fit int32, so the values travel WHOLE that far and are split into 16-bit halves only for
// the last two row steps and the read-lanes (round 4; rounds 2 / 3 ran two complete half-chains: 24 vector instructions against 18).
// Both half sums are below 2^24 in magnitude, so they are exact as floats and fma(hi, 65536, lo) rounds the exact integer sum
// once: the value of (float)(double)(int64 sum).
__device__ __forceinline__ void rs_wave_sum2(int v0, int v1, float& s0, float& s1)
{
    const auto sw = __builtin_amdgcn_permlane32_swap((unsigned)v0, (unsigned)v1, false, false);
    int x = (int)sw[0] + (int)sw[1];
    x += agt_dpp_i32<0xB1>(x);
    x += agt_dpp_i32<0x4E>(x);
    int xl = x & 0xffff, xh = x >> 16;
    xl += agt_dpp_i32<0x141>(xl); xh += agt_dpp_i32<0x141>(xh);
    xl += agt_dpp_i32<0x140>(xl); xh += agt_dpp_i32<0x140>(xh);
    const int lo0 = __builtin_amdgcn_readlane(xl, 0) + __builtin_amdgcn_readlane(xl, 16), lo1 = __builtin_amdgcn_readlane(xl, 32) + __builtin_amdgcn_readlane(xl, 48);
    const int hi0 = __builtin_amdgcn_readlane(xh, 0) + __builtin_amdgcn_readlane(xh, 16), hi1 = __builtin_amdgcn_readlane(xh, 32) + __builtin_amdgcn_readlane(xh, 48);
    s0 = __builtin_fmaf((float)hi0, 65536.f, (float)lo0);
    s1 = __builtin_fmaf((float)hi1, 65536.f, (float)lo1);
}

__device__ __forceinline__ const uint8_t* rs_uniform_ptr(const uint8_t* p)
{
    const unsigned long long v = (unsigned long long)p;
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)v), hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(v >> 32));
    return (const uint8_t*)(((unsigned long long)hi << 32) | lo);
}

// the one-wave shape's grid of un-rounded interpolations in LDS: 23 rows (window rows + 2) of 24 int32 (23 columns + pad)
constexpr int RS_BROWS = 23, RS_BP = 24, RS_B_BYTES = RS_BROWS * RS_BP * 4;

// LDS bytes of one corner: the level tiles only (no derivative tile, no reduction slots)
__host__ __device__ constexpr size_t lk_rs_lds_bytes(int levels) { return (size_t)levels * LkCfg<21, 1>::LEVEL_LDS; }

// true when every derivative position the 21x21 window of `pt` touches lies inside the image at every level
// (NLEV > 0: the trips are unrolled and predicated -- straight-line scalar code at the head of every corner's critical path)
template <int NLEV = 0>
__device__ __forceinline__ bool rs_interior(float ppx, float ppy, int max_level, int w0, int h0)
{
    bool ok = lk_pt_ok(ppx, ppy);              // (a NaN converts to 0, "inside": agt_lk_body.h lk_pt_ok)
    int w = w0, h = h0;
    if constexpr (NLEV > 0) {
#pragma unroll
        for (int l = 0; l < NLEV; l++) {
            const float scale = lk_level_scale(l);
            const int ipx = (int)floorf(ppx * scale - 10.f), ipy = (int)floorf(ppy * scale - 10.f);
            ok = ok && (l > max_level || (ipx >= 0 && ipx + 21 < w && ipy >= 0 && ipy + 21 < h));
            w = (w + 1) / 2; h = (h + 1) / 2;
        }
        return ok;
    }
    for (int l = 0; l <= max_level; l++) {
        const float scale = lk_level_scale(l);
        const int ipx = (int)floorf(ppx * scale - 10.f), ipy = (int)floorf(ppy * scale - 10.f);
        ok = ok && ipx >= 0 && ipx + 21 < w && ipy >= 0 && ipy + 21 < h;
        w = (w + 1) / 2; h = (h + 1) / 2;
    }
    return ok;
}

// the number of FINE levels 0 .. k - 1 at which every derivative position of the window lies inside the image (0: none, max_level + 1:
// all = rs_interior).  A window that leaves the image at some level leaves it at every coarser one (it covers twice the ground there),
// so the levels split into a coarse run for the general body and a fine run for the row-segment body (agt_lk.hip lk_kernel).
template <int NLEV>
__device__ __forceinline__ int rs_interior_levels(float ppx, float ppy, int max_level, int w0, int h0)
{
    if (!lk_pt_ok(ppx, ppy)) return 0;
    int w = w0, h = h0, n = 0;
    bool run = true;
#pragma unroll
    for (int l = 0; l < NLEV; l++) {
        const float scale = lk_level_scale(l);
        const int ipx = (int)floorf(ppx * scale - 10.f), ipy = (int)floorf(ppy * scale - 10.f);
        run = run && l <= max_level && ipx >= 0 && ipx + 21 < w && ipy >= 0 && ipy + 21 < h;
        n += run ? 1 : 0;
        w = (w + 1) / 2; h = (h + 1) / 2;
    }
    return n;
}

// lane -> (window row, segment) maps of the two shapes: one wave per corner (throughput: 7 consecutive pixels per lane,
// 63 lanes) and four waves per corner (latency: 2 pixels per lane, 231 lanes; the last segment's second pixel is column 21,
// outside the window, and is masked)
template <int NW>
struct RsCfg {
    static constexpr int PX = NW == 1 ? 7 : 2;              // window pixels per lane
    static constexpr int SEG = NW == 1 ? 3 : 11;            // segments per window row
    static constexpr int NLANE = 21 * SEG;                  // active lanes
    static constexpr int NB = PX + 2;                       // columns of the lane's B grid
    static constexpr int IDW = (3 + NB + 1 + 3) / 4;        // aligned dwords per I-tile row that cover NB + 1 bytes at any shift
    static constexpr int JDW = (3 + PX + 1 + 3) / 4;        // ... per J-tile row (PX + 1 bytes)
    static_assert(NW == 1 || NW == 4, "shapes built: 1 and 4 waves per corner");
};

// Track one corner through one frame with NW waves (all 64 * NW threads call).  Preconditions checked by the caller:
// rs_interior(...) holds and the corner's previous status is 1.  lds: lk_lds_bytes<21, NW>, 16-B aligned.
// level_top >= 0 (round 6): the corner's coarse levels max_level .. level_top + 1 were tracked by the general body (their windows touch the
// image border); this call carries the position it reached -- cx, cy, at the scale of level level_top + 1 -- through levels level_top .. 0.
template <int NW, int NLEV, typename PP>
__device__ __forceinline__ void lk_body_rs(PP P, int pt, int b, uint8_t* lds, const LkFrameIo<NLEV>& io, float ppx, float ppy,
                                           float& ox, float& oy, int& ost, const int level_top = -1, const float cx = 0.f, const float cy = 0.f)
{
    const bool cont = level_top >= 0;
    const int top = cont ? level_top : P->max_level;
    constexpr int WIN = 21;
    using C = LkCfg<WIN, NW>;
    using R = RsCfg<NW>;
    constexpr int T = AGT_WAVE * NW;
    constexpr int PX = R::PX, NB = R::NB;
    // (the thread index through an empty asm: nothing this body derives from it is shared with the general body that may have run the
    // coarse levels before it -- shared, those per-lane values stay alive across the whole general body and spill at 128 registers)
    int tid_raw = (int)threadIdx.x;
    asm volatile("" : "+v"(tid_raw));
    const int tid = NW == 1 ? (tid_raw & (AGT_WAVE - 1)) : tid_raw;
    const int lane = tid & (AGT_WAVE - 1), wave = tid / AGT_WAVE;
    const bool act = tid < R::NLANE;
    const int rr = act ? tid / R::SEG : 20, ss = act ? tid - rr * R::SEG : R::SEG - 1;     // window row, segment
    const int x0s = PX * ss;                                                               // first window column of the lane
    const int lane_off = __mul24(rr, C::JP) + x0s;                                         // ... and its byte offset from the window's origin in a search tile
    const long pidx = (long)b * P->n + pt;
    const float halfw = (WIN - 1) * 0.5f;
    const float FLT_SCALE = 1.f / (1 << 20);
    const float eps2_lo = (float)(P->eps2 * (1.0 - 1e-6)), eps2_hi = (float)(P->eps2 * (1.0 + 1e-6));
    long long* slots = reinterpret_cast<long long*>(lds + (P->max_level + 1) * C::LEVEL_LDS + ((C::DW * C::DW + 3) & ~3) * sizeof(int));
    if (NW == 4 && tid < 24) reinterpret_cast<int*>(slots)[tid] = 0;       // block_sum_exact's accumulators (a barrier precedes the first sum)
    int phase = 0;
    int nit = 0;                    // iterations over all levels (only kept where io.iters_out is set)

    float outx = cx, outy = cy;
    if (!cont && (P->flags & AGT_LK_USE_INITIAL_FLOW)) {
        outx = io.next_pts[pidx * 2]; outy = io.next_pts[pidx * 2 + 1];
        if (!agt_uniform((int)lk_pt_ok(outx, outy))) {        // a wild initial flow: lost, position carried (agt_lk_body.h lk_pt_ok)
            if (tid == 0) lk_publish(io, pidx, b, outx, outy, 0, 0.f);
            ox = outx; oy = outy; ost = 0;
            return;
        }
    }
    // where the search is expected to start, in level-0 coordinates (the search tiles are requested around it)
    const float up = __int_as_float((127 + top + 1) << 23);          // 2^(level_top + 1)
    const float gsx = cont ? cx * up : ((P->flags & AGT_LK_USE_INITIAL_FLOW) ? outx : ppx);
    const float gsy = cont ? cy * up : ((P->flags & AGT_LK_USE_INITIAL_FLOW) ? outy : ppy);

    // exact sums over the corner's lanes of two / three partials, rounded once to float
    auto sum2 = [&](int v0, int v1, float& s0, float& s1) {
        if constexpr (NW == 1) rs_wave_sum2(v0, v1, s0, s1);
        else {
            const int v[2] = { v0, v1 };
            long long t[2];
            block_sum_exact<NW, C::SUM_STEPS, 2, C::PAIR_OK>(v, t, slots, phase, wave, lane);
            s0 = (float)(double)t[0]; s1 = (float)(double)t[1];
        }
    };

    STAMP(0);
    // ---- prologue: request every level's tiles before touching any of them.
    // Fast path (the tile's dword-aligned footprint lies inside the image): buffer loads -- lane (row, dword) offsets
    // computed once per level, the tile origin rides in the scalar offset, no per-dword address or border arithmetic
    // (the general loader spends ~25 VALU per dword on indices, reflection and 64-bit addresses).
    {
        const int irow = tid / C::INDW, idw = tid - irow * C::INDW;
        const int jrow = tid / C::JNDW, jdw = tid - jrow * C::JNDW;
        constexpr int IR = T / C::INDW, JR = T / C::JNDW;                        // rows per load (9 / 5 with one wave)
        constexpr int IK = (C::IW + IR - 1) / IR, JK = (C::JT + JR - 1) / JR;    // loads per tile (3 / 8 with one wave, 1 / 2 with four)
        const bool ion = irow < IR, jon = jrow < JR;
        uint32_t fi[NLEV][IK], fj[NLEV][JK];
        bool fastI[NLEV], fastJ[NLEV];
#pragma unroll
        for (int l = 0; l < NLEV; l++) {
            fastI[l] = fastJ[l] = false;
            if (l <= top) {
                const float scale = lk_level_scale(l);
                const int ipx = (int)floorf(ppx * scale - halfw), ipy = (int)floorf(ppy * scale - halfw);
                const int jx0 = (int)floorf(gsx * scale - halfw) - C::MARGIN, jy0 = (int)floorf(gsy * scale - halfw) - C::MARGIN;
                AgtLevel LI = get_level(P->prev[l]);
                AgtLevel LJ = get_level(P->next[l]);
                // (frames 2.. of a group read their image pointers from an LDS table: uniform, but only provably so after this)
                if (io.grouped) { LI.ptr = rs_uniform_ptr(io.imgI[l]); LJ.ptr = rs_uniform_ptr(io.imgJ[l]); }
                {
                    const int ax0 = agt_uniform((ipx - 1) & ~3), ty0 = agt_uniform(ipy - 1);
                    fastI[l] = ax0 >= 0 && ax0 + 4 * C::INDW <= LI.w && ty0 >= 0 && ty0 + C::IW <= LI.h;
                    if (fastI[l]) {
                        const int pitch = (int)LI.pitch;
                        const auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(LI.ptr + (long)b * LI.bstride), 0, LI.h * pitch, 0x00020000);
                        const int vo = __mul24(irow, pitch) + 4 * idw;
#pragma unroll
                        for (int k = 0; k < IK; k++) {
                            fi[l][k] = 0;
                            if (ion && (k * IR + irow) < C::IW)
                                fi[l][k] = __builtin_amdgcn_raw_buffer_load_b32(rs, vo, (ty0 + k * IR) * pitch + ax0, 0);
                        }
                    } else {
                        // (a tile that touches the image border -- rare: loaded by the general loader and stored at once; kept in registers
                        // until the common store phase, these dwords -- up to 30 per lane over three levels, beside the 33 of the fast path -- were
                        // what the register allocator spilled first: round 6)
                        uint32_t ti[C::ILD];
                        tile_load<C::IW, C::INDW, T>(LI.ptr + (long)b * LI.bstride, LI.w, LI.h, LI.pitch, ipx - 1, ipy - 1, tid, ti);
                        tile_store<C::IW, C::INDW, T>(lds + l * C::LEVEL_LDS, tid, ti);
                    }
                }
                {
                    const int ax0 = agt_uniform(jx0 & ~3), ty0 = agt_uniform(jy0);
                    fastJ[l] = ax0 >= 0 && ax0 + 4 * C::JNDW <= LJ.w && ty0 >= 0 && ty0 + C::JT <= LJ.h;
                    if (fastJ[l]) {
                        const int pitch = (int)LJ.pitch;
                        const auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(LJ.ptr + (long)b * LJ.bstride), 0, LJ.h * pitch, 0x00020000);
                        const int vo = __mul24(jrow, pitch) + 4 * jdw;
#pragma unroll
                        for (int k = 0; k < JK; k++) {
                            fj[l][k] = 0;
                            if (jon && (k * JR + jrow) < C::JT)
                                fj[l][k] = __builtin_amdgcn_raw_buffer_load_b32(rs, vo, (ty0 + k * JR) * pitch + ax0, 0);
                        }
                    } else {
                        uint32_t tj[C::JLD];
                        tile_load<C::JT, C::JNDW, T>(LJ.ptr + (long)b * LJ.bstride, LJ.w, LJ.h, LJ.pitch, jx0, jy0, tid, tj);
                        tile_store<C::JT, C::JNDW, T>(lds + l * C::LEVEL_LDS + C::IW * C::IP, tid, tj);
                    }
                }
            }
        }
        STAMP(1);
#pragma unroll
        for (int l = 0; l < NLEV; l++) {
            if (l <= top) {
                uint8_t* sIl = lds + l * C::LEVEL_LDS;
                uint8_t* sJl = sIl + C::IW * C::IP;
                if (fastI[l]) {
#pragma unroll
                    for (int k = 0; k < IK; k++)
                        if (ion && (k * IR + irow) < C::IW) *reinterpret_cast<uint32_t*>(sIl + (k * IR + irow) * C::IP + 4 * idw) = fi[l][k];
                }
                if (fastJ[l]) {
#pragma unroll
                    for (int k = 0; k < JK; k++)
                        if (jon && (k * JR + jrow) < C::JT) *reinterpret_cast<uint32_t*>(sJl + (k * JR + jrow) * C::JP + 4 * jdw) = fj[l][k];
                }
            }
        }
    }
    block_sync<NW>();
    STAMP(2);

    int st = 1;
    float errv = 0.f;

    for (int level = top; level >= 0; level--) {
        STAMP(8 + level * 8 + 0);
        AgtLevel LJ = get_level(P->next[level]);
        if (io.grouped) {
            const uint8_t* q = io.imgJ[0];
#pragma unroll
            for (int l = 1; l < NLEV; l++) q = level == l ? io.imgJ[l] : q;
            LJ.ptr = q;
        }
        const uint8_t* imgJ = LJ.ptr + (long)b * LJ.bstride;
        const uint8_t* sI = lds + level * C::LEVEL_LDS;
        uint8_t* sJ = lds + level * C::LEVEL_LDS + C::IW * C::IP;
        const int sJ_lds = agt_uniform((int)(unsigned)(uintptr_t)(__attribute__((address_space(3))) uint8_t*)sJ);      // (the search tile's LDS address: a multiple of 4)
        const float scale = lk_level_scale(level);
        float prevx = ppx * scale, prevy = ppy * scale;
        float nextx, nexty;
        if (!cont && level == P->max_level) {
            if (P->flags & AGT_LK_USE_INITIAL_FLOW) { nextx = outx * scale; nexty = outy * scale; }
            else { nextx = prevx; nexty = prevy; }
        } else { nextx = outx * 2.f; nexty = outy * 2.f; }
        outx = nextx; outy = nexty;

        prevx -= halfw; prevy -= halfw;
        const int ipx = agt_uniform((int)floorf(prevx)), ipy = agt_uniform((int)floorf(prevy));
        // (interior by precondition: no bounds test of the I window)
        uint32_t W01, W23;                         // the level's I-side weights, then every iteration's J-side weights
        rs_weights_i16(prevx - (float)ipx, prevy - (float)ipy, W01, W23);

        // ---- I side in registers: B = un-rounded bilinear interpolation on the lane's 3 x NB grid (rows rr .. rr+2 of the
        // B grid = tile rows rr .. rr+3, columns x0s .. x0s + NB of the tile shifted by the alignment offset)
        int Iv[PX], Ix[PX], Iy[PX];
        {
            const int offI = (ipx - 1) - ((ipx - 1) & ~3);
            int Bv[3][NB];
            if constexpr (NW == 1) {
                // Round 4: every B value is interpolated ONCE.  The 23 x 23 grid of un-rounded interpolations the window's Scharr
                // taps touch is split over 46 lanes -- lane l = (row l >> 1, columns 12 (l & 1) .. + 11): two tile rows as four
                // aligned dwords each, 12 packed taps -- written to LDS as int32 (behind the tiles, where the general body keeps
                // its derivative tile) and read back by the lanes as their 3 x 9 neighbourhoods; the corner is one wave, LDS
                // operations of a wave execute in order, no barrier.  (Rounds 2 / 3: every lane interpolated its own 3 x 9 patch
                // -- 1,701 interpolations for 529 values, 132 vector instructions per lane and level against ~70 now.)
                constexpr int BP = RS_BP;
                int* sB = reinterpret_cast<int*>(lds + (P->max_level + 1) * C::LEVEL_LDS);
                block_sync<NW>();                               // (the previous level's readers are done with sB)
                {
                    const bool on = lane < 2 * RS_BROWS;
                    const int br = on ? lane >> 1 : RS_BROWS - 1, seg = lane & 1;
                    const int c0 = offI + 12 * seg, sh = c0 & 3;
                    uint32_t ea[4], eb[4];                      // bytes 0 .. 15 from column c0 of tile rows br, br + 1 (byte 12 is the last one used)
                    {
                        const uint32_t* pa = reinterpret_cast<const uint32_t*>(sI + __mul24(br, C::IP) + (c0 & ~3));
                        const uint32_t* pb = reinterpret_cast<const uint32_t*>(sI + __mul24(br + 1, C::IP) + (c0 & ~3));
                        uint32_t qa[4], qb[4];
#pragma unroll
                        for (int d = 0; d < 4; d++) { qa[d] = pa[d]; qb[d] = pb[d]; }
#pragma unroll
                        for (int d = 0; d < 3; d++) { ea[d] = __builtin_amdgcn_alignbyte(qa[d + 1], qa[d], sh); eb[d] = __builtin_amdgcn_alignbyte(qb[d + 1], qb[d], sh); }
                        ea[3] = __builtin_amdgcn_alignbyte(0u, qa[3], sh); eb[3] = __builtin_amdgcn_alignbyte(0u, qb[3], sh);
                    }
                    uint32_t QA[12], QB[12];
                    rs_row_pairs<12, 4>(ea, QA); rs_row_pairs<12, 4>(eb, QB);
                    int Bw[12];
#pragma unroll
                    for (int k = 0; k < 12; k++) Bw[k] = rs_tap2<false>(QA[k], QB[k], W01, W23);
                    if (on) {
                        int4* o = reinterpret_cast<int4*>(sB + br * BP + 12 * seg);
                        o[0] = make_int4(Bw[0], Bw[1], Bw[2], Bw[3]); o[1] = make_int4(Bw[4], Bw[5], Bw[6], Bw[7]); o[2] = make_int4(Bw[8], Bw[9], Bw[10], Bw[11]);
                    }
                }
                block_sync<NW>();
#pragma unroll
                for (int i = 0; i < 3; i++) {
                    const int* r = sB + (rr + i) * BP + x0s;
#pragma unroll
                    for (int k = 0; k < NB; k++) Bv[i][k] = r[k];
                }
            } else {
            const int c0 = offI + x0s, sh = c0 & 3;
            uint32_t e[4][3];                                   // four tile rows, bytes 0..11 from column c0 on
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const uint32_t* p = reinterpret_cast<const uint32_t*>(sI + __mul24(rr + i, C::IP) + (c0 & ~3));
                uint32_t q[4] = { 0, 0, 0, 0 };
#pragma unroll
                for (int d = 0; d < R::IDW; d++) q[d] = p[d];
                e[i][0] = __builtin_amdgcn_alignbyte(q[1], q[0], sh);
                e[i][1] = __builtin_amdgcn_alignbyte(q[2], q[1], sh);
                e[i][2] = __builtin_amdgcn_alignbyte(q[3], q[2], sh);
            }
            uint32_t Q[4][NB];
#pragma unroll
            for (int i = 0; i < 4; i++) rs_row_pairs<NB, 3>(e[i], Q[i]);
#pragma unroll
            for (int i = 0; i < 3; i++) {
#pragma unroll
                for (int k = 0; k < NB; k++) Bv[i][k] = rs_tap2<false>(Q[i][k], Q[i + 1][k], W01, W23);
            }
            }
            int Cs[NB], Es[NB];                                 // vertical Scharr halves per column
            // (3 x as shift-add, 10 x as a 24-bit multiply: B < 2^22, so every operand fits; v_mul_lo_u32 is quarter rate)
#pragma unroll
            for (int k = 0; k < NB; k++) {
                const int t = Bv[0][k] + Bv[2][k];
                Cs[k] = ((t << 1) + t) + __mul24(10, Bv[1][k]);
                Es[k] = Bv[2][k] - Bv[0][k];
            }
#pragma unroll
            for (int k = 0; k < PX; k++) {
                const int t = Es[k] + Es[k + 2];
                Iv[k] = descale(Bv[1][k + 1], W_BITS - 5);
                Ix[k] = descale(Cs[k + 2] - Cs[k], W_BITS);
                Iy[k] = descale(((t << 1) + t) + __mul24(10, Es[k + 1]), W_BITS);
            }
            if constexpr (NW != 1) {                            // the pixel of column 21 (last segment) is not in the window
                const bool in1 = x0s + 1 < WIN;
                Ix[PX - 1] = in1 ? Ix[PX - 1] : 0; Iy[PX - 1] = in1 ? Iy[PX - 1] : 0;
            }
        }
        STAMP(8 + level * 8 + 1);
        // (idle lanes duplicate the last active one; their derivatives are zeroed HERE, once per level, so that every product they
        // enter -- the level's three, each iteration's two -- is zero without a select of its own: round 6)
#pragma unroll
        for (int k = 0; k < PX; k++) { Ix[k] = act ? Ix[k] : 0; Iy[k] = act ? Iy[k] : 0; }
        int a11 = 0, a12 = 0, a22 = 0;
#pragma unroll
        for (int k = 0; k < PX; k++) { a11 += __mul24(Ix[k], Ix[k]); a12 += __mul24(Ix[k], Iy[k]); a22 += __mul24(Iy[k], Iy[k]); }
        float A11, A12, A22;
        if constexpr (NW == 1) {
            float sdummy;
            rs_wave_sum2(a11, a12, A11, A12);
            rs_wave_sum2(a22, 0, A22, sdummy);
        } else {
            const int v[3] = { a11, a12, a22 };
            long long t[3];
            block_sum_exact<NW, C::SUM_STEPS, 3, C::PAIR_OK>(v, t, slots, phase, wave, lane);
            A11 = (float)(double)t[0]; A12 = (float)(double)t[1]; A22 = (float)(double)t[2];
        }
        A11 *= FLT_SCALE; A12 *= FLT_SCALE; A22 *= FLT_SCALE;

        float D = A11 * A22 - A12 * A12;
        const float minEig = (A22 + A11 - sqrtf((A11 - A22) * (A11 - A22) + 4.f * A12 * A12)) / (float)(2 * WIN * WIN);
        if (P->flags & AGT_LK_GET_MIN_EIGENVALS) errv = minEig;
        if (agt_uniform((int)((double)minEig < P->min_eig_threshold || D < FLT_EPSILON))) {
            if (level == 0) st = 0;
            continue;
        }
        D = 1.f / D;
        // (b1, b2 are scaled by 2^-20 before they enter the 2 x 2 solve: a power of two commutes with every rounding on the way to the
        // step -- (A12 * (s2 * 2^-20) - A22 * (s1 * 2^-20)) * D = (A12 * s2 - A22 * s1) * (D * 2^-20) bit for bit (D * 2^-20 is exact and
        // far from the denormals: D <= 1 / FLT_EPSILON) -- so the scale rides in D: one multiply less per iteration)
        const float Ds = D * FLT_SCALE;
        STAMP(8 + level * 8 + 2);

        nextx -= halfw; nexty -= halfw;
        float pdx = 0.f, pdy = 0.f;
        // (the position a level hands on is formed once, behind the loop: nextPts = nextPt + halfWin of the last step taken -- and, after an
        // oscillation, minus half that step -- exactly OpenCV's two assignments; a level that takes no step hands on what it was given)
        int moved = 0, half_back = 0;
        float hbx = 0.f, hby = 0.f;
        int jx0 = agt_uniform((int)floorf(gsx * scale - halfw)) - C::MARGIN, jy0 = agt_uniform((int)floorf(gsy * scale - halfw)) - C::MARGIN;      // (scalar: the tile origin enters every iteration's window address)
        auto restage_j = [&](int inx, int iny) {
            jx0 = inx - C::MARGIN; jy0 = iny - C::MARGIN;
            uint32_t t[C::JLD];
            block_sync<NW>();
            tile_load<C::JT, C::JNDW, T>(imgJ, LJ.w, LJ.h, LJ.pitch, jx0, jy0, tid, t);
            tile_store<C::JT, C::JNDW, T>(sJ, tid, t);
            block_sync<NW>();
        };
        // IvR[k] = 256 - (Iv[k] << 9): floor((raw + 256 - 512 Iv) / 512) = floor((raw + 256) / 512) - Iv, so the subtraction of
        // the patch value rides in the accumulator input of the low dot4 (mod 2^32) and the arithmetic shift yields the difference
        int IvR[PX];
#pragma unroll
        for (int k = 0; k < PX; k++) IvR[k] = (1 << (W_BITS - 5 - 1)) - (Iv[k] << (W_BITS - 5));
        // the lane's temporal differences J - I (values * 32) at window position (inx, iny), weights in W01 / W23
        // (requesting the six dwords BEFORE the weights are computed, as the four-wave bodies do, measured slower here: 41.5-41.7
        // against 41.0-41.1 us per 64-stream step)
        auto window_taps = [&](int inx, int iny, int (&Jv)[PX]) {
            // (the window's origin inside the tile is wave-uniform; the lane's own offset is a constant)
            // (window origin: scalar arithmetic on scalar operands -- one vector add per iteration brings in the lane's part; the tile's
            // base is a multiple of 4, so the byte shift is that of the offset alone)
            typedef const __attribute__((address_space(3))) uint32_t* L32;
            const int a = rs_pin((iny - jy0) * C::JP + (inx - (jx0 & ~3)) + sJ_lds) + lane_off;
            const int shj = a & 3;
            const L32 p0 = (L32)(uintptr_t)(unsigned)(a & ~3);
            const L32 p1 = (L32)(uintptr_t)(unsigned)((a & ~3) + C::JP);
            uint32_t d[3] = { 0, 0, 0 }, f[3] = { 0, 0, 0 };
#pragma unroll
            for (int i = 0; i < R::JDW; i++) { d[i] = p0[i]; f[i] = p1[i]; }
            const uint32_t ra[2] = { __builtin_amdgcn_alignbyte(d[1], d[0], shj), __builtin_amdgcn_alignbyte(d[2], d[1], shj) };
            const uint32_t rb[2] = { __builtin_amdgcn_alignbyte(f[1], f[0], shj), __builtin_amdgcn_alignbyte(f[2], f[1], shj) };
            uint32_t QA[PX], QB[PX];
            rs_row_pairs<PX, 2>(ra, QA); rs_row_pairs<PX, 2>(rb, QB);
#pragma unroll
            for (int k = 0; k < PX; k++) Jv[k] = rs_tap2(QA[k], QB[k], W01, W23, IvR[k]) >> (W_BITS - 5);
        };
        // While the window's corner stays in this box it is inside the image band and inside the search tile: one float test per
        // iteration instead of the two integer ones (which remain, word for word, behind it); and ONE vector -> scalar decision
        // per iteration: converged / oscillating / FP64 tie-break needed / next position outside the box, folded into one code.
        float bx0, bx1, by0, by1;
        auto set_box = [&]() {
            const int lx = jx0 > -WIN ? jx0 : -WIN, hx = (jx0 + C::JT - WIN - 1 < LJ.w - 1 ? jx0 + C::JT - WIN - 1 : LJ.w - 1) + 1;
            const int ly = jy0 > -WIN ? jy0 : -WIN, hy = (jy0 + C::JT - WIN - 1 < LJ.h - 1 ? jy0 + C::JT - WIN - 1 : LJ.h - 1) + 1;
            bx0 = rs_scalar((float)lx); bx1 = rs_scalar((float)hx); by0 = rs_scalar((float)ly); by1 = rs_scalar((float)hy);     // (scalar operands of the compares; no loop-carried vector copies)
        };
        set_box();
        int slow = agt_uniform((int)!(nextx >= bx0 && nextx < bx1 && nexty >= by0 && nexty < by1));
        for (int j = 0; j < P->max_count; j++) {
            nit++;
            // (the floor as a float feeds the fractions, as an integer the address: nextx - (float)(int)floorf(nextx) = nextx - floorf(nextx)
            // exactly at image coordinates)
            const float flx = floorf(nextx), fly = floorf(nexty);
            const int inx = agt_uniform((int)flx), iny = agt_uniform((int)fly);
            if (slow) {
                if (inx < -WIN || inx >= LJ.w || iny < -WIN || iny >= LJ.h) {
                    if (level == 0) st = 0;
                    break;
                }
                if (inx < jx0 || inx + WIN >= jx0 + C::JT || iny < jy0 || iny + WIN >= jy0 + C::JT) { restage_j(inx, iny); set_box(); }
            }
            rs_weights_i16(nextx - flx, nexty - fly, W01, W23);
            int Jv[PX];
            window_taps(inx, iny, Jv);
            int b1 = 0, b2 = 0;
#pragma unroll
            for (int k = 0; k < PX; k++) { b1 = rs_mad24(Jv[k], Ix[k], b1); b2 = rs_mad24(Jv[k], Iy[k], b2); }
            float sb1, sb2;
            sum2(b1, b2, sb1, sb2);
            const float dx = (A12 * sb2 - A22 * sb1) * Ds;
            const float dy = (A12 * sb1 - A11 * sb2) * Ds;
            nextx += dx; nexty += dy;
            moved = 1;
            if (j == 0) STAMP(8 + level * 8 + 3);
#ifdef AGT_LK_STAMPS
            if (pidx == 0 && threadIdx.x == 0) agt_lk_stamps[8 + level * 8 + 6] = j + 1;
#endif
            // OpenCV tests  (double)dx*dx + (double)dy*dy <= eps^2  in FP64; the float sum is within 2e-7 of it, so the
            // FP64 evaluation is only needed inside a 1e-6 band around the threshold (code 4: wave-uniform, rare);
            // fabs((double)f) < 0.01  <=>  fabsf(f) <= 0.01f  (0.01f is the largest float below 0.01)
            const float d2 = dx * dx + dy * dy;
            const bool osc = j > 0 && fabsf(dx + pdx) <= 0.01f && fabsf(dy + pdy) <= 0.01f;
            const bool out_of_box = !(nextx >= bx0 && nextx < bx1 && nexty >= by0 && nexty < by1);
            int code = out_of_box ? 3 : 0;
            code = osc ? 2 : code;
            code = d2 < eps2_lo ? 1 : code;
            code = (!(d2 < eps2_lo) && !(d2 > eps2_hi)) ? 4 : code;
            code = agt_uniform(code);
            if (code == 4) {
                const bool conv = (double)dx * dx + (double)dy * dy <= P->eps2;
                code = agt_uniform(conv ? 1 : (osc ? 2 : (out_of_box ? 3 : 0)));
            }
            if (code == 1) break;
            if (code == 2) { half_back = 1; hbx = dx * 0.5f; hby = dy * 0.5f; break; }
            slow = code == 3;
            pdx = dx; pdy = dy;
        }

        if (moved) {
            outx = nextx + halfw; outy = nexty + halfw;
            if (half_back) { outx -= hbx; outy -= hby; }
        }
        STAMP(8 + level * 8 + 4);
        if (st && io.err && level == 0 && !(P->flags & AGT_LK_GET_MIN_EIGENVALS)) {
            const float npx = outx - halfw, npy = outy - halfw;
            const int inx = agt_uniform((int)floorf(npx)), iny = agt_uniform((int)floorf(npy));
            if (inx < -WIN || inx >= LJ.w || iny < -WIN || iny >= LJ.h) { st = 0; continue; }
            if (inx < jx0 || inx + WIN >= jx0 + C::JT || iny < jy0 || iny + WIN >= jy0 + C::JT) restage_j(inx, iny);
            rs_weights_i16(npx - (float)inx, npy - (float)iny, W01, W23);
            int Jv[PX];
            window_taps(inx, iny, Jv);
            int e = 0;
#pragma unroll
            for (int k = 0; k < PX; k++) {
                const int diff = Jv[k], ad = diff < 0 ? -diff : diff;
                e += (NW == 1 || x0s + k < WIN) ? ad : 0;
            }
            e = act ? e : 0;
            float se, sdum;
            sum2(e, 0, se, sdum);
            errv = se * 1.f / (float)(32 * WIN * WIN);
        }
    }

    STAMP(3);
    if (tid == 0) lk_publish(io, pidx, b, outx, outy, st, errv);
    if (tid == 0 && io.iters_out) io.iters_out[pidx] = (uint8_t)(nit > 255 ? 255 : nit);
#ifdef AGT_LK_STAMPS
    if (tid == 0 && !io.grouped && pidx < AGT_LK_CORNER_LOG) agt_lk_corner_log[pidx][2] = (unsigned long long)nit;
#endif
    ox = outx; oy = outy; ost = st;
}

}  // namespace agt_lk
