// agt_lk_chain_body.h -- the LK role of the fused step when one corner has a whole workgroup (four waves): ALL frames of a
// launch group are tracked by the same workgroup, and what one frame leaves behind is what the next one starts from.
//
// Same arithmetic as agt_lk_body.h (cv::calcOpticalFlowPyrLK, bit-identical to oracle/cv_lk.c), different schedule.  One
// frame of the general body is a serial chain of ~16 us: tile loads (3 us), then per level Scharr -> patch + covariance sums
// (1.35 us, three barriers) and the iterations (~1 us each).  Here:
//   * the previous image's tiles are never loaded twice: frame k+1 differentiates image k, and image k is already in LDS
//     as frame k's 40 x 40 search tiles -- the 24 x 24 neighbourhood of the tracked position lies inside them unless the
//     corner moved more than ~8 px at that level (then the tile is re-staged).  Two tile buffers alternate;
//   * the NEXT frame's search tiles are requested when the last level of THIS frame starts iterating, centred on the
//     estimate of that moment (the 9 px margin absorbs the last corrections): the HBM latency runs under the iterations;
//   * the image-(k-1) side of all levels (Scharr, interpolated patch, covariance sums, their inverse) does not depend on the
//     flow, so it is evaluated for every level at once, as straight-line code: the partial sums are transposed through LDS
//     and wave w finishes level w -- three barriers per frame instead of nine;
//   * the level loop is unrolled (per-level quantities are plain registers), the iteration makes one vector -> scalar decision,
//     its sums cross the waves as int32 hi / lo pairs;
//   * the result is stored at once (device-scope stores when the PnP role of the same launch waits for it) and counted into
//     the arrival counter one frame later, under the next frame's image-side work.
// Measured on production code with fixed iteration counts: 5.8 us per frame + 0.62 us per iteration (profiles/r02_summary.md).
// Only what the tracker uses is covered: flags == 0, no error output (lk_role keeps lk_body otherwise); the frames of a group
// share their geometry, the image before the group may differ in pitch.  A one-frame group is fine: its image-(k-1) tiles are
// loaded on demand like the first frame's of any group.
#pragma once
#include "agt_lk_body.h"

// Timeline of corner 0 over the first four frames of a launch (diagnostic build only: -DAGT_STEP_STAMPS; tools/chainstamps.py)
#ifdef AGT_STEP_STAMPS
__device__ unsigned long long agt_chain_stamps[4 * 16];
__device__ unsigned agt_chain_counts[8];      // corner-frames | with a previous-image tile reload | with a search tile load | re-stages | iterations
#define CCOUNT(i, n) do { if (threadIdx.x == 0) atomicAdd(&agt_chain_counts[i], (unsigned)(n)); } while (0)
#define CSTAMP(i) do { if (pidx == 0 && threadIdx.x == 0 && k < 4) agt_chain_stamps[k * 16 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
extern "C" int agt_debug_chain_stamps(unsigned long long* host64)
{
    return (int)hipMemcpyFromSymbol(host64, HIP_SYMBOL(agt_chain_stamps), sizeof(agt_chain_stamps));
}
extern "C" int agt_debug_chain_counts(unsigned* host8)
{
    return (int)hipMemcpyFromSymbol(host8, HIP_SYMBOL(agt_chain_counts), sizeof(agt_chain_counts));
}
#else
#define CSTAMP(i)
#define CCOUNT(i, n)
#endif

// Diagnostic build only (-DAGT_CHAIN_REPS): run a section of the per-frame work n times (results unchanged) to read its cost
// off the frame rate -- agt_chain_reps[0] Scharr pass, [1] patch + partial sums, [2] per-level reduction, [3] geometry, [4] next
// frame's tile requests, [5] hand-over tile stores, [6] result stores
#ifdef AGT_CHAIN_REPS
__device__ int agt_chain_reps[8] = { 1, 1, 1, 1, 1, 1, 1, 1 };
extern "C" int agt_debug_chain_reps(const int* host8) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(agt_chain_reps), host8, 32); }
#define CREPS(i) for (int rep_ = 0, n_ = *(volatile int*)&agt_chain_reps[i]; rep_ < n_; rep_++)
#else
#define CREPS(i)
#endif

namespace agt_lk {

template <int NLEV>
struct ChainCfg {
    using C = LkCfg<21, 4>;
    static constexpr int TILE = C::JT * C::JP;                          // one search tile (40 rows x 44 B)
    static constexpr int SD = ((C::DW * C::DW + 3) & ~3) * 4;           // one derivative tile (22 x 22 ints)
    static constexpr int OFF_SD = 2 * NLEV * TILE;
    static constexpr int OFF_SLOTS = OFF_SD + NLEV * SD;                // iteration sums: three rotating int4 accumulators (256 B reserved)
    static constexpr int OFF_RED = OFF_SLOTS + 2 * 4 * 4 * 8;           // covariance partials: [3 * NLEV][256 threads] int
    static constexpr int OFF_LVL = OFF_RED + 3 * NLEV * 256 * 4;        // per level: A11, A12, A22, 1 / det, usable (8 floats)
    static constexpr int BYTES = OFF_LVL + NLEV * 8 * 4;
    static_assert(TILE % 16 == 0 && SD % 16 == 0, "16-byte aligned LDS sections");
};

template <int NLEV>
__host__ __device__ constexpr size_t lk_chain_lds_bytes() { return (size_t)ChainCfg<NLEV>::BYTES; }

// LDS-only workgroup barrier: the plain __syncthreads() also waits for every outstanding GLOBAL load of the wave, which would
// put the prefetched tiles of the next frame back on the critical path of the first iteration after their request
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// 64-lane int32 sum (no overflow checks: callers bound the operands), wave-uniform result
__device__ __forceinline__ int wave_sum_i32(int v)
{
    v += agt_dpp_i32<0xB1>(v);
    v += agt_dpp_i32<0x4E>(v);
    v += agt_dpp_i32<0x141>(v);
    v += agt_dpp_i32<0x140>(v);
    return __builtin_amdgcn_readlane(v, 0) + __builtin_amdgcn_readlane(v, 16) + __builtin_amdgcn_readlane(v, 32) + __builtin_amdgcn_readlane(v, 48);
}

// two 64-lane int32 sums in one DPP chain (v_permlane32_swap: value 0 ends up in the lower half-wave, value 1 in the upper one)
__device__ __forceinline__ void wave_sum2_i32(int v0, int v1, int& t0, int& t1)
{
    const auto sw = __builtin_amdgcn_permlane32_swap((unsigned)v0, (unsigned)v1, false, false);
    int x = (int)sw[0] + (int)sw[1];
    x += agt_dpp_i32<0xB1>(x);
    x += agt_dpp_i32<0x4E>(x);
    x += agt_dpp_i32<0x141>(x);
    x += agt_dpp_i32<0x140>(x);
    t0 = __builtin_amdgcn_readlane(x, 0) + __builtin_amdgcn_readlane(x, 16);
    t1 = __builtin_amdgcn_readlane(x, 32) + __builtin_amdgcn_readlane(x, 48);
}

// The two mismatch sums of one iteration over the four waves, as the floats the 2 x 2 solve needs.  Exact: the partial sums are
// split x = 65536 * hi + lo before they could leave int32, the hi and lo parts are accumulated separately (|sum hi| < 2^18,
// sum lo < 2^20 over the workgroup), and fmaf(hi, 65536, lo) rounds the exact total once -- the same float as
// (float)(double)(int64 total) in block_sum_exact's caller.  v_permlane32_swap puts both sums in one DPP chain (value 0 in
// the lower half-wave, value 1 in the upper one).
__device__ __forceinline__ void iter_sum2(const int (&v)[2], float& f0, float& f1, int* slots, int& phase, int wave, int lane)
{
    // |v| < 2^26 per thread: the pair sum across the half-waves (2^27) and three DPP steps (x 8: < 2^30) still fit int32, so the
    // values travel whole that far (value 0 in the lower half-wave, value 1 in the upper one) and are split into 16-bit halves
    // only for the last DPP step and the cross-wave accumulation (lo < 2^20, |hi| < 2^18 over the workgroup) -- 11 instructions
    // instead of the 18 of two full half-chains
    const auto sw = __builtin_amdgcn_permlane32_swap((unsigned)v[0], (unsigned)v[1], false, false);
    int x = (int)sw[0] + (int)sw[1];
    x += agt_dpp_i32<0xB1>(x);
    x += agt_dpp_i32<0x4E>(x);
    x += agt_dpp_i32<0x141>(x);
    int xl = x & 0xffff, xh = x >> 16;
    xl += agt_dpp_i32<0x140>(xl); xh += agt_dpp_i32<0x140>(xh);
    // After the four DPP steps every lane of a 16-lane row holds its row's sum; rows 0 / 1 belong to value 0, rows 2 / 3 to value 1
    // (the swap).  The first lane of each row adds its row sum straight into the workgroup's accumulator with an LDS atomic
    // (integer: order-free, still exact): { lo0, hi0, lo1, hi1 }, three rotating slots -- the slot of iteration i + 2 is cleared
    // by thread 0 right after the barrier of iteration i, when its last readers (iteration i - 1) are past it.  No read-lanes,
    // no scalar adds, and after the barrier every wave reads ONE 16-byte word instead of four.
    // (round 2: read-lanes + ds_write_b128 per wave + four ds_read_b128 and twelve adds after the barrier; finishing the sums
    // with row_bcast DPP steps and a store from lanes 31 / 63 had measured 11 % slower than that)
    int* s = slots + phase * 4;
    if ((lane & 15) == 0) {
        __attribute__((address_space(3))) int* a = (__attribute__((address_space(3))) int*)(s + ((lane >> 5) << 1));
        __hip_atomic_fetch_add(a, xl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __hip_atomic_fetch_add(a + 1, xh, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    lds_barrier();
    const int4 t = *reinterpret_cast<const int4*>(s);
    const int nz = phase == 0 ? 2 : phase - 1;                     // (phase + 2) % 3
    if (threadIdx.x == 0) *reinterpret_cast<int4*>(slots + nz * 4) = make_int4(0, 0, 0, 0);
    f0 = fmaf((float)t.y, 65536.f, (float)t.x);
    f1 = fmaf((float)t.w, 65536.f, (float)t.z);
    phase = phase == 2 ? 0 : phase + 1;
}

// Request one 40 x 44 B tile (origin tx0, ty0) into registers.  A tile that lies inside the image (wave-uniform test; nearly
// always) costs one multiply-add and one global load per dword: (row, 4 * column) of this thread's dwords are computed once
// per kernel, the base address is scalar.  Otherwise the general, reflecting tile_load.
template <int N>
__device__ __forceinline__ void tile_request(const uint8_t* img, int w, int h, long pitch, int tx0, int ty0, int tid,
                                             const int (&tr)[N], const int (&tc)[N], const bool (&tv)[N], uint32_t (&v)[N])
{
    using C = LkCfg<21, 4>;
    const int ax0 = tx0 & ~3;
    if (agt_uniform((int)(ax0 >= 0 && ax0 + 4 * C::JNDW <= w && ty0 >= 0 && ty0 + C::JT <= h))) {
        typedef const __attribute__((address_space(1))) uint8_t* G8;
        typedef const __attribute__((address_space(1))) uint32_t* G32;
        const G8 base = (G8)img + (long)ty0 * pitch + ax0;
        const int p32 = (int)pitch;
#pragma unroll
        for (int k = 0; k < N; k++) {
            v[k] = 0;
            if (tv[k]) v[k] = *(G32)(base + (tr[k] * p32 + tc[k]));
        }
    } else {
        tile_load<C::JT, C::JNDW, C::T, N, true>(img, w, h, pitch, tx0, ty0, tid, v);
    }
}

// frame(k) -> LkFrameIo<NLEV> of frame k of the group (image pointers, outputs, arrival counter)
template <int NLEV, typename PP, typename FrameFn>
__device__ __forceinline__ void lk_frames_w4(PP P, int pt, int b, uint8_t* lds, int nf, FrameFn&& frame)
{
    constexpr int WIN = 21, NW = 4;
    using C = LkCfg<WIN, NW>;
    using K = ChainCfg<NLEV>;
    constexpr int T = C::T, JP = C::JP, JT = C::JT, DW = C::DW, MARGIN = C::MARGIN;
    constexpr int SRC_PAD = (JT - C::IW) / 2;                   // 8: the 24 x 24 neighbourhood centred in a 40 x 40 tile
    const int tid = (int)threadIdx.x, lane = tid & (AGT_WAVE - 1), wave = tid / AGT_WAVE;
    const long pidx = (long)b * P->n + pt;
    const int maxl = P->max_level;
    // (criteria read once: a scalar load from the kernel-argument segment inside the frame loop is a ~200-cycle wait)
    const int max_count = P->max_count;
    const double min_eig_threshold = P->min_eig_threshold, eps2 = P->eps2;
    int* sDall = reinterpret_cast<int*>(lds + K::OFF_SD);
    int* slots = reinterpret_cast<int*>(lds + K::OFF_SLOTS);      // three rotating int4 accumulators of the iteration sums (iter_sum2)
    if (threadIdx.x < 12) slots[threadIdx.x] = 0;                 // (visible long before the first iteration: three barriers come first)
    int* red = reinterpret_cast<int*>(lds + K::OFF_RED);
    float* lvl = reinterpret_cast<float*>(lds + K::OFF_LVL);
    int phase = 0;

    // this thread's window pixels / derivative positions as offsets into a 44-B-pitch tile resp. the 22-int-pitch derivative
    // tile (threads without a k-th element point at offset 0 and are masked arithmetically)
    // window pixels: thread t < 231 owns the horizontal pair (2s, y), (2s + 1, y), s = t mod 11, y = t / 11 (the eleventh
    // pair of a row has one pixel): the two pixels share two of their bilinear taps per row, in the image tiles (six
    // byte reads per pair instead of eight) as in the derivative tile (an 8-byte and a 4-byte read per row instead of four)
    static_assert(C::NPX == 2 && ((WIN + 1) / 2) * WIN <= T, "pairing of the window pixels");
    int oW[C::NPX], oD[C::NPX];
    bool pv[C::NPX];
    {
        constexpr int PR = (WIN + 1) / 2;
        const bool v = tid < PR * WIN;
        const int y = v ? tid / PR : 0, x = v ? 2 * (tid - y * PR) : 0;
        pv[0] = v; pv[1] = v && x + 1 < WIN;
        oW[0] = y * JP + x; oW[1] = oW[0] + 1;
        oD[0] = y * DW + x; oD[1] = oD[0] + 1;
    }
    // Scharr: one horizontal PAIR of derivative positions per thread and level (22 x 11 pairs = 242 threads): the 3 x 4
    // pixels under a pair are three unaligned dwords -- two aligned LDS reads + one v_alignbyte each -- and every tap sum is a
    // v_dot4_u32_u8 with constant weights (positive and negative parts apart: the pixels are unsigned)
    static_assert(DW % 2 == 0 && (DW / 2) * DW <= T && JP % 4 == 0, "pairing of the derivative positions");
    const bool sv = tid < (DW / 2) * DW;
    const int sY = sv ? tid / (DW / 2) : 0, sX = sv ? 2 * (tid - sY * (DW / 2)) : 0;     // the pair starts at (sX, sY)

    int tr[C::JLD], tc[C::JLD];                // (row, byte column) of this thread's dwords of a 40 x 44 B tile
    bool tv[C::JLD];
#pragma unroll
    for (int k = 0; k < C::JLD; k++) {
        const int i = tid + k * T;
        tv[k] = i < JT * C::JNDW;
        tr[k] = tv[k] ? i / C::JNDW : 0; tc[k] = tv[k] ? 4 * (i - tr[k] * C::JNDW) : 0;
    }

    // level geometry, read ONCE: the parameters live in the kernel-argument segment, every scalar load from it inside the
    // frame loop is a ~200-cycle wait in the middle of the chain.  The frames of a group share their geometry; only the image
    // BEFORE the group (the previous image of its first frame) may sit in a buffer with another pitch / stream stride.
    int gw[NLEV], gh[NLEV], gpitch[NLEV], gpitch0[NLEV];      // (row pitches fit 31 bits)
    long gbs[NLEV], gbs0[NLEV];
#pragma unroll
    for (int l = 0; l < NLEV; l++) {
        gw[l] = l <= maxl ? P->next[l].w : 0; gh[l] = l <= maxl ? P->next[l].h : 0;
        gpitch[l] = l <= maxl ? (int)P->next[l].pitch : 0; gbs[l] = l <= maxl ? P->next[l].bstride : 0;
        gpitch0[l] = l <= maxl ? (int)P->prev[l].pitch : 0; gbs0[l] = l <= maxl ? P->prev[l].bstride : 0;
    }
    bool all_safe = true;                      // every level can hold a whole tile at its origin (see the prefetch)
#pragma unroll
    for (int l = 0; l < NLEV; l++) all_safe = all_safe && (l > maxl || (4 * C::JNDW <= gw[l] && JT <= gh[l]));
    int toff[NLEV][C::JLD];                    // the same as byte offsets inside each level's image
#pragma unroll
    for (int l = 0; l < NLEV; l++) {
        const int p32 = gpitch[l];
#pragma unroll
        for (int k = 0; k < C::JLD; k++) toff[l][k] = tr[k] * p32 + tc[k];
    }

    const float halfw = (WIN - 1) * 0.5f;
    const float FLT_SCALE = 1.f / (1 << 20);
    const float eps2_lo = (float)(eps2 * (1.0 - 1e-6)), eps2_hi = (float)(eps2 * (1.0 + 1e-6));

    // ---- state carried from frame to frame (all wave-uniform)
    float px = 0.f, py = 0.f;
    int pst = 1;
    int cur = 0;                               // tile buffer that holds THIS frame's search tiles (image k); cur ^ 1: image k-1
    int jox[NLEV], joy[NLEV], sox[NLEV], soy[NLEV];        // tile origins (x as requested, the stored tile starts at x & ~3)
    int jmask = 0, smask = 0;                  // levels whose tile in buf[cur] / buf[cur ^ 1] is valid
    unsigned* owed = nullptr;                  // arrival counter of the previous frame, not yet counted into (see the frame's end)
#pragma unroll
    for (int l = 0; l < NLEV; l++) { jox[l] = joy[l] = sox[l] = soy[l] = 0; }

    for (int k = 0; k < nf; k++) {
        if (k) lds_barrier();                  // everyone is done with the previous frame's LDS; the table copy is visible
        const LkFrameIo<NLEV> io = frame(k);
        if (io.bad) {                          // the frame's table entries cannot be addresses: the group ends here for this corner (uniform over
            if (tid == 0 && owed) lk_arrive(owed, b);      // the workgroup); whoever waits for its frames gives up and reports (agt_step.hip)
            return;
        }
        if (k == 0) {
            if (io.have_pos) { px = io.px; py = io.py; pst = io.pst; }       // (lk_reseed_kernel: the start position was computed in the launch)
            else {
                px = io.prev_pts[pidx * 2]; py = io.prev_pts[pidx * 2 + 1];
                pst = P->prev_status ? (int)P->prev_status[pidx] : 1;
            }
            px = agt_uniform(px); py = agt_uniform(py); pst = agt_uniform(pst);
            if (!lk_pt_ok(px, py)) pst = 0;    // (a wild position is a lost corner: agt_lk_body.h lk_pt_ok)
        }
#ifdef AGT_STEP_STAMPS
        if (pidx == 0 && threadIdx.x == 0 && k < 4) agt_chain_stamps[k * 16 + 14] = 0;
#endif
        CSTAMP(0);
        if (!pst) {                            // sticky: a lost corner stays lost, position carried (see lk_body)
            if (tid == 0) {
                if (owed) lk_arrive(owed, b);
                lk_publish(io, pidx, b, px, py, 0, 0.f);
            }
            owed = nullptr;
            jmask = smask = 0;
            continue;
        }

        // ---- geometry of the image-(k-1) side (straight-line code: the levels are independent and meant to overlap); tiles
        // that are not in LDS yet (first frame of the group, a corner that moved ~8 px at a level)
        int ipx[NLEV], ipy[NLEV];
        float fa[NLEV], fb[NLEV];              // bilinear fractions of the previous position
        int lvA = 0, needI = 0, needJ = 0;
        CREPS(3) {
        lvA = needI = needJ = 0;
#pragma unroll
        for (int l = 0; l < NLEV; l++) {
            const float scale = lk_level_scale(l);
            const float prevx = px * scale - halfw, prevy = py * scale - halfw;
            ipx[l] = agt_uniform((int)floorf(prevx)); ipy[l] = agt_uniform((int)floorf(prevy));
            fa[l] = prevx - (float)ipx[l]; fb[l] = prevy - (float)ipy[l];
            const int w = gw[l], h = gh[l];
            const bool a = l <= maxl && !(ipx[l] < -WIN || ipx[l] >= w || ipy[l] < -WIN || ipy[l] >= h);
            const bool inside = ((smask >> l) & 1) && ipx[l] - 1 >= sox[l] && ipx[l] - 1 + C::IW <= sox[l] + JT &&
                                ipy[l] - 1 >= soy[l] && ipy[l] - 1 + C::IW <= soy[l] + JT;
            lvA |= a ? 1 << l : 0;
            needI |= (a && !inside) ? 1 << l : 0;
            needJ |= (a && !((jmask >> l) & 1)) ? 1 << l : 0;
        }
        }
        CSTAMP(7);
        CCOUNT(0, 1); CCOUNT(1, needI != 0); CCOUNT(2, needJ != 0);
        CSTAMP(15);
        if (agt_uniform(needI | needJ)) {
            uint32_t ti[NLEV][C::JLD], tj[NLEV][C::JLD];
#pragma unroll
            for (int l = 0; l < NLEV; l++) {
                if ((needI >> l) & 1) {
                    sox[l] = ipx[l] - 1 - SRC_PAD; soy[l] = ipy[l] - 1 - SRC_PAD;
                    tile_request(io.imgI[l] + (long)b * (k ? gbs[l] : gbs0[l]), gw[l], gh[l], k ? gpitch[l] : gpitch0[l], sox[l], soy[l], tid, tr, tc, tv, ti[l]);
                }
                if ((needJ >> l) & 1) {
                    jox[l] = ipx[l] - MARGIN; joy[l] = ipy[l] - MARGIN;           // centred on the initial guess = previous position
                    tile_request(io.imgJ[l] + (long)b * gbs[l], gw[l], gh[l], gpitch[l], jox[l], joy[l], tid, tr, tc, tv, tj[l]);
                }
            }
            CSTAMP(1);
#pragma unroll
            for (int l = 0; l < NLEV; l++) {
                if ((needI >> l) & 1) tile_store<JT, C::JNDW, T>(lds + ((cur ^ 1) * NLEV + l) * K::TILE, tid, ti[l]);
                if ((needJ >> l) & 1) tile_store<JT, C::JNDW, T>(lds + (cur * NLEV + l) * K::TILE, tid, tj[l]);
            }
            smask |= needI; jmask |= needJ;
            block_sync<NW>();
        }
        CSTAMP(2);

        // where each level's window starts inside its image-(k-1) tile (a level without a window points at a harmless spot)
        int offS[NLEV];
#pragma unroll
        for (int l = 0; l < NLEV; l++) {
            const int inside = (ipy[l] - soy[l]) * JP + (ipx[l] - (sox[l] & ~3));
            offS[l] = ((cur ^ 1) * NLEV + l) * K::TILE + (((lvA >> l) & 1) ? inside : JP + 4);
        }

        // ---- image-(k-1) side of every level: Scharr -> derivative tiles
        CREPS(0)
#pragma unroll
        for (int l = 0; l < NLEV; l++) {
            const int w = gw[l], h = gh[l];
            // byte address of the pixel left of and above the pair's first position; rows are JP = 44 B apart, so the three
            // rows share their alignment
            const int a = offS[l] + (sY - 1) * JP + (sX - 1);
            const int sh8 = a & 3;
            const uint32_t* p = reinterpret_cast<const uint32_t*>(lds + (a & ~3));
            uint32_t r[3];
#pragma unroll
            for (int i = 0; i < 3; i++) r[i] = __builtin_amdgcn_alignbyte(p[i * (JP / 4) + 1], p[i * (JP / 4)], sh8);   // pixels x-1 .. x+2 of row i
            // position 0 uses bytes 0..2, position 1 bytes 1..3: dx = (3, 10, 3) . (right - left), dy = (3, 10, 3) . (below - above)
            int val[2];
#pragma unroll
            for (int j = 0; j < 2; j++) {
                const uint32_t wr = j ? 0x03000000u : 0x00030000u, wl = j ? 0x00000300u : 0x00000003u;          // weight 3 on the right / left byte
                const uint32_t wr10 = j ? 0x0A000000u : 0x000A0000u, wl10 = j ? 0x00000A00u : 0x0000000Au;
                const uint32_t wrow = j ? 0x030A0300u : 0x00030A03u;
                unsigned xp = __builtin_amdgcn_udot4(r[0], wr, 0u, false);
                xp = __builtin_amdgcn_udot4(r[1], wr10, xp, false);
                xp = __builtin_amdgcn_udot4(r[2], wr, xp, false);
                unsigned xn = __builtin_amdgcn_udot4(r[0], wl, 0u, false);
                xn = __builtin_amdgcn_udot4(r[1], wl10, xn, false);
                xn = __builtin_amdgcn_udot4(r[2], wl, xn, false);
                const int dx = (int)xp - (int)xn;
                const int dy = (int)__builtin_amdgcn_udot4(r[2], wrow, 0u, false) - (int)__builtin_amdgcn_udot4(r[0], wrow, 0u, false);
                // the derivative image has a ZERO (BORDER_CONSTANT) border
                const int gx = ipx[l] + sX + j, gy = ipy[l] + sY;
                val[j] = (gx >= 0 && gx < w && gy >= 0 && gy < h) ? ((dx & 0xffff) | (dy << 16)) : 0;
            }
            if (sv) *reinterpret_cast<int2*>(sDall + l * (K::SD / 4) + sY * DW + sX) = make_int2(val[0], val[1]);
        }
        lds_barrier();
        CSTAMP(3);

        // ---- interpolated patch of every level (registers) + this thread's share of the covariance sums
        int Iv[NLEV][C::NPX], Ix[NLEV][C::NPX], Iy[NLEV][C::NPX];
        CREPS(1)
#pragma unroll
        for (int l = 0; l < NLEV; l++) {
            int iw00, iw01, iw10, iw11;
            bilinear_weights(fa[l], fb[l], iw00, iw01, iw10, iw11);
            const uint8_t* s0 = lds + offS[l];
            const int* sD = sDall + l * (K::SD / 4);
            int a0 = 0, a1 = 0, a2 = 0;
#pragma unroll
            for (int q = 0; q < C::NPX; q++) {
                const uint8_t* p = s0 + oW[q];
                const int iv = descale(bil4(p[0], p[1], p[JP], p[JP + 1], iw00, iw01, iw10, iw11), W_BITS - 5);
                const int* d = sD + oD[q];
                const int d00 = d[0], d01 = d[1], d10 = d[DW], d11 = d[DW + 1];
                const int ix = descale(bil4((short)d00, (short)d01, (short)d10, (short)d11, iw00, iw01, iw10, iw11), W_BITS);
                const int iy = descale(bil4(d00 >> 16, d01 >> 16, d10 >> 16, d11 >> 16, iw00, iw01, iw10, iw11), W_BITS);
                Iv[l][q] = pv[q] ? iv : 0; Ix[l][q] = pv[q] ? ix : 0; Iy[l][q] = pv[q] ? iy : 0;
                a0 += __mul24(Ix[l][q], Ix[l][q]); a1 += __mul24(Ix[l][q], Iy[l][q]); a2 += __mul24(Iy[l][q], Iy[l][q]);
            }
            // (|a| <= 2 * 4080^2 < 2^25)
            red[(3 * l) * T + tid] = a0; red[(3 * l + 1) * T + tid] = a1; red[(3 * l + 2) * T + tid] = a2;
        }
        lds_barrier();
        // ---- wave w finishes level w (w + 4, ..): exact sums over the 256 partials, the 2 x 2 system and its eigenvalue test
        CREPS(2)
        for (int l = wave; l < NLEV; l += NW) {
            // exact total = 65536 * sum(hi) + sum(lo) per value, both sums < 2^24 in magnitude: one rounding in the fma, the same
            // float as (float)(double)(int64 total); the six int32 sums go through three pair chains
            int s4[3];
#pragma unroll
            for (int i = 0; i < 3; i++) {
                const int* r = red + (3 * l + i) * T + lane;
                s4[i] = r[0] + r[64] + r[128] + r[192];                          // < 2^27
            }
            int lo[3], hi[3];
            wave_sum2_i32(s4[0] & 0xffff, s4[1] & 0xffff, lo[0], lo[1]);
            wave_sum2_i32(s4[2] & 0xffff, s4[0] >> 16, lo[2], hi[0]);
            wave_sum2_i32(s4[1] >> 16, s4[2] >> 16, hi[1], hi[2]);
            float A[3];
#pragma unroll
            for (int i = 0; i < 3; i++) A[i] = fmaf((float)hi[i], 65536.f, (float)lo[i]) * FLT_SCALE;
            const float D = A[0] * A[2] - A[1] * A[1];
            const float minEig = (A[2] + A[0] - sqrtf((A[0] - A[2]) * (A[0] - A[2]) + 4.f * A[1] * A[1])) / (float)(2 * WIN * WIN);
            const bool ok = ((lvA >> l) & 1) && !((double)minEig < min_eig_threshold || D < FLT_EPSILON);
            if (lane == 0) {
                float* o = lvl + l * 8;
                o[0] = A[0]; o[1] = A[1]; o[2] = A[2]; o[3] = ok ? 1.f / D : 0.f; o[4] = ok ? 1.f : 0.f;
            }
        }
        lds_barrier();
        // the previous frame's result was stored ~3 us ago: its stores have long been acknowledged, count the corner in now
        if (owed && tid == 0) lk_arrive(owed, b);
        owed = nullptr;
        CSTAMP(4);

        // ---- the levels, coarse to fine: the iterations against image k
        int st = 1;
        float outx = 0.f, outy = 0.f;
        bool pre = false;                      // next frame's search tiles requested
        uint32_t tn[NLEV][C::JLD];
        int nox[NLEV], noy[NLEV], nmask = 0;
#pragma unroll
        for (int l = 0; l < NLEV; l++) { nox[l] = noy[l] = 0; }
        // (unrolled: with a compile-time level every per-level quantity below is a plain register -- as a run-time loop each
        // level paid ~0.7 us of selects, read-first-lanes and their bubbles before its first iteration)
#pragma unroll
        for (int level = NLEV - 1; level >= 0; level--) {
            if (level > maxl) continue;
            const float scale = lk_level_scale(level);
            float nextx, nexty;
            if (level == maxl) { nextx = px * scale; nexty = py * scale; }
            else { nextx = outx * 2.f; nexty = outy * 2.f; }
            outx = nextx; outy = nexty;
            CSTAMP(8 + level * 2);
            if (level == 0 && k + 1 < nf) {
                // ---- request frame k+1's search tiles (image k+1 around where this corner is about to end up).  Straight-line
                // code, every load unconditional (a level whose tile is not wholly inside its image loads a harmless tile
                // at the image origin and is left to the next frame's on-demand path): any branch around a load makes the
                // compiler copy the register array at the merge, and the copy waits for the loads issued so far.
                const LkFrameIo<NLEV> nio = frame(k + 1);
                if (agt_uniform((int)(all_safe && !nio.bad))) {
                    CREPS(4) {
                    nmask = 0;
#pragma unroll
                    for (int l = 0; l < NLEV; l++) {
                        typedef const __attribute__((address_space(1))) uint8_t* G8;
                        typedef const __attribute__((address_space(1))) uint32_t* G32;
                        const float sc = lk_level_scale(l);
                        const int cx = agt_uniform((int)floorf(outx * sc - halfw)), cy = agt_uniform((int)floorf(outy * sc - halfw));
                        const int w = gw[l], h = gh[l];
                        const long pitch = gpitch[l], bstride = gbs[l];
                        const int ox = cx - MARGIN, oy = cy - MARGIN, ax0 = ox & ~3;
                        const bool ok = l <= maxl && ax0 >= 0 && ax0 + 4 * C::JNDW <= w && oy >= 0 && oy + JT <= h;
                        const int lx = ok ? ax0 : 0, ly = ok ? oy : 0;
                        const uint8_t* img = l <= maxl ? nio.imgJ[l] : nio.imgJ[0];
                        const G8 base = (G8)img + (long)b * (l <= maxl ? bstride : 0) + (long)ly * (l <= maxl ? pitch : 0) + lx;
#pragma unroll
                        for (int q = 0; q < C::JLD; q++) tn[l][q] = *(G32)(base + toff[l][q]);
                        nox[l] = ox; noy[l] = oy;
                        nmask |= ok ? 1 << l : 0;
                    }
                    }
                    pre = true;
                }
            }
            if (!((lvA >> level) & 1)) { if (level == 0) st = 0; continue; }
            const float4 lv4 = *reinterpret_cast<const float4*>(lvl + level * 8);
            if (!agt_uniform((int)(lvl[level * 8 + 4] != 0.f))) { if (level == 0) st = 0; continue; }
            // (`level` is a compile-time constant here: plain indexing, no selects)
            AgtLevel LJ;
            LJ.ptr = nullptr; LJ.w = gw[level]; LJ.h = gh[level]; LJ.pitch = gpitch[level]; LJ.bstride = gbs[level];
            const uint8_t* imgJ = io.imgJ[level] + (long)b * LJ.bstride;
            uint8_t* sJ = lds + (cur * NLEV + level) * K::TILE;
            const float a11 = agt_uniform(lv4.x), a12 = agt_uniform(lv4.y), a22 = agt_uniform(lv4.z), D = agt_uniform(lv4.w);
            int iv[C::NPX], ix[C::NPX], iy[C::NPX];
#pragma unroll
            for (int q = 0; q < C::NPX; q++) { iv[q] = Iv[level][q]; ix[q] = Ix[level][q]; iy[q] = Iy[level][q]; }
            int jx0 = jox[level], jy0 = joy[level];

            nextx -= halfw; nexty -= halfw;
            float pdx = 0.f, pdy = 0.f;
            int iw00, iw01, iw10, iw11;
            // while the window's corner stays in this box, it is inside the image band and inside the search tile: one float
            // test per iteration instead of the two integer ones (which remain, word for word, behind it)
            float bx0, bx1, by0, by1;
            auto set_box = [&]() {
                const int lx = jx0 > -WIN ? jx0 : -WIN, hx = (jx0 + JT - WIN - 1 < LJ.w - 1 ? jx0 + JT - WIN - 1 : LJ.w - 1) + 1;
                const int ly = jy0 > -WIN ? jy0 : -WIN, hy = (jy0 + JT - WIN - 1 < LJ.h - 1 ? jy0 + JT - WIN - 1 : LJ.h - 1) + 1;
                bx0 = (float)lx; bx1 = (float)hx; by0 = (float)ly; by1 = (float)hy;
            };
            set_box();
            // One vector -> scalar decision per iteration.  Every readfirstlane + branch is a ~50-cycle bubble on this chain, so
            // the tests of an iteration (converged, oscillating, FP64 tie-break needed, next window position outside the box)
            // are folded into one code that crosses to the scalar unit once.
            int slow = agt_uniform((int)!(nextx >= bx0 && nextx < bx1 && nexty >= by0 && nexty < by1));
            for (int j = 0; j < max_count; j++) {
                const float fx = floorf(nextx), fy = floorf(nexty);
                if (slow) {
                    const int inx = agt_uniform((int)fx), iny = agt_uniform((int)fy);
                    if (inx < -WIN || inx >= LJ.w || iny < -WIN || iny >= LJ.h) {
                        if (level == 0) st = 0;
                        break;
                    }
                    if (inx < jx0 || inx + WIN >= jx0 + JT || iny < jy0 || iny + WIN >= jy0 + JT) {
                        jx0 = inx - MARGIN; jy0 = iny - MARGIN;
                        uint32_t t[C::JLD];
                        block_sync<NW>();
                        tile_load<JT, C::JNDW, T, C::JLD, true>(imgJ, LJ.w, LJ.h, LJ.pitch, jx0, jy0, tid, t);
                        tile_store<JT, C::JNDW, T>(sJ, tid, t);
                        // every load of this (rare) path has landed -- said in a form the compiler's wait-count bookkeeping reads:
                        // left implicit, the lanes that loaded nothing keep "a load into t may be pending" alive up to the merge
                        // with the common path, which then waits for ALL outstanding loads (the next frame's tiles) every iteration
                        __builtin_amdgcn_s_waitcnt(0x0F70);        // vmcnt(0)
                        block_sync<NW>();
                        set_box();
                    }
                }
                const int inx = (int)fx, iny = (int)fy;
                // the six tap bytes are requested FIRST (their address needs only the integer part of the position) and the
                // bilinear weights -- ~20 dependent float operations on the fraction -- are computed while the LDS answers
                const uint8_t* q0 = sJ + __mul24(iny - jy0, JP) + (inx - jx0) + (jx0 - (jx0 & ~3));
                int tap[C::NPX][4];
#pragma unroll
                for (int q = 0; q < C::NPX; q++) {
                    const uint8_t* c = q0 + oW[q];
                    tap[q][0] = c[0]; tap[q][1] = c[1]; tap[q][2] = c[JP]; tap[q][3] = c[JP + 1];
                }
                __builtin_amdgcn_sched_barrier(0);
                bilinear_weights(nextx - fx, nexty - fy, iw00, iw01, iw10, iw11);
                int bsum[2] = { 0, 0 };
#pragma unroll
                for (int q = 0; q < C::NPX; q++) {
                    const int diff = descale(bil4(tap[q][0], tap[q][1], tap[q][2], tap[q][3], iw00, iw01, iw10, iw11), W_BITS - 5) - iv[q];
                    bsum[0] += __mul24(diff, ix[q]); bsum[1] += __mul24(diff, iy[q]);          // ix = iy = 0 where !pv
                }
                float fb1, fb2;
                iter_sum2(bsum, fb1, fb2, slots, phase, wave, lane);
                fb1 *= FLT_SCALE; fb2 *= FLT_SCALE;
                const float dx = (a12 * fb2 - a22 * fb1) * D;
                const float dy = (a12 * fb1 - a11 * fb2) * D;
                nextx += dx; nexty += dy;
                outx = nextx + halfw; outy = nexty + halfw;
#ifdef AGT_STEP_STAMPS
                if (pidx == 0 && threadIdx.x == 0 && k < 4 && level < 3) agt_chain_stamps[k * 16 + 14] += 1ull << (level * 8);
#endif
                // (double)dx * dx + (double)dy * dy <= eps2: FP64 only inside a 1e-6 band around the threshold (code 4, rare);
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
                    const bool conv = (double)dx * dx + (double)dy * dy <= eps2;
                    code = agt_uniform(conv ? 1 : (osc ? 2 : (out_of_box ? 3 : 0)));
                }
                if (code == 1) break;
                if (code == 2) { outx -= dx * 0.5f; outy -= dy * 0.5f; break; }
                slow = code == 3;
                pdx = dx; pdy = dy;
            }
            jox[level] = agt_uniform(jx0); joy[level] = agt_uniform(jy0);
            CSTAMP(9 + level * 2);
        }
        CSTAMP(5);
        // the result: stores now; the arrival is counted after their acknowledgement -- at once for the last frame of the group
        // (its pose solve is the launch's tail), otherwise under the next frame's image-side work
        if (io.done) {
            CREPS(6) if (tid == 0) lk_publish_stores(io, pidx, outx, outy, st);
            if (k + 1 < nf) owed = io.done;
            else if (tid == 0) lk_arrive(io.done, b);
        } else if (tid == 0) lk_publish(io, pidx, b, outx, outy, st, 0.f);

        // ---- hand-over: image k's search tiles become the next frame's previous-image tiles; the requested tiles of image
        // k+1 go into the buffer the image-(k-1) side no longer needs (its last reader was before this frame's barriers)
#pragma unroll
        for (int l = 0; l < NLEV; l++) { sox[l] = jox[l]; soy[l] = joy[l]; }
        smask = jmask;
        jmask = 0;
        if (pre) {
            CREPS(5)
#pragma unroll
            for (int l = 0; l < NLEV; l++) {
                if ((nmask >> l) & 1) {
                    tile_store<JT, C::JNDW, T>(lds + ((cur ^ 1) * NLEV + l) * K::TILE, tid, tn[l]);
                    jox[l] = nox[l]; joy[l] = noy[l];
                }
            }
            jmask = nmask;
        }
        cur ^= 1;
        px = outx; py = outy; pst = st;
        CSTAMP(6);
    }
}

}  // namespace agt_lk
