// agt_lk_body.h -- device body of the cv::calcOpticalFlowPyrLK per-point tracker for gfx950 (north-star step;
// no call site in the reference, belongs at the hole detect_pose.py:573-574).
// Semantics: OpenCV modules/video/src/lkpyramid.cpp LKTrackerInvoker + calcSharrDeriv,
// restated on the CPU in oracle/cv_lk.c; this kernel is bit-identical to that oracle
// (CVO_ACC_EXACT mode) on nextPts, status and err.
//
// Mapping: one workgroup of NW waves per corner (NW = 4 when few corners are in flight and
// latency matters, NW = 1 for big batches where throughput matters), all pyramid levels and
// all iterations inside one launch.  Thread t owns window pixels t, t + 64*NW, ... and keeps
// their patch values (I, Ix, Iy) in registers.  A single wave issues ~1 instruction per 10
// cycles here, so the per-iteration critical path is cut by spreading the 441 pixels over
// more lanes; sums cross waves through 8-byte LDS slots and one barrier per iteration.
//   * the 24x24 I neighbourhood is staged in LDS with aligned dword loads (reflect-101 at
//     the image edge), Scharr is evaluated on the fly from that tile (no full-frame
//     derivative image is ever written to HBM -- the CPU path writes 4 B/px/level);
//   * the J search tile (40x40) is re-staged only if the window leaves it;
//   * ALL levels' I and J tiles are requested at kernel entry (every global load in flight at
//     once, then written to per-level LDS slots), so the kernel pays one HBM latency instead
//     of one per staging-loop trip; J tiles are centred on the initial guess;
//   * the 2x2 normal equations are EXACT integer sums: int32 while provably safe (DPP row
//     ops), then 64-bit scalars (v_readlane + SALU) and LDS slots across waves, so the result
//     is independent of the reduction order and uniform; every OpenCV build approximates this
//     sum with float adds in its own SIMD order.
// Algorithmic bytes per point per level: 24*24 (I) + 40*40 (J tile) u8.
#pragma once
#include "agt_device.h"
#include "agt_kernels.h"

// In-kernel cycle stamps (diagnostic builds only: make dbg; never in the shipped library).
#ifdef AGT_LK_STAMPS
__device__ unsigned long long agt_lk_stamps[64];
#define STAMP(i) do { if (pidx == 0 && threadIdx.x == 0) agt_lk_stamps[i] = __builtin_amdgcn_s_memtime(); } while (0)
extern "C" int agt_debug_lk_stamps(unsigned long long* host64)
{
    return (int)hipMemcpyFromSymbol(host64, HIP_SYMBOL(agt_lk_stamps), sizeof(agt_lk_stamps));
}
// per-corner log of the stand-alone launch (tools/lkcorners.py): entry and exit time of every corner's wave(s) and the
// iterations it took -- the launch lasts as long as its slowest corner
#define AGT_LK_CORNER_LOG 8192
__device__ unsigned long long agt_lk_corner_log[AGT_LK_CORNER_LOG][4];        // entry time, exit time, iterations, HW_ID | XCC_ID << 32 (where the wave ran)
extern "C" int agt_debug_lk_corner_log(unsigned long long* host, int n)
{
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(agt_lk_corner_log), sizeof(unsigned long long) * 4 * (size_t)(n < AGT_LK_CORNER_LOG ? n : AGT_LK_CORNER_LOG));
}
#else
#define STAMP(i)
#endif

namespace agt_lk {

template <int WIN, int NW>
struct LkCfg {
    static constexpr int T = AGT_WAVE * NW;             // threads per corner
    static constexpr int NPX = (WIN * WIN + T - 1) / T; // window pixels per thread
    static constexpr int IW = WIN + 3;                  // I tile: window + bilinear + Scharr halo
    static constexpr int INDW = (IW + 6) / 4;           // aligned dwords per I-tile row (incl. <=3 B shift)
    static constexpr int IP = INDW * 4;                 // LDS pitch
    static constexpr int DW = WIN + 1;                  // derivative tile
    // search margin around the window's expected position.  Four waves per corner (latency: a re-staged tile is a memory round trip on the
    // corner's critical path): 9 px.  One wave per corner (big batches: the launch moves 3.3 x its algorithmic bytes in whole 128-B lines and
    // shares HBM with the pyramid pass, tools/trace_blocks.py): 5 px -- a 32-row tile instead of 40, a fifth fewer lines; corners that
    // move further within a level re-stage, as ever (round 6)
#ifndef AGT_LK_MARGIN1
#define AGT_LK_MARGIN1 5
#endif
    static constexpr int MARGIN = NW == 1 ? AGT_LK_MARGIN1 : 9;
    static constexpr int JT = WIN + 1 + 2 * MARGIN;     // J search tile
    static constexpr int JNDW = (JT + 6) / 4;
    static constexpr int JP = JNDW * 4;
    static constexpr int LEVEL_LDS = IW * IP + JT * JP; // per-level LDS slot (I tile + J tile)
    static constexpr int ILD = (IW * INDW + T - 1) / T; // I-tile dwords per thread
    static constexpr int JLD = (JT * JNDW + T - 1) / T; // J-tile dwords per thread
    static constexpr int NSD = (DW * DW + T - 1) / T;   // derivative positions per thread
    // int32 partial sums stay exact over 2^SUM_STEPS lanes: NPX * 8160 * 4080 * 2^S < 2^31
    static constexpr int SUM_STEPS = NPX <= 2 ? 4 : (NPX <= 8 ? 3 : 2);
    // the pair-packed reduction of the two mismatch sums adds one more lane doubling before the 4 row steps
    static constexpr bool PAIR_OK = (long long)NPX * 8160 * 4080 * 32 < (1LL << 31);
    static_assert((long long)NPX * 8160 * 4080 * (1 << SUM_STEPS) < (1LL << 31), "int32 partial sums could overflow");
    static_assert(LEVEL_LDS % 16 == 0, "keep per-level slots 16-byte aligned");
};

// NW > 1: workgroup barrier.  NW == 1: the corner is one wave (possibly sharing its workgroup
// with other corners), LDS ops of a wave execute in issue order, so only the compiler must be
// kept from reordering across the point.
template <int NW>
__device__ __forceinline__ void block_sync()
{
    if (NW > 1) __syncthreads();
    else { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); }
}

constexpr int W_BITS = 14;
__device__ __forceinline__ int descale(int x, int n) { return (x + (1 << (n - 1))) >> n; }

// border path of the tile loads: one dword assembled from four reflect-101 bytes (rare; kept
// out of line so the common path stays small)
__device__ __noinline__ uint32_t load_dword_reflect(const uint8_t* __restrict__ row, int gx, int w)
{
    uint32_t v = 0;
    for (int k = 0; k < 4; k++) v |= (uint32_t)row[agt_reflect101(gx + k, w)] << (8 * k);
    return v;
}

// Tile = rows [ty0, ty0+TH) x the NDW aligned dwords starting at (tx0 & ~3); pixel (x, y) lands at
// s[(y - ty0) * (4*NDW) + (x - (tx0 & ~3))].  Loads and LDS stores are split so that every
// level's loads are in flight together.
// INL: the border path is expanded in place.  A call in the path makes the compiler assume at every later merge point that
// loads of unknown registers are outstanding (s_waitcnt vmcnt(0) before the next VALU write), which serialises a caller
// that keeps loads in flight across such a point (agt_lk_chain_body.h prefetches the next frame's tiles).
template <int TH, int NDW, int T, int N, bool INL = false>
__device__ __forceinline__ void tile_load(const uint8_t* __restrict__ img, int w, int h, long pitch,
                                          int tx0, int ty0, int tid, uint32_t (&v)[N])
{
    const int ax0 = tx0 & ~3;
#pragma unroll
    for (int k = 0; k < N; k++) {
        const int i = tid + k * T;
        v[k] = 0;
        if (i < TH * NDW) {
            const int r = i / NDW, c4 = i - r * NDW;
            const int gy = agt_reflect101(ty0 + r, h);
            const int gx = ax0 + 4 * c4;
            const uint8_t* row = img + (long)gy * pitch;
            if (INL) {
                // (explicitly GLOBAL loads: while a FLAT load is outstanding, every wait for LDS data also waits for it)
                typedef const __attribute__((address_space(1))) uint8_t* G8;
                typedef const __attribute__((address_space(1))) uint32_t* G32;
                const G8 grow = (G8)row;
                if (gx >= 0 && gx + 3 < w) v[k] = *(G32)(grow + gx);
                else {
                    uint32_t t = 0;
#pragma unroll
                    for (int q = 0; q < 4; q++) t |= (uint32_t)grow[agt_reflect101(gx + q, w)] << (8 * q);
                    v[k] = t;
                }
            } else
                v[k] = (gx >= 0 && gx + 3 < w) ? *reinterpret_cast<const uint32_t*>(row + gx) : load_dword_reflect(row, gx, w);
        }
    }
}

template <int TH, int NDW, int T, int N>
__device__ __forceinline__ void tile_store(uint8_t* s, int tid, const uint32_t (&v)[N])
{
#pragma unroll
    for (int k = 0; k < N; k++) {
        const int i = tid + k * T;
        if (i < TH * NDW) *reinterpret_cast<uint32_t*>(s + 4 * i) = v[k];
    }
}

__device__ __forceinline__ void bilinear_weights(float a, float b, int& iw00, int& iw01, int& iw10, int& iw11)
{
    iw00 = __float2int_rn((1.f - a) * (1.f - b) * (float)(1 << W_BITS));
    iw01 = __float2int_rn(a * (1.f - b) * (float)(1 << W_BITS));
    iw10 = __float2int_rn((1.f - a) * b * (float)(1 << W_BITS));
    iw11 = (1 << W_BITS) - iw00 - iw01 - iw10;
}

// all operands of the fixed-point products fit 24 bits (pixels 8, weights 15, patch values 14,
// derivatives 13): v_mul_i32_i24 / v_mad_i32_i24 are full rate, v_mul_lo_u32 is not
__device__ __forceinline__ int bil4(int v00, int v01, int v10, int v11, int iw00, int iw01, int iw10, int iw11)
{
    return __mul24(v00, iw00) + __mul24(v01, iw01) + __mul24(v10, iw10) + __mul24(v11, iw11);
}

// Exact 64-lane sum of one int per lane; identical (wave-uniform) in every lane.
// STEPS DPP steps in int32 (proved not to overflow), the rest in 64-bit scalars.
template <int STEPS>
__device__ __forceinline__ long long wave_sum_exact(int v)
{
    v += agt_dpp_i32<0xB1>(v);                         // quad_perm [1,0,3,2]
    v += agt_dpp_i32<0x4E>(v);                         // quad_perm [2,3,0,1]
    if (STEPS >= 3) v += agt_dpp_i32<0x141>(v);        // row_half_mirror: 8-lane sums
    if (STEPS >= 4) v += agt_dpp_i32<0x140>(v);        // row_mirror: 16-lane sums
    constexpr int G = 1 << STEPS;
    long long t = 0;
#pragma unroll
    for (int l = 0; l < AGT_WAVE; l += G) t += (long long)__builtin_amdgcn_readlane(v, l);
    return t;
}

// Exact sum over the whole workgroup of NV ints per thread; identical in every thread.
template <int NW, int STEPS, int NV, bool PAIR_OK = false>
__device__ __forceinline__ void block_sum_exact(const int (&v)[NV], long long (&out)[NV], long long* slots, int& phase,
                                                int wave, int lane)
{
    if constexpr (NW == 4) {
        // Four waves per corner (round 3): every value is split v = 65536 * hi + lo, two values share one DPP chain per half
        // (v_permlane32_swap: value 0 in the lower half-wave, value 1 in the upper one), and the FIRST LANE OF EACH 16-LANE
        // ROW adds its row sum straight into the workgroup's accumulators with LDS atomics (integer: order-free, exact):
        // no read-lanes, no per-wave slots, one or two 16-byte reads after the barrier.  |v| < 2^28 per thread keeps every
        // int32 partial in range (lo: 256 x 65535 < 2^24, hi: 256 x 2^12).  Three accumulator sets rotate; the set of sum
        // i + 2 is cleared by thread 0 right after the barrier of sum i, when its last readers (sum i - 1) are past it.
        // Accumulator layout: pair p of values -> ints [4 p + 2 * (value & 1) + { 0: lo, 1: hi }].
        int* acc = reinterpret_cast<int*>(slots) + phase * 8;
#pragma unroll
        for (int p = 0; p < (NV + 1) / 2; p++) {
            const int a = v[2 * p], b = 2 * p + 1 < NV ? v[2 * p + 1] : 0;
            const auto swl = __builtin_amdgcn_permlane32_swap((unsigned)(a & 0xffff), (unsigned)(b & 0xffff), false, false);
            const auto swh = __builtin_amdgcn_permlane32_swap((unsigned)(a >> 16), (unsigned)(b >> 16), false, false);
            int xl = (int)swl[0] + (int)swl[1], xh = (int)swh[0] + (int)swh[1];
            xl += agt_dpp_i32<0xB1>(xl); xh += agt_dpp_i32<0xB1>(xh);
            xl += agt_dpp_i32<0x4E>(xl); xh += agt_dpp_i32<0x4E>(xh);
            xl += agt_dpp_i32<0x141>(xl); xh += agt_dpp_i32<0x141>(xh);
            xl += agt_dpp_i32<0x140>(xl); xh += agt_dpp_i32<0x140>(xh);
            if ((lane & 15) == 0) {
                __attribute__((address_space(3))) int* q = (__attribute__((address_space(3))) int*)(acc + 4 * p + ((lane >> 5) << 1));
                __hip_atomic_fetch_add(q, xl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                __hip_atomic_fetch_add(q + 1, xh, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }
        block_sync<NW>();
        const int4 r0 = *reinterpret_cast<const int4*>(acc);
        out[0] = (long long)r0.y * 65536 + r0.x;
        if constexpr (NV > 1) out[1] = (long long)r0.w * 65536 + r0.z;
        if constexpr (NV > 2) { const int2 r1 = *reinterpret_cast<const int2*>(acc + 4); out[2] = (long long)r1.y * 65536 + r1.x; }
        const int nz = phase == 0 ? 2 : phase - 1;                 // (phase + 2) % 3
        if (threadIdx.x == 0) {
            *reinterpret_cast<int4*>(reinterpret_cast<int*>(slots) + nz * 8) = make_int4(0, 0, 0, 0);
            *reinterpret_cast<int4*>(reinterpret_cast<int*>(slots) + nz * 8 + 4) = make_int4(0, 0, 0, 0);
        }
        phase = phase == 2 ? 0 : phase + 1;
        return;
    }
    if constexpr ((NV == 2 || NV == 3) && PAIR_OK) {
        // two sums in one chain: v_permlane32_swap leaves { v0 of lanes 0-31 | v1 of lanes 0-31 } and { v0 of lanes
        // 32-63 | v1 of lanes 32-63 } side by side, so one add gives pair sums of v0 in the lower half-wave and of v1
        // in the upper one; four DPP row steps (still < 2^31, checked by the caller's PAIR_OK) and four readlanes follow
        const auto sw = __builtin_amdgcn_permlane32_swap((unsigned)v[0], (unsigned)v[1], false, false);
        int x = (int)sw[0] + (int)sw[1];
        x += agt_dpp_i32<0xB1>(x);
        x += agt_dpp_i32<0x4E>(x);
        x += agt_dpp_i32<0x141>(x);
        x += agt_dpp_i32<0x140>(x);
        out[0] = (long long)__builtin_amdgcn_readlane(x, 0) + (long long)__builtin_amdgcn_readlane(x, 16);
        out[1] = (long long)__builtin_amdgcn_readlane(x, 32) + (long long)__builtin_amdgcn_readlane(x, 48);
        if constexpr (NV == 3) out[2] = wave_sum_exact<STEPS>(v[2]);
    } else {
#pragma unroll
        for (int i = 0; i < NV; i++) out[i] = wave_sum_exact<STEPS>(v[i]);
    }
    if (NW > 1) {
        long long* s = slots + phase * (NW * 4);
        if (lane == 0) {
#pragma unroll
            for (int i = 0; i < NV; i++) s[wave * 4 + i] = out[i];
        }
        block_sync<NW>();
#pragma unroll
        for (int i = 0; i < NV; i++) {
            long long t = 0;
#pragma unroll
            for (int w = 0; w < NW; w++) t += s[w * 4 + i];
            out[i] = t;
        }
        phase ^= 1;          // double-buffered: the next sum may start before slow waves have read this one
    }
}

// LDS bytes one corner needs: [levels] x { I tile | J tile }, derivative tile, reduction slots
template <int WIN, int NW>
__host__ __device__ constexpr size_t lk_lds_bytes(int levels)
{
    using C = LkCfg<WIN, NW>;
    size_t behind = (size_t)((C::DW * C::DW + 3) & ~3) * sizeof(int) + (size_t)2 * NW * 4 * sizeof(long long);
    // (one wave per corner, 21 x 21: the row-segment body keeps a 23 x 24 int32 grid there instead -- agt_lk_rs_body.h RS_B_BYTES)
    if (WIN == 21 && NW == 1 && behind < (size_t)23 * 24 * 4) behind = (size_t)23 * 24 * 4;
    return (size_t)levels * C::LEVEL_LDS + behind;
}


// 1 / 2^l as a float: the exponent field written directly.  (As the division 1.f / (float)(1 << l) with a run-time l it is the full IEEE
// sequence -- v_div_scale x 2, v_rcp, four fused multiply-adds, v_div_fmas, v_div_fixup -- on the critical path of every level of every
// corner, and of every trip of rs_interior: round 6.)  Exact for 0 <= l <= 126.
__device__ __forceinline__ float lk_level_scale(int l) { return __int_as_float((127 - l) << 23); }

// A coordinate the tracker will not form an address from: non-finite or |x| >= 2^20.  OpenCV floors such a value to an integer far
// outside every image (cvFloor of a NaN or of anything beyond the int range is INT_MIN on x86) and so finds the window "outside the
// image" at every level: status 0, err 0, the position carried -- which is what the bodies below do with it up front (a GPU
// float -> int conversion turns a NaN into 0, i.e. INTO the image).  Finite positions below 2^20 go through the usual bounds tests.
__device__ __forceinline__ bool lk_pt_ok(float x, float y) { return fabsf(x) < 1048576.f && fabsf(y) < 1048576.f; }

// Where one frame of the tracker reads and writes.  grouped = image bases come from imgI / imgJ (the frame
// group of the fused step) instead of P.prev / P.next; have_pos = the previous position comes in registers
// (frame 2.. of a group: the corner was tracked by this same workgroup a moment ago) instead of from prev_pts.
template <int NLEV>
struct LkFrameIo {
    const uint8_t* imgI[NLEV];      // previous image, per level
    const uint8_t* imgJ[NLEV];      // next image, per level
    bool grouped;
    const float* prev_pts; float* next_pts; uint8_t* status; float* err;
    bool have_pos; float px, py; int pst;       // pst: the corner's status after the previous frame (with have_pos)
    unsigned* done = nullptr;                   // chained launch (agt_step.hip): arrival counters of this frame, [B]; see lk_publish
    bool bad = false;                           // frame group: the frame's table entries cannot be addresses (agt_step.hip lk_role) -- nothing of the frame is touched
    uint8_t* iters_out = nullptr;               // stand-alone launches of big batches: iterations the corner took, [B][n] (AgtLkParams::iters_out)
};

// The frame's result for one corner (called by one lane).  In a chained launch the PnP role of the SAME launch picks the
// corners up as soon as the frame's counter reaches the corner count, from a workgroup on another XCD (another L2): the
// result is written with device-scope stores (write-through past the L2) and the arrival is counted only after every one of
// them has been acknowledged.
// count the corner in: every store of the calling lane has been acknowledged first (lk_publish_stores of this frame)
// The ordering below is hand-written for the gfx9 family: stores are counted in vmcnt there, and the sc1 write-through stores
// of lk_publish_stores are acknowledged by memory before vmcnt drops.  A target with a separate store counter (vscnt,
// gfx10+) would turn this into a silent data race, so the file refuses to compile for anything else.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__) && !defined(__gfx942__)
#error "agt_lk_body.h: the chained-launch publish/arrive ordering (s_waitcnt vmcnt(0) + relaxed agent-scope add) is only valid on gfx942 / gfx950"
#endif
__device__ __forceinline__ void lk_arrive(unsigned* done, int b)
{
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __hip_atomic_fetch_add(done + b, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// chained launch, first half of lk_publish: the device-scope stores only; the caller counts the corner in (lk_arrive, same
// lane) once it can wait for their acknowledgement without standing still -- a write-through to HBM takes ~0.8 us
template <int NLEV>
__device__ __forceinline__ void lk_publish_stores(const LkFrameIo<NLEV>& io, long pidx, float x, float y, int st)
{
    __hip_atomic_store(io.next_pts + pidx * 2, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(io.next_pts + pidx * 2 + 1, y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(io.status + pidx, (uint8_t)st, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <int NLEV>
__device__ __forceinline__ void lk_publish(const LkFrameIo<NLEV>& io, long pidx, int b, float x, float y, int st, float errv)
{
    if (io.done) {
        lk_publish_stores(io, pidx, x, y, st);
        if (io.err) io.err[pidx] = errv;
        lk_arrive(io.done, b);
        return;
    }
    io.next_pts[pidx * 2] = x; io.next_pts[pidx * 2 + 1] = y;
    io.status[pidx] = (uint8_t)st;
    if (io.err) io.err[pidx] = errv;
}

// Track corner `pt` of stream `b` through one frame.  Called by all 64*NW threads of a workgroup; lds:
// lk_lds_bytes, 16-B aligned.  Returns the new position in (ox, oy) (identical in every thread).
// field-wise copy: L may live in the kernel-argument address space (no implicit struct copy from there)
template <typename LV>
__device__ __forceinline__ AgtLevel get_level(const LV& L)
{
    AgtLevel r;
    r.ptr = L.ptr; r.pitch = L.pitch; r.bstride = L.bstride; r.w = L.w; r.h = L.h;
    return r;
}

// PP: pointer to the parameters -- `const AgtLkParams*` (stand-alone launch) or a pointer into the kernel-argument
// segment (fused step, where the level tables are indexed with run-time levels inside a frame loop).
// STOP > 0 (round 6, agt_lk.hip lk_kernel): only the pyramid levels max_level .. level_stop are tracked here and nothing is
// published -- the position after level `level_stop` comes back in ox / oy (at that level's scale) for the row-segment body to carry
// through the finer levels; ost < 0 then says that the corner was finished (published) here after all (lost, wild).
// (STOP is a template parameter: as a run-time bound of the level loop it cost the one-wave kernel its register allocation -- spills at
// 128 registers -- and with it known, the prologue requests only the tiles of the levels tracked here.)
template <int WIN, int NW, int NLEV, typename PP, int STOP = 0>
__device__ __forceinline__ void lk_body(PP P, int pt, int b, uint8_t* lds, const LkFrameIo<NLEV>& io, float& ox, float& oy, int& ost)
{
    constexpr int level_stop = STOP;
    using C = LkCfg<WIN, NW>;
    constexpr int T = C::T;
    int* sD = reinterpret_cast<int*>(lds + (P->max_level + 1) * C::LEVEL_LDS);
    long long* slots = reinterpret_cast<long long*>(sD + ((C::DW * C::DW + 3) & ~3));
    int phase = 0;

    const int tid = NW == 1 ? (int)(threadIdx.x & (AGT_WAVE - 1)) : (int)threadIdx.x;   // thread index within the corner
    if (NW == 4 && tid < 24) reinterpret_cast<int*>(slots)[tid] = 0;       // block_sum_exact's accumulators (a barrier precedes the first sum)
    const int lane = tid & (AGT_WAVE - 1), wave = tid / AGT_WAVE;
    const long pidx = (long)b * P->n + pt;
    int nit = 0;                    // iterations over all levels (only kept where io.iters_out is set)

    // window pixels of this thread, as byte / element offsets into the three LDS tiles (computed once;
    // threads without a k-th pixel point at offset 0 and are masked arithmetically, not by branches)
    int oI[C::NPX], oD[C::NPX], oJ[C::NPX];
    bool pv[C::NPX];
#pragma unroll
    for (int k = 0; k < C::NPX; k++) {
        const int p = tid + k * T;
        pv[k] = p < WIN * WIN;
        const int y = pv[k] ? p / WIN : 0, x = pv[k] ? p - y * WIN : 0;
        oI[k] = (y + 1) * C::IP + (x + 1);
        oD[k] = y * C::DW + x;
        oJ[k] = y * C::JP + x;
    }

    const float halfw = (WIN - 1) * 0.5f;
    const float FLT_SCALE = 1.f / (1 << 20);
    if (io.bad) { ox = io.px; oy = io.py; ost = 0; return; }      // (wave-uniform; only frames 2.. of a group can be: have_pos holds)
    const float ppx = io.have_pos ? io.px : io.prev_pts[pidx * 2], ppy = io.have_pos ? io.py : io.prev_pts[pidx * 2 + 1];
    // tracker mode: a corner that was lost (left the image, flat patch) is not picked up again by whatever texture
    // sits at its last position -- it stays lost, position carried, until the corner set is re-seeded
    {
        int pst = 1;
        if (io.have_pos) pst = io.pst;
        else if (P->prev_status) pst = P->prev_status[pidx];
        if (!agt_uniform(pst)) {
            if (tid == 0) lk_publish(io, pidx, b, ppx, ppy, 0, 0.f);
            ox = ppx; oy = ppy; ost = level_stop > 0 ? -1 : 0;
            return;
        }
    }
    float outx = 0.f, outy = 0.f;              // nextPts[ptidx]
    if (P->flags & AGT_LK_USE_INITIAL_FLOW) { outx = io.next_pts[pidx * 2]; outy = io.next_pts[pidx * 2 + 1]; }
    {
        // a wild position, or a wild initial flow: the window is outside the image at every level -- status 0, err 0, and nextPts is
        // what OpenCV leaves there, the start of the search scaled down and up again: the flow if one was given, else the position
        const bool flow = (P->flags & AGT_LK_USE_INITIAL_FLOW) != 0;
        if (!agt_uniform((int)(lk_pt_ok(ppx, ppy) && (!flow || lk_pt_ok(outx, outy))))) {
            const float cx = flow ? outx : ppx, cy = flow ? outy : ppy;
            if (tid == 0) lk_publish(io, pidx, b, cx, cy, 0, 0.f);
            ox = cx; oy = cy; ost = level_stop > 0 ? -1 : 0;
            return;
        }
    }
    const float gsx = (P->flags & AGT_LK_USE_INITIAL_FLOW) ? outx : ppx;     // where the search is expected to start
    const float gsy = (P->flags & AGT_LK_USE_INITIAL_FLOW) ? outy : ppy;

    STAMP(0);
    // ---- prologue: request every level's tiles before touching any of them (four waves per corner, the latency form), or -- one wave per
    // corner, where the 3 + 7 dwords per lane and level are 30 registers of a 128-register kernel -- level by level (round 6: the corners
    // that still come here with one wave are the few whose windows touch the image border; the others run the row-segment body)
    if constexpr (NW == 1) {
#pragma unroll
        for (int l = 0; l < NLEV; l++) {
            if (l >= STOP && l <= P->max_level) {
                const float scale = lk_level_scale(l);
                const int ipx = (int)floorf(ppx * scale - halfw), ipy = (int)floorf(ppy * scale - halfw);
                const int jx0 = (int)floorf(gsx * scale - halfw) - C::MARGIN, jy0 = (int)floorf(gsy * scale - halfw) - C::MARGIN;
                AgtLevel LI = get_level(P->prev[l]);
                AgtLevel LJ = get_level(P->next[l]);
                if (io.grouped) { LI.ptr = io.imgI[l]; LJ.ptr = io.imgJ[l]; }
                uint32_t ti[C::ILD], tj[C::JLD];
#pragma unroll
                for (int k = 0; k < C::ILD; k++) ti[k] = 0;
#pragma unroll
                for (int k = 0; k < C::JLD; k++) tj[k] = 0;
                if (!(ipx < -WIN || ipx >= LI.w || ipy < -WIN || ipy >= LI.h)) {
                    tile_load<C::IW, C::INDW, T>(LI.ptr + (long)b * LI.bstride, LI.w, LI.h, LI.pitch, ipx - 1, ipy - 1, tid, ti);
                    tile_load<C::JT, C::JNDW, T>(LJ.ptr + (long)b * LJ.bstride, LJ.w, LJ.h, LJ.pitch, jx0, jy0, tid, tj);
                }
                tile_store<C::IW, C::INDW, T>(lds + l * C::LEVEL_LDS, tid, ti);
                tile_store<C::JT, C::JNDW, T>(lds + l * C::LEVEL_LDS + C::IW * C::IP, tid, tj);
            }
        }
    } else {
        uint32_t ti[NLEV][C::ILD], tj[NLEV][C::JLD];
#pragma unroll
        for (int l = 0; l < NLEV; l++) {
            if (l >= STOP && l <= P->max_level) {
                const float scale = lk_level_scale(l);
                const int ipx = (int)floorf(ppx * scale - halfw), ipy = (int)floorf(ppy * scale - halfw);
                const int jx0 = (int)floorf(gsx * scale - halfw) - C::MARGIN, jy0 = (int)floorf(gsy * scale - halfw) - C::MARGIN;
                AgtLevel LI = get_level(P->prev[l]);
                AgtLevel LJ = get_level(P->next[l]);
                if (io.grouped) { LI.ptr = io.imgI[l]; LJ.ptr = io.imgJ[l]; }
                if (!(ipx < -WIN || ipx >= LI.w || ipy < -WIN || ipy >= LI.h)) {
                    tile_load<C::IW, C::INDW, T>(LI.ptr + (long)b * LI.bstride, LI.w, LI.h, LI.pitch, ipx - 1, ipy - 1, tid, ti[l]);
                    tile_load<C::JT, C::JNDW, T>(LJ.ptr + (long)b * LJ.bstride, LJ.w, LJ.h, LJ.pitch, jx0, jy0, tid, tj[l]);
                }
            }
        }
        STAMP(1);
#pragma unroll
        for (int l = 0; l < NLEV; l++) {
            if (l >= STOP && l <= P->max_level) {
                tile_store<C::IW, C::INDW, T>(lds + l * C::LEVEL_LDS, tid, ti[l]);
                tile_store<C::JT, C::JNDW, T>(lds + l * C::LEVEL_LDS + C::IW * C::IP, tid, tj[l]);
            }
        }
    }
    block_sync<NW>();
    STAMP(2);

    int st = 1;
    float errv = 0.f;

    for (int level = P->max_level; level >= level_stop; level--) {
        const AgtLevel LI = get_level(P->prev[level]);
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
        const float scale = lk_level_scale(level);
        float prevx = ppx * scale, prevy = ppy * scale;
        float nextx, nexty;
        if (level == P->max_level) {
            if (P->flags & AGT_LK_USE_INITIAL_FLOW) { nextx = outx * scale; nexty = outy * scale; }
            else { nextx = prevx; nexty = prevy; }
        } else { nextx = outx * 2.f; nexty = outy * 2.f; }
        outx = nextx; outy = nexty;
        STAMP(8 + level * 8 + 0);

        prevx -= halfw; prevy -= halfw;
        const int ipx = agt_uniform((int)floorf(prevx)), ipy = agt_uniform((int)floorf(prevy));
        if (ipx < -WIN || ipx >= LI.w || ipy < -WIN || ipy >= LI.h) {
            if (STOP == 0 && level == 0) { st = 0; errv = 0.f; }
            continue;
        }
        int iw00, iw01, iw10, iw11;
        bilinear_weights(prevx - (float)ipx, prevy - (float)ipy, iw00, iw01, iw10, iw11);

        // ---- Scharr on the fly from the prefetched I tile -> LDS derivative tile
        const int offI = (ipx - 1) - ((ipx - 1) & ~3);
        block_sync<NW>();                      // previous level's readers of sD are done
        // (one wave per corner: the thread index through an empty asm, per level -- the row / column split of the NSD derivative positions
        // of a lane is then recomputed in every level instead of being kept alive across the level loop, where it was the value the
        // register allocator spilled at 128 registers: round 6.  Four waves per corner: as ever.)
        int tid_l = tid;
        if constexpr (NW == 1) asm volatile("" : "+v"(tid_l));
#pragma unroll
        for (int k = 0; k < C::NSD; k++) {
            const int idx = tid_l + k * T;
            if (idx < C::DW * C::DW) {
                const int dyy = idx / C::DW, dxx = idx - dyy * C::DW;
                const int gx = ipx + dxx, gy = ipy + dyy;
                int val = 0;      // derivative image has a ZERO (BORDER_CONSTANT) border
                if (gx >= 0 && gx < LI.w && gy >= 0 && gy < LI.h) {
                    const uint8_t* c = sI + (dyy + 1) * C::IP + (dxx + 1) + offI;
                    const int v00 = c[-C::IP - 1], v01 = c[-C::IP], v02 = c[-C::IP + 1];
                    const int v10 = c[-1], v12 = c[1];
                    const int v20 = c[C::IP - 1], v21 = c[C::IP], v22 = c[C::IP + 1];
                    const int dx = (3 * (v02 + v22) + 10 * v12) - (3 * (v00 + v20) + 10 * v10);
                    const int dy = 3 * ((v20 - v00) + (v22 - v02)) + 10 * (v21 - v01);
                    val = (dx & 0xffff) | (dy << 16);
                }
                sD[idx] = val;
            }
        }
        block_sync<NW>();
        STAMP(8 + level * 8 + 1);

        // ---- per-thread patch (registers) + exact covariance sums
        int Iv[C::NPX], Ix[C::NPX], Iy[C::NPX];
        int asum[3] = { 0, 0, 0 };
#pragma unroll
        for (int k = 0; k < C::NPX; k++) {
            const uint8_t* p = sI + oI[k] + offI;
            const int iv = descale(bil4(p[0], p[1], p[C::IP], p[C::IP + 1], iw00, iw01, iw10, iw11), W_BITS - 5);
            const int* d = sD + oD[k];
            const int d00 = d[0], d01 = d[1], d10 = d[C::DW], d11 = d[C::DW + 1];
            const int ix = descale(bil4((short)d00, (short)d01, (short)d10, (short)d11, iw00, iw01, iw10, iw11), W_BITS);
            const int iy = descale(bil4(d00 >> 16, d01 >> 16, d10 >> 16, d11 >> 16, iw00, iw01, iw10, iw11), W_BITS);
            Iv[k] = pv[k] ? iv : 0; Ix[k] = pv[k] ? ix : 0; Iy[k] = pv[k] ? iy : 0;
            asum[0] += __mul24(Ix[k], Ix[k]); asum[1] += __mul24(Ix[k], Iy[k]); asum[2] += __mul24(Iy[k], Iy[k]);
        }
        long long at[3];
        block_sum_exact<NW, C::SUM_STEPS, 3, C::PAIR_OK>(asum, at, slots, phase, wave, lane);     // |Ix Iy| <= 4080^2 < 8160 * 4080
        const float A11 = (float)(double)at[0] * FLT_SCALE;
        const float A12 = (float)(double)at[1] * FLT_SCALE;
        const float A22 = (float)(double)at[2] * FLT_SCALE;

        float D = A11 * A22 - A12 * A12;
        const float minEig = (A22 + A11 - sqrtf((A11 - A22) * (A11 - A22) + 4.f * A12 * A12)) / (float)(2 * WIN * WIN);
        if (P->flags & AGT_LK_GET_MIN_EIGENVALS) errv = minEig;
        if (agt_uniform((int)((double)minEig < P->min_eig_threshold || D < FLT_EPSILON))) {
            if (STOP == 0 && level == 0) st = 0;
            continue;
        }
        D = 1.f / D;
        STAMP(8 + level * 8 + 2);

        nextx -= halfw; nexty -= halfw;
        float pdx = 0.f, pdy = 0.f;
        // the prefetched J tile of this level
        int jx0 = (int)floorf(gsx * scale - halfw) - C::MARGIN, jy0 = (int)floorf(gsy * scale - halfw) - C::MARGIN;
        auto restage_j = [&](int inx, int iny) {
            jx0 = inx - C::MARGIN; jy0 = iny - C::MARGIN;
            uint32_t t[C::JLD];
            block_sync<NW>();
            tile_load<C::JT, C::JNDW, T>(imgJ, LJ.w, LJ.h, LJ.pitch, jx0, jy0, tid, t);
            tile_store<C::JT, C::JNDW, T>(sJ, tid, t);
            block_sync<NW>();
        };
        // sum of the window's temporal differences against the patch: {diff*Ix, diff*Iy} or {|diff|}
        auto window_pass = [&](int inx, int iny, bool want_abs, int (&acc)[2]) {
            acc[0] = 0; acc[1] = 0;
            const uint8_t* q0 = sJ + (iny - jy0) * C::JP + (inx - jx0) + (jx0 - (jx0 & ~3));
#pragma unroll
            for (int k = 0; k < C::NPX; k++) {
                const uint8_t* q = q0 + oJ[k];
                const int diff = descale(bil4(q[0], q[1], q[C::JP], q[C::JP + 1], iw00, iw01, iw10, iw11), W_BITS - 5) - Iv[k];
                if (want_abs) acc[0] += pv[k] ? (diff < 0 ? -diff : diff) : 0;
                else { acc[0] += __mul24(diff, Ix[k]); acc[1] += __mul24(diff, Iy[k]); }   // Ix = Iy = 0 where !pv
            }
        };
        for (int j = 0; j < P->max_count; j++) {
            nit++;
            if (j == 1) STAMP(39);
            const int inx = agt_uniform((int)floorf(nextx)), iny = agt_uniform((int)floorf(nexty));
            if (inx < -WIN || inx >= LJ.w || iny < -WIN || iny >= LJ.h) {
                if (STOP == 0 && level == 0) st = 0;
                break;
            }
            if (j == 1) STAMP(40);
            if (inx < jx0 || inx + WIN >= jx0 + C::JT || iny < jy0 || iny + WIN >= jy0 + C::JT) restage_j(inx, iny);
            bilinear_weights(nextx - (float)inx, nexty - (float)iny, iw00, iw01, iw10, iw11);
            if (j == 1) STAMP(41);
            int bsum[2];
            window_pass(inx, iny, false, bsum);
            if (j == 1) STAMP(42);
            long long bt[2];
            block_sum_exact<NW, C::SUM_STEPS, 2, C::PAIR_OK>(bsum, bt, slots, phase, wave, lane);
            if (j == 1) STAMP(43);
            const float fb1 = (float)(double)bt[0] * FLT_SCALE;
            const float fb2 = (float)(double)bt[1] * FLT_SCALE;
            const float dx = (A12 * fb2 - A22 * fb1) * D;
            const float dy = (A12 * fb1 - A11 * fb2) * D;
            nextx += dx; nexty += dy;
            outx = nextx + halfw; outy = nexty + halfw;
            if (j == 1) STAMP(44);
            if (j == 0) STAMP(8 + level * 8 + 3);
#ifdef AGT_LK_STAMPS
            if (pidx == 0 && threadIdx.x == 0) agt_lk_stamps[8 + level * 8 + 6] = j + 1;
#endif
            if (agt_uniform((int)((double)dx * dx + (double)dy * dy <= P->eps2))) break;
            if (j > 0 && agt_uniform((int)(fabs((double)(dx + pdx)) < 0.01 && fabs((double)(dy + pdy)) < 0.01))) {
                outx -= dx * 0.5f; outy -= dy * 0.5f;
                break;
            }
            pdx = dx; pdy = dy;
            if (j == 1) STAMP(45);
        }

        STAMP(8 + level * 8 + 4);
        if (STOP == 0 && st && io.err && level == 0 && !(P->flags & AGT_LK_GET_MIN_EIGENVALS)) {
            const float npx = outx - halfw, npy = outy - halfw;
            const int inx = agt_uniform((int)floorf(npx)), iny = agt_uniform((int)floorf(npy));
            if (inx < -WIN || inx >= LJ.w || iny < -WIN || iny >= LJ.h) { st = 0; continue; }
            if (inx < jx0 || inx + WIN >= jx0 + C::JT || iny < jy0 || iny + WIN >= jy0 + C::JT) restage_j(inx, iny);
            bilinear_weights(npx - (float)inx, npy - (float)iny, iw00, iw01, iw10, iw11);
            int esum[2];
            window_pass(inx, iny, true, esum);
            int e1[1] = { esum[0] };
            long long et[1];
            block_sum_exact<NW, C::SUM_STEPS, 1>(e1, et, slots, phase, wave, lane);
            errv = (float)(double)et[0] * 1.f / (float)(32 * WIN * WIN);
        }
    }

    STAMP(3);
    if (level_stop > 0) { ox = outx; oy = outy; ost = st; return; }       // (the finer levels and the result are the caller's)
    if (tid == 0) lk_publish(io, pidx, b, outx, outy, st, errv);
    if (tid == 0 && io.iters_out) io.iters_out[pidx] = (uint8_t)(nit > 255 ? 255 : nit);
    ox = outx; oy = outy; ost = st;
}

}  // namespace agt_lk
