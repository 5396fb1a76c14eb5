// agt_lk.hip -- stand-alone cv::calcOpticalFlowPyrLK launch (body and design notes: agt_lk_body.h).
#include <cstdlib>
#include "agt_lk_rs_body.h"
#include "agt_lk_any_body.h"

namespace {

// OCC: waves per SIMD the register allocation leaves room for.  The one-wave-per-corner 21x21 kernel sits 3
// registers above the 128 that allow a fourth wave; big batches are throughput-bound, so it is held to 128.
// XCD-aware corner order (agt_kernels.h agt_xcd_order): workgroups are dealt round-robin to the X XCDs (each with its own L2), so workgroup g takes corner
// (g % X) * per_xcd + g / X of the launch's n * B -- every XCD walks a CONTIGUOUS run of corners, i.e. whole streams: the tiles of
// a tag's four corners (and of neighbouring tags) overlap, and with consecutive corners on eight different XCDs each of those
// L2s fetched the shared lines from HBM for itself (FETCH_SIZE 2.9x the algorithmic bytes at 64 streams).
template <int WIN, int NW, int NLEV, int OCC>
#ifdef AGT_LK_NUM_VGPR       // experiment builds (tools/build_variant.sh): a register ceiling below the occupancy attribute's 128
#define AGT_LK_VGPR_ATTR __attribute__((amdgpu_num_vgpr(AGT_LK_NUM_VGPR)))
#else
#define AGT_LK_VGPR_ATTR
#endif
__global__ __launch_bounds__(AGT_WAVE * NW) __attribute__((amdgpu_waves_per_eu(OCC))) AGT_LK_VGPR_ATTR void lk_kernel(const AgtLkParams P, const int total)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
#ifdef AGT_LK_TOUCH_VGPR     // experiment builds: a clobber of one high register raises the kernel's allocation (e.g. v135: 136 registers = at most
                             // three of its waves on a SIMD, whatever the CU's LDS would allow)
#define AGT_STR2(x) #x
#define AGT_STR(x) AGT_STR2(x)
    if (NW == 1) asm volatile("" ::: "v" AGT_STR(AGT_LK_TOUCH_VGPR));
#endif
    // issue priority 1 for every tracker wave of a context that has DECLARED co-tenancy (agt_lk_occupancy_cu with a count: its launches share
    // the device with other contexts' pyramid passes, agt_api.hip lk_track_on sets the internal flag): cold pairs 44.4-44.6 -> 43.3-43.4 us per
    // step (priority 2: the same).  Not for the split pipeline's own launches: c3 37.3-37.6 -> 37.8-38.1 (profiles/r06_experiments.md 17)
    const bool cotenant = NW == 1 && (P.flags & AGT_LK_FLAG_COTENANT) != 0;
    if (cotenant) __builtin_amdgcn_s_setprio(1);
    const int cidx = agt_xcd_order((int)blockIdx.x, (int)gridDim.x, P.xshift);
    if (cidx >= total) return;
    const int bY = cidx / P.n, bX = cidx - bY * P.n;          // stream, corner
    agt_lk::LkFrameIo<NLEV> io;
    io.grouped = false; io.prev_pts = P.prev_pts; io.next_pts = P.next_pts; io.status = P.status; io.err = P.err;
    io.have_pos = false; io.px = io.py = 0.f; io.pst = 1;
    float ox, oy; int ost;
#ifdef AGT_LK_STAMPS
    struct CornerLog {
        int c; unsigned long long t0;
        __device__ CornerLog(int c_) : c(c_), t0(__builtin_amdgcn_s_memtime()) {}
        __device__ ~CornerLog() { if ((threadIdx.x & 63) == 0 && threadIdx.x == 0 && c < AGT_LK_CORNER_LOG) { agt_lk_corner_log[c][0] = t0; agt_lk_corner_log[c][1] = __builtin_amdgcn_s_memtime();
                                  agt_lk_corner_log[c][3] = (unsigned long long)__builtin_amdgcn_s_getreg(63492) | ((unsigned long long)__builtin_amdgcn_s_getreg(63508) << 32); } }      // HW_REG_HW_ID, HW_REG_XCC_ID
    } corner_log(cidx);
#endif
    if constexpr (WIN == 21 && (NW == 1 || NW == 4)) {
        // one wave per corner: the row-segment body (agt_lk_rs_body.h) while the window's derivative footprint stays inside
        // the image at every level and the corner is alive; the general body otherwise (wave-uniform choice)
        const long pidx = cidx;
        const float ppx = P.prev_pts[pidx * 2], ppy = P.prev_pts[pidx * 2 + 1];
        const int pst = P.prev_status ? P.prev_status[pidx] : 1;
        // (round 6, tools/lkcorners.py: on 64 streams a wave lives 16 .. 31 us on the row-segment body and 35 us on the general one, which the
        // ~3 % of corners whose window touches the image border at SOME level took for ALL levels -- and the launch lasts as long as its
        // slowest wave.  The border is mostly reached at the coarsest level only: the general body now tracks just the coarse levels that
        // need it and hands the position to the row-segment body for the fine ones -- two bodies in sequence, each register-allocated alone.)
        const int fine = (pst != 0 && !(P.flags & 0x10000)) ? agt_uniform(agt_lk::rs_interior_levels<NLEV>(ppx, ppy, P.max_level, P.prev[0].w, P.prev[0].h)) : 0;
        // One split point is compiled: the coarsest level alone (by far the common case -- its window covers four times the ground of the
        // finest one's) of a full-depth pyramid; every other corner that needs the general body somewhere gets it everywhere, as before.
        // One call site per body: the general one runs all levels, result included, or the coarsest alone; the row-segment one the rest.
        int top = -1;
        float cx = 0.f, cy = 0.f;
        if (fine <= P.max_level) {
            // (the launch lasts as long as its slowest wave, and that is one of these: the general body's waves issue ahead of the row-segment
            // waves they share a SIMD with -- those have slack, tools/lkcorners.py)
            if (NW == 1 && pst != 0) __builtin_amdgcn_s_setprio(3);
            if (fine != NLEV - 1 || P.max_level != NLEV - 1) { agt_lk::lk_body<WIN, NW, NLEV>(&P, bX, bY, lds, io, ox, oy, ost); return; }
            agt_lk::lk_body<WIN, NW, NLEV, const AgtLkParams*, NLEV - 1>(&P, bX, bY, lds, io, ox, oy, ost);
            if (ost < 0) return;                           // (finished there: a wild initial flow)
            agt_lk::block_sync<NW>();                      // (the row-segment body reuses the LDS behind the tiles)
            top = fine - 1; cx = ox; cy = oy;
            if (NW == 1) { if (cotenant) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0); }
        }
        agt_lk::lk_body_rs<NW, NLEV>(&P, bX, bY, lds, io, ppx, ppy, ox, oy, ost, top, cx, cy);
        return;
    }
    agt_lk::lk_body<WIN, NW, NLEV>(&P, bX, bY, lds, io, ox, oy, ost);
}

#ifdef AGT_DEBUG_KNOBS      // measured and not shipped: agt_api.hip lk_track_on
// HYBRID launch (round 5, win 21, big batches): 256-thread workgroups in two roles.  Workgroups [0, n4) are the FOUR-WAVE role, one
// per corner (XCD-aware order): the corner is tracked here if it took >= slow_thr iterations in the previous frame, else the
// workgroup exits at once.  Workgroups [n4, n4 + nquad) are the ONE-WAVE role: wave w tracks corner 4 q + w (q in XCD-aware order:
// the four corners of a tag share a workgroup, their overlapping tiles one CU's L1) unless the four-wave role has it.  The slow
// corners are dispatched first -- they are the launch's critical path.  Register budget of four waves per SIMD (the one-wave
// body's 128; the four-wave bodies need 67).
template <int NLEV>
__global__ __launch_bounds__(AGT_WAVE * 4) __attribute__((amdgpu_waves_per_eu(4))) void lk_hybrid_kernel(const AgtLkParams P, const int total, const int n4,
                                                                                                           const int nquad, const int per_wave_lds)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const int bid = (int)blockIdx.x;
    const bool four = bid < n4;
    int cidx;
    uint8_t* my = lds;
    if (four) cidx = agt_xcd_order(bid, n4, P.xshift);
    else {
        const int wave = agt_uniform((int)threadIdx.x >> 6);          // (scalar: the wave's LDS base stays out of the vector registers)
        cidx = agt_xcd_order(bid - n4, nquad, P.xshift) * 4 + wave;
        my = lds + wave * per_wave_lds;
    }
    if (cidx >= total) return;
    const float ppx = P.prev_pts[(long)cidx * 2], ppy = P.prev_pts[(long)cidx * 2 + 1];
    const int pst = P.prev_status ? P.prev_status[cidx] : 1;
    const bool rs = agt_uniform((int)(pst != 0 && !(P.flags & 0x10000) && agt_lk::rs_interior(ppx, ppy, P.max_level, P.prev[0].w, P.prev[0].h))) != 0;
    // four waves only for corners whose windows stay inside the image (the row-segment body); a slow corner at the image border keeps
    // its one wave and the general body -- three tracker bodies in this kernel instead of four (with four, 14 VGPRs spilled)
    const bool slow = rs && agt_uniform((int)(P.iters_prev[cidx] >= P.slow_thr)) != 0;
    if (slow != four) return;
    const int bY = cidx / P.n, bX = cidx - bY * P.n;          // stream, corner
    agt_lk::LkFrameIo<NLEV> io;
    io.grouped = false; io.prev_pts = P.prev_pts; io.next_pts = P.next_pts; io.status = P.status; io.err = P.err;
    io.have_pos = false; io.px = io.py = 0.f; io.pst = 1;
    io.iters_out = P.iters_out;
    float ox, oy; int ost;
    if (four) {
        agt_lk::lk_body_rs<4, NLEV>(&P, bX, bY, my, io, ppx, ppy, ox, oy, ost);
    } else {
        if (rs) agt_lk::lk_body_rs<1, NLEV>(&P, bX, bY, my, io, ppx, ppy, ox, oy, ost);
        else agt_lk::lk_body<21, 1, NLEV>(&P, bX, bY, my, io, ox, oy, ost);
    }
}

#endif

template <int WIN, int NW>
hipError_t launch_lk_t(hipStream_t stream, const AgtLkParams& p_in, int B)
{
    AgtLkParams p = p_in;
    p.xshift = agt_chip_current().xshift;
    size_t lds = agt_lk::lk_lds_bytes<WIN, NW>(p.max_level + 1);
    if (NW == 1 && (size_t)p.lds_min > lds) lds = (size_t)p.lds_min;          // agt_lk_occupancy_cu: fewer resident LK waves per CU
#ifdef AGT_DEBUG_KNOBS      // AGT_LK_LDS_PAD=bytes: extra LDS per workgroup = fewer LK waves per CU (room for other kernels' waves beside them)
    { static const long pad = [] { const char* e = getenv("AGT_LK_LDS_PAD"); return e ? atol(e) : 0L; }(); if (pad > 0 && NW == 1) lds += (size_t)pad; }
#endif
    const long total = (long)p.n * B;
    if (total <= 0 || total > (1L << 30)) return hipErrorInvalidValue;
    const dim3 grid(agt_xcd_grid(total, p.xshift)), block(AGT_WAVE * NW);
    constexpr int OCC = (WIN == 21 && NW == 1) ? 4 : 1;
    // (pyramids of more than three levels: the six-level one-wave body keeps six levels' tile bookkeeping alive and spilled 34 VGPRs
    // at the 128 registers of four waves per SIMD; it gets the 168 of three -- round 5, no scratch left in the library)
    constexpr int OCC6 = (WIN == 21 && NW == 1) ? 3 : 1;
    if (p.max_level < 3) hipLaunchKernelGGL((lk_kernel<WIN, NW, 3, OCC>), grid, block, lds, stream, p, (int)total);
    else hipLaunchKernelGGL((lk_kernel<WIN, NW, AGT_MAX_LEVELS, OCC6>), grid, block, lds, stream, p, (int)total);
    return hipGetLastError();
}

// any window (agt_lk_any_body.h): one workgroup of four waves per corner, run-time window size
__global__ __launch_bounds__(agt_lk::ANY_T) void lk_any_kernel(const AgtLkParams P, const int total, const int ww, const int wh)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const int cidx = agt_xcd_order((int)blockIdx.x, (int)gridDim.x, P.xshift);
    if (cidx >= total) return;
    const int bY = cidx / P.n, bX = cidx - bY * P.n;          // stream, corner
    agt_lk::lk_body_any(&P, bX, bY, lds, ww, wh);
}

hipError_t launch_lk_any(hipStream_t stream, const AgtLkParams& p_in, int ww, int wh, int B)
{
    AgtLkParams p = p_in;
    p.xshift = agt_chip_current().xshift;
    const long total = (long)p.n * B;
    if (total <= 0 || total > (1L << 30)) return hipErrorInvalidValue;
    const size_t lds = (size_t)agt_lk::any_geom(ww, wh).bytes;
    hipLaunchKernelGGL(lk_any_kernel, dim3(agt_xcd_grid(total, p.xshift)), dim3(agt_lk::ANY_T), lds, stream, p, (int)total, ww, wh);
    return hipGetLastError();
}

}  // namespace

// `win` of AgtConfig: a square window's side, or AGT_WIN_RECT(w, h) = w | h << 8.  15, 21 and 31 have compiled-in bodies (21 x 21, the
// north-star's window, the specialised ones); every other size from 3 x 3 to 63 x 63, square or not, runs the general body.
void agt_lk_window_size(int win, int* ww, int* wh) { *ww = win & 0xff; *wh = (win >> 8) ? (win >> 8) & 0xff : *ww; }
bool agt_lk_window_supported(int win)
{
    if (win <= 0 || (win >> 16)) return false;
    int ww, wh;
    agt_lk_window_size(win, &ww, &wh);
    return ww >= 3 && wh >= 3 && ww <= agt_lk::ANY_WIN_MAX && wh <= agt_lk::ANY_WIN_MAX;
}

// waves per corner: 4 while the launch cannot fill the chip with single-wave corners (latency
// matters), 1 for large batches (throughput matters)
bool agt_lk_wide(int n, int B)
{
#ifdef AGT_DEBUG_KNOBS
    static const long cap = [] { const char* e = getenv("AGT_LK_WIDE_MAX"); return e ? atol(e) : 1024L; }();
#else
    const long cap = 1024;
#endif
    return (long)n * B <= cap;
}

#ifdef AGT_DEBUG_KNOBS
hipError_t agt_launch_lk_hybrid(hipStream_t stream, const AgtLkParams& p_in, int B)
{
    AgtLkParams p = p_in;
    if (!p.iters_prev || p.slow_thr <= 0 || p.max_level >= 3 || p.err) return hipErrorInvalidValue;
    p.xshift = agt_chip_current().xshift;
    const long total = (long)p.n * B;
    if (total <= 0 || total > (1L << 28)) return hipErrorInvalidValue;
    const size_t lds1 = agt_lk::lk_lds_bytes<21, 1>(p.max_level + 1), lds4 = agt_lk::lk_lds_bytes<21, 4>(p.max_level + 1);
    const size_t per = (lds1 + 15) & ~(size_t)15;
    size_t lds = 4 * per > lds4 ? 4 * per : lds4;
    const unsigned n4 = agt_xcd_grid(total, p.xshift), nquad = agt_xcd_grid((total + 3) / 4, p.xshift);
    hipLaunchKernelGGL((lk_hybrid_kernel<3>), dim3(n4 + nquad), dim3(AGT_WAVE * 4), lds, stream, p, (int)total, (int)n4, (int)nquad, (int)per);
    return hipGetLastError();
}
#else
hipError_t agt_launch_lk_hybrid(hipStream_t, const AgtLkParams&, int) { return hipErrorInvalidValue; }
#endif

hipError_t agt_launch_lk(hipStream_t stream, const AgtLkParams& p_in, int win, int B, int waves)
{
    AgtLkParams p = p_in;
#ifdef AGT_DEBUG_KNOBS      // diagnostic library only: AGT_LK_RS=0 keeps every corner on the general body (flag bit 16, internal)
    { static const int rs = [] { const char* e = getenv("AGT_LK_RS"); return e ? atoi(e) : 1; }(); if (!rs) p.flags |= 0x10000; }
#endif
    switch (win) {
    // (2 and 8 waves per corner were measured too: 2 loses to 1 on big batches -- 60 vs 42 us at 64 streams --, 8 loses
    // to 4 on small ones -- 19.5 vs 17.8 us)
    case 21: return (waves ? waves == 4 : agt_lk_wide(p.n, B)) ? launch_lk_t<21, 4>(stream, p, B) : launch_lk_t<21, 1>(stream, p, B);
    case 15: return launch_lk_t<15, 1>(stream, p, B);
    case 31: return launch_lk_t<31, 1>(stream, p, B);
    default: {
        if (!agt_lk_window_supported(win)) return hipErrorInvalidValue;
        int ww, wh;
        agt_lk_window_size(win, &ww, &wh);
        return launch_lk_any(stream, p, ww, wh, B);
    }
    }
}
