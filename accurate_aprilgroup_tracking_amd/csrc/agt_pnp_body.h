// agt_pnp_body.h -- device body of cv::solvePnP(SOLVEPNP_ITERATIVE) + cv::projectPoints for gfx950.
//
// Replaces the reference's calls at
//   /root/reference/aprilgroup_tracking/aprilgroup_pose_estimation/detect_pose.py:509-515 (no guess: DLT init + LM)
//   detect_pose.py:517-526 (useExtrinsicGuess=True: LM only)
//   transform_helper.py:106-111 / detect_pose.py:455-461 (projectPoints)
//   transform_helper.py:98-121 (mean reprojection error, fused into the solve's epilogue)
// Semantics: OpenCV calibration.cpp cvFindExtrinsicCameraParams2 + compat_ptsetreg.cpp
// CvLevMarq (<= 20 iterations, eps = FLT_EPSILON, lambda = 10^-3 initially), all FP64;
// restated on the CPU in oracle/cv_pnp.c.
//
// Mapping: ONE 64-lane wave per problem; lane l owns correspondences l, l+64, ... and
// evaluates residual + 2x6 Jacobian in FP64.  The 21 + 6 + 1 partial sums
// (upper J^T J, J^T e, |e|^2) cross lanes through a transposed LDS slab (two lanes per
// sum, 32 adds each) -- no MFMA: the contraction is 6x6.  The whole LM loop, the
// damped LDL^T solve and the termination test run inside the launch; the host sees one
// kernel per batch of B independent problems.
#pragma once
#include <type_traits>
#include <cstddef>
#include "agt_device.h"
#include "agt_kernels.h"

// In-kernel cycle stamps (diagnostic builds only: make dbg; never in the shipped library).
#ifdef AGT_PNP_STAMPS
__device__ unsigned long long agt_pnp_stamps[64];
#define PSTAMP(i) do { if (b == 0 && threadIdx.x == 0) agt_pnp_stamps[i] = __builtin_amdgcn_s_memtime(); } while (0)
// ... with a marker instruction the disassembly can be cut at (tools/pnp_eval_isa.py): s_mov_b32 sN, 0xbeef00 + i
#define PSTAMPM(i) do { int mk_; asm volatile("s_mov_b32 %0, %1" : "=s"(mk_) : "n"(0xbeef00 + (i))); PSTAMP(i); } while (0)
extern "C" int agt_debug_pnp_stamps(unsigned long long* host64)
{
    return (int)hipMemcpyFromSymbol(host64, HIP_SYMBOL(agt_pnp_stamps), sizeof(agt_pnp_stamps));
}
#else
#define PSTAMP(i)
#define PSTAMPM(i)
#endif

#pragma clang fp contract(fast)      // FP64 pose code only, see agt_device.h

namespace agt_pnp {

constexpr int NACC = 28;            // 21 (upper JtJ) + 6 (JtErr) + 1 (|err|^2)
constexpr int SLAB = 65;            // padded lane stride of the reduction slab (doubles)
constexpr int MAX_PPL = 4;          // points per lane -> n <= 256

// Every solver of this file is ONE wave (its workgroup may hold other waves doing something else -- the fused step's PnP role
// alternates two waves over the frames): cross-lane exchange through LDS needs no workgroup barrier, a wave's LDS operations
// execute in issue order; only the compiler must be kept from moving accesses across the point.
__device__ __forceinline__ void pnp_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

struct PnpShared {
    double part[NACC * SLAB];
    double tot[NACC + 4];
    double LL[144];
    double vec[48];
    unsigned long long tab[5 * AGT_MAX_GROUP];     // fused step: copy of AgtPnpTables (img / mask / so / wait / target per frame)
    int seq;                                       // fused step, two alternating waves: frames of this launch whose state update is complete
    int late;                                      // fused step: one of the two waves gave up a chained wait (the other stops waiting too)
    int coop[2];                                   // four-wave role: "frame k is solved from a guess" latched by wave 0 for frame parity k & 1 (pnp_role_coop)
    AgtTrackState ts;                              // pipelined roles: the stream's tracker state while the launch runs (see pnp_role)
};

// tracker state in LDS (pnp_role keeps it there for the frames of a launch: the hand-over from frame k to frame k + 1 is an
// LDS write / read instead of global stores that have to be acknowledged and loads that miss -- ~1.2 us per frame)
typedef __attribute__((address_space(3))) AgtTrackState AgtTrackStateLds;

// sum K per-lane partials across the wave; totals land in sh.tot[0..K) and (READBACK) come back in
// vals[] of every lane
template <int K, bool READBACK = true>
__device__ __forceinline__ void wave_reduce_slab(double (&vals)[K], PnpShared& sh, int lane)
{
    static_assert(K <= NACC, "the slab holds NACC rows");
    pnp_sync();
#pragma unroll
    for (int k = 0; k < K; k++) sh.part[k * SLAB + lane] = vals[k];
    pnp_sync();
    const int k = lane & 31, h = lane >> 5;
    double s = 0.0;
    if (k < K) {
        const double* p = &sh.part[k * SLAB + 32 * h];
#pragma unroll 8
        for (int i = 0; i < 32; i++) s += p[i];
    }
    s += __shfl_xor(s, 32);
    if (lane < K) sh.tot[lane] = s;
    pnp_sync();
    if (READBACK) {
#pragma unroll
        for (int i = 0; i < K; i++) vals[i] = sh.tot[i];
    }
}

// ---- register butterfly: 32 per-lane partials -> 32 wave totals without LDS traffic
// Recursive halving over the six lane bits.  Stages over bits 5 and 4 use gfx950's v_permlane32_swap /
// v_permlane16_swap (the swap leaves "my half" and "the partner's half" side by side: two swaps and one add
// per value, no selects); stages over bits 3..1 pair lanes through DPP row_mirror / row_half_mirror /
// quad_perm[3,2,1,0] (each pairs lanes that differ in the stage's bit and agree on the higher ones); the last
// stage adds the xor-1 neighbour.  Afterwards lane l holds the total of value l >> 1.
// ~125 VALU instructions in 6 short dependent stages, against 28 LDS stores + 32 dependent LDS load/adds.
__device__ __forceinline__ double dswap32(double a, double b, double& b_out)
{
    const auto lo = __builtin_amdgcn_permlane32_swap((unsigned)__double2loint(a), (unsigned)__double2loint(b), false, false);
    const auto hi = __builtin_amdgcn_permlane32_swap((unsigned)__double2hiint(a), (unsigned)__double2hiint(b), false, false);
    b_out = __hiloint2double((int)hi[1], (int)lo[1]);
    return __hiloint2double((int)hi[0], (int)lo[0]);
}
__device__ __forceinline__ double dswap16(double a, double b, double& b_out)
{
    const auto lo = __builtin_amdgcn_permlane16_swap((unsigned)__double2loint(a), (unsigned)__double2loint(b), false, false);
    const auto hi = __builtin_amdgcn_permlane16_swap((unsigned)__double2hiint(a), (unsigned)__double2hiint(b), false, false);
    b_out = __hiloint2double((int)hi[1], (int)lo[1]);
    return __hiloint2double((int)hi[0], (int)lo[0]);
}
template <int CTRL>
__device__ __forceinline__ double ddpp(double v)
{
    return __hiloint2double(agt_dpp_i32<CTRL>(__double2hiint(v)), agt_dpp_i32<CTRL>(__double2loint(v)));
}
template <int CTRL, int N>
__device__ __forceinline__ void bfly_stage(double (&v)[32], bool upper)
{
#pragma unroll
    for (int i = 0; i < N; i++) {
        const double a = v[i], b = v[i + N];
        const double keep = upper ? b : a, send = upper ? a : b;
        v[i] = keep + ddpp<CTRL>(send);
    }
}
// one double per lane summed over the wave, the total in every lane: four DPP row steps, then the two cross-row
// steps with the gfx950 lane swaps (no LDS crossbar: __shfl_xor costs two ds_bpermute per step)
__device__ __forceinline__ double wave_sum_f64(double v)
{
    v += ddpp<0xB1>(v);            // quad_perm [1,0,3,2]
    v += ddpp<0x4E>(v);            // quad_perm [2,3,0,1]
    v += ddpp<0x141>(v);           // row_half_mirror
    v += ddpp<0x140>(v);           // row_mirror: every lane holds its row's sum
    double o;
    const double a = dswap16(v, v, o);     // a: (r0, r0, r2, r2), o: (r1, r1, r3, r3) per row
    v = a + o;
    const double c = dswap32(v, v, o);     // c: lower-half sums everywhere, o: upper-half sums
    return c + o;
}

// vals[K] (K <= 32) summed over the wave; totals to tot[0..K) (LDS).  Ends with a barrier.
template <int K>
__device__ __forceinline__ void wave_reduce_bfly(const double (&vals)[K], double* tot, int lane)
{
    static_assert(K <= 32, "one pass handles at most 32 sums");
    double v[32];
#pragma unroll
    for (int i = 0; i < 32; i++) v[i] = i < K ? vals[i] : 0.0;
#pragma unroll
    for (int i = 0; i < 16; i++) { double b; const double a = dswap32(v[i], v[i + 16], b); v[i] = a + b; }
#pragma unroll
    for (int i = 0; i < 8; i++) { double b; const double a = dswap16(v[i], v[i + 8], b); v[i] = a + b; }
    bfly_stage<0x140, 4>(v, (lane & 8) != 0);      // row_mirror        l <-> l ^ 15
    bfly_stage<0x141, 2>(v, (lane & 4) != 0);      // row_half_mirror   l <-> l ^ 7
    bfly_stage<0x1B, 1>(v, (lane & 2) != 0);       // quad_perm [3,2,1,0]  l <-> l ^ 3
    const double t = v[0] + ddpp<0xB1>(v[0]);      // quad_perm [1,0,3,2]  l <-> l ^ 1
    pnp_sync();                               // earlier readers of tot[] are done
    if (!(lane & 1) && (lane >> 1) < K) tot[lane >> 1] = t;
    pnp_sync();
}

template <typename T>
__device__ __forceinline__ void load_cam(const AgtCameraHost& h, AgtCamera& c)
{
    c.fx = h.fx; c.fy = h.fy; c.cx = h.cx; c.cy = h.cy;
#pragma unroll
    for (int i = 0; i < 12; i++) c.k[i] = h.k[i];
    c.tilt = h.tilt;
}

// cvUndistortPointsInternal, criteria (COUNT, 5), R = I, no P
__device__ __forceinline__ void undistort5(const AgtCamera& cam, double u, double v, double& xo, double& yo)
{
    const double* k = cam.k;
    double x = (u - cam.cx) * (1.0 / cam.fx), y = (v - cam.cy) * (1.0 / cam.fy);
    const double xr = x, yr = y;
    if (cam.tilt) agt_tilt_apply(cam.tilt + 9, x, y, x, y, nullptr);       // compensate tilt distortion (invMatTilt)
    const double x0 = x, y0 = y;
    for (int j = 0; j < 5; j++) {
        double r2 = x * x + y * y;
        double icdist = (1 + ((k[7] * r2 + k[6]) * r2 + k[5]) * r2) / (1 + ((k[4] * r2 + k[1]) * r2 + k[0]) * r2);
        if (icdist < 0) { x = xr; y = yr; break; }
        double deltaX = 2 * k[2] * x * y + k[3] * (r2 + 2 * x * x) + k[8] * r2 + k[9] * r2 * r2;
        double deltaY = k[2] * (r2 + 2 * y * y) + 2 * k[3] * x * y + k[10] * r2 + k[11] * r2 * r2;
        x = (x0 - deltaX) * icdist;
        y = (y0 - deltaY) * icdist;
    }
    xo = x; yo = y;
}

// Smallest eigenvector of the symmetric PSD n x n matrix in A (row-major, LDS), by
// shifted inverse iteration on a Cholesky factor.  Serial: call from ONE lane.
// Gives the same vector (up to sign) as the last row of V^T in OpenCV's SVD of A.
__device__ void smallest_eigvec(double* A, int n, double* v, double* tmp)
{
    double dmax = 0;
    for (int i = 0; i < n; i++) dmax = fmax(dmax, A[i * n + i]);
    const double mu = dmax * 1e-11 + DBL_MIN;
    for (int j = 0; j < n; j++) {
        double d = A[j * n + j] + mu;
        for (int k = 0; k < j; k++) d -= A[j * n + k] * A[j * n + k];
        d = d > mu * 1e-3 ? d : mu * 1e-3;
        d = sqrt(d);
        A[j * n + j] = d;
        const double id = 1.0 / d;
        for (int i = j + 1; i < n; i++) {
            double s = A[i * n + j];
            for (int k = 0; k < j; k++) s -= A[i * n + k] * A[j * n + k];
            A[i * n + j] = s * id;
        }
    }
    for (int i = 0; i < n; i++) v[i] = 1.0 / sqrt((double)n) * ((i & 1) ? 0.9 : 1.1);
    for (int it = 0; it < 12; it++) {
        for (int i = 0; i < n; i++) {
            double s = v[i];
            for (int k = 0; k < i; k++) s -= A[i * n + k] * tmp[k];
            tmp[i] = s / A[i * n + i];
        }
        for (int i = n - 1; i >= 0; i--) {
            double s = tmp[i];
            for (int k = i + 1; k < n; k++) s -= A[k * n + i] * v[k];
            v[i] = s / A[i * n + i];
        }
        double nn = 0;
        for (int i = 0; i < n; i++) nn += v[i] * v[i];
        nn = 1.0 / sqrt(nn);
        for (int i = 0; i < n; i++) v[i] *= nn;
    }
}

// Cholesky solve of the symmetric positive definite n x n system A x = b (A row-major in LDS, overwritten by its factor);
// inv_diag != null: the diagonal of A^-1 instead / as well.  Serial: call from ONE lane.  false = not positive definite.
// (Always expanded in place: n is a constant at both call sites and `col` then lives in registers; as a function of its own --
// the inliner's choice in a translation unit with few callers -- it indexes `col` dynamically and brings 112 B of scratch.)
__device__ __forceinline__ bool spd_solve(double* A, int n, const double* b, double* x, double* inv_diag)
{
    for (int j = 0; j < n; j++) {
        double d = A[j * n + j];
        for (int k = 0; k < j; k++) d -= A[j * n + k] * A[j * n + k];
        if (!(d > 0.0)) return false;
        d = sqrt(d);
        A[j * n + j] = d;
        const double id = 1.0 / d;
        for (int i = j + 1; i < n; i++) {
            double s = A[i * n + j];
            for (int k = 0; k < j; k++) s -= A[i * n + k] * A[j * n + k];
            A[i * n + j] = s * id;
        }
    }
    if (x) {
        for (int i = 0; i < n; i++) {
            double s = b[i];
            for (int k = 0; k < i; k++) s -= A[i * n + k] * x[k];
            x[i] = s / A[i * n + i];
        }
        for (int i = n - 1; i >= 0; i--) {
            double s = x[i];
            for (int k = i + 1; k < n; k++) s -= A[k * n + i] * x[k];
            x[i] = s / A[i * n + i];
        }
    }
    if (inv_diag) {
        // A^-1 = L^-T L^-1: column c of L^-1 by forward substitution of e_c; diag_i = sum_c (L^-1)_{c i}^2 over c >= i
        for (int i = 0; i < n; i++) inv_diag[i] = 0.0;
        for (int c = 0; c < n; c++) {
            double col[12];
            for (int i = 0; i < n; i++) {
                double s = i == c ? 1.0 : 0.0;
                for (int k = c; k < i; k++) s -= A[i * n + k] * col[k];
                col[i] = i < c ? 0.0 : s / A[i * n + i];
            }
            // (L^-1)_{i c} = col[i]; (A^-1)_{cc} = sum_{i >= c} (L^-1)_{i c}^2
            double dsum = 0.0;
            for (int i = c; i < n; i++) dsum += col[i] * col[i];
            inv_diag[c] = dsum;
        }
    }
    return true;
}

// broadcast lane `src`'s double to the whole wave through SGPRs (no LDS)
__device__ __forceinline__ double lane_bcast(double v, int src)
{
    const long long bits = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_readlane((int)bits, src), hi = __builtin_amdgcn_readlane((int)(bits >> 32), src);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

// Smallest eigenvector of the symmetric PSD N x N matrix A (row-major, LDS), by the WHOLE wave: the same shifted inverse
// iteration on a Cholesky factor as smallest_eigvec, with lane i owning row i -- column j of the factor needs one broadcast
// of row j, a triangular solve is N broadcast steps.  ~2,700 instructions against ~170 us of serial LDS round trips on one
// lane (cv2-shaped solvePnP without a guess: 211 -> 60 us per call).  A is used as scratch; v_out: N doubles in LDS.
// Ends with the result visible to every lane (barrier inside).
template <int N>
__device__ __forceinline__ void smallest_eigvec_wave(double* A, double* v_out, int lane)
{
    const int i = lane < N ? lane : N - 1;               // (lanes >= N mirror the last row; their results are dropped)
    double a[N], l[N], lt[N];
    double dmax = 0.0;
#pragma unroll
    for (int k = 0; k < N; k++) { a[k] = A[i * N + k]; dmax = fmax(dmax, A[k * N + k]); }
    const double mu = dmax * 1e-11 + DBL_MIN;
    // ---- Cholesky, A + mu I = L L^T; l[k] = L[i][k] (k <= i meaningful)
#pragma unroll
    for (int j = 0; j < N; j++) {
        double d = a[j] + mu, s = a[j];
#pragma unroll
        for (int k = 0; k < j; k++) {
            const double ljk = lane_bcast(l[k], j);      // L[j][k]
            d -= l[k] * l[k];                            // (lane j's own row)
            s -= l[k] * ljk;
        }
        d = d > mu * 1e-3 ? d : mu * 1e-3;
        d = sqrt(d);
        const double djj = lane_bcast(d, j);
        const double id = 1.0 / djj;
        l[j] = (i == j) ? djj : s * id;
    }
    // column i of L for the backward solves: transpose through LDS
    pnp_sync();
    if (lane < N) {
#pragma unroll
        for (int k = 0; k < N; k++) A[i * N + k] = l[k];
    }
    pnp_sync();
#pragma unroll
    for (int k = 0; k < N; k++) lt[k] = A[k * N + i];    // L[k][i], meaningful for k >= i
    double dii = 0.0;                                    // L[i][i]
#pragma unroll
    for (int k = 0; k < N; k++) dii = (i == k) ? l[k] : dii;
    const double idii = 1.0 / dii;                       // (the serial form divides; one reciprocal + multiplies here)
    double x = 1.0 / sqrt((double)N) * ((i & 1) ? 0.9 : 1.1);
    for (int it = 0; it < 12; it++) {
        // forward: L y = x
        double acc = x, y = 0.0;
#pragma unroll
        for (int k = 0; k < N; k++) {
            const double yk = lane_bcast(acc * idii, k);
            y = (i == k) ? yk : y;
            acc -= l[k] * yk;                            // (only lanes i > k use it later)
        }
        // backward: L^T x = y
        acc = y;
#pragma unroll
        for (int k = N - 1; k >= 0; k--) {
            const double xk = lane_bcast(acc * idii, k);
            x = (i == k) ? xk : x;
            acc -= lt[k] * xk;                           // (only lanes i < k use it later)
        }
        const double nn = wave_sum_f64(lane < N ? x * x : 0.0);
        x *= 1.0 / sqrt(nn);
    }
    if (lane < N) v_out[lane] = x;
    pnp_sync();
}

// ---- motion model of PoseDetector (wave-cooperative: independent trig runs on separate lanes) ------------------------------------
// A^T B
__device__ inline void mat3_tmul(const double A[9], const double B[9], double C[9])
{
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) C[i * 3 + j] = A[i] * B[j] + A[3 + i] * B[3 + j] + A[6 + i] * B[6 + j];
}
// A^T v
__device__ inline void mat3_tvec(const double A[9], const double v[3], double o[3])
{
#pragma unroll
    for (int i = 0; i < 3; i++) o[i] = A[i] * v[0] + A[3 + i] * v[1] + A[6 + i] * v[2];
}

// detect_pose.py:553-566: get_pose_vel_acc (245-301) + _update_buffers (229-243) +
// apply_vel_acc (303-349).  curr/prev = (rvec, tvec).  Called by EVERY lane of the wave with
// uniform arguments (ts_in = the state as read before this frame); lane 0 stores the result.
// Returns AGT_TRK_* flags (uniform).
// Rc = Rodrigues(curr rvec) is the rotation of the solve's last evaluation; Rp = Rodrigues(prev rvec) was kept
// from the solve that produced prev (AgtTrackState::prev_R).  `st` = LDS copy of { rot_vel[2][9], tran_vel[2][3],
// prev_R[9] } taken when the solve started (the global loads are long done by now).
template <typename TS>
__device__ inline int motion_model_update(TS* ts, int lane, const double curr[6], bool curr_t_f32,
                                          const double prev[6], bool prev_t_f32, const double Rc[9], const double* st)
{
    double Rp[9];
#pragma unroll
    for (int i = 0; i < 9; i++) Rp[i] = st[24 + i];
    // get_relative_trans (transform_helper.py:184): rot_mat.T @ (tvec0 - tvec1); numpy subtracts in
    // float32 when both operands are float32 arrays
    double d[3], tran_vel[3], rot_vel[9];
#pragma unroll
    for (int i = 0; i < 3; i++)
        d[i] = (curr_t_f32 && prev_t_f32) ? (double)((float)prev[3 + i] - (float)curr[3 + i]) : prev[3 + i] - curr[3 + i];
    mat3_tvec(Rc, d, tran_vel);
    mat3_tmul(Rc, Rp, rot_vel);                  // get_relative_rot: rmat1.T @ rmat0
    bool any_zero = false;
#pragma unroll
    for (int i = 0; i < 9; i++) any_zero |= rot_vel[i] == 0.0;
#pragma unroll
    for (int i = 0; i < 3; i++) any_zero |= tran_vel[i] == 0.0;
    if (any_zero) return AGT_TRK_ZERO_VELOCITY;  // reference raises ValueError here
    // _update_buffers: keep the last two velocities
    const int n_before = ts->n_vel;
    double old_rv[9], old_tv[3];
    const int old_slot = n_before >= 2 ? 1 : 0;  // the entry that becomes "previous velocity"
#pragma unroll
    for (int i = 0; i < 9; i++) old_rv[i] = st[old_slot * 9 + i];
#pragma unroll
    for (int i = 0; i < 3; i++) old_tv[i] = st[18 + old_slot * 3 + i];
    const int n_after = n_before >= 2 ? 2 : n_before + 1;
    if (lane == 0) {
        if (n_before >= 2) {
            for (int i = 0; i < 9; i++) ts->rot_vel[0][i] = old_rv[i];
            for (int i = 0; i < 3; i++) ts->tran_vel[0][i] = old_tv[i];
        }
        const int slot = n_after - 1;
        for (int i = 0; i < 9; i++) ts->rot_vel[slot][i] = rot_vel[i];
        for (int i = 0; i < 3; i++) ts->tran_vel[slot][i] = tran_vel[i];
        ts->n_vel = n_after;
    }
    if (n_after < 2) return 0;                   // success = False: the guess is left as is
    double dv[3], tran_acc[3], rot_acc[9];
#pragma unroll
    for (int i = 0; i < 3; i++) dv[i] = old_tv[i] - tran_vel[i];
    mat3_tvec(rot_vel, dv, tran_acc);
    mat3_tmul(rot_vel, old_rv, rot_acc);
    // rotation_matrix_to_euler_angles (transform_helper.py:239-259), three atan2 on three lanes
    const double sy = agt_sqrtp(rot_acc[0] * rot_acc[0] + rot_acc[3] * rot_acc[3]);
    const bool sing = sy < 1e-6;
    double ay, ax;
    if (lane == 0) { ay = sing ? -rot_acc[5] : rot_acc[7]; ax = sing ? rot_acc[4] : rot_acc[8]; }
    else if (lane == 1) { ay = -rot_acc[6]; ax = sy; }
    else { ay = sing ? 0.0 : rot_acc[3]; ax = sing ? 1.0 : rot_acc[0]; }
    // apply_vel_acc halves the Euler angles and euler_angles_to_rotation_matrix (transform_helper.py:215-236,
    // R = Rz Ry Rx) only needs sin / cos of the halves: with (c, s) = (ax, ay) / |(ax, ay)| = (cos, sin) of
    // atan2(ay, ax), the half-angle identities give them without atan2 + sincos (two transcendental chains less):
    //   c >= 0:  cos(t/2) = sqrt((1 + c) / 2),  sin(t/2) = s / (2 cos(t/2))
    //   c <  0:  sin(t/2) = copysign(sqrt((1 - c) / 2), ay),  cos(t/2) = |s| / (2 |sin(t/2)|)
    // (atan2's range (-pi, pi] makes cos(t/2) >= 0; atan2(+-0, negative) = +-pi keeps the sign of ay.)
    double sn, cs;
    {
        const double r = agt_sqrtp(ax * ax + ay * ay);
        if (r > 0.0) {
            const double ir = agt_rcp(r), c = ax * ir, sv = ay * ir;
            if (c >= 0.0) { cs = agt_sqrtp(0.5 * (1.0 + c)); sn = sv * agt_rcp(2.0 * cs); }
            else { const double a = agt_sqrtp(0.5 * (1.0 - c)); sn = copysign(a, ay); cs = fabs(sv) * agt_rcp(2.0 * a); }
        } else { cs = 1.0; sn = 0.0; }
    }
    const double sx = lane_bcast(sn, 0), cx = lane_bcast(cs, 0);
    const double sy2 = lane_bcast(sn, 1), cy = lane_bcast(cs, 1);
    const double sz = lane_bcast(sn, 2), cz = lane_bcast(cs, 2);
    const double Rx[9] = { 1, 0, 0, 0, cx, -sx, 0, sx, cx };
    const double Ry[9] = { cy, 0, sy2, 0, 1, 0, -sy2, 0, cy };
    const double Rz[9] = { cz, -sz, 0, sz, cz, 0, 0, 0, 1 };
    double Tm[9], RA[9], M[9], Rpred[9], tp[3], tpred[3];
    agt_mat3_mul(Ry, Rx, Tm);
    agt_mat3_mul(Rz, Tm, RA);
    agt_mat3_mul(RA, rot_vel, M);                // (acc @ vel) @ pose, on the PREVIOUS pose
    agt_mat3_mul(M, Rp, Rpred);
#pragma unroll
    for (int i = 0; i < 3; i++) tp[i] = prev[3 + i];
#pragma unroll
    for (int i = 0; i < 3; i++)
        tpred[i] = (M[i * 3] * tp[0] + M[i * 3 + 1] * tp[1] + M[i * 3 + 2] * tp[2]) +
                   (RA[i * 3] * tran_vel[0] + RA[i * 3 + 1] * tran_vel[1] + RA[i * 3 + 2] * tran_vel[2] + 0.5 * tran_acc[i]);
    double rpred[3];
    agt_rodrigues_inv(Rpred, rpred);
    if (lane == 0) {
        for (int i = 0; i < 3; i++) { ts->guess[i] = rpred[i]; ts->guess[3 + i] = (double)(float)tpred[i]; }
        ts->guess_t_f32 = 1;                     // get_rmat_tvec casts to float32 (transform_helper.py:158-159)
    }
    return 0;
}

// Solve problem `b`.  Called by ONE wave (any wave of its workgroup); sh: scratch this wave may use now.
struct PnpNoHook { __device__ __forceinline__ void operator()() const {} };

// before_state(): called once, after this frame's correspondences have been requested and counted and BEFORE the tracker state
// is read (the fused step's alternating PnP waves wait there for the previous frame's state update: the loads of frame k+1
// run under the tail of frame k)
// LDS_STATE: the tracker state of stream b is sh.ts (the caller loaded it and writes it back), not P.track[b]
// COOP > 1 (n > 64, solves that start from a guess): the COOP waves of the workgroup share ONE solve.  Wave w owns correspondences
// 64 w .. 64 w + 63 (one per lane), every evaluation's per-wave totals meet in LDS behind one workgroup barrier (two
// alternating slabs, summed in wave order by every wave: all waves hold bit-identical totals and walk the CvLevMarq state
// machine redundantly, no decision is communicated), wave 0 alone writes results and runs the state update.  Against one
// wave with four points per lane: a quarter of the FP64 chain per evaluation for one barrier + 4 x 28 LDS reads
// (240 corners: 21.3 -> ~11 us).  Every wave of the workgroup must call, with identical arguments; the caller guarantees a
// guess (the DLT initialisation is one-wave code: COOP callers route guess-less solves to the PPL = 4 body on wave 0).
// OPAQUE (a caller that runs the body inside a FRAME LOOP): the thread index goes through an empty volatile asm, so that nothing
// derived from it is loop-invariant to the compiler -- hoisted out of the loop those values (lane-selected constants, addresses)
// stay alive across both solver bodies and spill.
template <typename T, int PPL, typename Hook = PnpNoHook, bool LDS_STATE = false, int COOP = 1, bool OPAQUE = false>
__device__ __forceinline__ void pnp_body(const AgtPnpParams& P, int b, PnpShared& sh, const void* img_p, const uint8_t* mask_p,
                                         double* so_p, int extra_flags = 0, Hook before_state = Hook())
{
    using TS = typename std::conditional<LDS_STATE, AgtTrackStateLds, AgtTrackState>::type;
    static_assert(COOP == 1 || (PPL == 1 && COOP * 32 * 3 <= NACC * SLAB), "cooperating waves hold one point per lane; slabs live in sh.part");
    int tid_ = (int)threadIdx.x;
    if constexpr (OPAQUE) asm volatile("" : "+v"(tid_));
    const int lane = tid_ & (AGT_WAVE - 1);
    const int wave = COOP > 1 ? agt_uniform(tid_ >> 6) : 0;
    const bool master = wave == 0;
    const bool writer = lane == 0 && master;
    const int n = P.n;
    // cross-wave meeting point (COOP > 1): sh.part = [2 alternating slabs][COOP][32] per-wave totals, then [COOP][32] private totals
    int slab_sel = 0;
    double* const tot = COOP > 1 ? &sh.part[2 * COOP * 32 + wave * 32] : sh.tot;
    auto coop_sum = [&](double v) -> double {          // v: this wave's total (uniform) -> the workgroup's, in every lane
        if constexpr (COOP == 1) return v;
        else {
            double* slot = &sh.part[slab_sel * COOP * 32];
            if (lane == 0) slot[wave * 32] = v;
            __syncthreads();
            double s_ = slot[0];
#pragma unroll
            for (int w = 1; w < COOP; w++) s_ += slot[w * 32];
            slab_sel ^= 1;
            return s_;
        }
    };
    PSTAMP(0);
    AgtCamera cam;
    load_cam<T>(P.cam, cam);

    // ---- this lane's correspondences
    double X[PPL], Y[PPL], Z[PPL], mu_[PPL], mv_[PPL];
    bool use[PPL];
    const T* obj = reinterpret_cast<const T*>(P.obj) + (long)b * P.obj_bstride;
    const T* img = reinterpret_cast<const T*>(img_p) + (long)b * n * 2;
    const uint8_t* mask = mask_p ? mask_p + (long)b * n : nullptr;
    int cnt = 0;
#pragma unroll
    for (int q = 0; q < PPL; q++) {
        const int i = lane + (q * COOP + wave) * AGT_WAVE;
        use[q] = i < n && (!mask || mask[i] != 0);
        X[q] = Y[q] = Z[q] = mu_[q] = mv_[q] = 0.0;
        if (use[q]) {
            X[q] = (double)obj[i * 3]; Y[q] = (double)obj[i * 3 + 1]; Z[q] = (double)obj[i * 3 + 2];
            mu_[q] = (double)img[i * 2]; mv_[q] = (double)img[i * 2 + 1];
            cnt++;
        }
    }
    if (P.seed_pts) {
        // detector-fed frame of the tracker: the supplied table is the corner set LK of the NEXT frame starts from
        // (detect_pose.py:400-437 -> the north-star LK step), a corner the detector did not deliver is not trackable
        for (int i = lane + wave * AGT_WAVE; i < n; i += AGT_WAVE * COOP) {
            P.seed_pts[((long)b * n + i) * 2] = (float)img[i * 2]; P.seed_pts[((long)b * n + i) * 2 + 1] = (float)img[i * 2 + 1];
            P.seed_status[(long)b * n + i] = (!mask || mask[i] != 0) ? 1 : 0;
        }
    }
    if (P.tag_gate) {
        // the reference solves on whole tags (detect_pose.py:400-437 collects four corners per detection, :494-496 wants two tags):
        // a corner counts only while its tag's other three do.  Tag t = corners 4t..4t+3 = one aligned quad of lanes (same q).
        cnt = 0;
#pragma unroll
        for (int q = 0; q < PPL; q++) {
            int u = use[q] ? 1 : 0;
            u &= agt_dpp_i32<0xB1>(u);          // quad_perm [1,0,3,2]
            u &= agt_dpp_i32<0x4E>(u);          // quad_perm [2,3,0,1]
            use[q] = u != 0;
            if (!use[q]) X[q] = Y[q] = Z[q] = mu_[q] = mv_[q] = 0.0;
            cnt += u;
        }
    }
    const int n_used = (int)coop_sum((double)agt_wave_sum_i64(cnt));
    before_state();
    int flags = extra_flags;          // AGT_TRK_CHAIN_TIMEOUT from the chained launch, reported with the frame's record
    double param[6];
    TS* ts;
    if constexpr (LDS_STATE) ts = (TS*)&sh.ts;
    else ts = P.track ? P.track + b : nullptr;
    if (ts && (agt_uniform(ts->chain_fault) | (extra_flags & AGT_TRK_CHAIN_TIMEOUT))) {
        // Fail-stop: the chained wait for this stream's corners gave up, now or in an earlier frame.  Nothing is
        // solved on a possibly stale ring entry and the motion-model state is not touched: the record is invalid and flagged,
        // and so is every later one of the stream until agt_tracker_reset.
        if (writer) {
            ts->chain_fault = 1; ts->frame++;
            if (P.fault) __hip_atomic_store(P.fault, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            if (P.dense_pose) {
                for (int i = 0; i < AGT_DENSE_STRIDE; i++) P.dense_rec[(long)b * AGT_DENSE_STRIDE + i] = 0.0;
                P.dense_done[b] = 1;
            }
            if (so_p) {
                double* so = so_p + (long)b * AGT_STATE_STRIDE;
                for (int i = 0; i < AGT_STATE_STRIDE; i++) so[i] = 0.0;
                so[AGT_ST_NTRACK] = n_used; so[AGT_ST_FLAGS] = AGT_TRK_CHAIN_TIMEOUT;
            }
        }
        return;
    }
    bool use_guess = P.use_guess != 0;
    double unchanged_prev[6] = { 0, 0, 0, 0, 0, 0 };
    int had_guess = 0, guess_f32 = 0, prev_f32 = 0;
    if (ts) {
        // detect_pose.py:490 deepcopy(prev_transform); :508 guess selection
        had_guess = ts->has_guess; guess_f32 = ts->guess_t_f32; prev_f32 = ts->prev_t_f32;
        use_guess = had_guess && P.enhance_ape;
#pragma unroll
        for (int i = 0; i < 6; i++) { unchanged_prev[i] = ts->prev[i]; param[i] = ts->guess[i]; }
        // velocities and prev_R for the motion model: requested now, read from LDS after the solve
        static_assert(offsetof(AgtTrackState, tran_vel) - offsetof(AgtTrackState, rot_vel) == 18 * sizeof(double) &&
                      offsetof(AgtTrackState, prev_R) - offsetof(AgtTrackState, rot_vel) == 24 * sizeof(double), "state layout");
        if (had_guess && P.enhance_ape && lane < 33 && master) sh.vec[lane] = (&ts->rot_vel[0][0])[lane];
        if (n_used < P.min_points) {          // detect_pose.py:573-574: fewer than two tags
            if (writer) {
                ts->has_guess = 0; ts->frame++;
                if (P.dense_pose) {
                    for (int i = 0; i < AGT_DENSE_STRIDE; i++) P.dense_rec[(long)b * AGT_DENSE_STRIDE + i] = 0.0;
                    P.dense_done[b] = 1;
                }
                if (so_p) {
                    double* so = so_p + (long)b * AGT_STATE_STRIDE;
                    for (int i = 0; i < AGT_STATE_STRIDE; i++) so[i] = 0.0;
                    so[AGT_ST_NTRACK] = n_used; so[AGT_ST_FLAGS] = AGT_PNP_TOO_FEW | flags;
                }
            }
            return;
        }
    } else {
#pragma unroll
        for (int i = 0; i < 6; i++) param[i] = P.pose[(long)b * 6 + i];
    }

    const bool enough = use_guess ? n_used >= 3 : n_used >= 4;      // the DLT branch re-checks for >= 6
    if (!enough) {
        if (writer && P.info) {
            P.info[b * 4 + AGT_INFO_OK] = 0; P.info[b * 4 + AGT_INFO_ITERS] = 0;
            P.info[b * 4 + AGT_INFO_NUSED] = n_used; P.info[b * 4 + AGT_INFO_FLAGS] = AGT_PNP_TOO_FEW;
        }
        if (writer && P.err) P.err[b] = 0.0;
        return;
    }

    // ---- initialisation without a guess: cvFindExtrinsicCameraParams2, DLT branch (one-wave code, see COOP above)
    if constexpr (COOP > 1) { if (!use_guess) return; }
    if constexpr (COOP == 1) if (!use_guess) {
        double c4[4] = { 0, 0, 0, 0 };
#pragma unroll
        for (int q = 0; q < PPL; q++) if (use[q]) { c4[0] += X[q]; c4[1] += Y[q]; c4[2] += Z[q]; }
        wave_reduce_slab<4>(c4, sh, lane);
        const double mcx = c4[0] / n_used, mcy = c4[1] / n_used, mcz = c4[2] / n_used;
        double m6[6] = { 0, 0, 0, 0, 0, 0 };
#pragma unroll
        for (int q = 0; q < PPL; q++) if (use[q]) {
            const double dx = X[q] - mcx, dy = Y[q] - mcy, dz = Z[q] - mcz;
            m6[0] += dx * dx; m6[1] += dx * dy; m6[2] += dx * dz; m6[3] += dy * dy; m6[4] += dy * dz; m6[5] += dz * dz;
        }
        wave_reduce_slab<6>(m6, sh, lane);
        const double MM[9] = { m6[0], m6[1], m6[2], m6[1], m6[3], m6[4], m6[2], m6[4], m6[5] };
        double Wm[3], Um[9], Vm[9];
        agt_svd3(MM, Wm, Um, Vm);
        if (Wm[2] / Wm[1] < 1e-3) {
            // ---- planar structure: cvFindExtrinsicCameraParams2's homography branch.
            // cv::findHomography(method 0): both point sets converted to float32, normalised DLT, then -- for more than
            // four points -- the LMSolver polish (levmarq.cpp LMSolverImpl::run, <= 10 iterations); oracle/cv_pnp.c
            // cvo_find_homography is the CPU restatement.
            flags |= AGT_PNP_PLANAR;
            double Rt[9];
#pragma unroll
            for (int i = 0; i < 9; i++) Rt[i] = Vm[i];
            if (Vm[2] * Vm[2] + Vm[5] * Vm[5] < 1e-10) {
#pragma unroll
                for (int i = 0; i < 9; i++) Rt[i] = (i % 4 == 0) ? 1.0 : 0.0;
            }
            if (agt_det3(Rt) < 0) {
#pragma unroll
                for (int i = 0; i < 9; i++) Rt[i] = -Rt[i];
            }
            double Tt[3];
#pragma unroll
            for (int a = 0; a < 3; a++) Tt[a] = -(Rt[a * 3] * mcx + Rt[a * 3 + 1] * mcy + Rt[a * 3 + 2] * mcz);
            double px_[PPL], py_[PPL], nx_[PPL], ny_[PPL];
            double c8[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
#pragma unroll
            for (int q = 0; q < PPL; q++) {
                px_[q] = py_[q] = nx_[q] = ny_[q] = 0.0;
                if (use[q]) {
                    px_[q] = (double)(float)(Rt[0] * X[q] + Rt[1] * Y[q] + Rt[2] * Z[q] + Tt[0]);
                    py_[q] = (double)(float)(Rt[3] * X[q] + Rt[4] * Y[q] + Rt[5] * Z[q] + Tt[1]);
                    undistort5(cam, mu_[q], mv_[q], nx_[q], ny_[q]);
                    nx_[q] = (double)(float)nx_[q]; ny_[q] = (double)(float)ny_[q];
                    c8[0] += nx_[q]; c8[1] += ny_[q]; c8[2] += px_[q]; c8[3] += py_[q];
                }
            }
            wave_reduce_slab<8>(c8, sh, lane);
            const double cmx = c8[0] / n_used, cmy = c8[1] / n_used, cMx = c8[2] / n_used, cMy = c8[3] / n_used;
            double d8[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
#pragma unroll
            for (int q = 0; q < PPL; q++) if (use[q]) {
                d8[0] += fabs(nx_[q] - cmx); d8[1] += fabs(ny_[q] - cmy); d8[2] += fabs(px_[q] - cMx); d8[3] += fabs(py_[q] - cMy);
            }
            wave_reduce_slab<8>(d8, sh, lane);
            bool hom_ok = !(fabs(d8[0]) < DBL_EPSILON || fabs(d8[1]) < DBL_EPSILON || fabs(d8[2]) < DBL_EPSILON || fabs(d8[3]) < DBL_EPSILON);
            const double smx = n_used / d8[0], smy = n_used / d8[1], sMx = n_used / d8[2], sMy = n_used / d8[3];
            // L^T L of the 2N x 9 system has 24 distinct sums: w * P P^T, P = (X, Y, 1), w in {1, x, y, x^2 + y^2}
            double t0[6], tx[6], ty[6], tq[6];
#pragma unroll
            for (int i = 0; i < 6; i++) t0[i] = tx[i] = ty[i] = tq[i] = 0.0;
#pragma unroll
            for (int q = 0; q < PPL; q++) if (use[q]) {
                const double x = (nx_[q] - cmx) * smx, y = (ny_[q] - cmy) * smy;
                const double Xn = (px_[q] - cMx) * sMx, Yn = (py_[q] - cMy) * sMy;
                const double pp[6] = { Xn * Xn, Xn * Yn, Xn, Yn * Yn, Yn, 1.0 };
                const double w2 = x * x + y * y;
#pragma unroll
                for (int i = 0; i < 6; i++) { t0[i] += pp[i]; tx[i] += x * pp[i]; ty[i] += y * pp[i]; tq[i] += w2 * pp[i]; }
            }
            wave_reduce_slab<6>(t0, sh, lane);
            wave_reduce_slab<6>(tx, sh, lane);
            wave_reduce_slab<6>(ty, sh, lane);
            wave_reduce_slab<6>(tq, sh, lane);
            pnp_sync();
            if (lane == 0) {
                const int ui[3][3] = { { 0, 1, 2 }, { 1, 3, 4 }, { 2, 4, 5 } };
                double* LtL = sh.LL;              // 9 x 9
                for (int a = 0; a < 3; a++)
                    for (int c = 0; c < 3; c++) {
                        const int u = ui[a][c];
                        LtL[a * 9 + c] = t0[u];          LtL[a * 9 + 3 + c] = 0.0;            LtL[a * 9 + 6 + c] = -tx[u];
                        LtL[(3 + a) * 9 + c] = 0.0;      LtL[(3 + a) * 9 + 3 + c] = t0[u];    LtL[(3 + a) * 9 + 6 + c] = -ty[u];
                        LtL[(6 + a) * 9 + c] = -tx[u];   LtL[(6 + a) * 9 + 3 + c] = -ty[u];   LtL[(6 + a) * 9 + 6 + c] = tq[u];
                    }
            }
            pnp_sync();
            smallest_eigvec_wave<9>(sh.LL, sh.vec, lane);
            if (lane == 0) {
                double* hv = sh.vec;
                const double invHn[9] = { 1. / smx, 0, cmx, 0, 1. / smy, cmy, 0, 0, 1 };
                const double Hn2[9] = { sMx, 0, -cMx * sMx, 0, sMy, -cMy * sMy, 0, 0, 1 };
                double H0[9], Tm[9], h[9];
                for (int i = 0; i < 9; i++) H0[i] = hv[i];
                agt_mat3_mul(invHn, H0, Tm);
                agt_mat3_mul(Tm, Hn2, h);
                bool ok = hom_ok && h[8] != 0.0;
                for (int i = 0; i < 9; i++) ok = ok && (h[i] - h[i] == 0.0);
                const double s8 = ok ? 1.0 / h[8] : 0.0;
                for (int i = 0; i < 9; i++) sh.vec[32 + i] = h[i] * s8;      // normalised: H[8] = 1
                sh.vec[41] = ok ? 1.0 : 0.0;
            }
            pnp_sync();
            double hx[8];
#pragma unroll
            for (int i = 0; i < 8; i++) hx[i] = sh.vec[32 + i];
            const bool h_ok = sh.vec[41] != 0.0;
            if (h_ok && n_used > 4) {
                // ---- LMSolverImpl::run with HomographyRefineCallback (fundam.cpp), uniform across the wave
                // sums of one evaluation: upper J^T J (36) | J^T r (8) | |r|^2 (1), taken in two passes of HP <= NACC sums
                // (the reduction slab holds NACC rows)
                constexpr int HP = 24;
                static_assert(HP <= NACC && 2 * HP >= 45, "two passes cover the 45 sums");
                auto hom_sums = [&](const double (&h)[8], auto LO_, double (&acc)[HP]) {
                    constexpr int LO = decltype(LO_)::value;
#pragma unroll
                    for (int i = 0; i < HP; i++) acc[i] = 0.0;
#pragma unroll
                    for (int q = 0; q < PPL; q++) if (use[q]) {
                        const double Mx = px_[q], My = py_[q];
                        double ww = h[6] * Mx + h[7] * My + 1.0;
                        ww = fabs(ww) > DBL_EPSILON ? 1.0 / ww : 0.0;
                        const double xi = (h[0] * Mx + h[1] * My + h[2]) * ww, yi = (h[3] * Mx + h[4] * My + h[5]) * ww;
                        const double ex = xi - nx_[q], ey = yi - ny_[q];
                        const double Jx[8] = { Mx * ww, My * ww, ww, 0, 0, 0, -Mx * ww * xi, -My * ww * xi };
                        const double Jy[8] = { 0, 0, 0, Mx * ww, My * ww, ww, -Mx * ww * yi, -My * ww * yi };
                        int idx = 0;
#pragma unroll
                        for (int a = 0; a < 8; a++)
#pragma unroll
                            for (int c = a; c < 8; c++) { if (idx >= LO && idx < LO + HP) acc[idx - LO] += Jx[a] * Jx[c] + Jy[a] * Jy[c]; idx++; }
#pragma unroll
                        for (int a = 0; a < 8; a++) { if (idx >= LO && idx < LO + HP) acc[idx - LO] += Jx[a] * ex + Jy[a] * ey; idx++; }
                        if (idx >= LO && idx < LO + HP) acc[idx - LO] += ex * ex + ey * ey;
                    }
                };
                // |r|^2 and max |r| only
                auto hom_cost = [&](const double (&h)[8], double& rinf) -> double {
                    double S = 0.0, ri = 0.0;
#pragma unroll
                    for (int q = 0; q < PPL; q++) if (use[q]) {
                        const double Mx = px_[q], My = py_[q];
                        double ww = h[6] * Mx + h[7] * My + 1.0;
                        ww = fabs(ww) > DBL_EPSILON ? 1.0 / ww : 0.0;
                        const double ex = (h[0] * Mx + h[1] * My + h[2]) * ww - nx_[q], ey = (h[3] * Mx + h[4] * My + h[5]) * ww - ny_[q];
                        S += ex * ex + ey * ey;
                        ri = fmax(ri, fmax(fabs(ex), fabs(ey)));
                    }
#pragma unroll
                    for (int o = 32; o >= 1; o >>= 1) ri = fmax(ri, __shfl_xor(ri, o));
                    rinf = ri;
                    return wave_sum_f64(S);
                };
                double Au[36], vv[8], Dg[8], S, rinf;
                auto hom_eval = [&](const double (&h)[8]) {
                    double acc[HP];
                    hom_sums(h, std::integral_constant<int, 0>{}, acc);
                    wave_reduce_slab<HP>(acc, sh, lane);
#pragma unroll
                    for (int i = 0; i < HP; i++) Au[i] = acc[i];
                    hom_sums(h, std::integral_constant<int, HP>{}, acc);
                    wave_reduce_slab<HP>(acc, sh, lane);
#pragma unroll
                    for (int i = 0; i < 36 - HP; i++) Au[HP + i] = acc[i];
#pragma unroll
                    for (int i = 0; i < 8; i++) vv[i] = acc[36 - HP + i];
                    S = acc[44 - HP];
                };
                hom_eval(hx);
                (void)hom_cost(hx, rinf);
                {
                    int idx = 0;
#pragma unroll
                    for (int a = 0; a < 8; a++)
#pragma unroll
                        for (int c = a; c < 8; c++) { if (c == a) Dg[a] = Au[idx]; idx++; }
                }
                double lambda_h = 1.0, lc = 0.75;
                for (int it = 0;;) {
                    // solve (A + lambda D) d = v  -- cv::solve(DECOMP_EIG) in OpenCV; Cholesky here (A + lambda D is SPD)
                    pnp_sync();
                    if (lane == 0) {
                        double* Ap = sh.LL;
                        int idx = 0;
                        for (int a = 0; a < 8; a++)
                            for (int c = a; c < 8; c++) { Ap[a * 8 + c] = Au[idx]; Ap[c * 8 + a] = Au[idx]; idx++; }
                        for (int a = 0; a < 8; a++) { Ap[a * 9] += lambda_h * Dg[a]; sh.vec[16 + a] = vv[a]; }
                        sh.vec[15] = spd_solve(Ap, 8, sh.vec + 16, sh.vec, nullptr) ? 1.0 : 0.0;
                    }
                    pnp_sync();
                    if (sh.vec[15] == 0.0) break;
                    double dd[8], xd[8];
#pragma unroll
                    for (int i = 0; i < 8; i++) { dd[i] = sh.vec[i]; xd[i] = hx[i] - dd[i]; }
                    double rinf_d;
                    const double Sd = hom_cost(xd, rinf_d);
                    double dS = 0.0, tdv = 0.0;
                    {
                        // temp_d = -A d + 2 v;  dS = d . temp_d
                        double Ad[8];
#pragma unroll
                        for (int a = 0; a < 8; a++) Ad[a] = 0.0;
                        int idx = 0;
#pragma unroll
                        for (int a = 0; a < 8; a++)
#pragma unroll
                            for (int c = a; c < 8; c++) { Ad[a] += Au[idx] * dd[c]; if (c != a) Ad[c] += Au[idx] * dd[a]; idx++; }
#pragma unroll
                        for (int a = 0; a < 8; a++) { dS += dd[a] * (2.0 * vv[a] - Ad[a]); tdv += dd[a] * vv[a]; }
                    }
                    const double Rq = (S - Sd) / (fabs(dS) > DBL_EPSILON ? dS : 1.0);
                    if (Rq > 0.75) {
                        lambda_h *= 0.5;
                        if (lambda_h < lc) lambda_h = 0.0;
                    } else if (Rq < 0.25) {
                        double nu = (Sd - S) / (fabs(tdv) > DBL_EPSILON ? tdv : 1.0) + 2.0;
                        nu = fmin(fmax(nu, 2.0), 10.0);
                        if (lambda_h == 0.0) {
                            pnp_sync();
                            if (lane == 0) {
                                double* Ap = sh.LL;
                                int idx = 0;
                                for (int a = 0; a < 8; a++)
                                    for (int c = a; c < 8; c++) { Ap[a * 8 + c] = Au[idx]; Ap[c * 8 + a] = Au[idx]; idx++; }
                                double mv = DBL_EPSILON;
                                if (spd_solve(Ap, 8, nullptr, nullptr, sh.vec + 24))
                                    for (int a = 0; a < 8; a++) mv = fmax(mv, fabs(sh.vec[24 + a]));
                                sh.vec[14] = mv;
                            }
                            pnp_sync();
                            lambda_h = lc = 1.0 / sh.vec[14];
                            nu *= 0.5;
                        }
                        lambda_h *= nu;
                    }
                    if (Sd < S) {
#pragma unroll
                        for (int i = 0; i < 8; i++) hx[i] = xd[i];
                        hom_eval(hx);
                        rinf = rinf_d;
                    }
                    it++;
                    double dinf = 0.0;
#pragma unroll
                    for (int i = 0; i < 8; i++) dinf = fmax(dinf, fabs(dd[i]));
                    if (!(it < 10 && dinf >= (double)FLT_EPSILON && rinf >= (double)FLT_EPSILON)) break;
                }
            }
            pnp_sync();
            if (lane == 0) {
                double h[9];
                for (int i = 0; i < 8; i++) h[i] = hx[i];
                h[8] = sh.vec[40];                 // h33 * (1 / h33) as the DLT left it; not a parameter of the polish
                double Rm[9], tv[3];
                if (h_ok) {
                    const double h1n = sqrt(h[0] * h[0] + h[3] * h[3] + h[6] * h[6]);
                    const double h2n = sqrt(h[1] * h[1] + h[4] * h[4] + h[7] * h[7]);
                    const double s1 = 1. / fmax(h1n, DBL_EPSILON), s2 = 1. / fmax(h2n, DBL_EPSILON), s3 = 2. / fmax(h1n + h2n, DBL_EPSILON);
                    const double t3[3] = { h[2] * s3, h[5] * s3, h[8] * s3 };
                    h[0] *= s1; h[3] *= s1; h[6] *= s1;
                    h[1] *= s2; h[4] *= s2; h[7] *= s2;
                    h[2] = h[3] * h[7] - h[6] * h[4];
                    h[5] = h[6] * h[1] - h[0] * h[7];
                    h[8] = h[0] * h[4] - h[3] * h[1];
                    double rr[3], dummy[27];
                    agt_rodrigues_inv(h, rr);
                    agt_rodrigues<false>(rr, h, dummy);
                    for (int a = 0; a < 3; a++) tv[a] = h[a * 3] * Tt[0] + h[a * 3 + 1] * Tt[1] + h[a * 3 + 2] * Tt[2] + t3[a];
                    agt_mat3_mul(h, Rt, Rm);
                } else {
                    for (int i = 0; i < 9; i++) Rm[i] = (i % 4 == 0) ? 1.0 : 0.0;
                    tv[0] = tv[1] = tv[2] = 0.0;
                }
                double rv[3];
                agt_rodrigues_inv(Rm, rv);
                sh.vec[32] = rv[0]; sh.vec[33] = rv[1]; sh.vec[34] = rv[2];
                sh.vec[35] = tv[0]; sh.vec[36] = tv[1]; sh.vec[37] = tv[2];
            }
            pnp_sync();
#pragma unroll
            for (int i = 0; i < 6; i++) param[i] = sh.vec[32 + i];
        } else {
        if (n_used < 6) {                     // non-planar DLT needs six correspondences
            if (lane == 0 && P.info) {
                P.info[b * 4 + AGT_INFO_OK] = 0; P.info[b * 4 + AGT_INFO_ITERS] = 0;
                P.info[b * 4 + AGT_INFO_NUSED] = n_used; P.info[b * 4 + AGT_INFO_FLAGS] = AGT_PNP_TOO_FEW;
            }
            if (lane == 0 && P.err) P.err[b] = 0.0;
            if (lane == 0 && ts) {
                ts->has_guess = 0; ts->frame++;
                if (P.dense_pose) {
                    for (int i = 0; i < AGT_DENSE_STRIDE; i++) P.dense_rec[(long)b * AGT_DENSE_STRIDE + i] = 0.0;
                    P.dense_done[b] = 1;
                }
                if (so_p) {
                    double* so = so_p + (long)b * AGT_STATE_STRIDE;
                    for (int i = 0; i < AGT_STATE_STRIDE; i++) so[i] = 0.0;
                    so[AGT_ST_NTRACK] = n_used; so[AGT_ST_FLAGS] = AGT_PNP_TOO_FEW | flags;
                }
            }
            return;
        }
        // L^T L has only 40 distinct sums: sum w * Mt Mt^T with w in {1, x, y, x^2+y^2}
        double s0[10], sx[10], sy[10], sq[10];
#pragma unroll
        for (int i = 0; i < 10; i++) s0[i] = sx[i] = sy[i] = sq[i] = 0.0;
#pragma unroll
        for (int q = 0; q < PPL; q++) if (use[q]) {
            double xn, yn;
            undistort5(cam, mu_[q], mv_[q], xn, yn);
            const double x = -xn, y = -yn, w2 = x * x + y * y;
            const double pr[10] = { X[q] * X[q], X[q] * Y[q], X[q] * Z[q], X[q], Y[q] * Y[q], Y[q] * Z[q], Y[q], Z[q] * Z[q], Z[q], 1.0 };
#pragma unroll
            for (int i = 0; i < 10; i++) { s0[i] += pr[i]; sx[i] += x * pr[i]; sy[i] += y * pr[i]; sq[i] += w2 * pr[i]; }
        }
        wave_reduce_slab<10>(s0, sh, lane);
        wave_reduce_slab<10>(sx, sh, lane);
        wave_reduce_slab<10>(sy, sh, lane);
        wave_reduce_slab<10>(sq, sh, lane);
        pnp_sync();
        if (lane == 0) {
            const int ui[4][4] = { { 0, 1, 2, 3 }, { 1, 4, 5, 6 }, { 2, 5, 7, 8 }, { 3, 6, 8, 9 } };
            for (int a = 0; a < 4; a++)
                for (int c = 0; c < 4; c++) {
                    const int u = ui[a][c];
                    sh.LL[a * 12 + c] = s0[u];           sh.LL[a * 12 + 4 + c] = 0.0;          sh.LL[a * 12 + 8 + c] = sx[u];
                    sh.LL[(4 + a) * 12 + c] = 0.0;       sh.LL[(4 + a) * 12 + 4 + c] = s0[u];  sh.LL[(4 + a) * 12 + 8 + c] = sy[u];
                    sh.LL[(8 + a) * 12 + c] = sx[u];     sh.LL[(8 + a) * 12 + 4 + c] = sy[u];  sh.LL[(8 + a) * 12 + 8 + c] = sq[u];
                }
        }
        pnp_sync();
        smallest_eigvec_wave<12>(sh.LL, sh.vec, lane);
        if (lane == 0) {
            double* v = sh.vec;
            double RR[9] = { v[0], v[1], v[2], v[4], v[5], v[6], v[8], v[9], v[10] };
            double tt[3] = { v[3], v[7], v[11] };
            if (agt_det3(RR) < 0) {
                for (int i = 0; i < 9; i++) RR[i] = -RR[i];
                for (int i = 0; i < 3; i++) tt[i] = -tt[i];
            }
            double sc = 0;
            for (int i = 0; i < 9; i++) sc += RR[i] * RR[i];
            sc = sqrt(sc);
            double Wr[3], U[9], Vt[9], Rm[9];
            agt_svd3(RR, Wr, U, Vt);
            agt_mat3_mul(U, Vt, Rm);
            double nr = 0;
            for (int i = 0; i < 9; i++) nr += Rm[i] * Rm[i];
            nr = sqrt(nr);
            double rv[3];
            agt_rodrigues_inv(Rm, rv);
            sh.vec[32] = rv[0]; sh.vec[33] = rv[1]; sh.vec[34] = rv[2];
            sh.vec[35] = tt[0] * nr / sc; sh.vec[36] = tt[1] * nr / sc; sh.vec[37] = tt[2] * nr / sc;
        }
        pnp_sync();
#pragma unroll
        for (int i = 0; i < 6; i++) param[i] = sh.vec[32 + i];
        }   // non-planar (DLT) branch
    }

    // ---- CvLevMarq
    double JtJ[21], JtErr[6];
    double prevParam[6];
    double prevErrNorm = DBL_MAX, errNorm = 0.0;
    int lambdaLg10 = -3, iters = 0;
    double lambda = 1e-3;
    const int max_iter = 20;
    const double epsilon = (double)FLT_EPSILON;

    bool has_dist = false;
#pragma unroll
    for (int i = 0; i < 12; i++) has_dist |= cam.k[i] != 0.0;
    has_dist |= cam.tilt != nullptr;
    double rex[PPL], rey[PPL];       // residuals of the most recent evaluation (re-used by the epilogue)
#pragma unroll
    for (int q = 0; q < PPL; q++) { rex[q] = 0.0; rey[q] = 0.0; }

    // residual (+ Jacobian) at `param`; returns |err|^2.  DIST is a std::integral_constant tag.
    // mode 0: residuals only.  mode 1: + J^T J / J^T e into the registers.  mode 2: + J^T J / J^T e left in
    // sh.tot as a CANDIDATE (CvLevMarq asks for J at exactly these parameters next, unless the step
    // is rejected or the solve terminates; evaluating it now saves a second projection pass).
    double Rlast[9];
    auto evaluate_t = [&](int mode, auto DIST) -> double {
        constexpr bool D = decltype(DIST)::value;
        const bool needJ = mode != 0;
        double R[9], G[9];
        if (mode == 2) PSTAMPM(48);
        if (needJ) agt_rodrigues<true>(param, R, G); else agt_rodrigues<false>(param, R, G);
#pragma unroll
        for (int i = 0; i < 9; i++) Rlast[i] = R[i];
        if (mode == 2) PSTAMPM(49);
        if (needJ) {
            double acc[NACC];
#pragma unroll
            for (int i = 0; i < NACC; i++) acc[i] = 0.0;
            // More than one point per lane (N > 64): the points are evaluated WITHOUT a branch around each one -- an unused slot
            // holds the origin and its terms are multiplied by an exact 0 -- so that the compiler may interleave the independent
            // FP64 chains of a lane's points (behind `if (use[q])` each point was its own basic block: 4 x 0.39 us per evaluation
            // at 240 corners).  One point per lane keeps the branch (nothing to interleave with).
#pragma unroll
            for (int q = 0; q < PPL; q++) if (PPL > 1 || use[q]) {
                double u, v, jr[6], jt[6];
                agt_project<true, D>(cam, R, G, param + 3, X[q], Y[q], Z[q], u, v, jr, jt);
                const double m = (PPL > 1 && !use[q]) ? 0.0 : 1.0;
                const double ex = (u - mu_[q]) * m, ey = (v - mv_[q]) * m;
                rex[q] = ex; rey[q] = ey;
                const double Jx[6] = { jr[0] * m, jr[1] * m, jr[2] * m, jt[0] * m, jt[1] * m, jt[2] * m };
                const double Jy[6] = { jr[3] * m, jr[4] * m, jr[5] * m, jt[3] * m, jt[4] * m, jt[5] * m };
                int idx = 0;
#pragma unroll
                for (int a = 0; a < 6; a++) {
#pragma unroll
                    for (int c = a; c < 6; c++) acc[idx++] += Jx[a] * Jx[c] + Jy[a] * Jy[c];
                    acc[21 + a] += Jx[a] * ex + Jy[a] * ey;
                }
                acc[27] += ex * ex + ey * ey;
            }
            if (mode == 2) PSTAMPM(50);
            if constexpr (COOP == 1) wave_reduce_bfly<NACC>(acc, tot, lane);
            else {
                double* slot = &sh.part[slab_sel * COOP * 32];
                wave_reduce_bfly<NACC>(acc, slot + wave * 32, lane);
                __syncthreads();
                if (lane < NACC) {
                    double s_ = slot[lane];
#pragma unroll
                    for (int w = 1; w < COOP; w++) s_ += slot[w * 32 + lane];
                    tot[lane] = s_;
                }
                pnp_sync();
                slab_sel ^= 1;
            }
            if (mode == 2) {
                PSTAMPM(51);
                return tot[27];
            }
#pragma unroll
            for (int i = 0; i < 21; i++) JtJ[i] = tot[i];
#pragma unroll
            for (int i = 0; i < 6; i++) JtErr[i] = tot[21 + i];
            return tot[27];
        }
        double e2 = 0.0;
#pragma unroll
        for (int q = 0; q < PPL; q++) if (PPL > 1 || use[q]) {
            double u, v;
            agt_project<false, D>(cam, R, G, param + 3, X[q], Y[q], Z[q], u, v, nullptr, nullptr);
            const double m = (PPL > 1 && !use[q]) ? 0.0 : 1.0;
            const double ex = (u - mu_[q]) * m, ey = (v - mv_[q]) * m;
            rex[q] = ex; rey[q] = ey;
            e2 += ex * ex + ey * ey;
        }
        return coop_sum(wave_sum_f64(e2));
    };
    auto evaluate = [&](int mode) -> double {
        return has_dist ? evaluate_t(mode, std::true_type{}) : evaluate_t(mode, std::false_type{});
    };
    auto commit_candidate = [&]() {
#pragma unroll
        for (int i = 0; i < 21; i++) JtJ[i] = tot[i];
#pragma unroll
        for (int i = 0; i < 6; i++) JtErr[i] = tot[21 + i];
    };
    // residuals at `param` re-using the rotation of the last evaluation (only tvec changed)
    auto residuals_same_rotation = [&](auto DIST) {
        constexpr bool D = decltype(DIST)::value;
#pragma unroll
        for (int q = 0; q < PPL; q++) if (use[q]) {
            double u, v;
            agt_project<false, D>(cam, Rlast, nullptr, param + 3, X[q], Y[q], Z[q], u, v, nullptr, nullptr);
            rex[q] = u - mu_[q]; rey[q] = v - mv_[q];
        }
    };

    auto step = [&]() {
        // lambda tracks 10^lambdaLg10 incrementally (OpenCV: exp(lambdaLg10 * log(10)); equal to ~1 ulp)
        double A[36], dx[6];
        int idx = 0;
#pragma unroll
        for (int a = 0; a < 6; a++)
#pragma unroll
            for (int c = a; c < 6; c++) { A[a * 6 + c] = JtJ[idx]; A[c * 6 + a] = JtJ[idx]; idx++; }
#pragma unroll
        for (int a = 0; a < 6; a++) A[a * 7] *= 1.0 + lambda;
        if (!agt_solve6(A, JtErr, dx)) {
            flags |= AGT_PNP_SINGULAR;
#pragma unroll
            for (int a = 0; a < 6; a++) dx[a] = 0.0;
        }
#pragma unroll
        for (int a = 0; a < 6; a++) param[a] = prevParam[a] - dx[a];
    };

    PSTAMP(1);
    double e2 = evaluate(1);
    PSTAMP(2);
    for (;;) {
#pragma unroll
        for (int i = 0; i < 6; i++) prevParam[i] = param[i];
        if (iters == 0) prevErrNorm = sqrt(e2);
        double rel = 0.0, e2cand = 0.0;
        bool last = false;
        for (;;) {
            PSTAMP(8 + iters * 4 + 0);
            step();
            PSTAMP(8 + iters * 4 + 1);
            // CvLevMarq's termination test depends on the parameters only, so it is known before the
            // residuals are: a terminating step needs no Jacobian, any other one will need it
            double dn = 0.0, pn = 0.0;
#pragma unroll
            for (int i = 0; i < 6; i++) { const double d = param[i] - prevParam[i]; dn += d * d; pn += prevParam[i] * prevParam[i]; }
            rel = sqrt(dn) / (sqrt(pn) + DBL_EPSILON);
            last = iters + 1 >= max_iter || rel < epsilon;
            e2cand = evaluate(last ? 0 : 2);
            errNorm = sqrt(e2cand);
            PSTAMP(8 + iters * 4 + 2);
            if (errNorm > prevErrNorm) { ++lambdaLg10; lambda *= 10.0; if (lambdaLg10 <= 16) continue; }
            break;
        }
        if (lambdaLg10 - 1 >= -16) { lambdaLg10 -= 1; lambda *= 0.1; } else { lambdaLg10 = -16; lambda = 1e-16; }
        ++iters;
        if (last) break;
        prevErrNorm = errNorm;
        commit_candidate();          // J^T J, J^T e at the accepted parameters (was: a fresh evaluate(true))
        e2 = e2cand;
        PSTAMP(8 + (iters - 1) * 4 + 3);
    }
    PSTAMP(3);

    // cv2 writes the result INTO the guess arrays: a float32 guess tvec (from the motion model)
    // yields a float32-rounded tvec (solvepnp.cpp convertTo(tvec, tvec.depth()))
    const bool tvec_f32 = ts && use_guess && guess_f32;
    if (tvec_f32) {
#pragma unroll
        for (int i = 3; i < 6; i++) param[i] = (double)(float)param[i];
    }

    // ---- epilogue: mean reprojection error (transform_helper.py:98-121) at the solution
    // The LM loop always ends right after an error evaluation at the final parameters, so the
    // residuals are already in registers; only the float32 tvec case re-projects.
    double esum = 0.0;
    {
        if (tvec_f32) { if (has_dist) residuals_same_rotation(std::true_type{}); else residuals_same_rotation(std::false_type{}); }
#pragma unroll
        for (int q = 0; q < PPL; q++) if (use[q]) esum += sqrt(rex[q] * rex[q] + rey[q] * rey[q]);
        esum = coop_sum(wave_sum_f64(esum)) / n_used;
    }
    PSTAMP(4);
    if (!master) return;             // (COOP > 1: results, state update and corner refresh are wave 0's)
    if (ts) {
        // ---- PoseDetector._estimate_pose state update, detect_pose.py:528-574
        const bool accepted = esum < P.gate_px;
        int tflags = flags;
        if (writer && use_guess) {           // in-place result: the guess arrays now hold the pose
            for (int i = 0; i < 6; i++) ts->guess[i] = param[i];
        }
        if (accepted && had_guess && P.enhance_ape)
            tflags |= motion_model_update(ts, lane, param, tvec_f32, unchanged_prev, prev_f32 != 0, Rlast, sh.vec);
        if (writer) {
            if (accepted) {
                if (!had_guess || !P.enhance_ape) {
                    for (int i = 0; i < 6; i++) ts->guess[i] = param[i];
                    ts->guess_t_f32 = 0; ts->has_guess = 1;
                }
                if (!(tflags & AGT_TRK_ZERO_VELOCITY)) {
                    for (int i = 0; i < 6; i++) ts->prev[i] = param[i];
                    for (int i = 0; i < 9; i++) ts->prev_R[i] = Rlast[i];
                    ts->prev_t_f32 = tvec_f32 ? 1 : 0; ts->has_prev = 1;
                }
            } else {
                ts->has_guess = 0;
            }
            ts->frame++;
            PSTAMP(5);
            if (P.dense_pose) {                   // hand-over to the dense stage of the same frame
                double* rc = P.dense_rec + (long)b * AGT_DENSE_STRIDE;
                for (int i = 0; i < 6; i++) { P.dense_pose[(long)b * 6 + i] = param[i]; rc[i] = param[i]; }
                for (int i = 6; i < AGT_DENSE_STRIDE; i++) rc[i] = 0.0;
                P.dense_done[b] = accepted ? 0 : 1;
            }
            if (so_p) {
                double* so = so_p + (long)b * AGT_STATE_STRIDE;
                for (int i = 0; i < 6; i++) so[i] = param[i];
                so[AGT_ST_OK] = accepted ? 1.0 : 0.0; so[AGT_ST_ERR] = esum; so[AGT_ST_NTRACK] = n_used;
                so[AGT_ST_ITERS] = iters; so[AGT_ST_GUESS] = use_guess ? 1.0 : 0.0; so[AGT_ST_FLAGS] = tflags;
                so[AGT_ST_TVEC_F32] = tvec_f32 ? 1.0 : 0.0;
                for (int i = AGT_ST_TVEC_F32 + 1; i < AGT_STATE_STRIDE; i++) so[i] = 0.0;
            }
        }
        if (accepted && P.reproject && P.corners_rw) {
            // refresh the whole corner set with projectPoints(all_objpts) (detect_pose.py:455-461)
            double R[9], G[9];
            agt_rodrigues<false>(param, R, G);
            float* cw = P.corners_rw + (long)b * n * 2;
            for (int i = lane; i < n; i += AGT_WAVE) {
                double u, v;
                agt_project<false>(cam, R, G, param + 3, (double)obj[i * 3], (double)obj[i * 3 + 1], (double)obj[i * 3 + 2],
                                   u, v, nullptr, nullptr);
                cw[i * 2] = (float)u; cw[i * 2 + 1] = (float)v;
                if (P.status_rw) P.status_rw[(long)b * n + i] = 1;       // lost corners are re-seeded: trackable again
            }
        }
        return;
    }
    if (writer) {
#pragma unroll
        for (int i = 0; i < 6; i++) P.pose[(long)b * 6 + i] = param[i];
        if (P.info) {
            P.info[b * 4 + AGT_INFO_OK] = 1; P.info[b * 4 + AGT_INFO_ITERS] = iters;
            P.info[b * 4 + AGT_INFO_NUSED] = n_used; P.info[b * 4 + AGT_INFO_FLAGS] = flags;
        }
        if (P.err) P.err[b] = esum;
    }
}

}  // namespace agt_pnp

#pragma clang fp contract(off)
