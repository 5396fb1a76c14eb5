// agt_dense_body.h -- the Gauss-Newton update of the dense refinement stage (agt_dense.hip: specification, mapping) and the
// stage's last step -- the final update and the corner re-seed -- as a device function, shared by dense_final_kernel and by the
// LK launch of the NEXT frame of a clip (agt_step.hip lk_reseed_kernel: every LK workgroup derives the refined pose from the
// block rows itself and projects its own corner, instead of a one-workgroup launch in the frame's serial chain).
#pragma once
#include "agt_pnp_body.h"
#ifndef DSTAMP
#define DSTAMP(i)
#endif

#pragma clang fp contract(fast)      // FP64 pose code only, see agt_device.h

namespace agt_dense {

constexpr int DN = 29;                 // 21 (upper JtJ) + 6 (Jt r) + r^2 + valid count
constexpr int DROW = 32;               // doubles per block row in the partials buffer

struct DenseParams {
    const uint8_t* img; long pitch, ibatch; int w, h;
    const float* mxyz; const float* mt; int M;
    const float* obj; const float* ipts; const uint8_t* mask; int N;
    AgtCameraHost cam;
    double* pose;                      // [B][6]
    double* partials;                  // [2][B][nblk + 1][DROW]: block rows, double-buffered by iteration parity
    double* ppose;                     // [2][B][8]: linearisation point of iteration k + 1 (published by block 0 of launch k + 1)
    long pstride;                      // doubles between the two row buffers
    int nblk;
    double* stats;                     // [B][stats_stride]: 5 values written per iteration
    int stats_stride;
    double* rec;                       // tracker stage: per-frame record [B][AGT_DENSE_STRIDE] (pose, refined flag, stats) or null
    int* done;                         // [B]
    double photo_weight, mu;
    int iter;
    float* seed_pts; uint8_t* seed_status;     // tracker stage with re-seed: the frame's corner set / LK status ([B][N][2], [B][N]) or null
    // clip submission (agt_track_frames_dense): the two-level pyramid pass of the NEXT frame rides in the first accumulate launch
    // as extra workgroups (blockIdx.x > nblk) -- it depends on nothing this frame computes, and alone it was a 6.5 us launch
    // in the frame's serial chain
    AgtPyrArgs py0, py1;
    int n_pyr;                                 // tiles per stream (0 = none)
};

struct DenseShared {
    double wtot[4][DROW];
    double rows8[8][DROW];             // update prologue: partial sums of the previous iteration's rows
    double geo[DROW];                  // ... and its corner (geometric) row
    double totw[4][2 * DROW];          // per wave: totals of the photometric rows | the corner row
};

// The Gauss-Newton update of iteration P.iter - 1 from its block rows: called by all 256 threads of a block; on return
// pose_new[0..5] holds the new pose in EVERY thread and the return value says "stop" -- the same bits in every thread of every
// block.  `publish`: this block also writes pose / statistics / record / done word to global memory.
// One global round trip: the block rows, the corner row and the previous pose are all requested up front; after the one
// barrier every wave reduces and solves for itself (redundantly: no second barrier, no LDS hand-over of the result).
// done_word != null: the stream's done word is tested here, AFTER the loads have been requested (its round trip runs beside
// theirs instead of in front); a set word returns "stop" before anything is written.
__device__ __forceinline__ bool dense_update(const DenseParams& P, DenseShared& sh, int b, const double* rows, const double* pose_in,
                                             double* pose_out, bool publish, int iter_done, double (&pose_new)[6], const int* done_word = nullptr,
                                             bool* solved = nullptr, bool write_done = true)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int k = threadIdx.x & 31, g = threadIdx.x >> 5;
    const int done_v = done_word ? *done_word : 0;
    double param[6];
#pragma unroll
    for (int q = 0; q < 6; q++) param[q] = pose_in[q];
    const double geo = (g == 0 && P.N > 0) ? rows[(long)P.nblk * DROW + k] : 0.0;
    {
        // every row load of the thread in flight at once (a rolled loop issues them one L2 round trip at a time: 7 us at 240
        // rows; round 2's 16 per trip still needed two trips for the 240 rows of BASELINE configs[4]), summed in row order: the
        // result does not depend on timing
        double s = 0.0;
        for (int j0 = g; j0 < P.nblk; j0 += 256) {
            double v[32];
#pragma unroll
            for (int u = 0; u < 32; u++) { const int j = j0 + 8 * u; v[u] = j < P.nblk ? rows[(long)j * DROW + k] : 0.0; }
            if (done_v) return true;
#pragma unroll
            for (int u = 0; u < 32; u++) s += v[u];
        }
        DSTAMP(8);
        sh.rows8[g][k] = s;
        if (g == 0) sh.geo[k] = geo;
    }
    __syncthreads();
    DSTAMP(9);
    // per wave: lanes 0..31 total the eight group rows, lanes 32..63 fetch the corner row; the wave's own copy in LDS, then
    // every lane reads what it needs (broadcast reads)
    double* tw = sh.totw[wave];
    if (lane < DROW)
        tw[lane] = ((sh.rows8[0][lane] + sh.rows8[1][lane]) + (sh.rows8[2][lane] + sh.rows8[3][lane])) +
                   ((sh.rows8[4][lane] + sh.rows8[5][lane]) + (sh.rows8[6][lane] + sh.rows8[7][lane]));
    else tw[lane] = sh.geo[lane - DROW];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    double A[36], gv[6], dx[6];
    int idx = 0;
#pragma unroll
    for (int q = 0; q < 6; q++)
#pragma unroll
        for (int c = q; c < 6; c++) { const double v = tw[DROW + idx] + tw[idx]; A[q * 6 + c] = v; A[c * 6 + q] = v; idx++; }
#pragma unroll
    for (int q = 0; q < 6; q++) { gv[q] = tw[DROW + 21 + q] + tw[21 + q]; A[q * 7] *= 1.0 + P.mu; }
    DSTAMP(10);
    const bool ok = agt_solve6(A, gv, dx);
    if (solved) *solved = ok;
    DSTAMP(11);
    double dn = 0.0, pn = 0.0;
#pragma unroll
    for (int q = 0; q < 6; q++) { dn += dx[q] * dx[q]; pn += param[q] * param[q]; }
    const bool stop = !ok || sqrt(dn) / (sqrt(pn) + DBL_EPSILON) < (double)FLT_EPSILON;
#pragma unroll
    for (int q = 0; q < 6; q++) pose_new[q] = ok ? param[q] - dx[q] : param[q];
    if (publish && threadIdx.x == 0) {
        const double ph_r2 = tw[27], ph_n = tw[28], geo_r2 = tw[DROW + 27], n_used = tw[DROW + 28];
        double* st = P.stats + (long)b * P.stats_stride;
        st[0] = ph_n > 0.0 ? sqrt(ph_r2 / ph_n) : 0.0;
        st[1] = n_used > 0.0 ? sqrt(geo_r2 / (2.0 * n_used)) : 0.0;
        st[2] = ph_n; st[3] = (double)iter_done; st[4] = n_used;
        if (!P.rec) st[5] = st[6] = st[7] = 0.0;
        if (ok) {
#pragma unroll
            for (int q = 0; q < 6; q++) { P.pose[(long)b * 6 + q] = pose_new[q]; if (pose_out) pose_out[q] = pose_new[q]; }
            if (P.rec) {
                double* rc = P.rec + (long)b * AGT_DENSE_STRIDE;
#pragma unroll
                for (int q = 0; q < 6; q++) rc[q] = pose_new[q];
                rc[AGT_DN_REFINED] = 1.0;
            }
        }
        if (stop && write_done) P.done[b] = 1;
    }
    return stop;
}

// The stage's last step for stream b, called by all 256 threads of a workgroup: the update of the last iteration (when the stream
// is not done) from the rows of launch P.iter - 1.  On return every thread holds the final pose in param[0..5]; the return value
// says whether the pose is a REFINED one (at least one Gauss-Newton step applied: the corner re-seed applies).  publish: this
// workgroup writes pose / statistics / record; write_done: ... and the done word -- only where no other workgroup of the same
// launch reads it (dense_final_kernel: one workgroup per stream).
__device__ __forceinline__ bool dense_finish(const DenseParams& P, DenseShared& sh, int b, int nstreams, bool publish, bool write_done, double (&param)[6])
{
    const int par = (P.iter - 1) & 1;                 // P.iter = iterations launched
    // the record as earlier launches left it (requested before the update: its round trip runs beside the row loads)
    double rec0[7];
    const double* rc = P.rec ? P.rec + (long)b * AGT_DENSE_STRIDE : nullptr;
#pragma unroll
    for (int k = 0; k < 7; k++) rec0[k] = rc ? rc[k < 6 ? k : AGT_DN_REFINED] : 0.0;
    bool refined = rec0[6] != 0.0;
#pragma unroll
    for (int k = 0; k < 6; k++) param[k] = rec0[k];
    if (!P.done[b] && P.iter > 0) {
        // the last update: every thread holds the new pose, the re-seed needs no trip through the record thread 0 writes
        double fin[6];
        bool ok = false;
        dense_update(P, sh, b, P.partials + (long)par * P.pstride + (long)b * (P.nblk + 1) * DROW, P.ppose + ((long)par * nstreams + b) * 8,
                     nullptr, publish, P.iter, fin, nullptr, &ok, write_done);
        if (ok) {
            refined = true;
#pragma unroll
            for (int k = 0; k < 6; k++) param[k] = fin[k];
        }
    }
    return refined;
}

}  // namespace agt_dense

#pragma clang fp contract(off)
