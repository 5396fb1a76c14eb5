// agt_dense.hip -- dense photometric + geometric pose refinement (BASELINE.json configs[4],
// SURVEY.md row a12 / 8f rank 2).  There is NO reference code for this step; the semantics are the
// specification at the top of oracle/cv_dense.c, which this file implements term for term:
//   E(p) = sum_j |project(obj_j; p) - img_j|^2 + photo_weight * sum_i (I~(project(X_i; p)) - T_i)^2
// damped Gauss-Newton (mu = 1e-3 on the diagonal, no step rejection), FP64 throughout.
//
// Mapping (the "large-N Jacobian reduce" of config 5: 61,440 samples, 29 sums each):
//   dense_accum_kernel  grid (ceil(M / 256) + 1, B): a thread owns one sample -- projection + 2x6
//       Jacobian in FP64, four bilinear taps + the eight central-difference taps of the gradient
//       straight from the frame (u8, L2/HBM), 21 + 6 + 1 + 1 partial sums.  Each wave reduces
//       through a transposed LDS slab (as the PnP kernel), the four waves through LDS again, and
//       the block writes ONE row of 32 doubles: no atomics, bit-reproducible.  The last block of a stream
//       evaluates the corner (geometric) rows instead, one corner per thread, into a row of the same layout
//       (round 1 did them serially inside the update kernel: 16 us per iteration at 240 corners, now 4).
//   the update (sum the block rows in 8 interleaved groups, fixed order; add the corner row; solve the damped 6x6 system;
//       new pose; `done` when the relative step falls under FLT_EPSILON) is the PROLOGUE of the next accumulate launch
//       (round 3): every block of iteration k + 1 re-derives the step of iteration k from the 241 rows (62 KB, L2-resident,
//       same order in every block, so all blocks hold the same bits) and goes on accumulating at the new pose; block 0 also
//       publishes pose, statistics and the done word.  Rows and intermediate poses are double-buffered by iteration parity (a
//       fast block of launch k + 1 writes rows while a slow one still reads those of launch k).  dense_final_kernel does the
//       last update and, as a tracker stage, the corner re-seed (projectPoints of the object points at the refined pose).
// iters Gauss-Newton iterations = iters + 1 launches (round 2: 2 * iters + 1 re-seed launch); nothing synchronises.
// No MFMA: the contraction is 6x6.
#undef AGT_PNP_STAMPS
#include <cstring>
#include "agt_pnp_body.h"
#include "agt_pyramid2_body.h"

// In-kernel cycle stamps of one accumulate launch (diagnostic builds only: make dbg): block 1 of stream 0, iteration >= 1
#ifdef AGT_DENSE_STAMPS
__device__ unsigned long long agt_dense_stamps[16];
#define DSTAMP(i) do { if (blockIdx.x == 1 && blockIdx.y == 0 && threadIdx.x == 0 && P.iter == 2) agt_dense_stamps[i] = __builtin_amdgcn_s_memtime(); } while (0)
extern "C" int agt_debug_dense_stamps(unsigned long long* host16)
{
    return (int)hipMemcpyFromSymbol(host16, HIP_SYMBOL(agt_dense_stamps), sizeof(agt_dense_stamps));
}
#else
#define DSTAMP(i)
#endif

#include "agt_dense_body.h"

#pragma clang fp contract(fast)      // FP64 pose code only, see agt_device.h

namespace {

using namespace agt_dense;

__global__ __launch_bounds__(256) void dense_accum_kernel(const DenseParams P)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t lds_raw[];
    DenseShared& sh = *reinterpret_cast<DenseShared*>(lds_raw);
    const int b = blockIdx.y;
    if ((int)blockIdx.x > P.nblk) {
        // next frame's pyramid tile (only in launches that carry the job; independent of the done word)
        const int tile = (int)blockIdx.x - P.nblk - 1;
        if (tile < P.n_pyr) {
            const int by = tile / P.py0.gx, bx = tile - by * P.py0.gx;
            agt_pyr2::pyr_down2_body(P.py0, P.py1, bx, by, P.py0.src + (long)b * P.py0.sbatch, P.py0.dst + (long)b * P.py0.dbatch,
                                     P.py1.dst + (long)b * P.py1.dbatch, lds_raw);
        }
        return;
    }
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool geo = (int)blockIdx.x == P.nblk;        // the last block of a stream evaluates the corner (geometric) rows
    const int par = P.iter & 1;
    // this thread's sample (or corner): requested before anything else, it does not depend on the pose -- the round trip runs
    // beside those of the update prologue (done word, block rows) instead of behind its solve
    DSTAMP(0);
    const int smp = blockIdx.x * 256 + tid;
    // (branch-free: behind a branch the compiler waits for the loads at the join.  Indices are clamped, pointers selected per block,
    // a block without a mask reads a byte it ignores.)
    const int si = smp < P.M ? smp : P.M - 1, ci = tid < P.N ? tid : (P.N > 0 ? P.N - 1 : 0);
    const float* p3 = geo ? P.obj + ci * 3 : P.mxyz + (long)si * 3;
    const float* p2 = geo ? P.ipts + ((long)b * P.N + ci) * 2 : P.mt + si;
    const bool masked = geo && P.mask != nullptr;
    const uint8_t* pm = masked ? P.mask + (long)b * P.N + ci : reinterpret_cast<const uint8_t*>(P.mt);
    float sX = p3[0], sY = p3[1], sZ = p3[2];
    const float f0 = p2[0], f1 = p2[geo ? 1 : 0];
    const uint8_t mb = *pm;
    float sT = f0, cu = f0, cv = f1;
    bool c_on = geo && tid < P.N && (!masked || mb != 0);
    double param[6];
    if (P.iter == 0 && P.done[b]) return;
    if (P.iter > 0) {
        // the update of iteration iter - 1, re-derived by every block (same rows, same order: same bits everywhere)
        const double* rows_prev = P.partials + (long)(par ^ 1) * P.pstride + (long)b * (P.nblk + 1) * DROW;
        // ppose[k & 1] = linearisation point of iteration k: written by block 0 of launch k, read by every block of launch k + 1
        // (launch 0 copies the caller's start pose there: P.pose itself is overwritten by the publishing block)
        const double* pose_prev = P.ppose + ((long)(par ^ 1) * gridDim.y + b) * 8;
        double* pose_pub = blockIdx.x == 0 ? P.ppose + ((long)par * gridDim.y + b) * 8 : nullptr;
        if (dense_update(P, sh, b, rows_prev, pose_prev, pose_pub, blockIdx.x == 0, P.iter, param, P.done + b)) return;
    } else {
#pragma unroll
        for (int k = 0; k < 6; k++) param[k] = P.pose[(long)b * 6 + k];
        if (blockIdx.x == 0 && tid < 6) P.ppose[((long)par * gridDim.y + b) * 8 + tid] = param[tid];
    }
    DSTAMP(1);
    AgtCamera cam;
    agt_pnp::load_cam<float>(P.cam, cam);
    bool has_dist = false;
#pragma unroll
    for (int k = 0; k < 12; k++) has_dist |= cam.k[k] != 0.0;
    has_dist |= cam.tilt != nullptr;
    double R[9], G[9];
    agt_rodrigues<true>(param, R, G);
    DSTAMP(2);
    const double tvec[3] = { param[3], param[4], param[5] };      // (param + 3 as an argument kept three doubles in scratch: 1.5 MB of writes per launch)

    double acc[DN];
#pragma unroll
    for (int k = 0; k < DN; k++) acc[k] = 0.0;
    if (!geo) {
        if (smp < P.M) {
            const double X = (double)sX, Y = (double)sY, Z = (double)sZ;
            double u, v, jr[6], jt[6];
            if (has_dist) agt_project<true, true>(cam, R, G, tvec, X, Y, Z, u, v, jr, jt);
            else agt_project<true, false>(cam, R, G, tvec, X, Y, Z, u, v, jr, jt);
            DSTAMP(3);
            const double fx0 = floor(u), fy0 = floor(v);
            if (fx0 >= 1.0 && fx0 <= (double)(P.w - 3) && fy0 >= 1.0 && fy0 <= (double)(P.h - 3)) {
                const int x0 = (int)fx0, y0 = (int)fy0;
                const double a = u - fx0, bb = v - fy0;
                const uint8_t* p = P.img + (long)b * P.ibatch + (long)y0 * P.pitch + x0;
                const long s = P.pitch;
                // the 4 x 4 neighbourhood minus its corners
                const double pm0 = p[-1], p00 = p[0], p10 = p[1], p20 = p[2];
                const double pm1 = p[s - 1], p01 = p[s], p11 = p[s + 1], p21 = p[s + 2];
                const double p0m = p[-s], p1m = p[-s + 1], p02 = p[2 * s], p12 = p[2 * s + 1];
                const double w00 = (1 - a) * (1 - bb), w01 = a * (1 - bb), w10 = (1 - a) * bb, w11 = a * bb;
                const double I = w00 * p00 + w01 * p10 + w10 * p01 + w11 * p11;
                const double gx = w00 * (p10 - pm0) * 0.5 + w01 * (p20 - p00) * 0.5 + w10 * (p11 - pm1) * 0.5 + w11 * (p21 - p01) * 0.5;
                const double gy = w00 * (p01 - p0m) * 0.5 + w01 * (p11 - p1m) * 0.5 + w10 * (p02 - p00) * 0.5 + w11 * (p12 - p10) * 0.5;
                DSTAMP(4);
                const double r = I - (double)sT;
                const double J[6] = { gx * jr[0] + gy * jr[3], gx * jr[1] + gy * jr[4], gx * jr[2] + gy * jr[5],
                                      gx * jt[0] + gy * jt[3], gx * jt[1] + gy * jt[4], gx * jt[2] + gy * jt[5] };
                int idx = 0;
#pragma unroll
                for (int q = 0; q < 6; q++) {
#pragma unroll
                    for (int c = q; c < 6; c++) acc[idx++] = P.photo_weight * J[q] * J[c];
                    acc[21 + q] = P.photo_weight * J[q] * r;
                }
                acc[27] = r * r;
                acc[28] = 1.0;
            }
        }
    } else {
        // geometric rows, one corner per thread per trip (N <= 256: one trip); same row layout, unweighted,
        // [27] = sum of squared residuals, [28] = corners used
        for (int i = tid; i < P.N; i += 256) {
            if (i >= 256) {             // (second and later trips: N > 256)
                c_on = !P.mask || P.mask[(long)b * P.N + i];
                sX = P.obj[i * 3]; sY = P.obj[i * 3 + 1]; sZ = P.obj[i * 3 + 2];
                cu = P.ipts[((long)b * P.N + i) * 2]; cv = P.ipts[((long)b * P.N + i) * 2 + 1];
            }
            if (!c_on) continue;
            double u, v, jr[6], jt[6];
            const double X = (double)sX, Y = (double)sY, Z = (double)sZ;
            if (has_dist) agt_project<true, true>(cam, R, G, tvec, X, Y, Z, u, v, jr, jt);
            else agt_project<true, false>(cam, R, G, tvec, X, Y, Z, u, v, jr, jt);
            const double ex = u - (double)cu, ey = v - (double)cv;
            const double Jx[6] = { jr[0], jr[1], jr[2], jt[0], jt[1], jt[2] };
            const double Jy[6] = { jr[3], jr[4], jr[5], jt[3], jt[4], jt[5] };
            int idx = 0;
#pragma unroll
            for (int q = 0; q < 6; q++) {
#pragma unroll
                for (int c = q; c < 6; c++) acc[idx++] += Jx[q] * Jx[c] + Jy[q] * Jy[c];
                acc[21 + q] += Jx[q] * ex + Jy[q] * ey;
            }
            acc[27] += ex * ex + ey * ey;
            acc[28] += 1.0;
        }
    }
    // wave: register butterfly of the 29 sums (the PnP kernel's: two permlane-swap stages, three DPP stages; lane l ends up
    // with the total of value l >> 1) -- round 2 went through a transposed LDS slab (29 stores + 32 dependent load / adds per
    // lane: ~1 us of every launch); block: four wave rows
    DSTAMP(5);
    {
        using namespace agt_pnp;
        double v[32];
#pragma unroll
        for (int i = 0; i < 32; i++) v[i] = i < DN ? acc[i] : 0.0;
#pragma unroll
        for (int i = 0; i < 16; i++) { double b2; const double a2 = dswap32(v[i], v[i + 16], b2); v[i] = a2 + b2; }
#pragma unroll
        for (int i = 0; i < 8; i++) { double b2; const double a2 = dswap16(v[i], v[i + 8], b2); v[i] = a2 + b2; }
        bfly_stage<0x140, 4>(v, (lane & 8) != 0);
        bfly_stage<0x141, 2>(v, (lane & 4) != 0);
        bfly_stage<0x1B, 1>(v, (lane & 2) != 0);
        const double tot = v[0] + ddpp<0xB1>(v[0]);
        if (!(lane & 1) && (lane >> 1) < DN) sh.wtot[wave][lane >> 1] = tot;
    }
    DSTAMP(6);
    __syncthreads();
    DSTAMP(7);
    if (tid < DROW) {
        const double t = tid < DN ? ((sh.wtot[0][tid] + sh.wtot[1][tid]) + (sh.wtot[2][tid] + sh.wtot[3][tid])) : 0.0;
        P.partials[(long)par * P.pstride + ((long)b * (P.nblk + 1) + blockIdx.x) * DROW + tid] = t;
    }
}

// After the last accumulate launch, one block per stream: the update of the last iteration and, as a stage of the tracker
// with re-seed, projectPoints(all object points; refined pose) into the frame's corner set, every corner trackable again
// (stops the drift of raw LK chaining).
__global__ __launch_bounds__(256) void dense_final_kernel(const DenseParams P)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t lds_raw[];
    DenseShared& sh = *reinterpret_cast<DenseShared*>(lds_raw);
    const int b = blockIdx.x;
    double param[6];
    const bool refined = dense_finish(P, sh, b, (int)gridDim.x, true, true, param);
    if (!P.seed_pts || !P.rec || !refined) return;
    AgtCamera cam;
    agt_pnp::load_cam<float>(P.cam, cam);
    double R[9], G[9];
    agt_rodrigues<false>(param, R, G);
    for (int i = threadIdx.x; i < P.N; i += 256) {
        double u, v;
        agt_project<false>(cam, R, G, param + 3, (double)P.obj[i * 3], (double)P.obj[i * 3 + 1], (double)P.obj[i * 3 + 2], u, v, nullptr, nullptr);
        P.seed_pts[((long)b * P.N + i) * 2] = (float)u; P.seed_pts[((long)b * P.N + i) * 2 + 1] = (float)v;
        P.seed_status[(long)b * P.N + i] = 1;
    }
}

}  // namespace

// rec == null: plain agt_dense_refine (done words cleared here, stats [B][8]).  rec != null: stage of the tracker -- the
// done words and the start poses were written by the PnP epilogue of the same frame (done = pose not accepted), the
// statistics go into the record; seed_pts / seed_status != null: the corner re-seed rides in the final launch.
// partials: agt_dense_doubles(M, B) doubles.
hipError_t agt_launch_dense(hipStream_t stream, const uint8_t* img, long pitch, long ibatch, int w, int h,
                            const float* mxyz, const float* mt, int M,
                            const float* obj, const float* ipts, const uint8_t* mask, int N,
                            const AgtCameraHost& cam, double* pose, double* partials, double* stats, int* done,
                            int B, int iters, double photo_weight, double mu, double* rec, float* seed_pts, uint8_t* seed_status,
                            hipEvent_t* ev, int n_ev, const AgtPyrArgs* next_pyr, AgtDenseFinal* defer_final)
{
    static_assert(sizeof(DenseParams) <= sizeof(AgtDenseFinal::bytes), "AgtDenseFinal holds a DenseParams");
    DenseParams P;
    P.n_pyr = 0;
    P.py0 = AgtPyrArgs(); P.py1 = AgtPyrArgs();
    P.rec = rec; P.stats_stride = rec ? AGT_DENSE_STRIDE : 8;
    if (rec) stats = rec + AGT_DN_PHOTO_RMS;
    P.img = img; P.pitch = pitch; P.ibatch = ibatch; P.w = w; P.h = h;
    P.mxyz = mxyz; P.mt = mt; P.M = M; P.obj = obj; P.ipts = ipts; P.mask = mask; P.N = N;
    P.cam = cam; P.pose = pose; P.nblk = (M + 255) / 256; P.stats = stats; P.done = done;
    P.partials = partials; P.pstride = (long)B * (P.nblk + 1) * DROW; P.ppose = partials + 2 * P.pstride;
    P.photo_weight = photo_weight; P.mu = mu;
    P.seed_pts = seed_pts; P.seed_status = seed_status;
    hipError_t e = rec ? hipSuccess : hipMemsetAsync(done, 0, (size_t)B * sizeof(int), stream);
    for (int it = 0; it < iters && e == hipSuccess; it++) {
        P.iter = it;
        // (the geometric block's slot is always in the grid when a pyramid job rides along: block indices above nblk are tiles)
        unsigned gx = (unsigned)(P.nblk + (N > 0 ? 1 : 0));
        size_t lds = sizeof(DenseShared);
        P.n_pyr = 0;
        // (in the SECOND launch when there is one: the first has no update prologue and is shorter than a pyramid tile's ~6 us --
        // riding there stretched it by 3.2 us; the later launches last ~7 us and hide the tiles completely)
        if (it == (iters > 1 ? 1 : 0) && next_pyr) {
            P.py0 = next_pyr[0]; P.py1 = next_pyr[1];
            P.n_pyr = P.py0.gx * P.py0.gy;
            gx = (unsigned)(P.nblk + 1 + P.n_pyr);
            lds = lds > (size_t)agt_pyr2::PYR2_LDS_BYTES ? lds : (size_t)agt_pyr2::PYR2_LDS_BYTES;
        }
        hipLaunchKernelGGL(dense_accum_kernel, dim3(gx, B), dim3(256), lds, stream, P);
        // profiling only: ev[0] closes the Gauss-Newton launches (launch i carries the update of iteration i - 1), ev[1] the final launch
        if (ev && n_ev >= 2 && it == iters - 1) (void)hipEventRecord(ev[0], stream);
        e = hipGetLastError();
    }
    if (e == hipSuccess && defer_final) {
        // clip submission: the last update and the re-seed ride in the next frame's LK launch (agt_step.hip lk_reseed_kernel)
        P.iter = iters; P.n_pyr = 0;
        memcpy(defer_final->bytes, &P, sizeof(P));
        return e;
    }
    if (e == hipSuccess) {
        P.iter = iters;
        hipLaunchKernelGGL(dense_final_kernel, dim3(B), dim3(256), sizeof(DenseShared), stream, P);
        if (ev && n_ev >= 2 && iters > 0) (void)hipEventRecord(ev[1], stream);
        e = hipGetLastError();
    }
    return e;
}

hipError_t agt_launch_dense_final(hipStream_t stream, const AgtDenseFinal& F, int B)
{
    DenseParams P;
    memcpy(&P, F.bytes, sizeof(P));
    hipLaunchKernelGGL(dense_final_kernel, dim3(B), dim3(256), sizeof(DenseShared), stream, P);
    return hipGetLastError();
}

// rows of block partials per stream: the photometric blocks and one geometric (corner) block
int agt_dense_blocks(int M) { return (M + 255) / 256 + 1; }
// doubles of scratch a call needs: two row buffers (iteration parity) + two pose slots per stream
size_t agt_dense_doubles(int M, int B) { return (size_t)2 * B * (size_t)agt_dense_blocks(M) * DROW + (size_t)2 * B * 8; }
