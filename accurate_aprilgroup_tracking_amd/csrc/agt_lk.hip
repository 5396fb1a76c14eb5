// agt_lk.hip -- cv::calcOpticalFlowPyrLK per-point tracker for gfx950 (north-star step;
// no call site in the reference, belongs at the hole detect_pose.py:573-574).
// Semantics: OpenCV modules/video/src/lkpyramid.cpp LKTrackerInvoker + calcSharrDeriv,
// restated on the CPU in oracle/cv_lk.c; this kernel is bit-identical to that oracle
// (CVO_ACC_EXACT mode) on nextPts, status and err.
//
// Mapping: ONE 64-lane wave per corner, all pyramid levels and all iterations inside one
// launch.  A WIN x WIN window is WIN rows x R runs of RUN pixels (21x21 -> 21 x 3 x 7 =
// 63 lanes); each lane keeps its RUN patch values (I, Ix, Iy) in registers.
//   * the 24x24 I neighbourhood is staged in LDS with aligned dword loads (reflect-101 at
//     the image edge), Scharr is evaluated on the fly from that tile (no full-frame
//     derivative image is ever written to HBM -- the CPU path writes 4 B/px/level);
//   * the J search tile (40x40) is staged once per level and re-staged only if the window
//     leaves it;
//   * the 2x2 normal equations are EXACT integer sums (|terms| < 2^31 per lane, int64
//     across lanes) reduced with DPP row ops + v_readlane, so the result is independent of
//     the reduction order and wave-uniform; every OpenCV build approximates this sum with
//     float adds in its own SIMD order.
// Algorithmic bytes per point per level: 24*24 (I) + 40*40 (J tile) u8.
#include "agt_device.h"
#include "agt_kernels.h"

namespace {

template <int WIN>
struct LkCfg {
    static constexpr int R = 64 / WIN;                 // runs per window row
    static constexpr int RUN = (WIN + R - 1) / R;      // pixels per lane
    static constexpr int IW = WIN + 3;                 // I tile: window + bilinear + Scharr halo
    static constexpr int IP = ((IW + 6) / 4) * 4;      // LDS pitch incl. <=3 B alignment shift
    static constexpr int DW = WIN + 1;                 // derivative tile
    static constexpr int MARGIN = 9;
    static constexpr int JT = WIN + 1 + 2 * MARGIN;    // J search tile
    static constexpr int JP = ((JT + 6) / 4) * 4;
    static_assert(R >= 1 && WIN * R <= 64, "window does not fit one wave");
};

constexpr int W_BITS = 14;
__device__ __forceinline__ int descale(int x, int n) { return (x + (1 << (n - 1))) >> n; }

// Stage [tx0, tx0+tw) x [ty0, ty0+th) of a reflect-101 padded image into LDS.
// Pixel (x, y) lands at s[(y - ty0) * sp + (x - (tx0 & ~3))].
__device__ __forceinline__ void stage_tile(const uint8_t* __restrict__ img, int w, int h, long pitch,
                                           int tx0, int ty0, int tw, int th,
                                           uint8_t* s, int sp, int lane)
{
    const int ax0 = tx0 & ~3;
    const int ndw = (tx0 + tw - ax0 + 3) >> 2;
    const int total = th * ndw;
    for (int i = lane; i < total; i += AGT_WAVE) {
        const int r = i / ndw, c4 = i - r * ndw;
        const int gy = agt_reflect101(ty0 + r, h);
        const int gx = ax0 + 4 * c4;
        const uint8_t* row = img + (long)gy * pitch;
        uint32_t v;
        if (gx >= 0 && gx + 3 < w) {
            v = *reinterpret_cast<const uint32_t*>(row + gx);
        } else {
            v = (uint32_t)row[agt_reflect101(gx, w)] | ((uint32_t)row[agt_reflect101(gx + 1, w)] << 8) |
                ((uint32_t)row[agt_reflect101(gx + 2, w)] << 16) | ((uint32_t)row[agt_reflect101(gx + 3, w)] << 24);
        }
        *reinterpret_cast<uint32_t*>(s + r * sp + 4 * c4) = v;
    }
}

__device__ __forceinline__ void bilinear_weights(float a, float b, int& iw00, int& iw01, int& iw10, int& iw11)
{
    iw00 = __float2int_rn((1.f - a) * (1.f - b) * (float)(1 << W_BITS));
    iw01 = __float2int_rn(a * (1.f - b) * (float)(1 << W_BITS));
    iw10 = __float2int_rn((1.f - a) * b * (float)(1 << W_BITS));
    iw11 = (1 << W_BITS) - iw00 - iw01 - iw10;
}

template <int WIN>
__global__ __launch_bounds__(AGT_WAVE) void lk_kernel(const AgtLkParams P)
{
    using C = LkCfg<WIN>;
    __shared__ __attribute__((aligned(16))) uint8_t sI[C::IW * C::IP];
    __shared__ int sD[C::DW * C::DW];
    __shared__ __attribute__((aligned(16))) uint8_t sJ[C::JT * C::JP];

    const int lane = threadIdx.x;
    const int b = blockIdx.y;
    const long pidx = (long)b * P.n + blockIdx.x;
    const int row = lane / C::R;
    const int x0 = (lane - row * C::R) * C::RUN;
    const bool active = row < WIN;
    const int nx = active ? (WIN - x0 < C::RUN ? WIN - x0 : C::RUN) : 0;

    const float halfw = (WIN - 1) * 0.5f;
    const float FLT_SCALE = 1.f / (1 << 20);
    const float ppx = P.prev_pts[pidx * 2], ppy = P.prev_pts[pidx * 2 + 1];
    float outx = 0.f, outy = 0.f;              // nextPts[ptidx]
    if (P.flags & AGT_LK_USE_INITIAL_FLOW) { outx = P.next_pts[pidx * 2]; outy = P.next_pts[pidx * 2 + 1]; }
    int st = 1;
    float errv = 0.f;

    for (int level = P.max_level; level >= 0; level--) {
        const AgtLevel LI = P.prev[level];
        const AgtLevel LJ = P.next[level];
        const uint8_t* imgI = LI.ptr + (long)b * LI.bstride;
        const uint8_t* imgJ = LJ.ptr + (long)b * LJ.bstride;
        const float scale = 1.f / (float)(1 << level);
        float prevx = ppx * scale, prevy = ppy * scale;
        float nextx, nexty;
        if (level == P.max_level) {
            if (P.flags & AGT_LK_USE_INITIAL_FLOW) { nextx = outx * scale; nexty = outy * scale; }
            else { nextx = prevx; nexty = prevy; }
        } else { nextx = outx * 2.f; nexty = outy * 2.f; }
        outx = nextx; outy = nexty;

        prevx -= halfw; prevy -= halfw;
        const int ipx = agt_uniform((int)floorf(prevx)), ipy = agt_uniform((int)floorf(prevy));
        if (ipx < -WIN || ipx >= LI.w || ipy < -WIN || ipy >= LI.h) {
            if (level == 0) { st = 0; errv = 0.f; }
            continue;
        }
        int iw00, iw01, iw10, iw11;
        bilinear_weights(prevx - (float)ipx, prevy - (float)ipy, iw00, iw01, iw10, iw11);

        // ---- I neighbourhood -> LDS, Scharr on the fly -> LDS
        __syncthreads();
        stage_tile(imgI, LI.w, LI.h, LI.pitch, ipx - 1, ipy - 1, C::IW, C::IW, sI, C::IP, lane);
        __syncthreads();
        const int offI = (ipx - 1) - ((ipx - 1) & ~3);
        for (int idx = lane; idx < C::DW * C::DW; idx += AGT_WAVE) {
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
        __syncthreads();

        // ---- per-lane patch (registers) + exact covariance sums
        int Iv[C::RUN], Ix[C::RUN], Iy[C::RUN];
        int a11 = 0, a12 = 0, a22 = 0;
#pragma unroll
        for (int k = 0; k < C::RUN; k++) {
            Iv[k] = 0; Ix[k] = 0; Iy[k] = 0;
            if (k < nx) {
                const int x = x0 + k;
                const uint8_t* p = sI + (row + 1) * C::IP + (x + 1) + offI;
                Iv[k] = descale(p[0] * iw00 + p[1] * iw01 + p[C::IP] * iw10 + p[C::IP + 1] * iw11, W_BITS - 5);
                const int d00 = sD[row * C::DW + x], d01 = sD[row * C::DW + x + 1];
                const int d10 = sD[(row + 1) * C::DW + x], d11 = sD[(row + 1) * C::DW + x + 1];
                Ix[k] = descale((short)d00 * iw00 + (short)d01 * iw01 + (short)d10 * iw10 + (short)d11 * iw11, W_BITS);
                Iy[k] = descale((d00 >> 16) * iw00 + (d01 >> 16) * iw01 + (d10 >> 16) * iw10 + (d11 >> 16) * iw11, W_BITS);
                a11 += Ix[k] * Ix[k]; a12 += Ix[k] * Iy[k]; a22 += Iy[k] * Iy[k];
            }
        }
        const float A11 = (float)(double)agt_wave_sum_i64(a11) * FLT_SCALE;
        const float A12 = (float)(double)agt_wave_sum_i64(a12) * FLT_SCALE;
        const float A22 = (float)(double)agt_wave_sum_i64(a22) * FLT_SCALE;

        float D = A11 * A22 - A12 * A12;
        const float minEig = (A22 + A11 - sqrtf((A11 - A22) * (A11 - A22) + 4.f * A12 * A12)) / (float)(2 * WIN * WIN);
        if (P.flags & AGT_LK_GET_MIN_EIGENVALS) errv = minEig;
        if (agt_uniform((int)((double)minEig < P.min_eig_threshold || D < FLT_EPSILON))) {
            if (level == 0) st = 0;
            continue;
        }
        D = 1.f / D;

        nextx -= halfw; nexty -= halfw;
        float pdx = 0.f, pdy = 0.f;
        int jx0 = 0, jy0 = 0, jvalid = 0;
        for (int j = 0; j < P.max_count; j++) {
            const int inx = agt_uniform((int)floorf(nextx)), iny = agt_uniform((int)floorf(nexty));
            if (inx < -WIN || inx >= LJ.w || iny < -WIN || iny >= LJ.h) {
                if (level == 0) st = 0;
                break;
            }
            if (!jvalid || inx < jx0 || inx + WIN >= jx0 + C::JT || iny < jy0 || iny + WIN >= jy0 + C::JT) {
                jx0 = inx - C::MARGIN; jy0 = iny - C::MARGIN; jvalid = 1;
                __syncthreads();
                stage_tile(imgJ, LJ.w, LJ.h, LJ.pitch, jx0, jy0, C::JT, C::JT, sJ, C::JP, lane);
                __syncthreads();
            }
            bilinear_weights(nextx - (float)inx, nexty - (float)iny, iw00, iw01, iw10, iw11);
            int b1 = 0, b2 = 0;
            if (active) {
                const uint8_t* q = sJ + (iny - jy0 + row) * C::JP + (inx - jx0 + x0) + (jx0 - (jx0 & ~3));
#pragma unroll
                for (int k = 0; k < C::RUN; k++) {
                    if (k < nx) {
                        const int diff = descale(q[k] * iw00 + q[k + 1] * iw01 + q[C::JP + k] * iw10 + q[C::JP + k + 1] * iw11,
                                                 W_BITS - 5) - Iv[k];
                        b1 += diff * Ix[k]; b2 += diff * Iy[k];
                    }
                }
            }
            const float fb1 = (float)(double)agt_wave_sum_i64(b1) * FLT_SCALE;
            const float fb2 = (float)(double)agt_wave_sum_i64(b2) * FLT_SCALE;
            const float dx = (A12 * fb2 - A22 * fb1) * D;
            const float dy = (A12 * fb1 - A11 * fb2) * D;
            nextx += dx; nexty += dy;
            outx = nextx + halfw; outy = nexty + halfw;
            if (agt_uniform((int)((double)dx * dx + (double)dy * dy <= P.eps2))) break;
            if (j > 0 && agt_uniform((int)(fabs((double)(dx + pdx)) < 0.01 && fabs((double)(dy + pdy)) < 0.01))) {
                outx -= dx * 0.5f; outy -= dy * 0.5f;
                break;
            }
            pdx = dx; pdy = dy;
        }

        if (st && P.err && level == 0 && !(P.flags & AGT_LK_GET_MIN_EIGENVALS)) {
            const float npx = outx - halfw, npy = outy - halfw;
            const int inx = agt_uniform((int)floorf(npx)), iny = agt_uniform((int)floorf(npy));
            if (inx < -WIN || inx >= LJ.w || iny < -WIN || iny >= LJ.h) { st = 0; continue; }
            if (!jvalid || inx < jx0 || inx + WIN >= jx0 + C::JT || iny < jy0 || iny + WIN >= jy0 + C::JT) {
                jx0 = inx - C::MARGIN; jy0 = iny - C::MARGIN; jvalid = 1;
                __syncthreads();
                stage_tile(imgJ, LJ.w, LJ.h, LJ.pitch, jx0, jy0, C::JT, C::JT, sJ, C::JP, lane);
                __syncthreads();
            }
            bilinear_weights(npx - (float)inx, npy - (float)iny, iw00, iw01, iw10, iw11);
            int e = 0;
            if (active) {
                const uint8_t* q = sJ + (iny - jy0 + row) * C::JP + (inx - jx0 + x0) + (jx0 - (jx0 & ~3));
#pragma unroll
                for (int k = 0; k < C::RUN; k++) {
                    if (k < nx) {
                        const int diff = descale(q[k] * iw00 + q[k + 1] * iw01 + q[C::JP + k] * iw10 + q[C::JP + k + 1] * iw11,
                                                 W_BITS - 5) - Iv[k];
                        e += diff < 0 ? -diff : diff;
                    }
                }
            }
            errv = (float)(double)agt_wave_sum_i64(e) * 1.f / (float)(32 * WIN * WIN);
        }
    }

    if (lane == 0) {
        P.next_pts[pidx * 2] = outx;
        P.next_pts[pidx * 2 + 1] = outy;
        P.status[pidx] = (uint8_t)st;
        if (P.err) P.err[pidx] = errv;
    }
}

}  // namespace

bool agt_lk_window_supported(int win) { return win == 21 || win == 15 || win == 31; }

hipError_t agt_launch_lk(hipStream_t stream, const AgtLkParams& p, int win, int B)
{
    dim3 grid(p.n, B), block(AGT_WAVE);
    switch (win) {
    case 21: hipLaunchKernelGGL(lk_kernel<21>, grid, block, 0, stream, p); break;
    case 15: hipLaunchKernelGGL(lk_kernel<15>, grid, block, 0, stream, p); break;
    case 31: hipLaunchKernelGGL(lk_kernel<31>, grid, block, 0, stream, p); break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}
