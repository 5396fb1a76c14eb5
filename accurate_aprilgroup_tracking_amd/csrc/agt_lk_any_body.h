// agt_lk_any_body.h -- cv::calcOpticalFlowPyrLK for ANY window: winSize = (ww, wh), 3 <= ww, wh <= 63, square or not (round 6).
// The compiled-in windows (agt_lk_body.h: 15, 21, 31 -- the north-star's 21 x 21 with its specialised bodies) fix every tile size, loop
// count and register array at compile time; this body is the same algorithm with run-time sizes: one workgroup of four waves per corner,
// one pyramid level at a time, the I patch (value and both derivatives of every window pixel) kept in LDS instead of registers, plain
// strided loops over tiles and window.  It is the GENERAL path, not a tuned one: bit-identical results (oracle: oracle/cv_lk.c with
// win_w / win_h); measured against the compiled-in bodies on the same windows (tools/lkanybench.py, 1280x720, 48 corners per stream): one stream
// 36 against 18 us (21 x 21), 32 against 21 (15), 41 against 41 (31); 64 streams 151 against 28 us, 134 against 30, 191 against 89 -- four waves
// per corner cost a big batch its throughput.  Reference call shape: cv.calcOpticalFlowPyrLK(prev, next, pts, None,
// winSize=(w, h), ...) -- the reference has no call site (the hole at detect_pose.py:573-574); OpenCV: lkpyramid.cpp LKTrackerInvoker.
#pragma once
#include "agt_lk_body.h"

namespace agt_lk {

constexpr int ANY_T = 4 * AGT_WAVE;         // threads per corner
constexpr int ANY_MARGIN = 9;               // search margin of the J tile (as the four-wave bodies)
constexpr int ANY_WIN_MAX = 63;

struct AnyGeom {
    int ww, wh;
    int iw, ih, indw, ip;                   // I tile: (ww + 3) x (wh + 3) pixels, rows of indw aligned dwords
    int jw, jh, jndw, jp;                   // J tile: (ww + 1 + 2 M) x (wh + 1 + 2 M)
    int dw, dh;                             // derivative tile: (ww + 1) x (wh + 1) packed (dx | dy << 16)
    int off_j, off_d, off_p, off_s, bytes;  // LDS byte offsets: I tile at 0
};

__host__ __device__ inline AnyGeom any_geom(int ww, int wh)
{
    AnyGeom g;
    g.ww = ww; g.wh = wh;
    g.iw = ww + 3; g.ih = wh + 3; g.indw = (g.iw + 6) / 4; g.ip = 4 * g.indw;
    g.jw = ww + 1 + 2 * ANY_MARGIN; g.jh = wh + 1 + 2 * ANY_MARGIN; g.jndw = (g.jw + 6) / 4; g.jp = 4 * g.jndw;
    g.dw = ww + 1; g.dh = wh + 1;
    auto up = [](int v) { return (v + 15) & ~15; };
    g.off_j = up(g.ih * g.ip);
    g.off_d = g.off_j + up(g.jh * g.jp);
    g.off_p = g.off_d + up(g.dw * g.dh * 4);
    g.off_s = g.off_p + up(ww * wh * 8);                // patch: { iv | ix << 16, iy } per window pixel
    g.bytes = g.off_s + 2 * 4 * 3 * (int)sizeof(long long);     // two phases x four waves x three sums
    return g;
}

// rows [ty0, ty0 + th) x ndw aligned dwords from (tx0 & ~3), reflect-101 outside the image (tile_load / tile_store with run-time sizes)
__device__ __forceinline__ void any_tile(uint8_t* s, const uint8_t* __restrict__ img, int w, int h, long pitch, int tx0, int ty0, int th, int ndw, int tid)
{
    const int ax0 = tx0 & ~3;
    for (int i = tid; i < th * ndw; i += ANY_T) {
        const int r = i / ndw, c4 = i - r * ndw;
        const int gy = agt_reflect101(ty0 + r, h), gx = ax0 + 4 * c4;
        const uint8_t* row = img + (long)gy * pitch;
        *reinterpret_cast<uint32_t*>(s + 4 * i) = (gx >= 0 && gx + 3 < w) ? *reinterpret_cast<const uint32_t*>(row + gx) : load_dword_reflect(row, gx, w);
    }
}

// exact sums over the workgroup of NV ints per thread (per-thread partials fit int32: <= 16 window pixels per thread, each product below
// 8160 * 4080), identical in every thread; double-buffered slots: one barrier per sum
template <int NV>
__device__ __forceinline__ void any_block_sum(const int (&v)[NV], long long (&out)[NV], long long* slots, int& phase, int wave, int lane)
{
    long long* s = slots + phase * 12;
#pragma unroll
    for (int i = 0; i < NV; i++) {
        long long t = (long long)v[i];
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) t += __shfl_xor(t, off, AGT_WAVE);
        if (lane == 0) s[wave * 3 + i] = t;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NV; i++) out[i] = s[i] + s[3 + i] + s[6 + i] + s[9 + i];
    phase ^= 1;
}

// Track corner `pt` of stream `b`; all 256 threads of the workgroup.  lds: any_geom(ww, wh).bytes, 16-byte aligned.
__device__ __forceinline__ void lk_body_any(const AgtLkParams* P, int pt, int b, uint8_t* lds, int ww, int wh)
{
    const AnyGeom G = any_geom(ww, wh);
    uint8_t* sI = lds;
    uint8_t* sJ = lds + G.off_j;
    int* sD = reinterpret_cast<int*>(lds + G.off_d);
    int2* sP = reinterpret_cast<int2*>(lds + G.off_p);
    long long* slots = reinterpret_cast<long long*>(lds + G.off_s);
    int phase = 0;
    const int tid = (int)threadIdx.x, lane = tid & (AGT_WAVE - 1), wave = tid / AGT_WAVE;
    const long pidx = (long)b * P->n + pt;
    LkFrameIo<1> io;
    io.grouped = false; io.prev_pts = P->prev_pts; io.next_pts = P->next_pts; io.status = P->status; io.err = P->err;
    io.have_pos = false; io.px = io.py = 0.f; io.pst = 1;

    const float halfx = (ww - 1) * 0.5f, halfy = (wh - 1) * 0.5f;
    const float FLT_SCALE = 1.f / (1 << 20);
    const float ppx = P->prev_pts[pidx * 2], ppy = P->prev_pts[pidx * 2 + 1];
    // tracker mode: a lost corner stays lost, position carried (agt_lk_body.h)
    if (P->prev_status && !agt_uniform((int)P->prev_status[pidx])) {
        if (tid == 0) lk_publish(io, pidx, b, ppx, ppy, 0, 0.f);
        return;
    }
    float outx = 0.f, outy = 0.f;
    const bool flow = (P->flags & AGT_LK_USE_INITIAL_FLOW) != 0;
    if (flow) { outx = P->next_pts[pidx * 2]; outy = P->next_pts[pidx * 2 + 1]; }
    if (!agt_uniform((int)(lk_pt_ok(ppx, ppy) && (!flow || lk_pt_ok(outx, outy))))) {        // wild coordinates: lost, flow or position carried
        const float cx = flow ? outx : ppx, cy = flow ? outy : ppy;
        if (tid == 0) lk_publish(io, pidx, b, cx, cy, 0, 0.f);
        return;
    }

    int st = 1;
    float errv = 0.f;
    for (int level = P->max_level; level >= 0; level--) {
        const AgtLevel LI = get_level(P->prev[level]);
        const AgtLevel LJ = get_level(P->next[level]);
        const uint8_t* imgI = LI.ptr + (long)b * LI.bstride;
        const uint8_t* imgJ = LJ.ptr + (long)b * LJ.bstride;
        const float scale = lk_level_scale(level);
        float prevx = ppx * scale, prevy = ppy * scale;
        float nextx, nexty;
        if (level == P->max_level) {
            if (flow) { nextx = outx * scale; nexty = outy * scale; }
            else { nextx = prevx; nexty = prevy; }
        } else { nextx = outx * 2.f; nexty = outy * 2.f; }
        outx = nextx; outy = nexty;

        prevx -= halfx; prevy -= halfy;
        const int ipx = agt_uniform((int)floorf(prevx)), ipy = agt_uniform((int)floorf(prevy));
        if (ipx < -ww || ipx >= LI.w || ipy < -wh || ipy >= LI.h) {
            if (level == 0) { st = 0; errv = 0.f; }
            continue;
        }
        int iw00, iw01, iw10, iw11;
        bilinear_weights(prevx - (float)ipx, prevy - (float)ipy, iw00, iw01, iw10, iw11);

        nextx -= halfx; nexty -= halfy;
        const int inx0 = agt_uniform((int)floorf(nextx)), iny0 = agt_uniform((int)floorf(nexty));
        int jx0 = inx0 - ANY_MARGIN, jy0 = iny0 - ANY_MARGIN;
        __syncthreads();                                       // the previous level's readers of every tile are done
        any_tile(sI, imgI, LI.w, LI.h, LI.pitch, ipx - 1, ipy - 1, G.ih, G.indw, tid);
        // (a search that starts outside the image -- a far initial flow -- ends at the first iteration's bounds test: no tile for it, and no
        // reflection of coordinates a million pixels out)
        if (!(inx0 < -ww || inx0 >= LJ.w || iny0 < -wh || iny0 >= LJ.h)) any_tile(sJ, imgJ, LJ.w, LJ.h, LJ.pitch, jx0, jy0, G.jh, G.jndw, tid);
        __syncthreads();

        // ---- Scharr from the I tile -> derivative tile (zero outside the image: the derivative image has a BORDER_CONSTANT border)
        const int offI = (ipx - 1) - ((ipx - 1) & ~3);
        for (int idx = tid; idx < G.dw * G.dh; idx += ANY_T) {
            const int dyy = idx / G.dw, dxx = idx - dyy * G.dw;
            const int gx = ipx + dxx, gy = ipy + dyy;
            int val = 0;
            if (gx >= 0 && gx < LI.w && gy >= 0 && gy < LI.h) {
                const uint8_t* c = sI + (dyy + 1) * G.ip + (dxx + 1) + offI;
                const int v00 = c[-G.ip - 1], v01 = c[-G.ip], v02 = c[-G.ip + 1];
                const int v10 = c[-1], v12 = c[1];
                const int v20 = c[G.ip - 1], v21 = c[G.ip], v22 = c[G.ip + 1];
                const int dx = (3 * (v02 + v22) + 10 * v12) - (3 * (v00 + v20) + 10 * v10);
                const int dy = 3 * ((v20 - v00) + (v22 - v02)) + 10 * (v21 - v01);
                val = (dx & 0xffff) | (dy << 16);
            }
            sD[idx] = val;
        }
        __syncthreads();

        // ---- the I patch (to LDS) + exact covariance sums
        int asum[3] = { 0, 0, 0 };
        for (int p = tid; p < ww * wh; p += ANY_T) {
            const int y = p / ww, x = p - y * ww;
            const uint8_t* q = sI + (y + 1) * G.ip + (x + 1) + offI;
            const int iv = descale(bil4(q[0], q[1], q[G.ip], q[G.ip + 1], iw00, iw01, iw10, iw11), W_BITS - 5);
            const int* d = sD + y * G.dw + x;
            const int d00 = d[0], d01 = d[1], d10 = d[G.dw], d11 = d[G.dw + 1];
            const int ix = descale(bil4((short)d00, (short)d01, (short)d10, (short)d11, iw00, iw01, iw10, iw11), W_BITS);
            const int iy = descale(bil4(d00 >> 16, d01 >> 16, d10 >> 16, d11 >> 16, iw00, iw01, iw10, iw11), W_BITS);
            sP[p] = make_int2((iv & 0xffff) | (ix << 16), iy);
            asum[0] += __mul24(ix, ix); asum[1] += __mul24(ix, iy); asum[2] += __mul24(iy, iy);
        }
        long long at[3];
        any_block_sum<3>(asum, at, slots, phase, wave, lane);
        const float A11 = (float)(double)at[0] * FLT_SCALE;
        const float A12 = (float)(double)at[1] * FLT_SCALE;
        const float A22 = (float)(double)at[2] * FLT_SCALE;

        float D = A11 * A22 - A12 * A12;
        const float minEig = (A22 + A11 - sqrtf((A11 - A22) * (A11 - A22) + 4.f * A12 * A12)) / (float)(2 * ww * wh);
        if (P->flags & AGT_LK_GET_MIN_EIGENVALS) errv = minEig;
        if (agt_uniform((int)((double)minEig < P->min_eig_threshold || D < FLT_EPSILON))) {
            if (level == 0) st = 0;
            continue;
        }
        D = 1.f / D;

        float pdx = 0.f, pdy = 0.f;
        auto restage_j = [&](int inx, int iny) {
            jx0 = inx - ANY_MARGIN; jy0 = iny - ANY_MARGIN;
            __syncthreads();
            any_tile(sJ, imgJ, LJ.w, LJ.h, LJ.pitch, jx0, jy0, G.jh, G.jndw, tid);
            __syncthreads();
        };
        // sums of the window's temporal differences against the patch: { diff * Ix, diff * Iy } or { |diff| }
        auto window_pass = [&](int inx, int iny, bool want_abs, int (&acc)[2]) {
            acc[0] = 0; acc[1] = 0;
            const uint8_t* q0 = sJ + (iny - jy0) * G.jp + (inx - jx0) + (jx0 - (jx0 & ~3));
            for (int p = tid; p < ww * wh; p += ANY_T) {
                const int y = p / ww, x = p - y * ww;
                const uint8_t* q = q0 + y * G.jp + x;
                const int2 pv = sP[p];
                const int diff = descale(bil4(q[0], q[1], q[G.jp], q[G.jp + 1], iw00, iw01, iw10, iw11), W_BITS - 5) - (int)(short)(pv.x & 0xffff);
                if (want_abs) acc[0] += diff < 0 ? -diff : diff;
                else { acc[0] += __mul24(diff, pv.x >> 16); acc[1] += __mul24(diff, pv.y); }
            }
        };
        auto outside_tile = [&](int inx, int iny) { return inx < jx0 || inx + ww >= jx0 + G.jw || iny < jy0 || iny + wh >= jy0 + G.jh; };
        for (int j = 0; j < P->max_count; j++) {
            const int inx = agt_uniform((int)floorf(nextx)), iny = agt_uniform((int)floorf(nexty));
            if (inx < -ww || inx >= LJ.w || iny < -wh || iny >= LJ.h) {
                if (level == 0) st = 0;
                break;
            }
            if (outside_tile(inx, iny)) restage_j(inx, iny);
            bilinear_weights(nextx - (float)inx, nexty - (float)iny, iw00, iw01, iw10, iw11);
            int bsum[2];
            window_pass(inx, iny, false, bsum);
            long long bt[2];
            any_block_sum<2>(bsum, bt, slots, phase, wave, lane);
            const float fb1 = (float)(double)bt[0] * FLT_SCALE;
            const float fb2 = (float)(double)bt[1] * FLT_SCALE;
            const float dx = (A12 * fb2 - A22 * fb1) * D;
            const float dy = (A12 * fb1 - A11 * fb2) * D;
            nextx += dx; nexty += dy;
            outx = nextx + halfx; outy = nexty + halfy;
            if (agt_uniform((int)((double)dx * dx + (double)dy * dy <= P->eps2))) break;
            if (j > 0 && agt_uniform((int)(fabs((double)(dx + pdx)) < 0.01 && fabs((double)(dy + pdy)) < 0.01))) {
                outx -= dx * 0.5f; outy -= dy * 0.5f;
                break;
            }
            pdx = dx; pdy = dy;
        }

        if (st && io.err && level == 0 && !(P->flags & AGT_LK_GET_MIN_EIGENVALS)) {
            const float npx = outx - halfx, npy = outy - halfy;
            const int inx = agt_uniform((int)floorf(npx)), iny = agt_uniform((int)floorf(npy));
            if (inx < -ww || inx >= LJ.w || iny < -wh || iny >= LJ.h) { st = 0; continue; }
            if (outside_tile(inx, iny)) restage_j(inx, iny);
            bilinear_weights(npx - (float)inx, npy - (float)iny, iw00, iw01, iw10, iw11);
            int esum[2];
            window_pass(inx, iny, true, esum);
            int e1[1] = { esum[0] };
            long long et[1];
            any_block_sum<1>(e1, et, slots, phase, wave, lane);
            errv = (float)(double)et[0] * 1.f / (float)(32 * ww * wh);
        }
    }
    if (tid == 0) lk_publish(io, pidx, b, outx, outy, st, errv);
}

}  // namespace agt_lk
