/*
 * cv_lk.c -- CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE).  See cv_oracle.h.
 *
 * Restates cv::calcOpticalFlowPyrLK for 8-bit single-channel images:
 *   pyrDown            -> OpenCV modules/imgproc/src/pyramids.cpp  pyrDown_<FixPtCast<uchar,8>>
 *   pyramid + borders  -> OpenCV modules/video/src/lkpyramid.cpp   buildOpticalFlowPyramid
 *   Scharr image       -> lkpyramid.cpp calcSharrDeriv
 *   per-point tracker  -> lkpyramid.cpp cv::detail::LKTrackerInvoker::operator()
 * The reference has no call site for this step (north-star; hole at
 * /root/reference/aprilgroup_tracking/aprilgroup_pose_estimation/detect_pose.py:573-574).
 * PARITY UNPINNED against real cv2.
 *
 * Data flow is kept as OpenCV's (full padded pyramids, a full-frame Scharr image per
 * level, per-point fixed-point tracker, points parallelised with OpenMP as parallel_for_
 * does) so that timing this code is a fair stand-in for the reference CPU path.
 *
 * Compile with -ffp-contract=off: the float expressions below must evaluate exactly as
 * written (OpenCV's baseline x86-64 build has no FMA contraction in this file).
 */
#include "cv_oracle.h"
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <float.h>
#ifdef _OPENMP
#include <omp.h>
#endif

struct cvo_level {
    int w, h;        /* image size (un-padded) */
    int padx, pady;  /* border on each side */
    int stride;      /* bytes per padded row */
    uint8_t* buf;    /* padded buffer */
};
struct cvo_pyramid {
    int nlevels;     /* number of stored levels = max usable level + 1 */
    struct cvo_level lv[CVO_MAX_LEVELS];
};

/* threads for the full-frame passes (pyrDown, Scharr, the derivative buffer's zero fill): bands of rows, as OpenCV's
 * parallel_for_ stripes.  Set by cvo_set_num_threads / cvo_track_frame; 1 = serial.  Band edges do not change any
 * value: every output row depends on its own source rows only. */
static int g_frame_threads = 1;
void cvo_set_num_threads(int n) { g_frame_threads = n > 0 ? n : 1; }
int cvo_get_num_threads(void) { return g_frame_threads; }

/* cv::borderInterpolate(p, len, BORDER_REFLECT_101) */
static inline int reflect101(int p, int len)
{
    if ((unsigned)p < (unsigned)len) return p;
    if (len == 1) return 0;
    do {
        if (p < 0) p = -p;
        else p = 2 * (len - 1) - p;
    } while ((unsigned)p >= (unsigned)len);
    return p;
}

/* ------------------------------------------------------------------------- */
/* pyrDown (pyramids.cpp): separable [1 4 6 4 1], rows cached in a 5-slot ring */
int cvo_pyr_down_u8(const uint8_t* src, int sw, int sh, int sstride,
                    uint8_t* dst, int dstride)
{
    if (!src || !dst || sw <= 0 || sh <= 0) return -1;
    const int dw = (sw + 1) / 2, dh = (sh + 1) / 2;
    int nb = g_frame_threads < dh / 8 ? g_frame_threads : dh / 8;
    if (nb < 1) nb = 1;
    int* rings = (int*)malloc((size_t)nb * 5 * dw * sizeof(int));
    int* xtab = (int*)malloc((size_t)5 * dw * sizeof(int));
    if (!rings || !xtab) { free(rings); free(xtab); return -2; }
    for (int x = 0; x < dw; x++)
        for (int k = 0; k < 5; k++)
            xtab[x * 5 + k] = reflect101(2 * x - 2 + k, sw);

#ifdef _OPENMP
#pragma omp parallel for num_threads(nb) schedule(static)
#endif
    for (int band = 0; band < nb; band++) {
    int* ring = rings + (size_t)band * 5 * dw;
    int slot_row[5] = { -1, -1, -1, -1, -1 };
    const int y_lo = (int)((long)dh * band / nb), y_hi = (int)((long)dh * (band + 1) / nb);
    for (int y = y_lo; y < y_hi; y++) {
        const int* rows[5];
        for (int k = 0; k < 5; k++) {
            int sy = reflect101(2 * y - 2 + k, sh);
            int slot = sy % 5;
            int* r = ring + (size_t)slot * dw;
            if (slot_row[slot] != sy) {
                const uint8_t* s = src + (size_t)sy * sstride;
                /* interior fast path, borders through the table */
                for (int x = 0; x < dw; x++) {
                    const int* t = xtab + x * 5;
                    r[x] = s[t[2]] * 6 + (s[t[1]] + s[t[3]]) * 4 + s[t[0]] + s[t[4]];
                }
                slot_row[slot] = sy;
            }
            rows[k] = r;
        }
        uint8_t* d = dst + (size_t)y * dstride;
        for (int x = 0; x < dw; x++) {
            int v = rows[2][x] * 6 + (rows[1][x] + rows[3][x]) * 4 + rows[0][x] + rows[4][x];
            d[x] = (uint8_t)((v + 128) >> 8);   /* FixPtCast<uchar, 8> */
        }
    }
    }
    free(rings); free(xtab);
    return 0;
}

/* copyMakeBorder(..., BORDER_REFLECT_101) of the level's own interior */
static void fill_border101(struct cvo_level* L)
{
    uint8_t* base = L->buf + (size_t)L->pady * L->stride + L->padx;
    for (int y = 0; y < L->h; y++) {
        uint8_t* row = base + (size_t)y * L->stride;
        for (int x = -L->padx; x < 0; x++) row[x] = row[reflect101(x, L->w)];
        for (int x = L->w; x < L->w + L->padx; x++) row[x] = row[reflect101(x, L->w)];
    }
    for (int y = -L->pady; y < L->h + L->pady; y++) {
        if (y >= 0 && y < L->h) continue;
        int sy = reflect101(y, L->h);
        memcpy(base + (ptrdiff_t)y * L->stride - L->padx,
               base + (ptrdiff_t)sy * L->stride - L->padx, (size_t)L->w + 2 * L->padx);
    }
}

static int level_alloc(struct cvo_level* L, int w, int h, int padx, int pady)
{
    L->w = w; L->h = h; L->padx = padx; L->pady = pady;
    L->stride = (w + 2 * padx + 15) & ~15;
    L->buf = (uint8_t*)malloc((size_t)L->stride * (h + 2 * pady));
    return L->buf ? 0 : -1;
}

static inline const uint8_t* level_origin(const struct cvo_level* L)
{
    return L->buf + (size_t)L->pady * L->stride + L->padx;
}

/* lkpyramid.cpp buildOpticalFlowPyramid(img, pyr, winSize, maxLevel, withDerivatives=false,
 * pyrBorder=BORDER_REFLECT_101, ...) */
cvo_pyramid* cvo_pyramid_build(const uint8_t* img, int w, int h, int stride,
                               int win_w, int win_h, int max_level)
{
    if (!img || w <= 0 || h <= 0 || max_level < 0 || max_level >= CVO_MAX_LEVELS) return NULL;
    cvo_pyramid* P = (cvo_pyramid*)calloc(1, sizeof(*P));
    if (!P) return NULL;
    int cw = w, ch = h;
    for (int level = 0; level <= max_level; level++) {
        struct cvo_level* L = &P->lv[level];
        if (level_alloc(L, cw, ch, win_w, win_h)) { cvo_pyramid_free(P); return NULL; }
        uint8_t* o = (uint8_t*)level_origin(L);
        if (level == 0) {
            for (int y = 0; y < ch; y++) memcpy(o + (size_t)y * L->stride, img + (size_t)y * stride, (size_t)cw);
        } else {
            const struct cvo_level* Pv = &P->lv[level - 1];
            cvo_pyr_down_u8(level_origin(Pv), Pv->w, Pv->h, Pv->stride, o, L->stride);
        }
        fill_border101(L);
        P->nlevels = level + 1;
        /* early stop exactly as OpenCV: the NEXT level would be <= winSize */
        cw = (cw + 1) / 2; ch = (ch + 1) / 2;
        if (cw <= win_w || ch <= win_h) break;
    }
    return P;
}

void cvo_pyramid_free(cvo_pyramid* p)
{
    if (!p) return;
    for (int i = 0; i < CVO_MAX_LEVELS; i++) free(p->lv[i].buf);
    free(p);
}

int cvo_pyramid_levels(const cvo_pyramid* p) { return p ? p->nlevels - 1 : -1; }

int cvo_pyramid_level_size(const cvo_pyramid* p, int level, int* w, int* h)
{
    if (!p || level < 0 || level >= p->nlevels) return -1;
    if (w) *w = p->lv[level].w;
    if (h) *h = p->lv[level].h;
    return 0;
}

int cvo_pyramid_level_copy(const cvo_pyramid* p, int level, uint8_t* dst, int dstride)
{
    if (!p || level < 0 || level >= p->nlevels) return -1;
    const struct cvo_level* L = &p->lv[level];
    const uint8_t* o = level_origin(L);
    for (int y = 0; y < L->h; y++) memcpy(dst + (size_t)y * dstride, o + (size_t)y * L->stride, (size_t)L->w);
    return 0;
}

/* ------------------------------------------------------------------------- */
/* lkpyramid.cpp calcSharrDeriv: un-normalised Scharr, reflect-101 inside the image */
int cvo_scharr_deriv(const uint8_t* src, int w, int h, int sstride,
                     int16_t* dst, int dstride)
{
    if (!src || !dst || w <= 0 || h <= 0) return -1;
    int nb = g_frame_threads < h / 8 ? g_frame_threads : h / 8;
    if (nb < 1) nb = 1;
    int16_t* bufs = (int16_t*)malloc((size_t)nb * 2 * (w + 2) * sizeof(int16_t));
    if (!bufs) return -2;
#ifdef _OPENMP
#pragma omp parallel for num_threads(nb) schedule(static)
#endif
    for (int band = 0; band < nb; band++) {
    int16_t* buf = bufs + (size_t)band * 2 * (w + 2);
    int16_t* trow0 = buf + 1;
    int16_t* trow1 = buf + (w + 2) + 1;
    const int y_lo = (int)((long)h * band / nb), y_hi = (int)((long)h * (band + 1) / nb);
    for (int y = y_lo; y < y_hi; y++) {
        const uint8_t* srow0 = src + (size_t)(y > 0 ? y - 1 : h > 1 ? 1 : 0) * sstride;
        const uint8_t* srow1 = src + (size_t)y * sstride;
        const uint8_t* srow2 = src + (size_t)(y < h - 1 ? y + 1 : h > 1 ? h - 2 : 0) * sstride;
        int16_t* drow = dst + (size_t)y * dstride;
        for (int x = 0; x < w; x++) {
            int t0 = (srow0[x] + srow2[x]) * 3 + srow1[x] * 10;
            int t1 = srow2[x] - srow0[x];
            trow0[x] = (int16_t)t0;
            trow1[x] = (int16_t)t1;
        }
        int x0 = w > 1 ? 1 : 0, x1 = w > 1 ? w - 2 : 0;
        trow0[-1] = trow0[x0]; trow0[w] = trow0[x1];
        trow1[-1] = trow1[x0]; trow1[w] = trow1[x1];
        for (int x = 0; x < w; x++) {
            int16_t t0 = (int16_t)(trow0[x + 1] - trow0[x - 1]);
            int16_t t1 = (int16_t)((trow1[x + 1] + trow1[x - 1]) * 3 + trow1[x] * 10);
            drow[x * 2] = t0; drow[x * 2 + 1] = t1;
        }
    }
    }
    free(bufs);
    return 0;
}

/* ------------------------------------------------------------------------- */
#define W_BITS 14
#define DESCALE(x, n) (((x) + (1 << ((n) - 1))) >> (n))

static inline int cv_round_f(float v) { return (int)lrintf(v); }  /* round-half-even, as cvRound */
static inline int cv_floor_f(float v) { return (int)floorf(v); }

struct lk_level_ctx {
    const struct cvo_level* I;
    const struct cvo_level* J;
    const int16_t* deriv;      /* padded derivative image, origin pointer */
    int dstep;                 /* int16 elements per padded derivative row */
    int level, max_level;
    int win_w, win_h;
    int max_count; double eps2;
    int flags; double min_eig_threshold;
    int acc_mode;
};

/* development counters (not thread-safe; read with nthreads=1): iterations and points per level */
long cvo_dbg_iters[CVO_MAX_LEVELS], cvo_dbg_points[CVO_MAX_LEVELS];
void cvo_dbg_reset(void) { memset(cvo_dbg_iters, 0, sizeof(cvo_dbg_iters)); memset(cvo_dbg_points, 0, sizeof(cvo_dbg_points)); }

/* ---- CVO_ACC_FLOAT_SIMD: the accumulation order of OpenCV's CV_SIMD128 (x86 SSE baseline, no FMA) loops of LKTrackerInvoker
 * [OpenCV-knowledge: 4.x modules/video/src/lkpyramid.cpp, universal intrinsics; the apt / pip builds the reference runs on].
 * A window row is consumed 8 pixels at a time while x <= ww - 8; the remaining ww % 8 pixels go through the scalar loop into the
 * scalar float accumulators (iA11.., ib1, ib2), which are added to the horizontal sum of the vector accumulators at the end.
 *   covariance: two groups of four pixels per step; lane l of qA11 / qA12 / qA22 (four float lanes, alive across rows) takes
 *     pixel 4 * group + l:  qA = qA + fx * fy  with the product rounded to float before the add (v_muladd without FMA);
 *   mismatch:   v_dotprod on int16 pairs gives EXACT int32 pair sums  d[k] * I[k] + d[k + 4] * I[k + 4]  (k = 0..3 of the
 *     8-pixel step), converted to float and added:  k = 0, 1 into qb0 (lanes x, y, x, y), k = 2, 3 into qb1;  at the end
 *     q = qb0 + qb1,  ib1 += (q[0] + 0) + (q[2] + 0),  ib2 += (q[1] + 0) + (q[3] + 0);
 *   v_reduce_sum(float32x4) = (l0 + l2) + (l1 + l3)  (SSE: add the high half onto the low half, then lane 1 onto lane 0).
 * The level-0 error sum has no vector form there: scalar order. */
struct simd_acc3 { float q11[4], q12[4], q22[4]; float t11, t12, t22; };
static inline float simd_reduce4(const float* q) { float lo = q[0] + q[2], hi = q[1] + q[3]; return lo + hi; }

/* LKTrackerInvoker::operator() for one point */
static void lk_track_point(const struct lk_level_ctx* c, const float* prev_pts, float* next_pts,
                           uint8_t* status, float* err, int ptidx, int16_t* patch /* 3*win area */)
{
    const int ww = c->win_w, wh = c->win_h, level = c->level;
    const float halfx = (ww - 1) * 0.5f, halfy = (wh - 1) * 0.5f;
    const float scale = (float)(1. / (1 << level));
    const int Icols = c->I->w, Irows = c->I->h;
    const int Jcols = c->J->w, Jrows = c->J->h;
    const int stepI = c->I->stride, stepJ = c->J->stride, dstep = c->dstep;
    const uint8_t* Ibase = level_origin(c->I);
    const uint8_t* Jbase = level_origin(c->J);
    const float FLT_SCALE = 1.f / (1 << 20);

    float prevx = prev_pts[ptidx * 2] * scale, prevy = prev_pts[ptidx * 2 + 1] * scale;
    float nextx, nexty;
    if (level == c->max_level) {
        if (c->flags & CVO_LK_USE_INITIAL_FLOW) {
            nextx = next_pts[ptidx * 2] * scale; nexty = next_pts[ptidx * 2 + 1] * scale;
        } else { nextx = prevx; nexty = prevy; }
    } else {
        nextx = next_pts[ptidx * 2] * 2.f; nexty = next_pts[ptidx * 2 + 1] * 2.f;
    }
    next_pts[ptidx * 2] = nextx; next_pts[ptidx * 2 + 1] = nexty;

    prevx -= halfx; prevy -= halfy;
    int ipx = cv_floor_f(prevx), ipy = cv_floor_f(prevy);
    if (ipx < -ww || ipx >= Icols || ipy < -wh || ipy >= Irows) {
        if (level == 0) { status[ptidx] = 0; if (err) err[ptidx] = 0; }
        return;
    }
    float a = prevx - ipx, b = prevy - ipy;
    int iw00 = cv_round_f((1.f - a) * (1.f - b) * (1 << W_BITS));
    int iw01 = cv_round_f(a * (1.f - b) * (1 << W_BITS));
    int iw10 = cv_round_f((1.f - a) * b * (1 << W_BITS));
    int iw11 = (1 << W_BITS) - iw00 - iw01 - iw10;

    int16_t* Iwin = patch;                 /* ww*wh */
    int16_t* dIwin = patch + ww * wh;      /* 2*ww*wh */
    int64_t sA11 = 0, sA12 = 0, sA22 = 0;
    float fA11 = 0, fA12 = 0, fA22 = 0;
    struct simd_acc3 qa; memset(&qa, 0, sizeof(qa));
    const int simd_w = c->acc_mode == CVO_ACC_FLOAT_SIMD ? (ww / 8) * 8 : 0;     /* window columns the 8-pixel vector loop covers */
    for (int y = 0; y < wh; y++) {
        const uint8_t* src = Ibase + (ptrdiff_t)(y + ipy) * stepI + ipx;
        const int16_t* dsrc = c->deriv + (ptrdiff_t)(y + ipy) * dstep + ipx * 2;
        int16_t* Ip = Iwin + y * ww;
        int16_t* dIp = dIwin + y * ww * 2;
        for (int x = 0; x < ww; x++, dsrc += 2, dIp += 2) {
            int ival = DESCALE(src[x] * iw00 + src[x + 1] * iw01 + src[x + stepI] * iw10 + src[x + stepI + 1] * iw11, W_BITS - 5);
            int ixval = DESCALE(dsrc[0] * iw00 + dsrc[2] * iw01 + dsrc[dstep] * iw10 + dsrc[dstep + 2] * iw11, W_BITS);
            int iyval = DESCALE(dsrc[1] * iw00 + dsrc[3] * iw01 + dsrc[dstep + 1] * iw10 + dsrc[dstep + 3] * iw11, W_BITS);
            Ip[x] = (int16_t)ival; dIp[0] = (int16_t)ixval; dIp[1] = (int16_t)iyval;
            if (c->acc_mode == CVO_ACC_EXACT) {
                sA11 += (int64_t)ixval * ixval; sA12 += (int64_t)ixval * iyval; sA22 += (int64_t)iyval * iyval;
            } else if (x < simd_w) {
                /* vector lane x % 4: float(ix) * float(iy) rounded, then added (no FMA) */
                const float fx = (float)ixval, fy = (float)iyval;
                const float p11 = fx * fx, p12 = fx * fy, p22 = fy * fy;
                qa.q11[x & 3] = qa.q11[x & 3] + p11; qa.q12[x & 3] = qa.q12[x & 3] + p12; qa.q22[x & 3] = qa.q22[x & 3] + p22;
            } else {
                fA11 += (float)(ixval * ixval); fA12 += (float)(ixval * iyval); fA22 += (float)(iyval * iyval);
            }
        }
    }
    float A11, A12, A22;
    if (c->acc_mode == CVO_ACC_EXACT) {
        /* exact sum (< 2^53) rounded once to float */
        A11 = (float)(double)sA11 * FLT_SCALE; A12 = (float)(double)sA12 * FLT_SCALE; A22 = (float)(double)sA22 * FLT_SCALE;
    } else {
        if (simd_w) { fA11 += simd_reduce4(qa.q11); fA12 += simd_reduce4(qa.q12); fA22 += simd_reduce4(qa.q22); }
        A11 = fA11 * FLT_SCALE; A12 = fA12 * FLT_SCALE; A22 = fA22 * FLT_SCALE;
    }
    float D = A11 * A22 - A12 * A12;
    float minEig = (A22 + A11 - sqrtf((A11 - A22) * (A11 - A22) + 4.f * A12 * A12)) / (2 * ww * wh);
    if (err && (c->flags & CVO_LK_GET_MIN_EIGENVALS)) err[ptidx] = minEig;
    if ((double)minEig < c->min_eig_threshold || D < FLT_EPSILON) {
        if (level == 0) status[ptidx] = 0;
        return;
    }
    D = 1.f / D;

    nextx -= halfx; nexty -= halfy;
    float pdx = 0, pdy = 0;
    cvo_dbg_points[level]++;
    for (int j = 0; j < c->max_count; j++) {
        cvo_dbg_iters[level]++;
        int inx = cv_floor_f(nextx), iny = cv_floor_f(nexty);
        if (inx < -ww || inx >= Jcols || iny < -wh || iny >= Jrows) {
            if (level == 0) status[ptidx] = 0;
            break;
        }
        a = nextx - inx; b = nexty - iny;
        iw00 = cv_round_f((1.f - a) * (1.f - b) * (1 << W_BITS));
        iw01 = cv_round_f(a * (1.f - b) * (1 << W_BITS));
        iw10 = cv_round_f((1.f - a) * b * (1 << W_BITS));
        iw11 = (1 << W_BITS) - iw00 - iw01 - iw10;
        int64_t sb1 = 0, sb2 = 0; float fb1 = 0, fb2 = 0;
        float qb0[4] = { 0, 0, 0, 0 }, qb1[4] = { 0, 0, 0, 0 };
        for (int y = 0; y < wh; y++) {
            const uint8_t* Jp = Jbase + (ptrdiff_t)(y + iny) * stepJ + inx;
            const int16_t* Ip = Iwin + y * ww;
            const int16_t* dIp = dIwin + y * ww * 2;
            int x = 0;
            for (; x < simd_w; x += 8, dIp += 16) {
                int d8[8];
                for (int k = 0; k < 8; k++)
                    d8[k] = (int16_t)(DESCALE(Jp[x + k] * iw00 + Jp[x + k + 1] * iw01 + Jp[x + k + stepJ] * iw10 + Jp[x + k + stepJ + 1] * iw11, W_BITS - 5) - Ip[x + k]);
                /* v_dotprod: int32 pair sums of pixels (k, k + 4); k = 0, 1 -> qb0 lanes (x, y, x, y), k = 2, 3 -> qb1 */
                for (int k = 0; k < 4; k++) {
                    const int px = d8[k] * dIp[2 * k] + d8[k + 4] * dIp[2 * (k + 4)];
                    const int py = d8[k] * dIp[2 * k + 1] + d8[k + 4] * dIp[2 * (k + 4) + 1];
                    float* q = k < 2 ? qb0 : qb1;
                    q[2 * (k & 1)] = q[2 * (k & 1)] + (float)px; q[2 * (k & 1) + 1] = q[2 * (k & 1) + 1] + (float)py;
                }
            }
            for (; x < ww; x++, dIp += 2) {
                int diff = DESCALE(Jp[x] * iw00 + Jp[x + 1] * iw01 + Jp[x + stepJ] * iw10 + Jp[x + stepJ + 1] * iw11, W_BITS - 5) - Ip[x];
                if (c->acc_mode == CVO_ACC_EXACT) {
                    sb1 += (int64_t)diff * dIp[0]; sb2 += (int64_t)diff * dIp[1];
                } else {
                    fb1 += (float)(diff * dIp[0]); fb2 += (float)(diff * dIp[1]);
                }
            }
        }
        float b1, b2;
        if (c->acc_mode == CVO_ACC_EXACT) { b1 = (float)(double)sb1 * FLT_SCALE; b2 = (float)(double)sb2 * FLT_SCALE; }
        else {
            if (simd_w) {
                /* v_recombine(v_interleave_pairs(qb0 + qb1), 0): qf0 = (X0, X1, 0, 0), qf1 = (Y0, Y1, 0, 0); v_reduce_sum of each */
                float q[4]; for (int k = 0; k < 4; k++) q[k] = qb0[k] + qb1[k];
                const float qf0[4] = { q[0], q[2], 0.f, 0.f }, qf1[4] = { q[1], q[3], 0.f, 0.f };
                fb1 += simd_reduce4(qf0); fb2 += simd_reduce4(qf1);
            }
            b1 = fb1 * FLT_SCALE; b2 = fb2 * FLT_SCALE;
        }
        float dx = (A12 * b2 - A22 * b1) * D;
        float dy = (A12 * b1 - A11 * b2) * D;
        nextx += dx; nexty += dy;
        next_pts[ptidx * 2] = nextx + halfx; next_pts[ptidx * 2 + 1] = nexty + halfy;
        if ((double)dx * dx + (double)dy * dy <= c->eps2) break;     /* delta.ddot(delta) */
        if (j > 0 && fabs((double)(dx + pdx)) < 0.01 && fabs((double)(dy + pdy)) < 0.01) {
            next_pts[ptidx * 2] -= dx * 0.5f; next_pts[ptidx * 2 + 1] -= dy * 0.5f;
            break;
        }
        pdx = dx; pdy = dy;
    }

    if (status[ptidx] && err && level == 0 && !(c->flags & CVO_LK_GET_MIN_EIGENVALS)) {
        float npx = next_pts[ptidx * 2] - halfx, npy = next_pts[ptidx * 2 + 1] - halfy;
        int inx = cv_floor_f(npx), iny = cv_floor_f(npy);
        if (inx < -ww || inx >= Jcols || iny < -wh || iny >= Jrows) { status[ptidx] = 0; return; }
        float aa = npx - inx, bb = npy - iny;
        iw00 = cv_round_f((1.f - aa) * (1.f - bb) * (1 << W_BITS));
        iw01 = cv_round_f(aa * (1.f - bb) * (1 << W_BITS));
        iw10 = cv_round_f((1.f - aa) * bb * (1 << W_BITS));
        iw11 = (1 << W_BITS) - iw00 - iw01 - iw10;
        float errval = 0.f; int64_t serr = 0;
        for (int y = 0; y < wh; y++) {
            const uint8_t* Jp = Jbase + (ptrdiff_t)(y + iny) * stepJ + inx;
            const int16_t* Ip = Iwin + y * ww;
            for (int x = 0; x < ww; x++) {
                int diff = DESCALE(Jp[x] * iw00 + Jp[x + 1] * iw01 + Jp[x + stepJ] * iw10 + Jp[x + stepJ + 1] * iw11, W_BITS - 5) - Ip[x];
                if (c->acc_mode == CVO_ACC_EXACT) serr += diff < 0 ? -diff : diff;
                else errval += fabsf((float)diff);
            }
        }
        if (c->acc_mode == CVO_ACC_EXACT) errval = (float)(double)serr;
        err[ptidx] = errval * 1.f / (32 * ww * wh);
    }
}

int cvo_lk_on_pyramids(const cvo_pyramid* prev_pyr, const cvo_pyramid* next_pyr,
                       const float* prev_pts, float* next_pts,
                       uint8_t* status, float* err, int npoints,
                       int win_w, int win_h, int max_level,
                       int crit_type, int crit_max_count, double crit_eps,
                       int flags, double min_eig_threshold,
                       int acc_mode, int nthreads)
{
    if (!prev_pyr || !next_pyr || !prev_pts || !next_pts || !status) return -1;
    if (win_w <= 2 || win_h <= 2 || npoints < 0) return -1;
    if (npoints == 0) return 0;
    /* SparsePyrLKOpticalFlowImpl::calc: criteria normalisation */
    int max_count = (crit_type & CVO_TERM_COUNT) ? (crit_max_count < 0 ? 0 : crit_max_count > 100 ? 100 : crit_max_count) : 30;
    double eps = (crit_type & CVO_TERM_EPS) ? (crit_eps < 0. ? 0. : crit_eps > 10. ? 10. : crit_eps) : 0.01;
    eps *= eps;
    int ml = max_level;
    if (prev_pyr->nlevels - 1 < ml) ml = prev_pyr->nlevels - 1;
    if (next_pyr->nlevels - 1 < ml) ml = next_pyr->nlevels - 1;
    for (int i = 0; i < npoints; i++) status[i] = 1;
    if (err) for (int i = 0; i < npoints; i++) err[i] = 0.f;   /* OpenCV leaves it uninitialised; we define 0 */

    /* one derivative buffer sized for level 0, re-used per level (derivIBuf) */
    const struct cvo_level* L0 = &prev_pyr->lv[0];
    size_t dcap = (size_t)(L0->w + 2 * win_w) * 2 * (size_t)(L0->h + 2 * win_h);
    int16_t* dbuf = (int16_t*)malloc(dcap * sizeof(int16_t));
    if (!dbuf) return -2;
#ifdef _OPENMP
    int nt = nthreads > 0 ? nthreads : 1;
#else
    int nt = 1; (void)nthreads;
#endif
    int16_t* patches = (int16_t*)malloc((size_t)nt * 3 * win_w * win_h * sizeof(int16_t));
    if (!patches) { free(dbuf); return -2; }

    for (int level = ml; level >= 0; level--) {
        const struct cvo_level* I = &prev_pyr->lv[level];
        const struct cvo_level* J = &next_pyr->lv[level];
        int dstep = (I->w + 2 * win_w) * 2;
        /* copyMakeBorder(derivI, _derivI, ..., BORDER_CONSTANT): zero padding */
        {
            const int rows = I->h + 2 * win_h, nbz = g_frame_threads < rows / 8 ? (g_frame_threads > 0 ? g_frame_threads : 1) : 1;
#ifdef _OPENMP
#pragma omp parallel for num_threads(nbz) schedule(static)
#endif
            for (int band = 0; band < nbz; band++) {
                const int r0 = (int)((long)rows * band / nbz), r1 = (int)((long)rows * (band + 1) / nbz);
                memset(dbuf + (size_t)r0 * dstep, 0, (size_t)dstep * (r1 - r0) * sizeof(int16_t));
            }
        }
        int16_t* dorg = dbuf + (size_t)win_h * dstep + win_w * 2;
        cvo_scharr_deriv(level_origin(I), I->w, I->h, I->stride, dorg, dstep);

        struct lk_level_ctx c;
        c.I = I; c.J = J; c.deriv = dorg; c.dstep = dstep; c.level = level; c.max_level = ml;
        c.win_w = win_w; c.win_h = win_h; c.max_count = max_count; c.eps2 = eps;
        c.flags = flags; c.min_eig_threshold = min_eig_threshold; c.acc_mode = acc_mode;
#ifdef _OPENMP
#pragma omp parallel for num_threads(nt) schedule(static)
#endif
        for (int i = 0; i < npoints; i++) {
#ifdef _OPENMP
            int16_t* patch = patches + (size_t)omp_get_thread_num() * 3 * win_w * win_h;
#else
            int16_t* patch = patches;
#endif
            lk_track_point(&c, prev_pts, next_pts, status, err, i, patch);
        }
    }
    free(patches); free(dbuf);
    return 0;
}

int cvo_calc_optical_flow_pyr_lk(const uint8_t* prev_img, const uint8_t* next_img,
                                 int w, int h, int stride,
                                 const float* prev_pts, float* next_pts,
                                 uint8_t* status, float* err, int npoints,
                                 int win_w, int win_h, int max_level,
                                 int crit_type, int crit_max_count, double crit_eps,
                                 int flags, double min_eig_threshold,
                                 int acc_mode, int nthreads)
{
    cvo_pyramid* P = cvo_pyramid_build(prev_img, w, h, stride, win_w, win_h, max_level);
    cvo_pyramid* N = cvo_pyramid_build(next_img, w, h, stride, win_w, win_h, max_level);
    int rc = -2;
    if (P && N)
        rc = cvo_lk_on_pyramids(P, N, prev_pts, next_pts, status, err, npoints, win_w, win_h, max_level,
                                crit_type, crit_max_count, crit_eps, flags, min_eig_threshold, acc_mode, nthreads);
    cvo_pyramid_free(P); cvo_pyramid_free(N);
    return rc;
}
