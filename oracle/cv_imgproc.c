/*
 * cv_imgproc.c -- CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE).  See cv_oracle.h.
 *
 * Restates the frame pre-processing that precedes the pose path in the reference
 * (SURVEY.md section 8f rank 1):
 *   PoseDetector.undistort_frame   /root/reference/aprilgroup_tracking/aprilgroup_pose_estimation/detect_pose.py:147-183
 *       cv.getOptimalNewCameraMatrix(mtx, dist, (w,h), 1, (w,h))  -> calibration.cpp cvGetOptimalNewCameraMatrix + icvGetRectangles
 *       cv.undistort(frame, mtx, dist, None, newK)                -> undistort.dispatch.cpp initUndistortRectifyMap (CV_16SC2)
 *                                                                    + imgwarp.cpp remap(INTER_LINEAR, BORDER_CONSTANT)
 *       ROI crop
 *   cv.cvtColor(frame, COLOR_BGR2GRAY)   detect_pose.py:602    -> color_rgb RGB2Gray<uchar> (14-bit fixed point)
 * PARITY UNPINNED against real cv2.
 *
 * Documented deviations from the OpenCV sources:
 *   - the undistortion map evaluates the back-projected ray as j*ir[0] + (i*ir[1] + ir[2]) per pixel,
 *     OpenCV's scalar loop accumulates _x += ir[0] along the row (its AVX2 loop differs again);
 *     the two agree to ~1e-13 relative, i.e. a 1/32-pixel map entry can differ with
 *     probability ~1e-11 per pixel;
 *   - cv.undistort builds its maps stripe by stripe (shifting cy per stripe); here one map covers
 *     the frame.
 *   - the bilinear weight table is used in exact form; OpenCV's short table clamps the single
 *     weight 32768 to 32767 and adds the missing 1 to the opposite tap, which yields the same
 *     8-bit results.
 */
#include "cv_oracle.h"
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <float.h>

#define INTER_BITS 5
#define INTER_TAB_SIZE (1 << INTER_BITS)
#define REMAP_COEF_BITS 15

static int round_half_even_d(double v) { return (int)lrint(v); }     /* cvRound / saturate_cast<int>(double) */
static int round_half_even_f(float v) { return (int)lrintf(v); }

/* color_rgb.simd.hpp RGB2Gray<uchar>: (B*1868 + G*9617 + R*4899 + 2^13) >> 14 */
int cvo_cvt_bgr2gray(const uint8_t* src, int w, int h, int sstride, uint8_t* dst, int dstride)
{
    if (!src || !dst || w <= 0 || h <= 0) return -1;
    for (int y = 0; y < h; y++) {
        const uint8_t* s = src + (size_t)y * sstride;
        uint8_t* d = dst + (size_t)y * dstride;
        for (int x = 0; x < w; x++)
            d[x] = (uint8_t)((s[3 * x] * 1868 + s[3 * x + 1] * 9617 + s[3 * x + 2] * 4899 + (1 << 13)) >> 14);
    }
    return 0;
}

/* cvUndistortPointsInternal for one point with R = I and an optional new camera matrix P, criteria (COUNT, 5) */
static void undistort_point(double u, double v, const double K[9], const double k[14], int has_dist,
                            const double* P, double* ox, double* oy)
{
    const double fx = K[0], fy = K[4], ifx = 1. / fx, ify = 1. / fy, cx = K[2], cy = K[5];
    double x = (u - cx) * ifx, y = (v - cy) * ify;
    if (has_dist) {
        double Ti[9], vu[3];
        cvo_tilt_of(k, NULL, Ti);
        for (int r = 0; r < 3; r++) { double a = 0; a += Ti[r * 3] * x; a += Ti[r * 3 + 1] * y; a += Ti[r * 3 + 2] * 1; vu[r] = a; }
        const double invProj = vu[2] ? 1. / vu[2] : 1;
        x = invProj * vu[0]; y = invProj * vu[1];
        const double x0 = x, y0 = y;
        for (int j = 0; j < 5; j++) {
            double r2 = x * x + y * y;
            double icdist = (1 + ((k[7] * r2 + k[6]) * r2 + k[5]) * r2) / (1 + ((k[4] * r2 + k[1]) * r2 + k[0]) * r2);
            if (icdist < 0) { x = (u - cx) * ifx; y = (v - cy) * ify; break; }
            double deltaX = 2 * k[2] * x * y + k[3] * (r2 + 2 * x * x) + k[8] * r2 + k[9] * r2 * r2;
            double deltaY = k[2] * (r2 + 2 * y * y) + 2 * k[3] * x * y + k[10] * r2 + k[11] * r2 * r2;
            x = (x0 - deltaX) * icdist;
            y = (y0 - deltaY) * icdist;
        }
    }
    if (P) {
        double xx = P[0] * x + P[1] * y + P[2], yy = P[3] * x + P[4] * y + P[5], ww = 1. / (P[6] * x + P[7] * y + P[8]);
        x = xx * ww; y = yy * ww;
    }
    *ox = x; *oy = y;
}

/* calibration.cpp icvGetRectangles: 9x9 grid (float points), undistorted, inner / outer rectangles in float */
static void get_rectangles(const double K[9], const double k[14], int has_dist, const double* P, int w, int h,
                           float inner[4], float outer[4])
{
    const int N = 9;
    float iX0 = -FLT_MAX, iX1 = FLT_MAX, iY0 = -FLT_MAX, iY1 = FLT_MAX;
    float oX0 = FLT_MAX, oX1 = -FLT_MAX, oY0 = FLT_MAX, oY1 = -FLT_MAX;
    for (int y = 0; y < N; y++)
        for (int x = 0; x < N; x++) {
            float px = (float)x * w / (N - 1), py = (float)y * h / (N - 1);
            double ux, uy;
            undistort_point((double)px, (double)py, K, k, has_dist, P, &ux, &uy);
            float fx = (float)ux, fy = (float)uy;
            if (fx < oX0) oX0 = fx;
            if (fx > oX1) oX1 = fx;
            if (fy < oY0) oY0 = fy;
            if (fy > oY1) oY1 = fy;
            if (x == 0 && fx > iX0) iX0 = fx;
            if (x == N - 1 && fx < iX1) iX1 = fx;
            if (y == 0 && fy > iY0) iY0 = fy;
            if (y == N - 1 && fy < iY1) iY1 = fy;
        }
    inner[0] = iX0; inner[1] = iY0; inner[2] = iX1 - iX0; inner[3] = iY1 - iY0;
    outer[0] = oX0; outer[1] = oY0; outer[2] = oX1 - oX0; outer[3] = oY1 - oY0;
}

/* cv::getOptimalNewCameraMatrix(K, dist, (w,h), alpha, (new_w,new_h), centerPrincipalPoint = false)
 * -> newK (3x3 row-major), roi (x, y, width, height) */
int cvo_get_optimal_new_camera_matrix(const double K[9], const double* dist, int ndist, int w, int h, double alpha,
                                      int new_w, int new_h, double newK[9], int roi[4])
{
    if (!K || !newK || w <= 0 || h <= 0) return -1;
    if (ndist != 0 && ndist != 4 && ndist != 5 && ndist != 8 && ndist != 12 && ndist != 14) return -3;
    double k[14];
    memset(k, 0, sizeof(k));
    int has_dist = 0;
    for (int i = 0; i < ndist && i < 14; i++) { k[i] = dist[i]; }
    has_dist = dist != NULL && ndist > 0;
    if (new_w * new_h == 0) { new_w = w; new_h = h; }
    alpha = alpha < 0. ? 0. : alpha > 1. ? 1. : alpha;       /* cvGetOptimalNewCameraMatrix clamps with MIN/MAX */
    float inner[4], outer[4];
    get_rectangles(K, k, has_dist, NULL, w, h, inner, outer);
    double fx0 = (new_w - 1) / inner[2], fy0 = (new_h - 1) / inner[3];
    double cx0 = -fx0 * inner[0], cy0 = -fy0 * inner[1];
    double fx1 = (new_w - 1) / outer[2], fy1 = (new_h - 1) / outer[3];
    double cx1 = -fx1 * outer[0], cy1 = -fy1 * outer[1];
    memcpy(newK, K, 9 * sizeof(double));
    newK[0] = fx0 * (1 - alpha) + fx1 * alpha;
    newK[4] = fy0 * (1 - alpha) + fy1 * alpha;
    newK[2] = cx0 * (1 - alpha) + cx1 * alpha;
    newK[5] = cy0 * (1 - alpha) + cy1 * alpha;
    if (roi) {
        get_rectangles(K, k, has_dist, newK, w, h, inner, outer);
        int rx = round_half_even_f(inner[0]), ry = round_half_even_f(inner[1]);
        int rw = round_half_even_f(inner[2]), rh = round_half_even_f(inner[3]);
        /* r &= Rect(0, 0, new_w, new_h) */
        int x1 = rx > 0 ? rx : 0, y1 = ry > 0 ? ry : 0;
        int x2 = rx + rw < new_w ? rx + rw : new_w, y2 = ry + rh < new_h ? ry + rh : new_h;
        if (x2 <= x1 || y2 <= y1) { roi[0] = roi[1] = roi[2] = roi[3] = 0; }
        else { roi[0] = x1; roi[1] = y1; roi[2] = x2 - x1; roi[3] = y2 - y1; }
    }
    return 0;
}

static int inv3(const double A[9], double B[9])
{
    double c00 = A[4] * A[8] - A[5] * A[7], c01 = A[5] * A[6] - A[3] * A[8], c02 = A[3] * A[7] - A[4] * A[6];
    double det = A[0] * c00 + A[1] * c01 + A[2] * c02;
    if (det == 0) return -1;
    double id = 1. / det;
    B[0] = c00 * id; B[1] = (A[2] * A[7] - A[1] * A[8]) * id; B[2] = (A[1] * A[5] - A[2] * A[4]) * id;
    B[3] = c01 * id; B[4] = (A[0] * A[8] - A[2] * A[6]) * id; B[5] = (A[2] * A[3] - A[0] * A[5]) * id;
    B[6] = c02 * id; B[7] = (A[1] * A[6] - A[0] * A[7]) * id; B[8] = (A[0] * A[4] - A[1] * A[3]) * id;
    return 0;
}

/* cv::initUndistortRectifyMap(K, dist, R = I, newK, (w,h), CV_16SC2) -> map1 (w*h*2 int16: x, y), map2 (w*h uint16) */
int cvo_init_undistort_rectify_map(const double K[9], const double* dist, int ndist, const double newK[9],
                                   int w, int h, int16_t* map1, uint16_t* map2)
{
    if (!K || !newK || !map1 || !map2 || w <= 0 || h <= 0) return -1;
    if (ndist != 0 && ndist != 4 && ndist != 5 && ndist != 8 && ndist != 12 && ndist != 14) return -3;
    double k[14], ir[9];
    memset(k, 0, sizeof(k));
    for (int i = 0; i < ndist && i < 14; i++) k[i] = dist ? dist[i] : 0.;
    if (inv3(newK, ir)) return -1;
    const double u0 = K[2], v0 = K[5], fx = K[0], fy = K[4];
    const double k1 = k[0], k2 = k[1], p1 = k[2], p2 = k[3], k3 = k[4], k4 = k[5], k5 = k[6], k6 = k[7];
    const double s1 = k[8], s2 = k[9], s3 = k[10], s4 = k[11];
    double T[9];
    cvo_tilt_of(k, T, NULL);
    for (int i = 0; i < h; i++)
        for (int j = 0; j < w; j++) {
            double _x = j * ir[0] + (i * ir[1] + ir[2]), _y = j * ir[3] + (i * ir[4] + ir[5]), _w = j * ir[6] + (i * ir[7] + ir[8]);
            double ww = 1. / _w, x = _x * ww, y = _y * ww;
            double x2 = x * x, y2 = y * y, r2 = x2 + y2, _2xy = 2 * x * y;
            double kr = (1 + ((k3 * r2 + k2) * r2 + k1) * r2) / (1 + ((k6 * r2 + k5) * r2 + k4) * r2);
            double xd = (x * kr + p1 * _2xy + p2 * (r2 + 2 * x2) + s1 * r2 + s2 * r2 * r2);
            double yd = (y * kr + p1 * (r2 + 2 * y2) + p2 * _2xy + s3 * r2 + s4 * r2 * r2);
            double vt[3];
            for (int r = 0; r < 3; r++) { double a = 0; a += T[r * 3] * xd; a += T[r * 3 + 1] * yd; a += T[r * 3 + 2] * 1; vt[r] = a; }
            double invProj = vt[2] ? 1. / vt[2] : 1;
            double u = fx * invProj * vt[0] + u0, v = fy * invProj * vt[1] + v0;
            double us = u * INTER_TAB_SIZE, vs = v * INTER_TAB_SIZE;
            /* saturate_cast<int>(double) */
            int iu = us >= 2147483647. ? 2147483647 : us <= -2147483648. ? (-2147483647 - 1) : round_half_even_d(us);
            int iv = vs >= 2147483647. ? 2147483647 : vs <= -2147483648. ? (-2147483647 - 1) : round_half_even_d(vs);
            size_t o = (size_t)i * w + j;
            map1[o * 2] = (int16_t)(iu >> INTER_BITS);
            map1[o * 2 + 1] = (int16_t)(iv >> INTER_BITS);
            map2[o] = (uint16_t)((iv & (INTER_TAB_SIZE - 1)) * INTER_TAB_SIZE + (iu & (INTER_TAB_SIZE - 1)));
        }
    return 0;
}

/* imgwarp.cpp remapBilinear<FixedPtCast<int, uchar, 15>, ..., short>, BORDER_CONSTANT (0), cn channels.
 * dst has the size of the maps (dw x dh). */
int cvo_remap_bilinear_u8(const uint8_t* src, int sw, int sh, int sstride, int cn,
                          const int16_t* map1, const uint16_t* map2, int dw, int dh,
                          uint8_t* dst, int dstride)
{
    if (!src || !dst || !map1 || !map2 || cn < 1 || cn > 4) return -1;
    for (int y = 0; y < dh; y++) {
        uint8_t* D = dst + (size_t)y * dstride;
        for (int x = 0; x < dw; x++) {
            size_t o = (size_t)y * dw + x;
            int sx = map1[o * 2], sy = map1[o * 2 + 1];
            int fxq = map2[o] & (INTER_TAB_SIZE - 1), fyq = (map2[o] >> INTER_BITS) & (INTER_TAB_SIZE - 1);
            int w00 = (INTER_TAB_SIZE - fxq) * (INTER_TAB_SIZE - fyq) * 32, w01 = fxq * (INTER_TAB_SIZE - fyq) * 32;
            int w10 = (INTER_TAB_SIZE - fxq) * fyq * 32, w11 = fxq * fyq * 32;
            if ((unsigned)sx < (unsigned)(sw - 1) && (unsigned)sy < (unsigned)(sh - 1)) {
                const uint8_t* S = src + (size_t)sy * sstride + sx * cn;
                for (int c = 0; c < cn; c++)
                    D[x * cn + c] = (uint8_t)((S[c] * w00 + S[c + cn] * w01 + S[c + sstride] * w10 + S[c + sstride + cn] * w11 +
                                               (1 << (REMAP_COEF_BITS - 1))) >> REMAP_COEF_BITS);
            } else if (sx >= sw || sx + 1 < 0 || sy >= sh || sy + 1 < 0) {
                for (int c = 0; c < cn; c++) D[x * cn + c] = 0;
            } else {
                int in00 = sx >= 0 && sy >= 0 && sx < sw && sy < sh, in01 = sx + 1 >= 0 && sy >= 0 && sx + 1 < sw && sy < sh;
                int in10 = sx >= 0 && sy + 1 >= 0 && sx < sw && sy + 1 < sh, in11 = sx + 1 >= 0 && sy + 1 >= 0 && sx + 1 < sw && sy + 1 < sh;
                for (int c = 0; c < cn; c++) {
                    int v00 = in00 ? src[(size_t)sy * sstride + sx * cn + c] : 0;
                    int v01 = in01 ? src[(size_t)sy * sstride + (sx + 1) * cn + c] : 0;
                    int v10 = in10 ? src[(size_t)(sy + 1) * sstride + sx * cn + c] : 0;
                    int v11 = in11 ? src[(size_t)(sy + 1) * sstride + (sx + 1) * cn + c] : 0;
                    D[x * cn + c] = (uint8_t)((v00 * w00 + v01 * w01 + v10 * w10 + v11 * w11 + (1 << (REMAP_COEF_BITS - 1))) >> REMAP_COEF_BITS);
                }
            }
        }
    }
    return 0;
}

/* cv::undistort(src (BGR u8), K, dist, None, newK) -> dst of the same size */
int cvo_undistort_u8(const uint8_t* src, int w, int h, int sstride, int cn, const double K[9], const double* dist, int ndist,
                     const double newK[9], uint8_t* dst, int dstride)
{
    int16_t* m1 = (int16_t*)malloc((size_t)w * h * 2 * sizeof(int16_t));
    uint16_t* m2 = (uint16_t*)malloc((size_t)w * h * sizeof(uint16_t));
    if (!m1 || !m2) { free(m1); free(m2); return -2; }
    int rc = cvo_init_undistort_rectify_map(K, dist, ndist, newK, w, h, m1, m2);
    if (!rc) rc = cvo_remap_bilinear_u8(src, w, h, sstride, cn, m1, m2, w, h, dst, dstride);
    free(m1); free(m2);
    return rc;
}
