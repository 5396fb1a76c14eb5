/*
 * cv_dense.c -- CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE).  See cv_oracle.h.
 *
 * Dense photometric pose refinement ("DPR", BASELINE.json configs[4]).
 * THERE IS NO REFERENCE CODE for this step: the reference only links the DodecaPen paper
 * (/root/reference/README.md:20).  The semantics below are therefore DEFINED BY THIS BUILD; the
 * oracle and the HIP kernels (csrc/agt_dense.hip) implement exactly this specification.
 *
 * SPECIFICATION
 *   inputs   gray frame I (u8, w x h); M model samples (X_i in the object frame, f32 x 3, and a
 *            template intensity T_i, f32); N corner correspondences (object point f32 x 3, image
 *            point f32 x 2, optional u8 mask); camera K, dist (cv2 layout); pose p = (rvec, tvec), f64.
 *   residuals
 *     geometric   e_j = project(obj_j; p) - img_j                      (2 per unmasked corner, pixels;
 *                 projection and its 2x6 Jacobian as cv.projectPoints)
 *     photometric r_i = I~(u_i) - T_i,  u_i = project(X_i; p)          (1 per valid sample)
 *                 I~ is the bilinear interpolation of I at u_i = (x, y); with x0 = floor(x), y0 = floor(y)
 *                 the sample is VALID iff 1 <= x0 <= w - 3 and 1 <= y0 <= h - 3 (all taps inside);
 *                 invalid samples contribute nothing.  The image gradient is the bilinear
 *                 interpolation of central differences  gx(a,b) = (I(a+1,b) - I(a-1,b)) / 2,
 *                 gy(a,b) = (I(a,b+1) - I(a,b-1)) / 2  taken at the four integer neighbours.
 *                 Jacobian row: dr_i/dp = gx * du/dp + gy * dv/dp   (1 x 6).
 *   cost     E(p) = sum_j |e_j|^2 + photo_weight * sum_i r_i^2
 *   solver   damped Gauss-Newton, no step rejection:
 *              repeat `iters` times:
 *                 A = J^T J,  g = J^T r   over both residual sets (photometric rows scaled by photo_weight)
 *                 A_kk *= (1 + mu)                               (mu = 1e-3: CvLevMarq's initial damping)
 *                 solve A dx = g;  p <- p - dx
 *                 stop early when |dx| / (|p_before| + DBL_EPSILON) < FLT_EPSILON
 *   outputs  p; stats = { photometric RMS at the last linearisation point, geometric RMS there,
 *            valid samples there, iterations executed, used corners, 0, 0, 0 }.
 *   All arithmetic in FP64.  Sums run in index order here; the HIP kernels sum per block and then
 *   over blocks in index order, so poses agree to ~1e-12, not bitwise.
 */
#include "cv_oracle.h"
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <float.h>

static int solve6_ldl(const double A[36], const double b[6], double x[6])
{
    double L[6][6], D[6], y[6];
    for (int j = 0; j < 6; j++) {
        double d = A[j * 6 + j];
        for (int k = 0; k < j; k++) d -= L[j][k] * L[j][k] * D[k];
        if (!(d > 0.0)) return -1;
        D[j] = d;
        for (int i = j + 1; i < 6; i++) {
            double v = A[j * 6 + i];
            for (int k = 0; k < j; k++) v -= L[i][k] * L[j][k] * D[k];
            L[i][j] = v / d;
        }
    }
    for (int i = 0; i < 6; i++) { double v = b[i]; for (int k = 0; k < i; k++) v -= L[i][k] * y[k]; y[i] = v; }
    for (int i = 5; i >= 0; i--) { double v = y[i] / D[i]; for (int k = i + 1; k < 6; k++) v -= L[k][i] * x[k]; x[i] = v; }
    return 0;
}

/* photometric residual and 1x6 Jacobian row of one sample; returns 0 when the sample is invalid */
int cvo_dense_sample(const uint8_t* img, int w, int h, int stride, double u, double v, const double ju[6], const double jv[6],
                     double t, double* r, double jrow[6])
{
    const double fx0 = floor(u), fy0 = floor(v);
    if (!(fx0 >= 1.0 && fx0 <= (double)(w - 3) && fy0 >= 1.0 && fy0 <= (double)(h - 3))) return 0;
    const int x0 = (int)fx0, y0 = (int)fy0;
    const double a = u - fx0, b = v - fy0;
    const uint8_t* p = img + (size_t)y0 * stride + x0;
#define PX(dx, dy) ((double)p[(dy) * stride + (dx)])
    const double w00 = (1 - a) * (1 - b), w01 = a * (1 - b), w10 = (1 - a) * b, w11 = a * b;
    const double I = w00 * PX(0, 0) + w01 * PX(1, 0) + w10 * PX(0, 1) + w11 * PX(1, 1);
    const double gx = w00 * (PX(1, 0) - PX(-1, 0)) * 0.5 + w01 * (PX(2, 0) - PX(0, 0)) * 0.5 +
                      w10 * (PX(1, 1) - PX(-1, 1)) * 0.5 + w11 * (PX(2, 1) - PX(0, 1)) * 0.5;
    const double gy = w00 * (PX(0, 1) - PX(0, -1)) * 0.5 + w01 * (PX(1, 1) - PX(1, -1)) * 0.5 +
                      w10 * (PX(0, 2) - PX(0, 0)) * 0.5 + w11 * (PX(1, 2) - PX(1, 0)) * 0.5;
#undef PX
    *r = I - t;
    for (int k = 0; k < 6; k++) jrow[k] = gx * ju[k] + gy * jv[k];
    return 1;
}

int cvo_dense_refine(const uint8_t* img, int w, int h, int stride,
                     const float* model_xyz, const float* model_t, int M,
                     const float* obj, const float* img_pts, const uint8_t* mask, int N,
                     const double K[9], const double* dist, int ndist,
                     double pose[6], int iters, double photo_weight, double mu, double stats[8])
{
    if (!img || !pose || M < 0 || N < 0 || (M > 0 && (!model_xyz || !model_t)) || (N > 0 && (!obj || !img_pts))) return -1;
    if (ndist != 0 && ndist != 4 && ndist != 5 && ndist != 8 && ndist != 12 && ndist != 14) return -3;
    double* X = (double*)malloc((size_t)(M > N ? M : N) * 3 * sizeof(double) + 64);
    double* uv = (double*)malloc((size_t)(M > N ? M : N) * 2 * sizeof(double) + 64);
    double* dr = (double*)malloc((size_t)(M > N ? M : N) * 6 * sizeof(double) + 64);
    double* dt = (double*)malloc((size_t)(M > N ? M : N) * 6 * sizeof(double) + 64);
    if (!X || !uv || !dr || !dt) { free(X); free(uv); free(dr); free(dt); return -2; }
    double st_photo = 0, st_geo = 0, st_valid = 0, st_used = 0;
    int it_done = 0;
    for (int it = 0; it < iters; it++) {
        double JtJ[36], Jtr[6];
        memset(JtJ, 0, sizeof(JtJ)); memset(Jtr, 0, sizeof(Jtr));
        /* geometric rows */
        double e_geo = 0; int used = 0;
        if (N > 0) {
            for (int i = 0; i < N * 3; i++) X[i] = (double)obj[i];
            cvo_project_points(X, N, pose, pose + 3, K, dist, ndist, uv, dr, dt);
            for (int i = 0; i < N; i++) {
                if (mask && !mask[i]) continue;
                used++;
                for (int c = 0; c < 2; c++) {
                    double J[6] = { dr[i * 6 + c * 3], dr[i * 6 + c * 3 + 1], dr[i * 6 + c * 3 + 2],
                                    dt[i * 6 + c * 3], dt[i * 6 + c * 3 + 1], dt[i * 6 + c * 3 + 2] };
                    double e = uv[i * 2 + c] - (double)img_pts[i * 2 + c];
                    e_geo += e * e;
                    for (int a = 0; a < 6; a++) { Jtr[a] += J[a] * e; for (int b = a; b < 6; b++) JtJ[a * 6 + b] += J[a] * J[b]; }
                }
            }
        }
        /* photometric rows */
        double e_ph = 0; long valid = 0;
        if (M > 0) {
            for (int i = 0; i < M * 3; i++) X[i] = (double)model_xyz[i];
            cvo_project_points(X, M, pose, pose + 3, K, dist, ndist, uv, dr, dt);
            for (int i = 0; i < M; i++) {
                double ju[6] = { dr[i * 6], dr[i * 6 + 1], dr[i * 6 + 2], dt[i * 6], dt[i * 6 + 1], dt[i * 6 + 2] };
                double jv[6] = { dr[i * 6 + 3], dr[i * 6 + 4], dr[i * 6 + 5], dt[i * 6 + 3], dt[i * 6 + 4], dt[i * 6 + 5] };
                double r, J[6];
                if (!cvo_dense_sample(img, w, h, stride, uv[i * 2], uv[i * 2 + 1], ju, jv, (double)model_t[i], &r, J)) continue;
                valid++;
                e_ph += r * r;
                for (int a = 0; a < 6; a++) { Jtr[a] += photo_weight * J[a] * r; for (int b = a; b < 6; b++) JtJ[a * 6 + b] += photo_weight * J[a] * J[b]; }
            }
        }
        st_photo = valid ? sqrt(e_ph / valid) : 0; st_geo = used ? sqrt(e_geo / (2 * used)) : 0; st_valid = (double)valid; st_used = used;
        for (int a = 0; a < 6; a++) for (int b = 0; b < a; b++) JtJ[a * 6 + b] = JtJ[b * 6 + a];
        for (int a = 0; a < 6; a++) JtJ[a * 7] *= 1.0 + mu;
        double dx[6];
        it_done = it + 1;
        if (solve6_ldl(JtJ, Jtr, dx)) break;
        double dn = 0, pn = 0;
        for (int a = 0; a < 6; a++) { dn += dx[a] * dx[a]; pn += pose[a] * pose[a]; }
        for (int a = 0; a < 6; a++) pose[a] -= dx[a];
        if (sqrt(dn) / (sqrt(pn) + DBL_EPSILON) < FLT_EPSILON) break;
    }
    if (stats) { stats[0] = st_photo; stats[1] = st_geo; stats[2] = st_valid; stats[3] = it_done; stats[4] = st_used; stats[5] = stats[6] = stats[7] = 0; }
    free(X); free(uv); free(dr); free(dt);
    return 0;
}
