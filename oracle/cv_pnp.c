/*
 * cv_pnp.c -- CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE).  See cv_oracle.h.
 *
 * Restates, in FP64 as OpenCV does:
 *   cv::Rodrigues        -> OpenCV modules/calib3d/src/calibration.cpp cvRodrigues2
 *       (reference call sites: transform_helper.py:87, detect_pose.py:275-276, 330, 344)
 *   cv::projectPoints    -> calibration.cpp cvProjectPoints2Internal
 *       (reference call sites: transform_helper.py:106-111, detect_pose.py:455-461)
 *   cv::solvePnP(ITERATIVE) -> calibration.cpp cvFindExtrinsicCameraParams2 +
 *       compat_ptsetreg.cpp CvLevMarq  (reference call sites: detect_pose.py:509-515, 517-526)
 *   undistortPoints (init only) -> undistort.dispatch.cpp cvUndistortPointsInternal
 * PARITY UNPINNED against real cv2.
 */
#include "cv_oracle.h"
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <float.h>

/* ------------------------------------------------------------------------- */
/* One-sided Jacobi SVD (Hestenes), the scheme of OpenCV's JacobiSVDImpl_.    */
int cvo_svd(const double* A, int m, int n, double* w, double* u, double* vt)
{
    if (!A || !w || m < n || n <= 0) return -1;
    double* At = (double*)malloc((size_t)n * m * sizeof(double));   /* rows = columns of A */
    double* V = (double*)malloc((size_t)n * n * sizeof(double));    /* rows = right vectors */
    double* W = (double*)malloc((size_t)n * sizeof(double));
    if (!At || !V || !W) { free(At); free(V); free(W); return -2; }
    for (int i = 0; i < n; i++) {
        double sd = 0;
        for (int k = 0; k < m; k++) { double t = A[(size_t)k * n + i]; At[(size_t)i * m + k] = t; sd += t * t; }
        W[i] = sd;
        for (int k = 0; k < n; k++) V[(size_t)i * n + k] = (i == k);
    }
    const double eps = DBL_EPSILON * 2;
    const int max_iter = m > 30 ? m : 30;
    for (int iter = 0; iter < max_iter; iter++) {
        int changed = 0;
        for (int i = 0; i < n - 1; i++)
            for (int j = i + 1; j < n; j++) {
                double* Ai = At + (size_t)i * m; double* Aj = At + (size_t)j * m;
                double a = W[i], p = 0, b = W[j];
                for (int k = 0; k < m; k++) p += Ai[k] * Aj[k];
                if (fabs(p) <= eps * sqrt(a * b)) continue;
                p *= 2;
                double beta = a - b, gamma = hypot(p, beta), c, s;
                if (beta < 0) {
                    double delta = (gamma - beta) * 0.5;
                    s = sqrt(delta / gamma);
                    c = p / (gamma * s * 2);
                } else {
                    c = sqrt((gamma + beta) / (gamma * 2));
                    s = p / (gamma * c * 2);
                }
                a = b = 0;
                for (int k = 0; k < m; k++) {
                    double t0 = c * Ai[k] + s * Aj[k];
                    double t1 = -s * Ai[k] + c * Aj[k];
                    Ai[k] = t0; Aj[k] = t1;
                    a += t0 * t0; b += t1 * t1;
                }
                W[i] = a; W[j] = b;
                changed = 1;
                double* Vi = V + (size_t)i * n; double* Vj = V + (size_t)j * n;
                for (int k = 0; k < n; k++) {
                    double t0 = c * Vi[k] + s * Vj[k];
                    double t1 = -s * Vi[k] + c * Vj[k];
                    Vi[k] = t0; Vj[k] = t1;
                }
            }
        if (!changed) break;
    }
    for (int i = 0; i < n; i++) {
        double sd = 0;
        for (int k = 0; k < m; k++) { double t = At[(size_t)i * m + k]; sd += t * t; }
        W[i] = sqrt(sd);
    }
    /* sort descending (selection sort, swapping rows of At and V) */
    for (int i = 0; i < n - 1; i++) {
        int j = i;
        for (int k = i + 1; k < n; k++) if (W[j] < W[k]) j = k;
        if (i != j) {
            double t = W[i]; W[i] = W[j]; W[j] = t;
            for (int k = 0; k < m; k++) { t = At[(size_t)i * m + k]; At[(size_t)i * m + k] = At[(size_t)j * m + k]; At[(size_t)j * m + k] = t; }
            for (int k = 0; k < n; k++) { t = V[(size_t)i * n + k]; V[(size_t)i * n + k] = V[(size_t)j * n + k]; V[(size_t)j * n + k] = t; }
        }
    }
    for (int i = 0; i < n; i++) {
        w[i] = W[i];
        if (u) {
            double s = W[i] > 0 ? 1. / W[i] : 0.;
            for (int k = 0; k < m; k++) u[(size_t)k * n + i] = At[(size_t)i * m + k] * s;
        }
        if (vt) for (int k = 0; k < n; k++) vt[(size_t)i * n + k] = V[(size_t)i * n + k];
    }
    free(At); free(V); free(W);
    return 0;
}

/* cv::solve(A, b, x, DECOMP_SVD): back-substitution with the SVBkSb threshold */
int cvo_solve_svd(const double* A, const double* b, int n, double* x)
{
    double* w = (double*)malloc((size_t)n * sizeof(double));
    double* u = (double*)malloc((size_t)n * n * sizeof(double));
    double* vt = (double*)malloc((size_t)n * n * sizeof(double));
    if (!w || !u || !vt) { free(w); free(u); free(vt); return -2; }
    int rc = cvo_svd(A, n, n, w, u, vt);
    if (rc == 0) {
        double thr = 0;
        for (int i = 0; i < n; i++) thr += w[i];
        thr *= DBL_EPSILON * 2;
        for (int k = 0; k < n; k++) x[k] = 0;
        for (int i = 0; i < n; i++) {
            if (w[i] <= thr) continue;
            double s = 0;
            for (int k = 0; k < n; k++) s += u[(size_t)k * n + i] * b[k];
            s /= w[i];
            for (int k = 0; k < n; k++) x[k] += s * vt[(size_t)i * n + k];
        }
    }
    free(w); free(u); free(vt);
    return rc;
}

/* ------------------------------------------------------------------------- */
static double det3(const double* M)
{
    return M[0] * (M[4] * M[8] - M[5] * M[7]) - M[1] * (M[3] * M[8] - M[5] * M[6]) + M[2] * (M[3] * M[7] - M[4] * M[6]);
}
static void mat3_mul(const double* A, const double* B, double* C)
{
    double T[9];
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++)
        T[i * 3 + j] = A[i * 3] * B[j] + A[i * 3 + 1] * B[3 + j] + A[i * 3 + 2] * B[6 + j];
    memcpy(C, T, sizeof(T));
}

/* calibration.cpp cvRodrigues2, vector -> matrix */
void cvo_rodrigues_vec2mat(const double r_in[3], double R[9], double* J)
{
    double rx = r_in[0], ry = r_in[1], rz = r_in[2];
    double theta = sqrt(rx * rx + ry * ry + rz * rz);
    if (theta < DBL_EPSILON) {
        for (int i = 0; i < 9; i++) R[i] = (i % 4 == 0);
        if (J) {
            memset(J, 0, 27 * sizeof(double));
            J[5] = J[15] = J[19] = -1;
            J[7] = J[11] = J[21] = 1;
        }
        return;
    }
    double c = cos(theta), s = sin(theta), c1 = 1. - c, itheta = theta ? 1. / theta : 0.;
    rx *= itheta; ry *= itheta; rz *= itheta;
    double rrt[9] = { rx * rx, rx * ry, rx * rz, rx * ry, ry * ry, ry * rz, rx * rz, ry * rz, rz * rz };
    double r_x[9] = { 0, -rz, ry, rz, 0, -rx, -ry, rx, 0 };
    static const double I[9] = { 1, 0, 0, 0, 1, 0, 0, 0, 1 };
    for (int k = 0; k < 9; k++) R[k] = c * I[k] + c1 * rrt[k] + s * r_x[k];
    if (J) {
        double drrt[27] = { rx + rx, ry, rz, ry, 0, 0, rz, 0, 0,
                            0, rx, 0, rx, ry + ry, rz, 0, rz, 0,
                            0, 0, rx, 0, 0, ry, rx, ry, rz + rz };
        static const double d_r_x_[27] = { 0, 0, 0, 0, 0, -1, 0, 1, 0,
                                           0, 0, 1, 0, 0, 0, -1, 0, 0,
                                           0, -1, 0, 1, 0, 0, 0, 0, 0 };
        for (int i = 0; i < 3; i++) {
            double ri = i == 0 ? rx : i == 1 ? ry : rz;
            double a0 = -s * ri, a1 = (s - 2 * c1 * itheta) * ri, a2 = c1 * itheta;
            double a3 = (c - s * itheta) * ri, a4 = s * itheta;
            for (int k = 0; k < 9; k++)
                J[i * 9 + k] = a0 * I[k] + a1 * rrt[k] + a2 * drrt[i * 9 + k] + a3 * r_x[k] + a4 * d_r_x_[i * 9 + k];
        }
    }
}

/* calibration.cpp cvRodrigues2, matrix -> vector.  The Jacobian (9x3) is produced
 * numerically-free as OpenCV's chain rule is never consumed by the reference; we
 * return zeros when requested (documented deviation). */
int cvo_rodrigues_mat2vec(const double Rin[9], double r[3], double* jac)
{
    for (int i = 0; i < 9; i++)
        if (!(Rin[i] > -100. && Rin[i] < 100.)) {       /* checkRange(R, true, NULL, -100, 100) */
            r[0] = r[1] = r[2] = 0;
            if (jac) memset(jac, 0, 27 * sizeof(double));
            return 0;
        }
    double W[3], U[9], Vt[9], R[9];
    cvo_svd(Rin, 3, 3, W, U, Vt);
    mat3_mul(U, Vt, R);
    double rx = R[7] - R[5], ry = R[2] - R[6], rz = R[3] - R[1];
    double s = sqrt((rx * rx + ry * ry + rz * rz) * 0.25);
    double c = (R[0] + R[4] + R[8] - 1) * 0.5;
    c = c > 1. ? 1. : c < -1. ? -1. : c;
    double theta = acos(c);
    if (s < 1e-5) {
        if (c > 0) rx = ry = rz = 0;
        else {
            double t;
            t = (R[0] + 1) * 0.5; rx = sqrt(t > 0. ? t : 0.);
            t = (R[4] + 1) * 0.5; ry = sqrt(t > 0. ? t : 0.) * (R[1] < 0 ? -1. : 1.);
            t = (R[8] + 1) * 0.5; rz = sqrt(t > 0. ? t : 0.) * (R[2] < 0 ? -1. : 1.);
            if (fabs(rx) < fabs(ry) && fabs(rx) < fabs(rz) && (R[5] > 0) != (ry * rz > 0)) rz = -rz;
            theta /= sqrt(rx * rx + ry * ry + rz * rz);
            rx *= theta; ry *= theta; rz *= theta;
        }
    } else {
        double vth = 1 / (2 * s);
        vth *= theta;
        rx *= vth; ry *= vth; rz *= vth;
    }
    r[0] = rx; r[1] = ry; r[2] = rz;
    if (jac) memset(jac, 0, 27 * sizeof(double));
    return 1;
}

/* ------------------------------------------------------------------------- */
static void load_dist(const double* dist, int ndist, double k[14])
{
    memset(k, 0, 14 * sizeof(double));
    if (dist) for (int i = 0; i < ndist && i < 14; i++) k[i] = dist[i];
}

/* distortion_model.hpp detail::computeTiltProjectionMatrix<double> [OpenCV-knowledge]: the sensor plane tilted by tau_x about x and
 * tau_y about y -- rotate (R_y R_x), then project along z back onto z = 1: matTilt = matProjZ * matRotXY, its inverse
 * matRotXY^T * invMatProjZ.  Matx products accumulate s = 0; s += a(i,k) * b(k,j) over k (mat3_mul_cv). */
static void mat3_mul_cv(const double A[9], const double B[9], double C[9])
{
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
            double s = 0;
            for (int k = 0; k < 3; k++) s += A[i * 3 + k] * B[k * 3 + j];
            C[i * 3 + j] = s;
        }
}

void cvo_tilt_matrices(double tauX, double tauY, double matTilt[9], double invMatTilt[9])
{
    const double cTauX = cos(tauX), sTauX = sin(tauX), cTauY = cos(tauY), sTauY = sin(tauY);
    const double matRotX[9] = { 1, 0, 0, 0, cTauX, sTauX, 0, -sTauX, cTauX };
    const double matRotY[9] = { cTauY, 0, -sTauY, 0, 1, 0, sTauY, 0, cTauY };
    double matRotXY[9];
    mat3_mul_cv(matRotY, matRotX, matRotXY);
    if (matTilt) {
        const double matProjZ[9] = { matRotXY[8], 0, -matRotXY[2], 0, matRotXY[8], -matRotXY[5], 0, 0, 1 };
        mat3_mul_cv(matProjZ, matRotXY, matTilt);
    }
    if (invMatTilt) {
        const double inv = 1. / matRotXY[8];
        const double invMatProjZ[9] = { inv, 0, inv * matRotXY[2], 0, inv, inv * matRotXY[5], 0, 0, 1 };
        const double t[9] = { matRotXY[0], matRotXY[3], matRotXY[6], matRotXY[1], matRotXY[4], matRotXY[7], matRotXY[2], matRotXY[5], matRotXY[8] };
        mat3_mul_cv(t, invMatProjZ, invMatTilt);
    }
}

/* matTilt / invMatTilt of a coefficient vector: the identity unless k[12] or k[13] is non-zero (as every OpenCV user of the model) */
void cvo_tilt_of(const double k[14], double matTilt[9], double invMatTilt[9])
{
    static const double I[9] = { 1, 0, 0, 0, 1, 0, 0, 0, 1 };
    if (matTilt) memcpy(matTilt, I, sizeof(I));
    if (invMatTilt) memcpy(invMatTilt, I, sizeof(I));
    if (k[12] != 0 || k[13] != 0) cvo_tilt_matrices(k[12], k[13], matTilt, invMatTilt);
}

/* calibration.cpp cvProjectPoints2Internal, 14-coefficient model: radial (rational), tangential, thin prism, tilted sensor */
int cvo_project_points(const double* obj, int n, const double rvec[3], const double tvec[3],
                       const double Kc[9], const double* dist, int ndist,
                       double* m, double* dpdr, double* dpdt)
{
    if (!obj || !m || n < 0) return -1;
    if (ndist != 0 && ndist != 4 && ndist != 5 && ndist != 8 && ndist != 12 && ndist != 14) return -3;
    double R[9], dRdr[27], k[14];
    cvo_rodrigues_vec2mat(rvec, R, dRdr);
    load_dist(dist, ndist, k);
    double T[9];
    cvo_tilt_of(k, T, NULL);
    const double fx = Kc[0], fy = Kc[4], cx = Kc[2], cy = Kc[5];
    const double* t = tvec;
    for (int i = 0; i < n; i++) {
        double X = obj[i * 3], Y = obj[i * 3 + 1], Z = obj[i * 3 + 2];
        double x = R[0] * X + R[1] * Y + R[2] * Z + t[0];
        double y = R[3] * X + R[4] * Y + R[5] * Z + t[1];
        double z = R[6] * X + R[7] * Y + R[8] * Z + t[2];
        z = z ? 1. / z : 1;
        x *= z; y *= z;
        double r2 = x * x + y * y, r4 = r2 * r2, r6 = r4 * r2;
        double a1 = 2 * x * y, a2 = r2 + 2 * x * x, a3 = r2 + 2 * y * y;
        double cdist = 1 + k[0] * r2 + k[1] * r4 + k[4] * r6;
        double icdist2 = 1. / (1 + k[5] * r2 + k[6] * r4 + k[7] * r6);
        double xd0 = x * cdist * icdist2 + k[2] * a1 + k[3] * a2 + k[8] * r2 + k[9] * r4;
        double yd0 = y * cdist * icdist2 + k[2] * a3 + k[3] * a1 + k[10] * r2 + k[11] * r4;
        /* additional distortion by projecting onto a tilt plane: vecTilt = matTilt * (xd0, yd0, 1) */
        double vt[3];
        for (int r = 0; r < 3; r++) { double a = 0; a += T[r * 3] * xd0; a += T[r * 3 + 1] * yd0; a += T[r * 3 + 2] * 1; vt[r] = a; }
        double invProj = vt[2] ? 1. / vt[2] : 1;
        double xd = invProj * vt[0], yd = invProj * vt[1];
        m[i * 2] = xd * fx + cx;
        m[i * 2 + 1] = yd * fy + cy;
        /* d(xd, yd) / d(xd0, yd0) */
        double dT[4];
        for (int r = 0; r < 2; r++)
            for (int c = 0; c < 2; c++) dT[r * 2 + c] = T[r * 3 + c] * vt[2] - T[6 + c] * vt[r];
        const double invProjSquare = invProj * invProj;
        for (int q = 0; q < 4; q++) dT[q] *= invProjSquare;
        if (dpdt) {
            double* p = dpdt + (size_t)i * 6;
            double dxdt[3] = { z, 0, -x * z }, dydt[3] = { 0, z, -y * z };
            for (int j = 0; j < 3; j++) {
                double dr2dt = 2 * x * dxdt[j] + 2 * y * dydt[j];
                double dcdist_dt = k[0] * dr2dt + 2 * k[1] * r2 * dr2dt + 3 * k[4] * r4 * dr2dt;
                double dicdist2_dt = -icdist2 * icdist2 * (k[5] * dr2dt + 2 * k[6] * r2 * dr2dt + 3 * k[7] * r4 * dr2dt);
                double da1dt = 2 * (x * dydt[j] + y * dxdt[j]);
                double dmxdt = (dxdt[j] * cdist * icdist2 + x * dcdist_dt * icdist2 + x * cdist * dicdist2_dt +
                                k[2] * da1dt + k[3] * (dr2dt + 4 * x * dxdt[j]) + k[8] * dr2dt + 2 * r2 * k[9] * dr2dt);
                double dmydt = (dydt[j] * cdist * icdist2 + y * dcdist_dt * icdist2 + y * cdist * dicdist2_dt +
                                k[2] * (dr2dt + 4 * y * dydt[j]) + k[3] * da1dt + k[10] * dr2dt + 2 * r2 * k[11] * dr2dt);
                double dX = 0, dY = 0;
                dX += dT[0] * dmxdt; dX += dT[1] * dmydt;
                dY += dT[2] * dmxdt; dY += dT[3] * dmydt;
                p[j] = fx * dX;
                p[3 + j] = fy * dY;
            }
        }
        if (dpdr) {
            double* p = dpdr + (size_t)i * 6;
            double dx0dr[3] = { X * dRdr[0] + Y * dRdr[1] + Z * dRdr[2],
                                X * dRdr[9] + Y * dRdr[10] + Z * dRdr[11],
                                X * dRdr[18] + Y * dRdr[19] + Z * dRdr[20] };
            double dy0dr[3] = { X * dRdr[3] + Y * dRdr[4] + Z * dRdr[5],
                                X * dRdr[12] + Y * dRdr[13] + Z * dRdr[14],
                                X * dRdr[21] + Y * dRdr[22] + Z * dRdr[23] };
            double dz0dr[3] = { X * dRdr[6] + Y * dRdr[7] + Z * dRdr[8],
                                X * dRdr[15] + Y * dRdr[16] + Z * dRdr[17],
                                X * dRdr[24] + Y * dRdr[25] + Z * dRdr[26] };
            for (int j = 0; j < 3; j++) {
                double dxdr = z * (dx0dr[j] - x * dz0dr[j]);
                double dydr = z * (dy0dr[j] - y * dz0dr[j]);
                double dr2dr = 2 * x * dxdr + 2 * y * dydr;
                double dcdist_dr = (k[0] + 2 * k[1] * r2 + 3 * k[4] * r4) * dr2dr;
                double dicdist2_dr = -icdist2 * icdist2 * (k[5] + 2 * k[6] * r2 + 3 * k[7] * r4) * dr2dr;
                double da1dr = 2 * (x * dydr + y * dxdr);
                double dmxdr = (dxdr * cdist * icdist2 + x * dcdist_dr * icdist2 + x * cdist * dicdist2_dr +
                                k[2] * da1dr + k[3] * (dr2dr + 4 * x * dxdr) + (k[8] + 2 * r2 * k[9]) * dr2dr);
                double dmydr = (dydr * cdist * icdist2 + y * dcdist_dr * icdist2 + y * cdist * dicdist2_dr +
                                k[2] * (dr2dr + 4 * y * dydr) + k[3] * da1dr + (k[10] + 2 * r2 * k[11]) * dr2dr);
                double dX = 0, dY = 0;
                dX += dT[0] * dmxdr; dX += dT[1] * dmydr;
                dY += dT[2] * dmxdr; dY += dT[3] * dmydr;
                p[j] = fx * dX;
                p[3 + j] = fy * dY;
            }
        }
    }
    return 0;
}

/* undistort.dispatch.cpp cvUndistortPointsInternal with criteria (COUNT, 5), R = I, no P */
int cvo_undistort_points(const double* img, int n, const double Kc[9],
                         const double* dist, int ndist, double* out)
{
    if (!img || !out || n < 0) return -1;
    double k[14];
    load_dist(dist, ndist, k);
    double Ti[9];
    cvo_tilt_of(k, NULL, Ti);
    const double fx = Kc[0], fy = Kc[4], ifx = 1. / fx, ify = 1. / fy, cx = Kc[2], cy = Kc[5];
    for (int i = 0; i < n; i++) {
        double u = img[i * 2], v = img[i * 2 + 1];
        double x = (u - cx) * ifx, y = (v - cy) * ify;
        if (dist && ndist > 0) {
            /* compensate tilt distortion: vecUntilt = invMatTilt * (x, y, 1) */
            double vu[3];
            for (int r = 0; r < 3; r++) { double a = 0; a += Ti[r * 3] * x; a += Ti[r * 3 + 1] * y; a += Ti[r * 3 + 2] * 1; vu[r] = a; }
            const double invProj = vu[2] ? 1. / vu[2] : 1;
            x = invProj * vu[0]; y = invProj * vu[1];
            double x0 = x, y0 = y;
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
        out[i * 2] = x; out[i * 2 + 1] = y;
    }
    return 0;
}

/* ------------------------------------------------------------------------- */
/* cv::findHomography(method = 0) initial estimate (fundam.cpp HomographyEstimatorCallback::runKernel);
 * for more than four points OpenCV refines it with LMSolver (homography_refine_lm below). */
static int homography_dlt(const double* M, const double* m, int count, double H[9])
{
    double cMx = 0, cMy = 0, cmx = 0, cmy = 0, sMx = 0, sMy = 0, smx = 0, smy = 0;
    for (int i = 0; i < count; i++) { cmx += m[i * 2]; cmy += m[i * 2 + 1]; cMx += M[i * 2]; cMy += M[i * 2 + 1]; }
    cmx /= count; cmy /= count; cMx /= count; cMy /= count;
    for (int i = 0; i < count; i++) {
        smx += fabs(m[i * 2] - cmx); smy += fabs(m[i * 2 + 1] - cmy);
        sMx += fabs(M[i * 2] - cMx); sMy += fabs(M[i * 2 + 1] - cMy);
    }
    if (fabs(smx) < DBL_EPSILON || fabs(smy) < DBL_EPSILON || fabs(sMx) < DBL_EPSILON || fabs(sMy) < DBL_EPSILON) return -1;
    smx = count / smx; smy = count / smy; sMx = count / sMx; sMy = count / sMy;
    double invHnorm[9] = { 1. / smx, 0, cmx, 0, 1. / smy, cmy, 0, 0, 1 };
    double Hnorm2[9] = { sMx, 0, -cMx * sMx, 0, sMy, -cMy * sMy, 0, 0, 1 };
    double LtL[81];
    memset(LtL, 0, sizeof(LtL));
    for (int i = 0; i < count; i++) {
        double x = (m[i * 2] - cmx) * smx, y = (m[i * 2 + 1] - cmy) * smy;
        double X = (M[i * 2] - cMx) * sMx, Y = (M[i * 2 + 1] - cMy) * sMy;
        double Lx[9] = { X, Y, 1, 0, 0, 0, -x * X, -x * Y, -x };
        double Ly[9] = { 0, 0, 0, X, Y, 1, -y * X, -y * Y, -y };
        for (int j = 0; j < 9; j++) for (int kk = j; kk < 9; kk++) LtL[j * 9 + kk] += Lx[j] * Lx[kk] + Ly[j] * Ly[kk];
    }
    for (int j = 0; j < 9; j++) for (int kk = 0; kk < j; kk++) LtL[j * 9 + kk] = LtL[kk * 9 + j];
    double w[9], vt[81];
    if (cvo_svd(LtL, 9, 9, w, NULL, vt)) return -1;
    double T[9];
    mat3_mul(invHnorm, vt + 8 * 9, T);
    mat3_mul(T, Hnorm2, H);
    if (H[8] == 0 || !isfinite(H[8])) return -1;
    double s = 1. / H[8];
    for (int i = 0; i < 9; i++) H[i] *= s;
    return 0;
}

/* x = pinv(A) b for a symmetric PSD n x n A, as cv::solve(A, b, x, DECOMP_EIG): eigen-decomposition, components
 * whose eigenvalue is <= 2 * DBL_EPSILON * sum(eigenvalues) dropped (matrix.cpp SVBkSb).  n <= 8. */
static int sym_solve_eig(const double* A, int n, const double* b, double* x, double* inv_diag)
{
    double w[8], vt[64];
    if (n > 8 || cvo_svd(A, n, n, w, NULL, vt)) return -1;     /* SVD of a symmetric PSD matrix = its eigen-decomposition */
    double thr = 0;
    for (int i = 0; i < n; i++) thr += w[i];
    thr *= DBL_EPSILON * 2;
    if (x) for (int i = 0; i < n; i++) x[i] = 0;
    if (inv_diag) for (int i = 0; i < n; i++) inv_diag[i] = 0;
    for (int k = 0; k < n; k++) {
        if (fabs(w[k]) <= thr) continue;
        const double* v = vt + k * n;
        double iw = 1. / w[k];
        if (x) {
            double s = 0;
            for (int i = 0; i < n; i++) s += v[i] * b[i];
            s *= iw;
            for (int i = 0; i < n; i++) x[i] += s * v[i];
        }
        if (inv_diag) for (int i = 0; i < n; i++) inv_diag[i] += v[i] * v[i] * iw;
    }
    return 0;
}

/* fundam.cpp HomographyRefineCallback::compute: reprojection residuals of H = [h0..h7, 1] and their 2 x 8 Jacobian rows,
 * accumulated into A = J^T J (8 x 8), v = J^T r, S = |r|^2; returns max |r| in *rinf */
static double hom_eval(const double* M, const double* m, int count, const double h[8], double* A, double* v, double* rinf)
{
    double S = 0, ri = 0;
    if (A) { memset(A, 0, 64 * sizeof(double)); memset(v, 0, 8 * sizeof(double)); }
    for (int i = 0; i < count; i++) {
        double Mx = M[i * 2], My = M[i * 2 + 1];
        double ww = h[6] * Mx + h[7] * My + 1.;
        ww = fabs(ww) > DBL_EPSILON ? 1. / ww : 0;
        double xi = (h[0] * Mx + h[1] * My + h[2]) * ww;
        double yi = (h[3] * Mx + h[4] * My + h[5]) * ww;
        double ex = xi - m[i * 2], ey = yi - m[i * 2 + 1];
        S += ex * ex + ey * ey;
        if (fabs(ex) > ri) ri = fabs(ex);
        if (fabs(ey) > ri) ri = fabs(ey);
        if (A) {
            double Jx[8] = { Mx * ww, My * ww, ww, 0, 0, 0, -Mx * ww * xi, -My * ww * xi };
            double Jy[8] = { 0, 0, 0, Mx * ww, My * ww, ww, -Mx * ww * yi, -My * ww * yi };
            for (int a = 0; a < 8; a++) {
                for (int b = 0; b < 8; b++) A[a * 8 + b] += Jx[a] * Jx[b] + Jy[a] * Jy[b];
                v[a] += Jx[a] * ex + Jy[a] * ey;
            }
        }
    }
    if (rinf) *rinf = ri;
    return S;
}

/* calib3d levmarq.cpp LMSolverImpl::run(param) with HomographyRefineCallback, maxIters = 10, epsx = epsf = FLT_EPSILON:
 * what cv::findHomography(method 0) does to the DLT estimate when there are more than four points. */
static void homography_refine_lm(const double* M, const double* m, int count, double H[9])
{
    double x[8], xd[8], A[64], Ap[64], v[8], d[8], D[8], tmp[8];
    for (int i = 0; i < 8; i++) x[i] = H[i];            /* H[8] == 1 after the DLT's normalisation */
    double rinf;
    double S = hom_eval(M, m, count, x, A, v, &rinf);
    for (int i = 0; i < 8; i++) D[i] = A[i * 9];
    const double Rlo = 0.25, Rhi = 0.75;
    double lambda = 1, lc = 0.75;
    const int max_iters = 10;
    const double epsx = FLT_EPSILON, epsf = FLT_EPSILON;
    for (int iter = 0;;) {
        memcpy(Ap, A, sizeof(Ap));
        for (int i = 0; i < 8; i++) Ap[i * 9] += lambda * D[i];
        if (sym_solve_eig(Ap, 8, v, d, NULL)) break;
        for (int i = 0; i < 8; i++) xd[i] = x[i] - d[i];
        double Sd = hom_eval(M, m, count, xd, NULL, NULL, NULL);
        double dS = 0;
        for (int i = 0; i < 8; i++) {
            double t = 2 * v[i];
            for (int j = 0; j < 8; j++) t -= A[i * 8 + j] * d[j];
            tmp[i] = t;                                /* temp_d = -A d + 2 v */
        }
        for (int i = 0; i < 8; i++) dS += d[i] * tmp[i];
        double R = (S - Sd) / (fabs(dS) > DBL_EPSILON ? dS : 1);
        if (R > Rhi) {
            lambda *= 0.5;
            if (lambda < lc) lambda = 0;
        } else if (R < Rlo) {
            double t = 0;
            for (int i = 0; i < 8; i++) t += d[i] * v[i];
            double nu = (Sd - S) / (fabs(t) > DBL_EPSILON ? t : 1) + 2;
            nu = fmin(fmax(nu, 2.), 10.);
            if (lambda == 0) {
                double idg[8] = { 0, 0, 0, 0, 0, 0, 0, 0 }, maxval = DBL_EPSILON;
                sym_solve_eig(A, 8, NULL, NULL, idg);          /* invert(A, Ap, DECOMP_EIG): only its diagonal is used */
                for (int i = 0; i < 8; i++) maxval = fmax(maxval, fabs(idg[i]));
                lambda = lc = 1. / maxval;
                nu *= 0.5;
            }
            lambda *= nu;
        }
        if (Sd < S) {
            S = Sd;
            for (int i = 0; i < 8; i++) x[i] = xd[i];
            S = hom_eval(M, m, count, x, A, v, &rinf);      /* same value as Sd; refreshes r, J */
        }
        iter++;
        double dinf = 0;
        for (int i = 0; i < 8; i++) dinf = fmax(dinf, fabs(d[i]));
        if (!(iter < max_iters && dinf >= epsx && rinf >= epsf)) break;
    }
    for (int i = 0; i < 8; i++) H[i] = x[i];
}

int cvo_find_homography(const double* M, const double* m, int count, int refine, double H[9])
{
    /* cv::findHomography converts both point sets to CV_32F first (fundam.cpp: p.reshape(2, npoints).convertTo(m, CV_32F)) */
    if (!M || !m || !H || count < 4) return -1;
    double* b = (double*)malloc((size_t)count * 4 * sizeof(double));
    if (!b) return -2;
    for (int i = 0; i < count * 2; i++) { b[i] = (double)(float)M[i]; b[count * 2 + i] = (double)(float)m[i]; }
    int rc = homography_dlt(b, b + count * 2, count, H);
    if (rc == 0 && refine && count > 4) homography_refine_lm(b, b + count * 2, count, H);
    free(b);
    return rc;
}

/* cvFindExtrinsicCameraParams2, the !useExtrinsicGuess branch */
int cvo_pnp_init(const double* obj, const double* img, int n,
                 const double Kc[9], const double* dist, int ndist,
                 double rvec[3], double tvec[3])
{
    if (!obj || !img || n < 4) return -1;
    double* mn = (double*)malloc((size_t)n * 2 * sizeof(double));
    if (!mn) return -2;
    cvo_undistort_points(img, n, Kc, dist, ndist, mn);
    double Mc[3] = { 0, 0, 0 }, MM[9], W[3], V[9], R[9];
    for (int i = 0; i < n; i++) { Mc[0] += obj[i * 3]; Mc[1] += obj[i * 3 + 1]; Mc[2] += obj[i * 3 + 2]; }
    Mc[0] /= n; Mc[1] /= n; Mc[2] /= n;
    memset(MM, 0, sizeof(MM));
    for (int i = 0; i < n; i++) {
        double d[3] = { obj[i * 3] - Mc[0], obj[i * 3 + 1] - Mc[1], obj[i * 3 + 2] - Mc[2] };
        for (int a = 0; a < 3; a++) for (int b = 0; b < 3; b++) MM[a * 3 + b] += d[a] * d[b];
    }
    cvo_svd(MM, 3, 3, W, NULL, V);    /* V holds V^T (CV_SVD_V_T) */
    int rc = 0;
    if (W[2] / W[1] < 1e-3) {
        /* planar structure */
        double tt[3], h[9];
        double* Rt = V;
        if (V[2] * V[2] + V[5] * V[5] < 1e-10) for (int i = 0; i < 9; i++) Rt[i] = (i % 4 == 0);
        if (det3(Rt) < 0) for (int i = 0; i < 9; i++) Rt[i] = -Rt[i];
        for (int a = 0; a < 3; a++) tt[a] = -(Rt[a * 3] * Mc[0] + Rt[a * 3 + 1] * Mc[1] + Rt[a * 3 + 2] * Mc[2]);
        double* Mxy = (double*)malloc((size_t)n * 2 * sizeof(double));
        if (!Mxy) { free(mn); return -2; }
        for (int i = 0; i < n; i++) {
            const double* s = obj + i * 3;
            Mxy[i * 2] = Rt[0] * s[0] + Rt[1] * s[1] + Rt[2] * s[2] + tt[0];
            Mxy[i * 2 + 1] = Rt[3] * s[0] + Rt[4] * s[1] + Rt[5] * s[2] + tt[1];
        }
        int ok = cvo_find_homography(Mxy, mn, n, 1, h) == 0;
        free(Mxy);
        for (int i = 0; ok && i < 9; i++) if (!isfinite(h[i])) ok = 0;
        if (ok) {
            double h1n = sqrt(h[0] * h[0] + h[3] * h[3] + h[6] * h[6]);
            double h2n = sqrt(h[1] * h[1] + h[4] * h[4] + h[7] * h[7]);
            double s1 = 1. / fmax(h1n, DBL_EPSILON), s2 = 1. / fmax(h2n, DBL_EPSILON);
            double s3 = 2. / fmax(h1n + h2n, DBL_EPSILON);
            double t[3] = { h[2] * s3, h[5] * s3, h[8] * s3 };
            h[0] *= s1; h[3] *= s1; h[6] *= s1;
            h[1] *= s2; h[4] *= s2; h[7] *= s2;
            h[2] = h[3] * h[7] - h[6] * h[4];
            h[5] = h[6] * h[1] - h[0] * h[7];
            h[8] = h[0] * h[4] - h[3] * h[1];
            double rr[3];
            cvo_rodrigues_mat2vec(h, rr, NULL);
            cvo_rodrigues_vec2mat(rr, h, NULL);
            for (int a = 0; a < 3; a++) tvec[a] = h[a * 3] * tt[0] + h[a * 3 + 1] * tt[1] + h[a * 3 + 2] * tt[2] + t[a];
            mat3_mul(h, Rt, R);
        } else {
            for (int i = 0; i < 9; i++) R[i] = (i % 4 == 0);
            tvec[0] = tvec[1] = tvec[2] = 0;
        }
        cvo_rodrigues_mat2vec(R, rvec, NULL);
    } else {
        /* non-planar: DLT */
        if (n < 6) { free(mn); return -4; }
        double LL[144], LW[12], LV[144];
        memset(LL, 0, sizeof(LL));
        for (int i = 0; i < n; i++) {
            double x = -mn[i * 2], y = -mn[i * 2 + 1];
            double X = obj[i * 3], Y = obj[i * 3 + 1], Z = obj[i * 3 + 2];
            double L0[12] = { X, Y, Z, 1, 0, 0, 0, 0, x * X, x * Y, x * Z, x };
            double L1[12] = { 0, 0, 0, 0, X, Y, Z, 1, y * X, y * Y, y * Z, y };
            for (int a = 0; a < 12; a++) for (int b = a; b < 12; b++) LL[a * 12 + b] += L0[a] * L0[b] + L1[a] * L1[b];
        }
        for (int a = 0; a < 12; a++) for (int b = 0; b < a; b++) LL[a * 12 + b] = LL[b * 12 + a];
        cvo_svd(LL, 12, 12, LW, NULL, LV);
        double RRt[12];
        memcpy(RRt, LV + 11 * 12, sizeof(RRt));
        double RR[9] = { RRt[0], RRt[1], RRt[2], RRt[4], RRt[5], RRt[6], RRt[8], RRt[9], RRt[10] };
        if (det3(RR) < 0) { for (int i = 0; i < 12; i++) RRt[i] = -RRt[i]; for (int i = 0; i < 9; i++) RR[i] = -RR[i]; }
        double sc = 0;
        for (int i = 0; i < 9; i++) sc += RR[i] * RR[i];
        sc = sqrt(sc);
        if (!(fabs(sc) > DBL_EPSILON)) { free(mn); return -5; }
        double U[9], Vt[9], Wr[3];
        cvo_svd(RR, 3, 3, Wr, U, Vt);
        mat3_mul(U, Vt, R);
        double nr = 0;
        for (int i = 0; i < 9; i++) nr += R[i] * R[i];
        nr = sqrt(nr);
        tvec[0] = RRt[3] * nr / sc; tvec[1] = RRt[7] * nr / sc; tvec[2] = RRt[11] * nr / sc;
        cvo_rodrigues_mat2vec(R, rvec, NULL);
    }
    free(mn);
    return rc;
}

/* ------------------------------------------------------------------------- */
/* CvLevMarq state machine (compat_ptsetreg.cpp), update() variant that owns J and err */
enum { LM_DONE = 0, LM_STARTED = 1, LM_CALC_J = 2, LM_CHECK_ERR = 3 };

static double vec_norm(const double* v, int n)
{
    double s = 0;
    for (int i = 0; i < n; i++) s += v[i] * v[i];
    return sqrt(s);
}

int cvo_solve_pnp_iterative(const double* obj, const double* img, int n,
                            const double Kc[9], const double* dist, int ndist,
                            double rvec[3], double tvec[3], int use_guess,
                            int* iters_out)
{
    if (!obj || !img || !rvec || !tvec) return -1;
    if (!(n >= 4 || (n == 3 && use_guess))) return -1;
    if (ndist != 0 && ndist != 4 && ndist != 5 && ndist != 8 && ndist != 12 && ndist != 14) return -3;
    double param[6], prevParam[6] = { 0, 0, 0, 0, 0, 0 };
    if (use_guess) {
        for (int i = 0; i < 3; i++) { param[i] = rvec[i]; param[3 + i] = tvec[i]; }
    } else {
        int rc = cvo_pnp_init(obj, img, n, Kc, dist, ndist, param, param + 3);
        if (rc) return rc;
    }
    const int max_iter = 20;
    const double epsilon = FLT_EPSILON;
    double* J = (double*)malloc((size_t)2 * n * 6 * sizeof(double));
    double* err = (double*)malloc((size_t)2 * n * sizeof(double));
    double* dpdr = (double*)malloc((size_t)2 * n * 3 * sizeof(double));
    double* dpdt = (double*)malloc((size_t)2 * n * 3 * sizeof(double));
    if (!J || !err || !dpdr || !dpdt) { free(J); free(err); free(dpdr); free(dpdt); return -2; }
    double JtJ[36], JtErr[6], A[36], dx[6];
    double prevErrNorm = DBL_MAX, errNorm = 0;
    int lambdaLg10 = -3, iters = 0, state = LM_STARTED;
    const double LOG10 = log(10.);

    for (;;) {
        int needJ = 0, needErr = 0;
        /* ---- CvLevMarq::update ---- */
        if (state == LM_DONE) break;
        if (state == LM_STARTED) {
            needJ = needErr = 1; state = LM_CALC_J;
        } else if (state == LM_CALC_J) {
            memset(JtJ, 0, sizeof(JtJ)); memset(JtErr, 0, sizeof(JtErr));
            for (int i = 0; i < 2 * n; i++) {
                const double* Ji = J + (size_t)i * 6;
                for (int a = 0; a < 6; a++) {
                    JtErr[a] += Ji[a] * err[i];
                    for (int b = a; b < 6; b++) JtJ[a * 6 + b] += Ji[a] * Ji[b];
                }
            }
            for (int a = 0; a < 6; a++) for (int b = 0; b < a; b++) JtJ[a * 6 + b] = JtJ[b * 6 + a];
            memcpy(prevParam, param, sizeof(param));
            goto do_step;
        } else {   /* LM_CHECK_ERR */
            errNorm = vec_norm(err, 2 * n);
            if (errNorm > prevErrNorm) {
                if (++lambdaLg10 <= 16) goto do_step_keep;
            }
            lambdaLg10 = lambdaLg10 - 1 > -16 ? lambdaLg10 - 1 : -16;
            {
                double d[6];
                for (int i = 0; i < 6; i++) d[i] = param[i] - prevParam[i];
                double rel = vec_norm(d, 6) / (vec_norm(prevParam, 6) + DBL_EPSILON);
                if (++iters >= max_iter || rel < epsilon) { state = LM_DONE; break; }
            }
            prevErrNorm = errNorm;
            needJ = needErr = 1; state = LM_CALC_J;
        }
        goto evaluate;
do_step:
        /* first pass through CALC_J records prevErrNorm after step() */
        {
            double lambda = exp(lambdaLg10 * LOG10);
            memcpy(A, JtJ, sizeof(A));
            for (int a = 0; a < 6; a++) A[a * 7] *= 1. + lambda;
            cvo_solve_svd(A, JtErr, 6, dx);
            for (int a = 0; a < 6; a++) param[a] = prevParam[a] - dx[a];
            if (iters == 0) prevErrNorm = vec_norm(err, 2 * n);
            needErr = 1; state = LM_CHECK_ERR;
        }
        goto evaluate;
do_step_keep:
        {
            double lambda = exp(lambdaLg10 * LOG10);
            memcpy(A, JtJ, sizeof(A));
            for (int a = 0; a < 6; a++) A[a * 7] *= 1. + lambda;
            cvo_solve_svd(A, JtErr, 6, dx);
            for (int a = 0; a < 6; a++) param[a] = prevParam[a] - dx[a];
            needErr = 1; state = LM_CHECK_ERR;
        }
evaluate:
        /* ---- caller side of the loop in cvFindExtrinsicCameraParams2 ---- */
        if (!needErr) break;
        cvo_project_points(obj, n, param, param + 3, Kc, dist, ndist, err, needJ ? dpdr : NULL, needJ ? dpdt : NULL);
        for (int i = 0; i < 2 * n; i++) err[i] -= img[i];
        if (needJ)
            for (int i = 0; i < n; i++)
                for (int r = 0; r < 2; r++)
                    for (int a = 0; a < 3; a++) {
                        J[((size_t)2 * i + r) * 6 + a] = dpdr[(size_t)i * 6 + r * 3 + a];
                        J[((size_t)2 * i + r) * 6 + 3 + a] = dpdt[(size_t)i * 6 + r * 3 + a];
                    }
    }
    for (int i = 0; i < 3; i++) { rvec[i] = param[i]; tvec[i] = param[3 + i]; }
    if (iters_out) *iters_out = iters;
    free(J); free(err); free(dpdr); free(dpdt);
    return 0;
}

/* transform_helper.py:98-121 get_reprojection_error: mean of per-point L2 norms */
double cvo_mean_reproj_error(const double* obj, const double* img, int n,
                             const double rvec[3], const double tvec[3],
                             const double Kc[9], const double* dist, int ndist)
{
    double* m = (double*)malloc((size_t)n * 2 * sizeof(double));
    if (!m) return -1.;
    cvo_project_points(obj, n, rvec, tvec, Kc, dist, ndist, m, NULL, NULL);
    double s = 0;
    for (int i = 0; i < n; i++) {
        double dx = img[i * 2] - m[i * 2], dy = img[i * 2 + 1] - m[i * 2 + 1];
        s += sqrt(dx * dx + dy * dy);
    }
    free(m);
    return s / n;
}
