// agt_device.h -- device-side helpers shared by the gfx950 kernels.
// Wave = 64 lanes everywhere (CDNA4); no 32-wide assumptions.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <float.h>

#define AGT_WAVE 64

// cv::borderInterpolate(p, len, BORDER_REFLECT_101); valid for any p
__device__ __forceinline__ int agt_reflect101(int p, int len)
{
    if ((unsigned)p < (unsigned)len) return p;
    if (len == 1) return 0;
    do {
        p = p < 0 ? -p : 2 * (len - 1) - p;
    } while ((unsigned)p >= (unsigned)len);
    return p;
}

// ---------------------------------------------------------------------------
// DPP helpers: sum within a 16-lane row without touching LDS, then combine the
// four rows through SGPRs (v_readlane) so the total is wave-uniform.
template <int CTRL>
__device__ __forceinline__ int agt_dpp_i32(int v)
{
    return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, true);
}

__device__ __forceinline__ long long agt_dpp_add_i64_step(long long v, int lo, int hi)
{
    long long o = ((long long)hi << 32) | (unsigned int)lo;
    return v + o;
}

#define AGT_DPP_STEP_I64(v, CTRL)                                            \
    do {                                                                      \
        int lo_ = agt_dpp_i32<CTRL>((int)(v));                                \
        int hi_ = agt_dpp_i32<CTRL>((int)((v) >> 32));                        \
        (v) = agt_dpp_add_i64_step((v), lo_, hi_);                            \
    } while (0)

// exact 64-lane integer sum; result identical (and uniform) in every lane
__device__ __forceinline__ long long agt_wave_sum_i64(long long v)
{
    AGT_DPP_STEP_I64(v, 0xB1);   // quad_perm [1,0,3,2]
    AGT_DPP_STEP_I64(v, 0x4E);   // quad_perm [2,3,0,1]
    AGT_DPP_STEP_I64(v, 0x141);  // row_half_mirror
    AGT_DPP_STEP_I64(v, 0x140);  // row_mirror  -> every lane holds its row's total
    int lo = (int)v, hi = (int)(v >> 32);
    long long t = 0;
#pragma unroll
    for (int r = 0; r < 4; r++) {
        unsigned int l = (unsigned int)__builtin_amdgcn_readlane(lo, r * 16);
        int h = __builtin_amdgcn_readlane(hi, r * 16);
        t += ((long long)h << 32) | l;
    }
    return t;
}

// wave-uniform broadcast of lane 0's value (keeps loop control scalar)
__device__ __forceinline__ int agt_uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ float agt_uniform(float v)
{
    return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v)));
}

// ---------------------------------------------------------------------------
// FP64 geometry (OpenCV calibration.cpp semantics; see oracle/cv_pnp.c for the restatement)
//
// The library is compiled with -ffp-contract=off because the LK float expressions must round exactly
// as OpenCV's do.  The FP64 pose code has no such requirement (parity bar 1e-9 on the pose, far above
// FP64 round-off), and it is a serial dependency chain: contracting a*b+c into one v_fma_f64 halves the
// length of most of its links.  Contraction is therefore switched on for this section only.
#pragma clang fp contract(fast)

// 1 / x and sqrt(x) for the FP64 pose code where the last bit does not matter (values, not decisions): the hardware estimate
// (v_rcp_f64 / v_rsq_f64, ~26 bits) + two Newton steps -- 7 / 8 dependent instructions instead of the 11 / 14 of the
// correctly rounded sequences (v_div_scale / v_div_fmas / v_div_fixup; scaled sqrt).  Error <= 1 ulp; no range scaling: callers
// pass finite, normal, positive (agt_sqrtp: >= 0) operands.  The PnP parity bound is 1e-9 on the pose.
__device__ __forceinline__ double agt_rcp(double x)
{
    double r = __builtin_amdgcn_rcp(x);
    double e = __builtin_fma(-x, r, 1.0);
    r = __builtin_fma(r, e, r);
    e = __builtin_fma(-x, r, 1.0);
    return __builtin_fma(r, e, r);
}
// 1 / x with ONE Newton step on the hardware estimate (relative error ~2^-50): the pivots of the damped normal equations,
// where six of these sit in series on the chain of every LM step
__device__ __forceinline__ double agt_rcp1(double x)
{
    const double r = __builtin_amdgcn_rcp(x);
    return __builtin_fma(r, __builtin_fma(-x, r, 1.0), r);
}
__device__ __forceinline__ double agt_sqrtp(double x)
{
    if (x == 0.0) return 0.0;
    const double y = __builtin_amdgcn_rsq(x);
    double g = x * y, h = 0.5 * y;
    double r = __builtin_fma(-h, g, 0.5);
    g = __builtin_fma(g, r, g); h = __builtin_fma(h, r, h);
    const double d = __builtin_fma(-g, g, x);
    return __builtin_fma(d, h, g);
}


// cvRodrigues2 vector->matrix.  When JAC, G (3x3 row-major) is the LEFT JACOBIAN of SO(3) at r,
//     G = I + (1 - cos t)/t^2 [r]x + (t - sin t)/t^3 [r]x^2,      R(r + d) = exp([G d]x) R(r) + O(d^2),
// so that d(R X)/dr_j = G_j x (R X) (G_j = column j).  This is the same derivative OpenCV forms through its
// 3x9 dR/dr table (27 entries, ~170 FP64 operations, and 27 multiply-adds per point): analytically equal,
// a third of the work on the serial chain.  The oracle keeps OpenCV's table; parity is checked to 1e-9.
// sin and cos of x >= 0 together, for the rotation angles of this path (|x| < 2^20; anything else goes to the library).
// Cody-Waite reduction by pi/2 in two terms, then the fdlibm kernel polynomials on |r| <= pi/4 (< 1 ulp): two short
// interleaved Horner chains instead of the library's general-argument sincos (which sat at 0.3 us of every LM evaluation).
__device__ __forceinline__ void agt_sincos(double x, double& s, double& c)
{
    if (!(x < 1048576.0)) { sincos(x, &s, &c); return; }
    const double k = rint(x * 6.36619772367581382433e-01);               // 2 / pi
    double r = fma(-k, 1.57079632673412561417e+00, x);                    // pi/2, first 33 bits
    r = fma(-k, 6.07710050650619224932e-11, r);                           // pi/2 - the above
    const double z = r * r;
    const double sp = fma(z, fma(z, fma(z, fma(z, fma(z, 1.58969099521155010221e-10, -2.50507602534068634195e-08), 2.75573137070700676789e-06),
                                       -1.98412698298579493134e-04), 8.33333333332248946124e-03), -1.66666666666666324348e-01);
    const double cp = fma(z, fma(z, fma(z, fma(z, fma(z, -1.13596475577881948265e-11, 2.08757232129817482790e-09), -2.75573143513906633035e-07),
                                       2.48015872894767294178e-05), -1.38888888888741095749e-03), 4.16666666666666019037e-02);
    const double sr = fma(r * z, sp, r);
    const double cr = fma(z * z, cp, fma(z, -0.5, 1.0));
    const int q = (int)k & 3;
    const double ss = (q & 1) ? cr : sr, cc = (q & 1) ? sr : cr;
    s = (q & 2) ? -ss : ss;
    c = ((q + 1) & 2) ? -cc : cc;
}

template <bool JAC>
__device__ __forceinline__ void agt_rodrigues(const double r_in[3], double R[9], double G[9])
{
    double rx = r_in[0], ry = r_in[1], rz = r_in[2];
    const double t2 = rx * rx + ry * ry + rz * rz;
    const double theta = agt_sqrtp(t2);
    // theta < DBL_EPSILON: R = I (and G = I: dR/dr_j = [e_j]x at r = 0).  Written as selects on the results, not as an early
    // return: with two exits the optimiser kept R[0] and G[0] in memory behind a pointer phi -- 32 B of scratch per lane in every
    // kernel that calls this (1.5 MB of scratch writes per dense launch, a scratch round trip in each LM evaluation).
    const bool tiny = theta < DBL_EPSILON;
    double s, c;
    agt_sincos(theta, s, c);
    const double c1 = 1.0 - c, itheta = agt_rcp(tiny ? 1.0 : theta);
    if (JAC) {
        // a = (1 - cos t)/t^2, b = (t - sin t)/t^3; series below t = 1e-2 (cancellation), relative error < 1e-16 there
        double a, bq;
        if (theta < 1e-2) {
            a = 0.5 - t2 * (1.0 / 24.0 - t2 * (1.0 / 720.0 - t2 * (1.0 / 40320.0)));
            bq = 1.0 / 6.0 - t2 * (1.0 / 120.0 - t2 * (1.0 / 5040.0 - t2 * (1.0 / 362880.0)));
        } else {
            const double it2 = itheta * itheta;
            a = c1 * it2;
            bq = (theta - s) * it2 * itheta;
        }
        const double d = 1.0 - bq * t2;                       // [r]x^2 = r r^T - t^2 I
        const double g[9] = { d + bq * rx * rx, bq * rx * ry - a * rz, bq * rx * rz + a * ry,
                              bq * rx * ry + a * rz, d + bq * ry * ry, bq * ry * rz - a * rx,
                              bq * rx * rz - a * ry, bq * ry * rz + a * rx, d + bq * rz * rz };
#pragma unroll
        for (int i = 0; i < 9; i++) G[i] = tiny ? ((i % 4 == 0) ? 1.0 : 0.0) : g[i];
    }
    rx *= itheta; ry *= itheta; rz *= itheta;
    const double rrt[9] = { rx * rx, rx * ry, rx * rz, rx * ry, ry * ry, ry * rz, rx * rz, ry * rz, rz * rz };
    const double r_x[9] = { 0, -rz, ry, rz, 0, -rx, -ry, rx, 0 };
#pragma unroll
    for (int k = 0; k < 9; k++) {
        const double v = c * ((k % 4 == 0) ? 1.0 : 0.0) + c1 * rrt[k] + s * r_x[k];
        R[k] = tiny ? ((k % 4 == 0) ? 1.0 : 0.0) : v;
    }
}

struct AgtCamera {
    double fx, fy, cx, cy;
    double k[12];     // k1 k2 p1 p2 k3 k4 k5 k6 s1 s2 s3 s4
    // tilted sensor (tau_x, tau_y != 0): where matTilt / invMatTilt live (row-major, 9 + 9 doubles, uniform); null = no tilt.  Read only
    // inside the `if (cam.tilt)` branches of the distortion paths, so a camera without tilt costs a scalar compare per projection.
    const double* tilt;
};

// vecTilt = M (x, y, 1) dehomogenised, in OpenCV's accumulation order (s = 0; s += m * v); d: optional 2 x 2 Jacobian
__device__ __forceinline__ void agt_tilt_apply(const double* M, double x, double y, double& xo, double& yo, double* d)
{
    double v[3];
#pragma unroll
    for (int r = 0; r < 3; r++) { double a = 0.0; a += M[r * 3] * x; a += M[r * 3 + 1] * y; a += M[r * 3 + 2] * 1.0; v[r] = a; }
    const double ip = v[2] != 0.0 ? 1.0 / v[2] : 1.0;
    xo = ip * v[0]; yo = ip * v[1];
    if (d) {
        const double ip2 = ip * ip;
#pragma unroll
        for (int r = 0; r < 2; r++)
#pragma unroll
            for (int c = 0; c < 2; c++) d[r * 2 + c] = (M[r * 3 + c] * v[2] - M[6 + c] * v[r]) * ip2;
    }
}

// cvProjectPoints2Internal for one point.  jr/jt: rows (du/d., dv/d.) x 3.
// DIST = false is the exact specialisation for all-zero distortion coefficients (every term it
// drops is a multiplication by 0 or by 1): same values, ~40 % fewer FP64 instructions.
template <bool JAC, bool DIST = true>
__device__ __forceinline__ void agt_project(const AgtCamera& cam, const double R[9], const double G[9],
                                            const double t[3], double X, double Y, double Z,
                                            double& u, double& v, double jr[6], double jt[6])
{
    const double* k = cam.k;
    const double wx = R[0] * X + R[1] * Y + R[2] * Z;        // rotated point; d(R X)/dr_j = G_j x w
    const double wy = R[3] * X + R[4] * Y + R[5] * Z;
    const double wz = R[6] * X + R[7] * Y + R[8] * Z;
    double x = wx + t[0];
    double y = wy + t[1];
    double z = wz + t[2];
    z = z != 0.0 ? agt_rcp(z) : 1.0;
    x *= z; y *= z;
    if (!DIST) {
        u = x * cam.fx + cam.cx;
        v = y * cam.fy + cam.cy;
        if (JAC) {
            // d(x,y)/dt = (z, 0, -x z), (0, z, -y z); d(x,y)/dr_j = z (dY_j.xy - (x,y) dY_j.z)
            jt[0] = cam.fx * z; jt[1] = 0.0; jt[2] = cam.fx * (-x * z);
            jt[3] = 0.0; jt[4] = cam.fy * z; jt[5] = cam.fy * (-y * z);
#pragma unroll
            for (int j = 0; j < 3; j++) {
                const double gx = G[j], gy = G[3 + j], gz = G[6 + j];
                const double dx0 = gy * wz - gz * wy;
                const double dy0 = gz * wx - gx * wz;
                const double dz0 = gx * wy - gy * wx;
                jr[j] = cam.fx * (z * (dx0 - x * dz0));
                jr[3 + j] = cam.fy * (z * (dy0 - y * dz0));
            }
        }
        return;
    }
    double r2 = x * x + y * y, r4 = r2 * r2, r6 = r4 * r2;
    double a1 = 2 * x * y, a2 = r2 + 2 * x * x, a3 = r2 + 2 * y * y;
    double cdist = 1 + k[0] * r2 + k[1] * r4 + k[4] * r6;
    double icdist2 = 1.0 / (1 + k[5] * r2 + k[6] * r4 + k[7] * r6);
    double xd = x * cdist * icdist2 + k[2] * a1 + k[3] * a2 + k[8] * r2 + k[9] * r4;
    double yd = y * cdist * icdist2 + k[2] * a3 + k[3] * a1 + k[10] * r2 + k[11] * r4;
    // additional distortion by projecting onto a tilt plane (cvProjectPoints2Internal); dT = d(xd, yd)_tilted / d(xd, yd)
    double dT[4] = { 1.0, 0.0, 0.0, 1.0 };
    const bool tilted = cam.tilt != nullptr;
    if (tilted) agt_tilt_apply(cam.tilt, xd, yd, xd, yd, JAC ? dT : nullptr);
    u = xd * cam.fx + cam.cx;
    v = yd * cam.fy + cam.cy;
    if (JAC) {
        const double dxdt[3] = { z, 0, -x * z }, dydt[3] = { 0, z, -y * z };
#pragma unroll
        for (int j = 0; j < 3; j++) {
            double dr2dt = 2 * x * dxdt[j] + 2 * y * dydt[j];
            double dcdist_dt = k[0] * dr2dt + 2 * k[1] * r2 * dr2dt + 3 * k[4] * r4 * dr2dt;
            double dicdist2_dt = -icdist2 * icdist2 * (k[5] * dr2dt + 2 * k[6] * r2 * dr2dt + 3 * k[7] * r4 * dr2dt);
            double da1dt = 2 * (x * dydt[j] + y * dxdt[j]);
            double dmxdt = (dxdt[j] * cdist * icdist2 + x * dcdist_dt * icdist2 + x * cdist * dicdist2_dt +
                            k[2] * da1dt + k[3] * (dr2dt + 4 * x * dxdt[j]) + k[8] * dr2dt + 2 * r2 * k[9] * dr2dt);
            double dmydt = (dydt[j] * cdist * icdist2 + y * dcdist_dt * icdist2 + y * cdist * dicdist2_dt +
                            k[2] * (dr2dt + 4 * y * dydt[j]) + k[3] * da1dt + k[10] * dr2dt + 2 * r2 * k[11] * dr2dt);
            if (tilted) { const double a = dmxdt, b = dmydt; dmxdt = (0.0 + dT[0] * a) + dT[1] * b; dmydt = (0.0 + dT[2] * a) + dT[3] * b; }
            jt[j] = cam.fx * dmxdt;
            jt[3 + j] = cam.fy * dmydt;
        }
        double dx0dr[3], dy0dr[3], dz0dr[3];
#pragma unroll
        for (int j = 0; j < 3; j++) {
            const double gx = G[j], gy = G[3 + j], gz = G[6 + j];
            dx0dr[j] = gy * wz - gz * wy;
            dy0dr[j] = gz * wx - gx * wz;
            dz0dr[j] = gx * wy - gy * wx;
        }
#pragma unroll
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
            if (tilted) { const double a = dmxdr, b = dmydr; dmxdr = (0.0 + dT[0] * a) + dT[1] * b; dmydr = (0.0 + dT[2] * a) + dT[3] * b; }
            jr[j] = cam.fx * dmxdr;
            jr[3 + j] = cam.fy * dmydr;
        }
    }
}

// One-sided Jacobi SVD of a 3x3 (row-major A).  Outputs sorted descending; U columns = left
// vectors, Vt rows = right vectors (the layout of oracle cvo_svd).
__device__ inline void agt_svd3(const double A[9], double W[3], double U[9], double Vt[9])
{
    double At[3][3], V[3][3], w[3];
#pragma unroll
    for (int i = 0; i < 3; i++) {
        double sd = 0;
#pragma unroll
        for (int k = 0; k < 3; k++) { At[i][k] = A[k * 3 + i]; sd += At[i][k] * At[i][k]; V[i][k] = (i == k) ? 1.0 : 0.0; }
        w[i] = sd;
    }
    const double eps = DBL_EPSILON * 2;
    for (int iter = 0; iter < 30; iter++) {
        bool changed = false;
#pragma unroll
        for (int i = 0; i < 2; i++)
#pragma unroll
            for (int j = i + 1; j < 3; j++) {
                double a = w[i], b = w[j];
                double p = At[i][0] * At[j][0] + At[i][1] * At[j][1] + At[i][2] * At[j][2];
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
#pragma unroll
                for (int k = 0; k < 3; k++) {
                    double t0 = c * At[i][k] + s * At[j][k];
                    double t1 = -s * At[i][k] + c * At[j][k];
                    At[i][k] = t0; At[j][k] = t1;
                    a += t0 * t0; b += t1 * t1;
                    double v0 = c * V[i][k] + s * V[j][k];
                    double v1 = -s * V[i][k] + c * V[j][k];
                    V[i][k] = v0; V[j][k] = v1;
                }
                w[i] = a; w[j] = b;
                changed = true;
            }
        if (!changed) break;
    }
#pragma unroll
    for (int i = 0; i < 3; i++) w[i] = sqrt(At[i][0] * At[i][0] + At[i][1] * At[i][1] + At[i][2] * At[i][2]);
    // sort descending (3 elements: fixed compare-exchange network, static indices)
#define AGT_SVD3_CSWAP(i, j)                                                   \
    if (w[i] < w[j]) {                                                         \
        double t_ = w[i]; w[i] = w[j]; w[j] = t_;                              \
        for (int k = 0; k < 3; k++) {                                          \
            t_ = At[i][k]; At[i][k] = At[j][k]; At[j][k] = t_;                 \
            t_ = V[i][k]; V[i][k] = V[j][k]; V[j][k] = t_;                     \
        }                                                                      \
    }
    AGT_SVD3_CSWAP(0, 1) AGT_SVD3_CSWAP(0, 2) AGT_SVD3_CSWAP(1, 2)
#undef AGT_SVD3_CSWAP
#pragma unroll
    for (int i = 0; i < 3; i++) {
        W[i] = w[i];
        double s = w[i] > 0 ? 1.0 / w[i] : 0.0;
#pragma unroll
        for (int k = 0; k < 3; k++) { U[k * 3 + i] = At[i][k] * s; Vt[i * 3 + k] = V[i][k]; }
    }
}

__device__ __forceinline__ void agt_mat3_mul(const double A[9], const double B[9], double C[9])
{
    double T[9];
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) T[i * 3 + j] = A[i * 3] * B[j] + A[i * 3 + 1] * B[3 + j] + A[i * 3 + 2] * B[6 + j];
#pragma unroll
    for (int i = 0; i < 9; i++) C[i] = T[i];
}

__device__ __forceinline__ double agt_det3(const double M[9])
{
    return M[0] * (M[4] * M[8] - M[5] * M[7]) - M[1] * (M[3] * M[8] - M[5] * M[6]) + M[2] * (M[3] * M[7] - M[4] * M[6]);
}

// cvRodrigues2 matrix->vector (no Jacobian).  OpenCV first replaces R by its polar factor U V^T
// (SVD).  For an input that is orthonormal to ~1e-7 (always the case for the motion model's
// products of rotations) one Newton-Schulz step R (3I - R^T R) / 2 reproduces that factor to
// O(|R^T R - I|^2) < 1e-14 at a fraction of the Jacobi SVD's latency; otherwise the SVD runs.
__device__ inline void agt_rodrigues_inv(const double Rin[9], double r[3])
{
    double R[9];
    double G[9];          // R^T R
    double dev = 0.0;
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) {
            G[i * 3 + j] = Rin[i] * Rin[j] + Rin[3 + i] * Rin[3 + j] + Rin[6 + i] * Rin[6 + j];
            dev = fmax(dev, fabs(G[i * 3 + j] - (i == j ? 1.0 : 0.0)));
        }
    if (dev < 1e-7) {
        double Hm[9];
#pragma unroll
        for (int i = 0; i < 9; i++) Hm[i] = 0.5 * ((i % 4 == 0 ? 3.0 : 0.0) - G[i]);
        agt_mat3_mul(Rin, Hm, R);
    } else {
        double W[3], U[9], Vt[9];
        agt_svd3(Rin, W, U, Vt);
        agt_mat3_mul(U, Vt, R);
    }
    double rx = R[7] - R[5], ry = R[2] - R[6], rz = R[3] - R[1];
    double s = agt_sqrtp((rx * rx + ry * ry + rz * rz) * 0.25);
    double c = (R[0] + R[4] + R[8] - 1) * 0.5;
    c = c > 1.0 ? 1.0 : c < -1.0 ? -1.0 : c;
    double theta = acos(c);
    if (s < 1e-5) {
        if (c > 0) rx = ry = rz = 0;
        else {
            double t;
            t = (R[0] + 1) * 0.5; rx = sqrt(t > 0.0 ? t : 0.0);
            t = (R[4] + 1) * 0.5; ry = sqrt(t > 0.0 ? t : 0.0) * (R[1] < 0 ? -1.0 : 1.0);
            t = (R[8] + 1) * 0.5; rz = sqrt(t > 0.0 ? t : 0.0) * (R[2] < 0 ? -1.0 : 1.0);
            if (fabs(rx) < fabs(ry) && fabs(rx) < fabs(rz) && (R[5] > 0) != (ry * rz > 0)) rz = -rz;
            theta /= sqrt(rx * rx + ry * ry + rz * rz);
            rx *= theta; ry *= theta; rz *= theta;
        }
    } else {
        double vth = agt_rcp(2 * s);      // (s >= 1e-5 here)
        vth *= theta;
        rx *= vth; ry *= vth; rz *= vth;
    }
    r[0] = rx; r[1] = ry; r[2] = rz;
}

// Solve the damped 6x6 normal equations A x = b, A symmetric positive definite
// (upper triangle used).  LDL^T, no square roots.  Returns false on a non-positive pivot.
__device__ __forceinline__ bool agt_solve6(const double A[36], const double b[6], double x[6])
{
    double L[6][6], D[6], iD[6];
    bool ok = true;
#pragma unroll
    for (int j = 0; j < 6; j++) {
        double d = A[j * 6 + j];
#pragma unroll
        for (int k = 0; k < j; k++) d -= L[j][k] * L[j][k] * D[k];
        if (!(d > 0.0)) ok = false;
        D[j] = d;
        double id = agt_rcp1(d);          // (d > 0 checked above; a non-positive pivot already reports failure)
        iD[j] = id;
#pragma unroll
        for (int i = j + 1; i < 6; i++) {
            double v = A[j * 6 + i];
#pragma unroll
            for (int k = 0; k < j; k++) v -= L[i][k] * L[j][k] * D[k];
            L[i][j] = v * id;
        }
    }
    double y[6];
#pragma unroll
    for (int i = 0; i < 6; i++) {
        double v = b[i];
#pragma unroll
        for (int k = 0; k < i; k++) v -= L[i][k] * y[k];
        y[i] = v;
    }
#pragma unroll
    for (int i = 5; i >= 0; i--) {
        double v = y[i] * iD[i];          // one reciprocal per pivot serves both sweeps
#pragma unroll
        for (int k = i + 1; k < 6; k++) v -= L[k][i] * x[k];
        x[i] = v;
    }
    return ok;
}

#pragma clang fp contract(off)
