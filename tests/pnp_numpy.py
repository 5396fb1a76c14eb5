"""Independent numpy statement of cv::solvePnP(SOLVEPNP_ITERATIVE) WITH an extrinsic guess (TEST INFRASTRUCTURE).

Written from the prose of SURVEY.md Appendix B step 3 (the CvLevMarq state machine) and Appendix C (projectPoints with its
Jacobian through the 3 x 9 dR/dr table of Rodrigues), NOT from oracle/cv_pnp.c: vectorised numpy over the points,
numpy.linalg for the damped 6 x 6 solves (pseudo-inverse through the SVD, as DECOMP_SVD), the state machine as an explicit
two-state loop.  It exists so that the C oracle's LM loop -- when a step is accepted or rejected, how lambda moves, when the
iteration stops, how many iterations that takes -- has a second, structurally different implementation to be compared with
(tests/test_oracle.py::test_pnp_lm_oracle_equals_numpy_statement): equal iteration counts, poses to 2e-9 (measured 1e-10).
The second half states the initialisation WITHOUT a guess (Appendix B steps 1-2, non-planar DLT branch) the same way
(test_pnp_noguess_init_oracle_equals_numpy_statement).
"""
import numpy as np

FLT_EPSILON = float(np.finfo(np.float32).eps)
DBL_EPSILON = float(np.finfo(np.float64).eps)


def rodrigues(r):
    """-> (R (3,3), dRdr (3,9): row i = d vec(R) / d r_i, R row-major)"""
    r = np.asarray(r, np.float64).reshape(3)
    theta = float(np.linalg.norm(r))
    J = np.zeros((3, 9))
    if theta < DBL_EPSILON:
        J[0, 5] = J[1, 6] = J[2, 1] = -1.0
        J[0, 7] = J[1, 2] = J[2, 3] = 1.0
        return np.eye(3), J
    c, s = np.cos(theta), np.sin(theta)
    c1 = 1.0 - c
    k = r / theta
    kkt = np.outer(k, k)
    kx = np.array([[0.0, -k[2], k[1]], [k[2], 0.0, -k[0]], [-k[1], k[0], 0.0]])
    R = c * np.eye(3) + c1 * kkt + s * kx
    eye9, kkt9, kx9 = np.eye(3).reshape(9), kkt.reshape(9), kx.reshape(9)
    for i in range(3):
        e = np.zeros(3); e[i] = 1.0
        dkkt = (np.outer(e, k) + np.outer(k, e)).reshape(9)               # d(k k^T)_i (before the 1/theta factor)
        dkx = np.array([[0.0, -e[2], e[1]], [e[2], 0.0, -e[0]], [-e[1], e[0], 0.0]]).reshape(9)
        J[i] = (-s * k[i]) * eye9 + ((s - 2.0 * c1 / theta) * k[i]) * kkt9 + (c1 / theta) * dkkt \
            + ((c - s / theta) * k[i]) * kx9 + (s / theta) * dkx
    return R, J


def tilt_matrix(tau_x, tau_y):
    """Tilted-sensor homography of OpenCV's 14-coefficient model (calib3d documentation, "tilted sensor" / Scheimpflug term): the
    distorted normalised point is rotated by R = R_y(tau_y) R_x(tau_x) and projected back onto z = 1 along the ROTATED optical axis:
    [[R22, 0, -R02], [0, R22, -R12], [0, 0, 1]] R."""
    cx_, sx_, cy_, sy_ = np.cos(tau_x), np.sin(tau_x), np.cos(tau_y), np.sin(tau_y)
    Rx = np.array([[1, 0, 0], [0, cx_, sx_], [0, -sx_, cx_]])
    Ry = np.array([[cy_, 0, -sy_], [0, 1, 0], [sy_, 0, cy_]])
    R = Ry @ Rx
    return np.array([[R[2, 2], 0, -R[0, 2]], [0, R[2, 2], -R[1, 2]], [0, 0, 1]]) @ R


def _coeffs14(dist):
    k = np.zeros(14)
    if dist is not None:
        d = np.asarray(dist, np.float64).reshape(-1)
        k[:d.size] = d[:14]
    return k


def project(obj, r, t, K, dist=None, jacobian=False):
    """cv::projectPoints: obj (N,3) -> img (N,2) [, J (2N,6) = d(u,v)/d(r,t)]; dist: up to 14 coefficients (k1 k2 p1 p2 k3 k4 k5 k6
    s1 s2 s3 s4 tau_x tau_y)"""
    X = np.asarray(obj, np.float64).reshape(-1, 3)
    n = X.shape[0]
    k = _coeffs14(dist)
    T = tilt_matrix(k[12], k[13])
    fx, fy, cx, cy = K[0, 0], K[1, 1], K[0, 2], K[1, 2]
    R, dRdr = rodrigues(r)
    Y = X @ R.T + np.asarray(t, np.float64).reshape(3)
    z = np.where(Y[:, 2] != 0.0, 1.0 / np.where(Y[:, 2] != 0.0, Y[:, 2], 1.0), 1.0)
    x, y = Y[:, 0] * z, Y[:, 1] * z
    r2 = x * x + y * y
    r4, r6 = r2 * r2, r2 * r2 * r2
    a1, a2, a3 = 2 * x * y, r2 + 2 * x * x, r2 + 2 * y * y
    cdist = 1 + k[0] * r2 + k[1] * r4 + k[4] * r6
    icdist2 = 1.0 / (1 + k[5] * r2 + k[6] * r4 + k[7] * r6)
    xd = x * cdist * icdist2 + k[2] * a1 + k[3] * a2 + k[8] * r2 + k[9] * r4
    yd = y * cdist * icdist2 + k[2] * a3 + k[3] * a1 + k[10] * r2 + k[11] * r4
    # tilted sensor: (xd, yd, 1) -> T (xd, yd, 1), dehomogenised; its 2 x 2 Jacobian by the quotient rule
    hx, hy, hw = (T[q, 0] * xd + T[q, 1] * yd + T[q, 2] for q in range(3))
    xt, yt = hx / hw, hy / hw
    img = np.stack([fx * xt + cx, fy * yt + cy], axis=1)
    if not jacobian:
        return img
    J = np.zeros((n, 2, 6))
    D = [[(T[q, c] * hw - T[2, c] * (hx, hy)[q]) / (hw * hw) for c in range(2)] for q in range(2)]

    def chain(dx, dy):
        """d(xd, yd) for a perturbation (dx, dy) of the normalised point"""
        dr2 = 2 * x * dx + 2 * y * dy
        dcdist = k[0] * dr2 + 2 * k[1] * r2 * dr2 + 3 * k[4] * r4 * dr2
        dicdist2 = -icdist2 * icdist2 * (k[5] * dr2 + 2 * k[6] * r2 * dr2 + 3 * k[7] * r4 * dr2)
        da1 = 2 * (x * dy + y * dx)
        dxd = dx * cdist * icdist2 + x * dcdist * icdist2 + x * cdist * dicdist2 + k[2] * da1 + k[3] * (dr2 + 4 * x * dx) \
            + k[8] * dr2 + 2 * k[9] * r2 * dr2
        dyd = dy * cdist * icdist2 + y * dcdist * icdist2 + y * cdist * dicdist2 + k[2] * (dr2 + 4 * y * dy) + k[3] * da1 \
            + k[10] * dr2 + 2 * k[11] * r2 * dr2
        return fx * (D[0][0] * dxd + D[0][1] * dyd), fy * (D[1][0] * dxd + D[1][1] * dyd)
    # translation: dY = e_j
    for j in range(3):
        e = np.zeros(3); e[j] = 1.0
        dx = z * (e[0] - x * e[2]); dy = z * (e[1] - y * e[2])
        J[:, 0, 3 + j], J[:, 1, 3 + j] = chain(dx, dy)
    # rotation: dY = (dR/dr_j) X
    for j in range(3):
        dY = X @ dRdr[j].reshape(3, 3).T
        dx = z * (dY[:, 0] - x * dY[:, 2]); dy = z * (dY[:, 1] - y * dY[:, 2])
        J[:, 0, j], J[:, 1, j] = chain(dx, dy)
    return img, J.reshape(2 * n, 6)


def solve_pnp_guess(obj, img, K, dist, rvec, tvec, max_iter=20, epsilon=FLT_EPSILON, trace=None):
    """-> (rvec (3,), tvec (3,), iterations).  CvLevMarq(6, 2N, (ITER + EPS, 20, FLT_EPSILON), completeSymm = true)"""
    img = np.asarray(img, np.float64).reshape(-1, 2)
    param = np.concatenate([np.asarray(rvec, np.float64).reshape(3), np.asarray(tvec, np.float64).reshape(3)])
    prev_param = param.copy()
    lambda_lg10 = -3
    iters = 0
    prev_err_norm = np.inf
    JtJ = np.zeros((6, 6)); JtErr = np.zeros(6)

    def step():
        lam = np.exp(lambda_lg10 * np.log(10.0))
        A = JtJ.copy()
        A[np.diag_indices(6)] *= 1.0 + lam
        return prev_param - np.linalg.pinv(A) @ JtErr               # cvSolve(..., DECOMP_SVD)

    state = "CALC_J"
    while True:
        if state == "CALC_J":
            p, J = project(obj, param[:3], param[3:], K, dist, jacobian=True)
            err = (p - img).reshape(-1)
            JtJ = J.T @ J
            JtErr = J.T @ err
            prev_param = param.copy()
            param = step()
            if iters == 0:
                prev_err_norm = float(np.linalg.norm(err))
            state = "CHECK_ERR"
        else:
            err = (project(obj, param[:3], param[3:], K, dist) - img).reshape(-1)
            err_norm = float(np.linalg.norm(err))
            if err_norm > prev_err_norm:
                lambda_lg10 += 1
                if lambda_lg10 <= 16:
                    param = step()
                    if trace is not None:
                        trace.append(("reject", iters, lambda_lg10))
                    continue
            lambda_lg10 = max(lambda_lg10 - 1, -16)
            iters += 1
            if trace is not None:
                trace.append(("accept", iters, lambda_lg10, param.copy()))
            if iters >= max_iter or np.linalg.norm(param - prev_param) / np.linalg.norm(prev_param) < epsilon:
                return param[:3].copy(), param[3:].copy(), iters
            prev_err_norm = err_norm
            state = "CALC_J"


# ---- initialisation WITHOUT a guess, written from that prose: numpy.linalg.svd throughout

def undistort_points(img, K, dist=None, iters=5):
    """cv::undistortPoints with R = I, no P: pixel -> normalised coordinates, `iters` fixed-point iterations of the inverse
    Brown-Conrady map (rational + tangential + thin-prism terms)"""
    p = np.asarray(img, np.float64).reshape(-1, 2)
    k = _coeffs14(dist)
    x0 = (p[:, 0] - K[0, 2]) / K[0, 0]
    y0 = (p[:, 1] - K[1, 2]) / K[1, 1]
    x, y = x0.copy(), y0.copy()
    if dist is None:
        return np.stack([x, y], 1)
    # the sensor tilt is undone first (exact inverse homography), then the fixed-point iterations run on the un-tilted point
    h = np.linalg.inv(tilt_matrix(k[12], k[13])) @ np.stack([x0, y0, np.ones_like(x0)])
    x0, y0 = h[0] / h[2], h[1] / h[2]
    x, y = x0.copy(), y0.copy()
    for _ in range(iters):
        r2 = x * x + y * y
        icdist = (1 + ((k[7] * r2 + k[6]) * r2 + k[5]) * r2) / (1 + ((k[4] * r2 + k[1]) * r2 + k[0]) * r2)
        dx = 2 * k[2] * x * y + k[3] * (r2 + 2 * x * x) + k[8] * r2 + k[9] * r2 * r2
        dy = k[2] * (r2 + 2 * y * y) + 2 * k[3] * x * y + k[10] * r2 + k[11] * r2 * r2
        x = (x0 - dx) * icdist
        y = (y0 - dy) * icdist
    return np.stack([x, y], 1)


def rodrigues_inv(R):
    """rotation matrix -> rotation vector (the generic branch: the test scenes stay away from theta = 0 and pi)"""
    U, _, Vt = np.linalg.svd(np.asarray(R, np.float64))
    R = U @ Vt
    c = np.clip((np.trace(R) - 1) * 0.5, -1.0, 1.0)
    theta = np.arccos(c)
    v = np.array([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]])
    s = np.linalg.norm(v) * 0.5
    return v * (0.5 / s) * theta if s > 1e-12 else np.zeros(3)


def is_planar(obj):
    X = np.asarray(obj, np.float64).reshape(-1, 3)
    d = X - X.mean(0)
    w = np.linalg.svd(d.T @ d, compute_uv=False)
    return w[2] / w[1] < 1e-3


def dlt_init(obj, img, K, dist=None):
    """non-planar branch: the 2N x 12 system, last right-singular vector of L^T L as [R | t], sign by det R, R orthonormalised by
    its SVD, t rescaled by |R_new|_F / |R|_F -> (rvec, tvec)"""
    X = np.asarray(obj, np.float64).reshape(-1, 3)
    n = X.shape[0]
    assert n >= 6 and not is_planar(X)
    m = undistort_points(img, K, dist)
    L = np.zeros((2 * n, 12))
    L[0::2, 0:3] = X; L[0::2, 3] = 1.0
    L[0::2, 8:11] = -m[:, 0:1] * X; L[0::2, 11] = -m[:, 0]
    L[1::2, 4:7] = X; L[1::2, 7] = 1.0
    L[1::2, 8:11] = -m[:, 1:2] * X; L[1::2, 11] = -m[:, 1]
    _, _, Vt = np.linalg.svd(L.T @ L)
    P = Vt[-1].reshape(3, 4)
    R, t = P[:, :3].copy(), P[:, 3].copy()
    if np.linalg.det(R) < 0:
        R, t = -R, -t
    sc = np.linalg.norm(R)
    U, _, Vt2 = np.linalg.svd(R)
    Rn = U @ Vt2
    t = t * (np.linalg.norm(Rn) / sc)
    return rodrigues_inv(Rn), t


def solve_pnp_noguess(obj, img, K, dist=None):
    """cv::solvePnP(SOLVEPNP_ITERATIVE) without a guess, non-planar object: DLT initialisation, then the LM loop above"""
    r0, t0 = dlt_init(obj, img, K, dist)
    r, t, it = solve_pnp_guess(obj, img, K, dist, r0, t0)
    return r, t, it, (r0, t0)


# ---- initialisation WITHOUT a guess, PLANAR object (first branch) and cv::findHomography (method 0) with its
# LMSolver polish -- written in round 4 from the prose of Appendix B and from OpenCV 4.x's published fundam.cpp / levmarq.cpp as
# recalled, NOT from oracle/cv_pnp.c: numpy.linalg.eigh for the 9 x 9 null vector, numpy.linalg.solve / pinv for the damped
# systems, the LMSolver loop spelled out.  (test_oracle.py::test_pnp_planar_init_oracle_equals_numpy_statement)

def homography_dlt(src, dst):
    """HomographyEstimatorCallback::runKernel on float32 points: centroids, scale = 1 / mean absolute deviation per axis,
    the symmetric 9 x 9 L^T L, its eigenvector of the smallest eigenvalue, de-normalised, divided by H[2, 2]"""
    M = np.asarray(src, np.float32).astype(np.float64).reshape(-1, 2)
    m = np.asarray(dst, np.float32).astype(np.float64).reshape(-1, 2)
    n = M.shape[0]
    cM, cm = M.mean(0), m.mean(0)
    sM, sm = np.abs(M - cM).mean(0), np.abs(m - cm).mean(0)
    if (np.concatenate([sM, sm]) < DBL_EPSILON).any():
        return None
    sM, sm = 1.0 / sM, 1.0 / sm
    invHnorm = np.array([[1.0 / sm[0], 0, cm[0]], [0, 1.0 / sm[1], cm[1]], [0, 0, 1]])
    Hnorm2 = np.array([[sM[0], 0, -cM[0] * sM[0]], [0, sM[1], -cM[1] * sM[1]], [0, 0, 1]])
    LtL = np.zeros((9, 9))
    for i in range(n):
        x, y = (m[i] - cm) * sm
        X, Y = (M[i] - cM) * sM
        Lx = np.array([X, Y, 1, 0, 0, 0, -x * X, -x * Y, -x])
        Ly = np.array([0, 0, 0, X, Y, 1, -y * X, -y * Y, -y])
        LtL += np.outer(Lx, Lx) + np.outer(Ly, Ly)
    w, V = np.linalg.eigh(LtL)
    H0 = V[:, 0].reshape(3, 3)
    H = invHnorm @ H0 @ Hnorm2
    return H / H[2, 2]


def _homography_residuals(h, M, m, jacobian):
    """HomographyRefineCallback::compute: 2N residuals (projected - measured) and the 2N x 8 Jacobian w.r.t. h[0..7] (h[8] = 1)"""
    Mx, My = M[:, 0], M[:, 1]
    ww = 1.0 / (h[6] * Mx + h[7] * My + 1.0)
    xi = (h[0] * Mx + h[1] * My + h[2]) * ww
    yi = (h[3] * Mx + h[4] * My + h[5]) * ww
    r = np.empty(2 * len(Mx))
    r[0::2] = xi - m[:, 0]; r[1::2] = yi - m[:, 1]
    if not jacobian:
        return r, None
    J = np.zeros((2 * len(Mx), 8))
    J[0::2, 0] = Mx * ww; J[0::2, 1] = My * ww; J[0::2, 2] = ww
    J[0::2, 6] = -Mx * ww * xi; J[0::2, 7] = -My * ww * xi
    J[1::2, 3] = Mx * ww; J[1::2, 4] = My * ww; J[1::2, 5] = ww
    J[1::2, 6] = -Mx * ww * yi; J[1::2, 7] = -My * ww * yi
    return r, J


def lm_solver(x, compute, max_iters, eps=FLT_EPSILON):
    """cv::LMSolver::run (levmarq.cpp): Levenberg-Marquardt with the gain-ratio update of lambda (Rlo = 0.25, Rhi = 0.75),
    damping lambda * diag(J^T J), steps accepted when the squared error falls; stops after max_iters, or when the step's
    or the residual's largest component falls under eps"""
    x = np.array(x, np.float64)
    r, J = compute(x, True)
    S = float(r @ r)
    A = J.T @ J
    v = J.T @ r
    D = np.diag(A).copy()
    Rlo, Rhi = 0.25, 0.75
    lam, lc = 1.0, 0.75
    it = 0
    while True:
        Ap = A + np.diag(lam * D)
        d = np.linalg.pinv(Ap) @ v                      # solve(Ap, v, d, DECOMP_EIG): positive definite here
        xd = x - d
        rd, _ = compute(xd, False)
        Sd = float(rd @ rd)
        dS = float(d @ (2.0 * v - A @ d))
        R = (S - Sd) / (dS if abs(dS) > DBL_EPSILON else 1.0)
        if R > Rhi:
            lam *= 0.5
            if lam < lc:
                lam = 0.0
        elif R < Rlo:
            t = float(d @ v)
            nu = (Sd - S) / (t if abs(t) > DBL_EPSILON else 1.0) + 2.0
            nu = min(max(nu, 2.0), 10.0)
            if lam == 0.0:
                Ai = np.linalg.pinv(A)
                lam = lc = 1.0 / max(DBL_EPSILON, float(np.abs(np.diag(Ai)).max()))
                nu *= 0.5
            lam *= nu
        if Sd < S:
            S = Sd
            x = xd
            r, J = compute(x, True)
            A = J.T @ J
            v = J.T @ r
        it += 1
        if not (it < max_iters and np.abs(d).max() >= eps and np.abs(r).max() >= eps):
            return x, it


def find_homography(src, dst, refine=True):
    """cv::findHomography(src, dst, 0): float32 points, normalised DLT, for more than four points the LMSolver polish (10
    iterations) of the eight free entries, H[2, 2] = 1"""
    H = homography_dlt(src, dst)
    M = np.asarray(src, np.float32).astype(np.float64).reshape(-1, 2)
    m = np.asarray(dst, np.float32).astype(np.float64).reshape(-1, 2)
    if H is None or not refine or M.shape[0] <= 4:
        return H
    h, _ = lm_solver(H.reshape(9)[:8], lambda hh, jac: _homography_residuals(np.append(hh, 1.0), M, m, jac), 10)
    return np.append(h, 1.0).reshape(3, 3)


def rodrigues_matrix(R):
    """cv::Rodrigues(matrix) through rodrigues_inv, then back: the orthonormalising round trip of the planar branch"""
    r = rodrigues_inv(R)
    return rodrigues(r)[0]


def planar_init(obj, img, K, dist=None):
    """Appendix B step 2, planar branch: rotate the object points into their plane (rows of V^T of the scatter's SVD, det forced
    positive; identity when the plane is already z = const), homography plane -> normalised image, rotation from its first
    two columns (normalised; third = their cross product; orthonormalised by a Rodrigues round trip), translation from the
    third column scaled by 2 / (|h1| + |h2|) -> (rvec, tvec)"""
    X = np.asarray(obj, np.float64).reshape(-1, 3)
    Mc = X.mean(0)
    d = X - Mc
    _, w, Vt = np.linalg.svd(d.T @ d)
    assert w[2] / w[1] < 1e-3
    Rt = Vt.copy()
    # numpy's singular vectors carry arbitrary signs; OpenCV's own choice does not matter downstream EXCEPT through det < 0 -> -V^T:
    # a sign flip of a ROW pair leaves the final pose unchanged (the homography absorbs it), which the test asserts by value
    if Rt[0, 2] ** 2 + Rt[1, 2] ** 2 < 1e-10:
        Rt = np.eye(3)
    if np.linalg.det(Rt) < 0:
        Rt = -Rt
    Tt = -Rt @ Mc
    Mxy = (X @ Rt.T + Tt)[:, :2]
    mn = undistort_points(img, K, dist)
    H = find_homography(Mxy, mn)
    h1n, h2n = np.linalg.norm(H[:, 0]), np.linalg.norm(H[:, 1])
    h1 = H[:, 0] / max(h1n, DBL_EPSILON)
    h2 = H[:, 1] / max(h2n, DBL_EPSILON)
    t = H[:, 2] * (2.0 / max(h1n + h2n, DBL_EPSILON))
    Rh = np.stack([h1, h2, np.cross(h1, h2)], axis=1)
    Rh = rodrigues_matrix(Rh)
    t = Rh @ Tt + t
    R = Rh @ Rt
    return rodrigues_inv(R), t


def solve_pnp_noguess_planar(obj, img, K, dist=None):
    r0, t0 = planar_init(obj, img, K, dist)
    r, t, it = solve_pnp_guess(obj, img, K, dist, r0, t0)
    return r, t, it, (r0, t0)
