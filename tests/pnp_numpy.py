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


def project(obj, r, t, K, dist=None, jacobian=False):
    """cv::projectPoints: obj (N,3) -> img (N,2) [, J (2N,6) = d(u,v)/d(r,t)]"""
    X = np.asarray(obj, np.float64).reshape(-1, 3)
    n = X.shape[0]
    k = np.zeros(12)
    if dist is not None:
        d = np.asarray(dist, np.float64).reshape(-1)
        k[:d.size] = d[:12]
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
    img = np.stack([fx * xd + cx, fy * yd + cy], axis=1)
    if not jacobian:
        return img
    J = np.zeros((n, 2, 6))

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
        return fx * dxd, fy * dyd
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
    k = np.zeros(12)
    if dist is not None:
        d = np.asarray(dist, np.float64).reshape(-1)
        k[:min(d.size, 12)] = d[:12]
    x0 = (p[:, 0] - K[0, 2]) / K[0, 0]
    y0 = (p[:, 1] - K[1, 2]) / K[1, 1]
    x, y = x0.copy(), y0.copy()
    if dist is None:
        return np.stack([x, y], 1)
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
