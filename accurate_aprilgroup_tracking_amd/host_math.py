"""Host-side 3x3 / SE(3) algebra of the path (numpy).

The reference does this algebra on the host too: cv.Rodrigues at detect_pose.py:275-276,
330, 344 and transform_helper.py:87, and the numpy helpers of transform_helper.py:123-259.
Only `Rodrigues` lives here; the TransformHelper mirror is in transform_helper.py.
Semantics follow OpenCV calibration.cpp cvRodrigues2; the device
twin used inside the PnP kernel is csrc/agt_device.h agt_rodrigues / agt_rodrigues_inv.
"""
import numpy as np

_EPS = np.finfo(np.float64).eps


def _vec2mat(r):
    theta = float(np.sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]))
    J = np.zeros((3, 9))
    if theta < _EPS:
        J[0, 5] = J[1, 6] = J[2, 1] = -1.0
        J[0, 7] = J[1, 2] = J[2, 3] = 1.0
        return np.eye(3), J
    c, s = np.cos(theta), np.sin(theta)
    c1, it = 1.0 - c, 1.0 / theta
    k = r * it
    rrt = np.outer(k, k)
    r_x = np.array([[0.0, -k[2], k[1]], [k[2], 0.0, -k[0]], [-k[1], k[0], 0.0]])
    R = c * np.eye(3) + c1 * rrt + s * r_x
    drrt = np.array([[k[0] + k[0], k[1], k[2], k[1], 0, 0, k[2], 0, 0],
                     [0, k[0], 0, k[0], k[1] + k[1], k[2], 0, k[2], 0],
                     [0, 0, k[0], 0, 0, k[1], k[0], k[1], k[2] + k[2]]])
    d_r_x = np.array([[0, 0, 0, 0, 0, -1, 0, 1, 0],
                      [0, 0, 1, 0, 0, 0, -1, 0, 0],
                      [0, -1, 0, 1, 0, 0, 0, 0, 0]], dtype=np.float64)
    I9 = np.eye(3).reshape(9)
    for i in range(3):
        ri = k[i]
        a0, a1, a2 = -s * ri, (s - 2 * c1 * it) * ri, c1 * it
        a3, a4 = (c - s * it) * ri, s * it
        J[i] = a0 * I9 + a1 * rrt.reshape(9) + a2 * drrt[i] + a3 * r_x.reshape(9) + a4 * d_r_x[i]
    return R, J


def _mat2vec(Rin, jac=None):
    """jac: optional (3, 9) array that receives d(rvec)/d(R) (R row-major), as cvRodrigues2 forms it -- through the
    orthonormalised R, zeros in the singular branches (angle ~ 0 or ~ pi)."""
    if jac is not None:
        jac[:] = 0.0
    if not np.all((Rin > -100.0) & (Rin < 100.0)):
        return np.zeros(3)
    U, _, Vt = np.linalg.svd(Rin)
    R = U @ Vt
    r = np.array([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]])
    s = np.sqrt((r @ r) * 0.25)
    c = float(np.clip((np.trace(R) - 1.0) * 0.5, -1.0, 1.0))
    theta = np.arccos(c)
    if jac is not None and s >= 1e-5:
        # calibration.cpp cvRodrigues2, matrix -> vector: var = [om1; vth; theta] with om1 = (R32 - R23, R13 - R31, R21 - R12), vth = 1 / (2 s);
        # omega = om1 * vth * theta.  d omega / dR = (d omega / d var2) (d var2 / d var) (d var / dR): 3x4, 4x5, 5x9
        vth = 1.0 / (2.0 * s)
        dtheta_dtr = -1.0 / s
        dvth_dtheta = -vth * c / s
        d1, d2 = 0.5 * dvth_dtheta * dtheta_dtr, 0.5 * dtheta_dtr
        dvardR = np.array([[0, 0, 0, 0, 0, 1, 0, -1, 0],
                           [0, 0, -1, 0, 0, 0, 1, 0, 0],
                           [0, 1, 0, -1, 0, 0, 0, 0, 0],
                           [d1, 0, 0, 0, d1, 0, 0, 0, d1],
                           [d2, 0, 0, 0, d2, 0, 0, 0, d2]], dtype=np.float64)
        dvar2dvar = np.array([[vth, 0, 0, r[0], 0],
                              [0, vth, 0, r[1], 0],
                              [0, 0, vth, r[2], 0],
                              [0, 0, 0, 0, 1]], dtype=np.float64)
        domegadvar2 = np.array([[theta, 0, 0, r[0] * vth],
                                [0, theta, 0, r[1] * vth],
                                [0, 0, theta, r[2] * vth]], dtype=np.float64)
        J = domegadvar2 @ dvar2dvar @ dvardR
        jac[:] = J.reshape(3, 3, 3).transpose(0, 2, 1).reshape(3, 9)      # (cvRodrigues2 transposes every row, read as a 3 x 3 matrix)
    if s < 1e-5:
        if c > 0:
            return np.zeros(3)
        r = np.array([np.sqrt(max((R[0, 0] + 1) * 0.5, 0.0)),
                      np.sqrt(max((R[1, 1] + 1) * 0.5, 0.0)) * (-1.0 if R[0, 1] < 0 else 1.0),
                      np.sqrt(max((R[2, 2] + 1) * 0.5, 0.0)) * (-1.0 if R[0, 2] < 0 else 1.0)])
        if abs(r[0]) < abs(r[1]) and abs(r[0]) < abs(r[2]) and ((R[1, 2] > 0) != (r[1] * r[2] > 0)):
            r[2] = -r[2]
        return r * (theta / np.linalg.norm(r))
    return r * (theta / (2 * s))


def Rodrigues(src):
    """cv2.Rodrigues: 3-vector -> ((3,3), jac (3,9)); (3,3) -> ((3,1), jac (9,3) = d(rvec)/d(R) transposed, zeros in the singular
    branches).  Output depth follows the input depth (float32 in -> float32 out), as OpenCV.  (The reference only ever takes [0]:
    detect_pose.py:275-276, 330, 344; the matrix -> vector Jacobian was zeros until round 6.)"""
    a = np.asarray(src)
    out_dtype = np.float32 if a.dtype == np.float32 else np.float64
    if a.size == 3:
        R, J = _vec2mat(a.astype(np.float64).reshape(3))
        return R.astype(out_dtype), J.astype(out_dtype)
    if a.shape == (3, 3):
        J = np.zeros((3, 9))
        r = _mat2vec(a.astype(np.float64), J)
        return r.reshape(3, 1).astype(out_dtype), np.ascontiguousarray(J.T).astype(out_dtype)
    raise ValueError("Rodrigues: input must be a 3-vector or a 3x3 matrix, got shape %r" % (a.shape,))
