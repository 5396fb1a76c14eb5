"""ctypes loader for the CPU oracle (oracle/libcvoracle.so).

TEST INFRASTRUCTURE, NOT PRODUCT CODE: only tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg may import this module.  PARITY UNPINNED against real
cv2 (see oracle/cv_oracle.h).  The function names mirror the cv2 calls the reference
makes (detect_pose.py:509-526 solvePnP, transform_helper.py:87 Rodrigues,
transform_helper.py:106-111 projectPoints) plus calcOpticalFlowPyrLK (north-star).
"""
import ctypes as C
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# CVORACLE_LIB: another build of the same sources (the AddressSanitizer / UBSan build of `make -C oracle asan`, CPU only)
_LIB_PATH = os.environ.get("CVORACLE_LIB") or os.path.join(_HERE, "libcvoracle.so")

SOLVEPNP_ITERATIVE = 0
OPTFLOW_USE_INITIAL_FLOW = 4
OPTFLOW_LK_GET_MIN_EIGENVALS = 8
TERM_COUNT, TERM_EPS = 1, 2
ACC_EXACT, ACC_FLOAT_SCALAR, ACC_FLOAT_SIMD = 0, 1, 2


def build(force=False):
    """Compile the oracle with the committed Makefile (gcc)."""
    if force or not os.path.exists(_LIB_PATH):
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return _LIB_PATH


BUILD_FLAGS = "-O3 -march=x86-64-v3 -ffp-contract=off -fopenmp (portable build)"


def select_native():
    """bench.py's cpu_baseline: rebuild the same sources with -march=native ON THE HOST THAT RUNS THEM and bind that
    library instead (must be called before the first lib() of the process).  Falls back to the portable build when
    the host has no compiler.  Returns the flags in use."""
    global _LIB_PATH, BUILD_FLAGS
    assert _lib is None, "select_native() after the oracle was loaded"
    try:
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B", "native"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        _LIB_PATH = os.path.join(_HERE, "_native", "libcvoracle_native.so")
        BUILD_FLAGS = "-O3 -march=native -ffp-contract=off -fopenmp (built on this host)"
    except (subprocess.CalledProcessError, OSError):
        pass
    return BUILD_FLAGS


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        vp, i32, f64 = C.c_void_p, C.c_int, C.c_double
        L.cvo_pyr_down_u8.argtypes = [vp, i32, i32, i32, vp, i32]
        L.cvo_pyramid_build.restype = vp
        L.cvo_pyramid_build.argtypes = [vp, i32, i32, i32, i32, i32, i32]
        L.cvo_pyramid_free.argtypes = [vp]
        L.cvo_pyramid_levels.argtypes = [vp]
        L.cvo_pyramid_level_size.argtypes = [vp, i32, vp, vp]
        L.cvo_pyramid_level_copy.argtypes = [vp, i32, vp, i32]
        L.cvo_scharr_deriv.argtypes = [vp, i32, i32, i32, vp, i32]
        lk_tail = [i32, i32, i32, i32, i32, f64, i32, f64, i32, i32]
        L.cvo_calc_optical_flow_pyr_lk.argtypes = [vp, vp, i32, i32, i32, vp, vp, vp, vp, i32] + lk_tail
        L.cvo_lk_on_pyramids.argtypes = [vp, vp, vp, vp, vp, vp, i32] + lk_tail
        L.cvo_rodrigues_vec2mat.argtypes = [vp, vp, vp]
        L.cvo_rodrigues_vec2mat.restype = None
        L.cvo_rodrigues_mat2vec.argtypes = [vp, vp, vp]
        L.cvo_project_points.argtypes = [vp, i32, vp, vp, vp, vp, i32, vp, vp, vp]
        L.cvo_undistort_points.argtypes = [vp, i32, vp, vp, i32, vp]
        L.cvo_solve_pnp_iterative.argtypes = [vp, vp, i32, vp, vp, i32, vp, vp, i32, vp]
        L.cvo_pnp_init.argtypes = [vp, vp, i32, vp, vp, i32, vp, vp]
        L.cvo_mean_reproj_error.restype = f64
        L.cvo_mean_reproj_error.argtypes = [vp, vp, i32, vp, vp, vp, vp, i32]
        L.cvo_svd.argtypes = [vp, i32, i32, vp, vp, vp]
        L.cvo_solve_svd.argtypes = [vp, vp, i32, vp]
        L.cvo_cvt_bgr2gray.argtypes = [vp, i32, i32, i32, vp, i32]
        L.cvo_get_optimal_new_camera_matrix.argtypes = [vp, vp, i32, i32, i32, f64, i32, i32, vp, vp]
        L.cvo_init_undistort_rectify_map.argtypes = [vp, vp, i32, vp, i32, i32, vp, vp]
        L.cvo_remap_bilinear_u8.argtypes = [vp, i32, i32, i32, i32, vp, vp, i32, i32, vp, i32]
        L.cvo_undistort_u8.argtypes = [vp, i32, i32, i32, i32, vp, vp, i32, vp, vp, i32]
        L.cvo_dense_refine.argtypes = [vp, i32, i32, i32, vp, vp, i32, vp, vp, vp, i32, vp, vp, i32, vp, i32, f64, f64, vp]
        L.cvo_track_frame.argtypes = [vp, vp, i32, i32, i32, vp, vp, vp, vp, i32, vp, vp, vp, i32,
                                      vp, vp, i32, i32, i32, i32, f64, i32, i32, vp]
        _lib = L
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _f64(a, shape=None):
    a = np.ascontiguousarray(np.asarray(a, dtype=np.float64))
    return a.reshape(shape) if shape is not None else a


def _dist(dist):
    if dist is None:
        return None, 0
    d = _f64(dist).reshape(-1)
    return (d, d.size) if d.size else (None, 0)


# --------------------------------------------------------------------------- images
def pyrDown(img):
    img = np.ascontiguousarray(img, dtype=np.uint8)
    h, w = img.shape
    out = np.empty(((h + 1) // 2, (w + 1) // 2), np.uint8)
    rc = lib().cvo_pyr_down_u8(_p(img), w, h, img.strides[0], _p(out), out.strides[0])
    assert rc == 0
    return out


def scharr(img):
    """-> (h, w, 2) int16, [...,0]=dx, [...,1]=dy (calcSharrDeriv)."""
    img = np.ascontiguousarray(img, dtype=np.uint8)
    h, w = img.shape
    out = np.empty((h, w, 2), np.int16)
    rc = lib().cvo_scharr_deriv(_p(img), w, h, img.strides[0], _p(out), 2 * w)
    assert rc == 0
    return out


class Pyramid:
    """Padded optical-flow pyramid (buildOpticalFlowPyramid)."""

    def __init__(self, img, win=21, max_level=2):
        img = np.ascontiguousarray(img, dtype=np.uint8)
        h, w = img.shape
        self.h = lib().cvo_pyramid_build(_p(img), w, h, img.strides[0], win, win, max_level)
        if not self.h:
            raise ValueError("pyramid build failed")
        self.win, self.max_level = win, max_level

    @classmethod
    def _adopt(cls, handle, win, max_level):
        o = cls.__new__(cls)
        o.h, o.win, o.max_level = handle, win, max_level
        return o

    @property
    def levels(self):
        return lib().cvo_pyramid_levels(self.h)

    def level(self, i):
        w, h = C.c_int(), C.c_int()
        if lib().cvo_pyramid_level_size(self.h, i, C.byref(w), C.byref(h)):
            raise IndexError(i)
        out = np.empty((h.value, w.value), np.uint8)
        lib().cvo_pyramid_level_copy(self.h, i, _p(out), out.strides[0])
        return out

    def __del__(self):
        if getattr(self, "h", None):
            lib().cvo_pyramid_free(self.h)
            self.h = None


def calcOpticalFlowPyrLK(prevImg, nextImg, prevPts, nextPts=None, winSize=(21, 21), maxLevel=3,
                         criteria=(TERM_COUNT | TERM_EPS, 30, 0.01), flags=0, minEigThreshold=1e-4,
                         acc_mode=ACC_EXACT, nthreads=1):
    """cv2.calcOpticalFlowPyrLK signature; prevImg/nextImg may be Pyramid objects."""
    pp = np.ascontiguousarray(np.asarray(prevPts, np.float32).reshape(-1, 2))
    n = pp.shape[0]
    if nextPts is not None and (flags & OPTFLOW_USE_INITIAL_FLOW):
        nx = np.ascontiguousarray(np.asarray(nextPts, np.float32).reshape(-1, 2)).copy()
    else:
        nx = np.zeros((n, 2), np.float32)
    st = np.zeros(n, np.uint8)
    er = np.zeros(n, np.float32)
    tail = (winSize[0], winSize[1], maxLevel, criteria[0], int(criteria[1]), float(criteria[2]),
            flags, float(minEigThreshold), acc_mode, nthreads)
    if isinstance(prevImg, Pyramid):
        rc = lib().cvo_lk_on_pyramids(prevImg.h, nextImg.h, _p(pp), _p(nx), _p(st), _p(er), n, *tail)
    else:
        a = np.ascontiguousarray(prevImg, dtype=np.uint8)
        b = np.ascontiguousarray(nextImg, dtype=np.uint8)
        assert a.shape == b.shape and a.strides == b.strides
        rc = lib().cvo_calc_optical_flow_pyr_lk(_p(a), _p(b), a.shape[1], a.shape[0], a.strides[0],
                                                _p(pp), _p(nx), _p(st), _p(er), n, *tail)
    if rc:
        raise ValueError("calcOpticalFlowPyrLK oracle error %d" % rc)
    return nx.reshape(-1, 1, 2), st.reshape(-1, 1), er.reshape(-1, 1)


COLOR_BGR2GRAY = 6


def cvtColor(src, code):
    if code != COLOR_BGR2GRAY:
        raise ValueError("only COLOR_BGR2GRAY is restated")
    a = np.ascontiguousarray(src, dtype=np.uint8)
    h, w, _ = a.shape
    out = np.empty((h, w), np.uint8)
    rc = lib().cvo_cvt_bgr2gray(_p(a), w, h, a.strides[0], _p(out), out.strides[0])
    assert rc == 0
    return out


def getOptimalNewCameraMatrix(cameraMatrix, distCoeffs, imageSize, alpha, newImgSize=(0, 0)):
    """cv2.getOptimalNewCameraMatrix -> (newK (3,3) f64, roi (x, y, w, h))"""
    K = _f64(cameraMatrix).reshape(9); d, nd = _dist(distCoeffs)
    newK = np.empty(9); roi = np.zeros(4, np.int32)
    rc = lib().cvo_get_optimal_new_camera_matrix(_p(K), _p(d), nd, int(imageSize[0]), int(imageSize[1]), float(alpha),
                                                 int(newImgSize[0]), int(newImgSize[1]), _p(newK), _p(roi))
    if rc:
        raise ValueError("getOptimalNewCameraMatrix oracle error %d" % rc)
    return newK.reshape(3, 3), tuple(int(v) for v in roi)


def initUndistortRectifyMap(cameraMatrix, distCoeffs, newCameraMatrix, size):
    """CV_16SC2 maps: map1 (h, w, 2) int16, map2 (h, w) uint16; R = I"""
    K = _f64(cameraMatrix).reshape(9); d, nd = _dist(distCoeffs); nK = _f64(newCameraMatrix).reshape(9)
    w, h = int(size[0]), int(size[1])
    m1 = np.empty((h, w, 2), np.int16); m2 = np.empty((h, w), np.uint16)
    rc = lib().cvo_init_undistort_rectify_map(_p(K), _p(d), nd, _p(nK), w, h, _p(m1), _p(m2))
    if rc:
        raise ValueError("initUndistortRectifyMap oracle error %d" % rc)
    return m1, m2


def undistort(src, cameraMatrix, distCoeffs, dst=None, newCameraMatrix=None):
    """cv2.undistort for 8-bit images with 1..4 channels (INTER_LINEAR, BORDER_CONSTANT)"""
    a = np.ascontiguousarray(src, dtype=np.uint8)
    h, w = a.shape[:2]; cn = 1 if a.ndim == 2 else a.shape[2]
    K = _f64(cameraMatrix).reshape(9); d, nd = _dist(distCoeffs)
    nK = K if newCameraMatrix is None else _f64(newCameraMatrix).reshape(9)
    out = np.empty_like(a)
    rc = lib().cvo_undistort_u8(_p(a), w, h, a.strides[0], cn, _p(K), _p(d), nd, _p(nK), _p(out), out.strides[0])
    if rc:
        raise ValueError("undistort oracle error %d" % rc)
    return out


# --------------------------------------------------------------------------- geometry
def Rodrigues(src):
    """cv2.Rodrigues: (3,1)/(1,3)/(3,) -> (3,3) with 3x9 jacobian; (3,3) -> (3,1).
    Output depth follows input depth (f32 in -> f32 out), as OpenCV."""
    a = np.asarray(src)
    out_dtype = np.float32 if a.dtype == np.float32 else np.float64
    if a.size == 3:
        r = _f64(a).reshape(3)
        R = np.empty((3, 3)); J = np.empty((3, 9))
        lib().cvo_rodrigues_vec2mat(_p(r), _p(R), _p(J))
        return R.astype(out_dtype), J.astype(out_dtype)
    if a.shape == (3, 3):
        R = _f64(a)
        r = np.empty(3); J = np.zeros((9, 3))
        lib().cvo_rodrigues_mat2vec(_p(R), _p(r), _p(J))
        return r.reshape(3, 1).astype(out_dtype), J.astype(out_dtype)
    raise ValueError("Rodrigues: bad input shape %r" % (a.shape,))


def projectPoints(objectPoints, rvec, tvec, cameraMatrix, distCoeffs, jacobian=False):
    """cv2.projectPoints -> (imagePoints (N,1,2), jacobian (2N, 6) or None).
    The jacobian holds only the d/dr, d/dt columns (all the reference's path consumes)."""
    o = np.asarray(objectPoints)
    out_dtype = np.float32 if o.dtype == np.float32 else np.float64
    obj = _f64(o).reshape(-1, 3)
    n = obj.shape[0]
    r = _f64(rvec).reshape(3); t = _f64(tvec).reshape(3); K = _f64(cameraMatrix).reshape(9)
    d, nd = _dist(distCoeffs)
    img = np.empty((n, 2))
    dr = np.empty((n, 2, 3)) if jacobian else None
    dt = np.empty((n, 2, 3)) if jacobian else None
    rc = lib().cvo_project_points(_p(obj), n, _p(r), _p(t), _p(K), _p(d), nd, _p(img), _p(dr), _p(dt))
    if rc:
        raise ValueError("projectPoints oracle error %d" % rc)
    jac = np.concatenate([dr, dt], axis=2).reshape(2 * n, 6) if jacobian else None
    return img.reshape(n, 1, 2).astype(out_dtype), jac


def undistortPoints(src, cameraMatrix, distCoeffs):
    m = _f64(src).reshape(-1, 2)
    K = _f64(cameraMatrix).reshape(9)
    d, nd = _dist(distCoeffs)
    out = np.empty_like(m)
    lib().cvo_undistort_points(_p(m), m.shape[0], _p(K), _p(d), nd, _p(out))
    return out.reshape(-1, 1, 2)


def solvePnP(objectPoints, imagePoints, cameraMatrix, distCoeffs, rvec=None, tvec=None,
             useExtrinsicGuess=False, flags=SOLVEPNP_ITERATIVE, return_iters=False):
    """cv2.solvePnP(flags=SOLVEPNP_ITERATIVE) -> (ok, rvec (3,1) f64, tvec (3,1) f64)."""
    if flags != SOLVEPNP_ITERATIVE:
        raise ValueError("only SOLVEPNP_ITERATIVE is restated")
    obj = _f64(objectPoints).reshape(-1, 3)
    img = _f64(imagePoints).reshape(-1, 2)
    n = obj.shape[0]
    if img.shape[0] != n:
        raise ValueError("solvePnP: point count mismatch")
    K = _f64(cameraMatrix).reshape(9)
    d, nd = _dist(distCoeffs)
    r = np.zeros(3); t = np.zeros(3)
    if useExtrinsicGuess:
        r[:] = _f64(rvec).reshape(3); t[:] = _f64(tvec).reshape(3)
    it = C.c_int(0)
    rc = lib().cvo_solve_pnp_iterative(_p(obj), _p(img), n, _p(K), _p(d), nd, _p(r), _p(t),
                                       1 if useExtrinsicGuess else 0, C.byref(it))
    if rc:
        raise ValueError("solvePnP oracle error %d" % rc)
    if useExtrinsicGuess and isinstance(rvec, np.ndarray) and isinstance(tvec, np.ndarray) \
            and rvec.dtype in (np.float32, np.float64) and tvec.dtype in (np.float32, np.float64):
        # cv2 writes the result into the guess arrays, in their dtype, and returns them
        rvec.reshape(-1)[:] = r
        tvec.reshape(-1)[:] = t
        out = (True, rvec, tvec)
    else:
        out = (True, r.reshape(3, 1).copy(), t.reshape(3, 1).copy())
    return out + (it.value,) if return_iters else out


def pnp_init(objectPoints, imagePoints, cameraMatrix, distCoeffs):
    obj = _f64(objectPoints).reshape(-1, 3); img = _f64(imagePoints).reshape(-1, 2)
    K = _f64(cameraMatrix).reshape(9); d, nd = _dist(distCoeffs)
    r = np.zeros(3); t = np.zeros(3)
    rc = lib().cvo_pnp_init(_p(obj), _p(img), obj.shape[0], _p(K), _p(d), nd, _p(r), _p(t))
    if rc:
        raise ValueError("pnp_init oracle error %d" % rc)
    return r.reshape(3, 1), t.reshape(3, 1)


def mean_reproj_error(objectPoints, imagePoints, rvec, tvec, cameraMatrix, distCoeffs):
    obj = _f64(objectPoints).reshape(-1, 3); img = _f64(imagePoints).reshape(-1, 2)
    K = _f64(cameraMatrix).reshape(9); d, nd = _dist(distCoeffs)
    r = _f64(rvec).reshape(3); t = _f64(tvec).reshape(3)
    return lib().cvo_mean_reproj_error(_p(obj), _p(img), obj.shape[0], _p(r), _p(t), _p(K), _p(d), nd)


def dense_refine(img, model_xyz, model_t, obj, img_pts, mask, K, dist, rvec, tvec, iters=5, photo_weight=0.01, mu=1e-3):
    """Dense photometric + geometric Gauss-Newton refinement (semantics: oracle/cv_dense.c).
    -> (rvec (3,), tvec (3,), stats dict)"""
    a = np.ascontiguousarray(img, dtype=np.uint8); h, w = a.shape
    mx = np.ascontiguousarray(np.asarray(model_xyz, np.float32).reshape(-1, 3)); mt = np.ascontiguousarray(np.asarray(model_t, np.float32).reshape(-1))
    M = mx.shape[0]
    if obj is None:
        o = ip = mk = None; N = 0
    else:
        o = np.ascontiguousarray(np.asarray(obj, np.float32).reshape(-1, 3)); ip = np.ascontiguousarray(np.asarray(img_pts, np.float32).reshape(-1, 2))
        N = o.shape[0]; mk = None if mask is None else np.ascontiguousarray(np.asarray(mask, np.uint8).reshape(-1))
    Kc = _f64(K).reshape(9); d, nd = _dist(dist)
    pose = np.concatenate([_f64(rvec).reshape(3), _f64(tvec).reshape(3)]).copy(); stats = np.zeros(8)
    rc = lib().cvo_dense_refine(_p(a), w, h, a.strides[0], _p(mx), _p(mt), M, _p(o), _p(ip), _p(mk), N, _p(Kc), _p(d), nd,
                                _p(pose), int(iters), float(photo_weight), float(mu), _p(stats))
    if rc:
        raise ValueError("dense_refine oracle error %d" % rc)
    return pose[:3].copy(), pose[3:].copy(), dict(photo_rms=stats[0], geo_rms=stats[1], valid=int(stats[2]), iters=int(stats[3]), used=int(stats[4]))


def svd(A):
    A = _f64(A); m, n = A.shape
    w = np.empty(n); u = np.empty((m, n)); vt = np.empty((n, n))
    rc = lib().cvo_svd(_p(A), m, n, _p(w), _p(u), _p(vt))
    assert rc == 0
    return w, u, vt


def solve_svd(A, b):
    A = _f64(A); b = _f64(b).reshape(-1); x = np.empty_like(b)
    rc = lib().cvo_solve_svd(_p(A), _p(b), b.size, _p(x))
    assert rc == 0
    return x


def findHomography(src, dst, refine=True):
    """cv2.findHomography(src, dst, method=0)[0]: (N,2) -> (N,2), 3x3 float64"""
    a = _f64(np.asarray(src, np.float64).reshape(-1, 2)); b = _f64(np.asarray(dst, np.float64).reshape(-1, 2))
    Hm = np.zeros(9)
    L = lib()
    L.cvo_find_homography.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    rc = L.cvo_find_homography(_p(a), _p(b), a.shape[0], 1 if refine else 0, _p(Hm))
    if rc:
        raise ValueError("findHomography failed (%d)" % rc)
    return Hm.reshape(3, 3)


def track_frame(prev_pyr, next_img, prev_pts, obj, K, dist, rvec, tvec, use_guess=True, win=21,
                max_level=2, max_count=30, eps=0.01, acc_mode=ACC_EXACT, nthreads=1):
    """Whole CPU frame step: pyramid(next) + LK + solvePnP(guess).  Returns
    (next_pyr, next_pts (N,2) f32, status (N,), err (N,), n_used, rvec(3,), tvec(3,))."""
    img = np.ascontiguousarray(next_img, dtype=np.uint8)
    h, w = img.shape
    pp = np.ascontiguousarray(np.asarray(prev_pts, np.float32).reshape(-1, 2))
    n = pp.shape[0]
    nx = np.zeros((n, 2), np.float32); st = np.zeros(n, np.uint8); er = np.zeros(n, np.float32)
    o = _f64(obj).reshape(-1, 3); Kc = _f64(K).reshape(9); d, nd = _dist(dist)
    r = _f64(rvec).reshape(3).copy(); t = _f64(tvec).reshape(3).copy()
    out = C.c_void_p()
    cnt = lib().cvo_track_frame(prev_pyr.h, _p(img), w, h, img.strides[0], _p(pp), _p(nx), _p(st), _p(er), n,
                                _p(o), _p(Kc), _p(d), nd, _p(r), _p(t), 1 if use_guess else 0,
                                win, max_level, max_count, float(eps), acc_mode, nthreads, C.byref(out))
    if cnt < 0:
        raise ValueError("track_frame oracle error %d" % cnt)
    return Pyramid._adopt(out.value, win, max_level), nx, st, er, cnt, r, t
