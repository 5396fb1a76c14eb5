"""cv2-shaped entry points backed by the gfx950 HIP library.

The reference's hot path is Python calling four cv2 functions; this
module offers the same four with the same argument meaning, return shapes and error
behaviour, so `import accurate_aprilgroup_tracking_amd.cv_hip as cv` drops in for the
calls at detect_pose.py:509-526 (solvePnP), transform_helper.py:106-111 (projectPoints),
transform_helper.py:87 (Rodrigues) and the north-star calcOpticalFlowPyrLK.

PyTorch-ROCm is used only as the device-buffer allocator / stream provider
(tensor.data_ptr(), torch.cuda.current_stream().cuda_stream).  There is no CPU path:
without the HIP library or a GPU these functions raise.
"""
import ctypes as C
import threading
import numpy as np
import torch

from . import hiplib as H
from .host_math import Rodrigues  # noqa: F401  (host-side 3x3 algebra, as in the reference)

SOLVEPNP_ITERATIVE = 0
COLOR_BGR2GRAY = 6
OPTFLOW_USE_INITIAL_FLOW = H.LK_USE_INITIAL_FLOW
OPTFLOW_LK_GET_MIN_EIGENVALS = H.LK_GET_MIN_EIGENVALS
TERM_CRITERIA_COUNT, TERM_CRITERIA_EPS = H.TERM_COUNT, H.TERM_EPS


class error(ValueError):
    """stands in for cv2.error (argument / shape violations)"""


_gpu_seen = False


def _require_gpu():
    global _gpu_seen
    if _gpu_seen:
        return
    if not torch.cuda.is_available():
        raise RuntimeError("accurate_aprilgroup_tracking_amd needs a ROCm GPU (no CPU fallback)")
    _gpu_seen = True


def _addr(a):
    """address of a numpy array's first element as a ctypes pointer argument (ndarray.ctypes.data_as costs twice as much, and the
    synchronous calls are short enough for that to show: bench.py per_call_latency_us)"""
    return C.c_void_p(a.__array_interface__["data"][0])


def _raw_stream(device):
    """the HIP stream torch launches on (torch.cuda.current_stream(device).cuda_stream without building the Stream object)"""
    try:
        return torch._C._cuda_getCurrentRawStream(device)
    except AttributeError:                      # (private torch API: fall back to the public one)
        return torch.cuda.current_stream(device).cuda_stream


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def _host_f64(a, n=None):
    if a is None:
        return None, 0
    if type(a) is np.ndarray and a.dtype == np.float64 and a.flags.c_contiguous:
        return a, a.size                        # (only the address is used: no reshape needed)
    a = np.ascontiguousarray(np.asarray(a, dtype=np.float64).reshape(-1))
    return a, a.size


class Context:
    """One agt_ctx: pyramid slots + tracker state for B streams of W x H frames."""

    def __init__(self, width, height, max_level=2, win=21, max_points=64, max_streams=1, device=None):
        _require_gpu()
        self.L = H.lib()
        self.device = torch.cuda.current_device() if device is None else device
        cfg = H.Config(self.device, width, height, max_level, win, max_points, max_streams)
        self.h = C.c_void_p()
        stream = torch.cuda.current_stream(self.device).cuda_stream
        H.check(self.L.agt_create(C.byref(cfg), C.c_void_p(stream), C.byref(self.h)), "agt_create")
        self.width, self.height, self.max_level, self.win = width, height, max_level, win
        self.max_points, self.max_streams = max_points, max_streams
        self._keep = {}          # frames aliased by pyramid level 0 must stay alive
        # a context is not re-entrant (include/agt_hip.h): the cv2-shaped functions below, which share cached contexts
        # between callers, hold this lock from staging their inputs to reading their outputs (cv2's own functions are
        # re-entrant; two threads calling cv_hip.solvePnP must not see each other's staging buffer)
        self.lock = threading.RLock()
        self._stream_set = None

    def close(self):
        if getattr(self, "h", None) and self.h.value:
            self.L.agt_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def use_current_stream(self):
        stream = _raw_stream(self.device)
        if stream != self._stream_set:          # (one foreign call less per cv2-shaped call while the stream stays the same)
            H.check(self.L.agt_set_stream(self.h, C.c_void_p(stream)), "agt_set_stream")
            self._stream_set = stream

    def synchronize(self):
        H.check(self.L.agt_synchronize(self.h), "agt_synchronize")

    @property
    def eff_max_level(self):
        return self.L.agt_pyramid_max_level(self.h)

    # ---- images
    def pyr_down(self, src):
        """src: cuda uint8 [B,H,W] (contiguous rows, W % 4 == 0) -> [B,(H+1)//2,(W+1)//2 padded to 4]"""
        assert src.dtype == torch.uint8 and src.is_cuda and src.dim() == 3
        B, h, w = src.shape
        dw, dh = (w + 1) // 2, (h + 1) // 2
        dp = (dw + 3) & ~3
        dst = torch.empty((B, dh, dp), dtype=torch.uint8, device=src.device)
        H.check(self.L.agt_pyr_down_u8(self.h, _ptr(src), w, h, src.stride(1), src.stride(0),
                                       _ptr(dst), dst.stride(1), dst.stride(0), B), "agt_pyr_down_u8")
        return dst[:, :, :dw]

    def pyramid_build(self, slot, frames):
        """frames: cuda uint8 [B,H,W]; level 0 aliases it (kept alive by this object)."""
        assert frames.dtype == torch.uint8 and frames.is_cuda and frames.dim() == 3
        B, h, w = frames.shape
        if (h, w) != (self.height, self.width):
            raise error("frame size %dx%d does not match the context (%dx%d)" % (w, h, self.width, self.height))
        H.check(self.L.agt_pyramid_build(self.h, slot, _ptr(frames), frames.stride(1), frames.stride(0), B),
                "agt_pyramid_build")
        self._keep[slot] = frames

    def pyramid_build_pair(self, prev, nxt):
        """both slots of a frame pair in one call (agt_pyramid_build_pair): slot 0 <- prev, slot 1 <- nxt, cuda uint8 [B,H,W] of one
        geometry (what cv2.calcOpticalFlowPyrLK(prev, next, ...) builds internally)"""
        for f in (prev, nxt):
            assert f.dtype == torch.uint8 and f.is_cuda and f.dim() == 3
        B, h, w = prev.shape
        if (h, w) != (self.height, self.width) or tuple(nxt.shape) != tuple(prev.shape) or nxt.stride() != prev.stride():
            raise error("frame pair: %s / %s do not match the context (%dx%d) or each other" % (tuple(prev.shape), tuple(nxt.shape), self.width, self.height))
        H.check(self.L.agt_pyramid_build_pair(self.h, _ptr(prev), _ptr(nxt), prev.stride(1), prev.stride(0), B), "agt_pyramid_build_pair")
        self._keep[0] = prev; self._keep[1] = nxt

    def pyramid_level(self, slot, level):
        """copy of one level of a built slot: numpy uint8 [B, h_l, w_l] (tests)"""
        p, w, h, pitch, bs = C.c_void_p(), C.c_int(), C.c_int(), C.c_size_t(), C.c_size_t()
        H.check(self.L.agt_pyramid_level(self.h, slot, level, C.byref(p), C.byref(w), C.byref(h), C.byref(pitch), C.byref(bs)),
                "agt_pyramid_level")
        B = self._keep[slot].shape[0]
        raw = np.empty(((B - 1) * bs.value + h.value * pitch.value,), np.uint8)
        H.check(self.L.agt_download(self.h, raw.ctypes.data_as(C.c_void_p), p, raw.size), "agt_download")     # (waits for the stream)
        return np.stack([np.lib.stride_tricks.as_strided(raw[b * bs.value:], (h.value, w.value), (pitch.value, 1)).copy() for b in range(B)])

    # ---- frame pre-processing (detect_pose.py:147-183, :602)
    def undistort_init(self, K, dist, newK, width, height):
        """Build the CV_16SC2 undistortion maps for this camera on the device (once per camera)."""
        Kh, _ = _host_f64(K); dh, nd = _host_f64(dist)
        nk = None if newK is None else _host_f64(newK)[0]
        H.check(self.L.agt_undistort_init(self.h, Kh.ctypes.data_as(C.c_void_p), dh.ctypes.data_as(C.c_void_p) if nd else None, nd,
                                          nk.ctypes.data_as(C.c_void_p) if nk is not None else None, width, height),
                "agt_undistort_init")
        self._map_size = (width, height)

    def undistort_maps(self):
        """(map1 [h,w,2] int16, map2 [h,w] uint16) copied to the host (tests)"""
        p1, p2, w, h = C.c_void_p(), C.c_void_p(), C.c_int(), C.c_int()
        H.check(self.L.agt_undistort_maps(self.h, C.byref(p1), C.byref(p2), C.byref(w), C.byref(h)), "agt_undistort_maps")
        m1 = np.empty((h.value, w.value, 2), np.int16)
        m2 = np.empty((h.value, w.value), np.uint16)
        H.check(self.L.agt_download(self.h, m1.ctypes.data_as(C.c_void_p), p1, m1.nbytes), "agt_download")
        H.check(self.L.agt_download(self.h, m2.ctypes.data_as(C.c_void_p), p2, m2.nbytes), "agt_download")
        return m1, m2

    def undistort_bgr(self, frames):
        """cv.undistort on cuda uint8 [B,H,W,3] -> same shape"""
        assert frames.dtype == torch.uint8 and frames.is_cuda and frames.dim() == 4 and frames.shape[3] == 3 and frames.is_contiguous()
        out = torch.empty_like(frames)
        H.check(self.L.agt_undistort_bgr(self.h, _ptr(frames), frames.stride(1), frames.stride(0), _ptr(out), out.stride(1),
                                         out.stride(0), frames.shape[0]), "agt_undistort_bgr")
        return out

    def preprocess_bgr(self, frames, roi=None, undistort=True, out=None):
        """fused undistort -> BGR2GRAY -> crop: cuda uint8 [B,H,W,3] -> [B,roi_h,roi_w] (row pitch padded to 16)"""
        assert frames.dtype == torch.uint8 and frames.is_cuda and frames.dim() == 4 and frames.shape[3] == 3 and frames.is_contiguous()
        B, h, w, _ = frames.shape
        rx, ry, rw, rh = (0, 0, w, h) if roi is None else roi
        if out is None:
            pitch = (rw + 15) & ~15
            out = torch.empty((B, rh, pitch), dtype=torch.uint8, device=frames.device)[:, :, :rw]
        H.check(self.L.agt_preprocess_bgr(self.h, _ptr(frames), frames.stride(1), frames.stride(0), w, h, B, int(bool(undistort)),
                                          rx, ry, rw, rh, _ptr(out), out.stride(1), out.stride(0)), "agt_preprocess_bgr")
        return out

    # ---- dense photometric + geometric refinement (BASELINE config 5; semantics: oracle/cv_dense.c)
    def dense_refine(self, frames, model_xyz, model_t, pose, K, dist, obj=None, img_pts=None, mask=None, iters=5, photo_weight=0.01):
        """frames cuda u8 [B,H,W]; model_xyz cuda f32 [M,3]; model_t cuda f32 [M]; pose cuda f64 [B,6] (in/out);
        obj cuda f32 [N,3] + img_pts cuda f32 [B,N,2] (+ mask u8 [B,N]) add the corner term.  Returns (pose, stats [B,8])."""
        assert frames.dtype == torch.uint8 and frames.is_cuda and frames.dim() == 3
        assert model_xyz.dtype == torch.float32 and model_xyz.is_contiguous() and model_t.dtype == torch.float32 and model_t.is_contiguous()
        assert pose.dtype == torch.float64 and pose.is_contiguous()
        B, h, w = frames.shape
        M = model_xyz.shape[0]
        N = 0 if obj is None else obj.shape[0]
        if N:
            assert obj.dtype == torch.float32 and obj.is_contiguous() and img_pts.dtype == torch.float32 and img_pts.is_contiguous()
            assert img_pts.shape == (B, N, 2)
        stats = torch.zeros((B, 8), dtype=torch.float64, device=frames.device)
        Kh, _ = _host_f64(K); dh, nd = _host_f64(dist)
        H.check(self.L.agt_dense_refine(self.h, _ptr(frames), frames.stride(1), frames.stride(0), w, h, _ptr(model_xyz), _ptr(model_t), M,
                                        _ptr(obj), _ptr(img_pts), _ptr(mask), N, Kh.ctypes.data_as(C.c_void_p),
                                        dh.ctypes.data_as(C.c_void_p) if nd else None, nd, _ptr(pose), B, int(iters),
                                        float(photo_weight), _ptr(stats)), "agt_dense_refine")
        return pose, stats

    # ---- LK
    def lk_track(self, prev_slot, next_slot, prev_pts, next_pts=None, criteria=(3, 30, 0.01), flags=0,
                 min_eig_threshold=1e-4, want_err=True):
        """prev_pts: cuda f32 [B,n,2].  Returns (next_pts [B,n,2] f32, status [B,n] u8, err [B,n] f32|None)."""
        assert prev_pts.dtype == torch.float32 and prev_pts.is_cuda and prev_pts.is_contiguous()
        B, n, _ = prev_pts.shape
        if next_pts is None:
            next_pts = torch.zeros_like(prev_pts)
        else:
            assert next_pts.dtype == torch.float32 and next_pts.is_contiguous() and next_pts.shape == prev_pts.shape
        status = torch.empty((B, n), dtype=torch.uint8, device=prev_pts.device)
        err = torch.empty((B, n), dtype=torch.float32, device=prev_pts.device) if want_err else None
        H.check(self.L.agt_lk_track(self.h, prev_slot, next_slot, _ptr(prev_pts), _ptr(next_pts), _ptr(status),
                                    _ptr(err), n, B, int(criteria[0]), int(criteria[1]), float(criteria[2]),
                                    int(flags), float(min_eig_threshold)), "agt_lk_track")
        return next_pts, status, err

    # ---- PnP
    def solve_pnp(self, obj, img, K, dist, pose=None, use_guess=False, mask=None):
        """obj: cuda [n,3] (shared) or [B,n,3]; img: cuda [B,n,2]; same float dtype.
        pose: cuda f64 [B,6] (rvec|tvec) in/out.  Returns (pose, info [B,4] i32, err [B] f64)."""
        assert obj.is_cuda and img.is_cuda and obj.dtype == img.dtype and obj.dtype in (torch.float32, torch.float64)
        assert obj.is_contiguous() and img.is_contiguous()
        B, n, _ = img.shape
        shared = obj.dim() == 2
        if (obj.shape[0] if shared else obj.shape[1]) != n:
            raise error("solvePnP: object/image point counts differ")
        dtype = H.F32 if obj.dtype == torch.float32 else H.F64
        if pose is None:
            pose = torch.zeros((B, 6), dtype=torch.float64, device=img.device)
        assert pose.dtype == torch.float64 and pose.is_contiguous() and pose.shape == (B, 6)
        info = torch.zeros((B, 4), dtype=torch.int32, device=img.device)
        err = torch.zeros((B,), dtype=torch.float64, device=img.device)
        Kh, _ = _host_f64(K)
        dh, nd = _host_f64(dist)
        if mask is not None:
            assert mask.dtype == torch.uint8 and mask.is_contiguous() and mask.shape == (B, n)
        H.check(self.L.agt_solve_pnp(self.h, _ptr(obj), 0 if shared else n * 3, _ptr(img), dtype, _ptr(mask), n, B,
                                     Kh.ctypes.data_as(C.c_void_p), dh.ctypes.data_as(C.c_void_p) if nd else None, nd,
                                     _ptr(pose), 1 if use_guess else 0, _ptr(info), _ptr(err)), "agt_solve_pnp")
        return pose, info, err

    def project_points(self, obj, pose, K, dist, jacobian=False):
        """obj cuda [n,3] or [B,n,3]; pose cuda f64 [B,6] -> (img [B,n,2] obj.dtype, jac [B,2n,6] f64|None)"""
        assert obj.is_cuda and obj.is_contiguous() and obj.dtype in (torch.float32, torch.float64)
        assert pose.dtype == torch.float64 and pose.is_contiguous()
        B = pose.shape[0]
        shared = obj.dim() == 2
        n = obj.shape[0] if shared else obj.shape[1]
        dtype = H.F32 if obj.dtype == torch.float32 else H.F64
        out = torch.empty((B, n, 2), dtype=obj.dtype, device=obj.device)
        jac = torch.empty((B, 2 * n, 6), dtype=torch.float64, device=obj.device) if jacobian else None
        Kh, _ = _host_f64(K)
        dh, nd = _host_f64(dist)
        H.check(self.L.agt_project_points(self.h, _ptr(obj), 0 if shared else n * 3, dtype, n, B, _ptr(pose),
                                          Kh.ctypes.data_as(C.c_void_p), dh.ctypes.data_as(C.c_void_p) if nd else None, nd,
                                          _ptr(out), _ptr(jac)), "agt_project_points")
        return out, jac


# ------------------------------------------------------------------------------------
# numpy-in / numpy-out functions with cv2's signatures (one synchronous call per frame,
# exactly how the reference uses cv2, detect_pose.py:669-681)
_ctx_cache = {}
_ctx_cache_lock = threading.Lock()


def _context(width, height, max_level, win, n):
    """cached context for one geometry; the caller takes ctx.lock around its use (and calls ctx.use_current_stream() inside)"""
    key = (width, height, max_level, win, torch.cuda.current_device())
    with _ctx_cache_lock:
        ctx = _ctx_cache.get(key)
        if ctx is None or ctx.max_points < n:
            ctx = Context(width, height, max_level=max_level, win=win, max_points=max(64, min(256, n)), max_streams=1)
            _ctx_cache[key] = ctx
    return ctx


def _geom_context(n):
    return _context(64, 64, 0, 21, n)


def getOptimalNewCameraMatrix(cameraMatrix, distCoeffs, imageSize, alpha, newImgSize=(0, 0)):
    """cv2.getOptimalNewCameraMatrix -> (newCameraMatrix (3,3) f64, roi (x, y, w, h)).  Host arithmetic
    (81 grid points), done by the C-ABI library like every other cv2 replacement."""
    Kh, _ = _host_f64(cameraMatrix)
    dh, nd = _host_f64(distCoeffs)
    newK = np.empty(9, np.float64); roi = np.zeros(4, np.int32)
    rc = H.lib().agt_get_optimal_new_camera_matrix(Kh.ctypes.data_as(C.c_void_p), dh.ctypes.data_as(C.c_void_p) if nd else None, nd,
                                                   int(imageSize[0]), int(imageSize[1]), float(alpha), int(newImgSize[0]),
                                                   int(newImgSize[1]), newK.ctypes.data_as(C.c_void_p), roi.ctypes.data_as(C.c_void_p))
    if rc:
        raise error("getOptimalNewCameraMatrix: AGT error %d" % rc)
    return newK.reshape(3, 3), tuple(int(v) for v in roi)


_undist_cache = {}


def _undistort_context(w, h, K, dist, newK):
    key = (w, h, np.asarray(K, np.float64).tobytes(), None if dist is None else np.asarray(dist, np.float64).tobytes(),
           None if newK is None else np.asarray(newK, np.float64).tobytes(), torch.cuda.current_device())
    ctx = _undist_cache.get(key)
    if ctx is None:
        ctx = Context(64, 64, max_level=0)
        ctx.undistort_init(K, dist, newK, w, h)
        _undist_cache.clear()
        _undist_cache[key] = ctx
    return ctx


def undistort(src, cameraMatrix, distCoeffs, dst=None, newCameraMatrix=None):
    """cv2.undistort for 8-bit BGR or gray frames (INTER_LINEAR, BORDER_CONSTANT): (H,W,3) u8 -> (H,W,3) u8, (H,W) -> (H,W).
    The remap acts per channel, so a gray frame goes through the 3-channel kernel as (g, g, g)."""
    _require_gpu()
    a = np.asarray(src)
    if a.dtype == np.uint8 and a.ndim == 2:
        return undistort(np.repeat(a[:, :, None], 3, axis=2), cameraMatrix, distCoeffs, None, newCameraMatrix)[:, :, 0].copy()
    if a.dtype != np.uint8 or a.ndim != 3 or a.shape[2] != 3:
        raise error("undistort: an 8-bit 1- or 3-channel frame is expected")
    h, w = a.shape[:2]
    with _ctx_cache_lock:
        ctx = _undistort_context(w, h, cameraMatrix, distCoeffs, newCameraMatrix)
    with ctx.lock:
        ctx.use_current_stream()
        out = ctx.undistort_bgr(torch.from_numpy(np.ascontiguousarray(a)).cuda().unsqueeze(0))
        return out[0].cpu().numpy()


def cvtColor(src, code):
    """cv2.cvtColor(frame, COLOR_BGR2GRAY) for 8-bit frames"""
    _require_gpu()
    if code != COLOR_BGR2GRAY:
        raise error("cvtColor: only COLOR_BGR2GRAY is built")
    a = np.asarray(src)
    if a.dtype != np.uint8 or a.ndim != 3 or a.shape[2] != 3:
        raise error("cvtColor: an 8-bit 3-channel frame is expected")
    ctx = _geom_context(1)
    with ctx.lock:
        ctx.use_current_stream()
        out = ctx.preprocess_bgr(torch.from_numpy(np.ascontiguousarray(a)).cuda().unsqueeze(0), None, undistort=False)
        return out[0].cpu().numpy()


def calcOpticalFlowPyrLK(prevImg, nextImg, prevPts, nextPts=None, winSize=(21, 21), maxLevel=3,
                         criteria=(TERM_CRITERIA_COUNT | TERM_CRITERIA_EPS, 30, 0.01), flags=0, minEigThreshold=1e-4):
    """cv2.calcOpticalFlowPyrLK for 8-bit single-channel images -> (nextPts (N,1,2) f32, status (N,1) u8, err (N,1) f32)."""
    _require_gpu()
    a = np.asarray(prevImg); b = np.asarray(nextImg)
    if a.dtype != np.uint8 or b.dtype != np.uint8 or a.ndim != 2 or a.shape != b.shape:
        raise error("calcOpticalFlowPyrLK: prevImg/nextImg must be equal-size 8-bit single-channel images")
    ww, wh = int(winSize[0]), int(winSize[1])
    if not (3 <= ww <= 63 and 3 <= wh <= 63):
        raise error("calcOpticalFlowPyrLK: winSize must be within 3 .. 63 in both dimensions")
    win = ww if ww == wh else (ww | (wh << 8))          # include/agt_hip.h AGT_WIN_RECT
    if maxLevel >= H.MAX_LEVELS:
        raise error("calcOpticalFlowPyrLK: maxLevel too large")
    pts = np.ascontiguousarray(np.asarray(prevPts, dtype=np.float32).reshape(-1, 2))
    n = pts.shape[0]
    h, w = a.shape
    if n == 0:
        return np.zeros((0, 1, 2), np.float32), np.zeros((0, 1), np.uint8), np.zeros((0, 1), np.float32)
    try:
        ctx = _context(w, h, maxLevel, win, n)
    except H.AgtError as e:
        raise error(str(e))
    with ctx.lock:
        ctx.use_current_stream()
        wp = (w + 3) & ~3
        dev = torch.device("cuda", ctx.device)
        fa = torch.zeros((1, h, wp), dtype=torch.uint8, device=dev); fa[0, :, :w] = torch.from_numpy(a).to(dev)
        fb = torch.zeros((1, h, wp), dtype=torch.uint8, device=dev); fb[0, :, :w] = torch.from_numpy(b).to(dev)
        ctx.pyramid_build_pair(fa[:, :, :w], fb[:, :, :w])      # views keep the padded pitch (multiple of 4); both pyramids, one call
        pp = torch.from_numpy(pts).to(dev).reshape(1, n, 2).contiguous()
        nx = None
        if nextPts is not None and (flags & OPTFLOW_USE_INITIAL_FLOW):
            nx = torch.from_numpy(np.ascontiguousarray(np.asarray(nextPts, np.float32).reshape(1, n, 2))).to(dev)
        nx, st, er = ctx.lk_track(0, 1, pp, nx, criteria=criteria, flags=flags, min_eig_threshold=minEigThreshold)
        return (nx.cpu().numpy().reshape(n, 1, 2), st.cpu().numpy().reshape(n, 1), er.cpu().numpy().reshape(n, 1))


def solvePnP(objectPoints, imagePoints, cameraMatrix, distCoeffs, rvec=None, tvec=None,
             useExtrinsicGuess=False, flags=SOLVEPNP_ITERATIVE):
    """cv2.solvePnP(flags=SOLVEPNP_ITERATIVE) -> (ok, rvec (3,1), tvec (3,1)).

    Without a guess the outputs are fresh float64 arrays.  With useExtrinsicGuess the result
    is written INTO the supplied rvec/tvec ndarrays, in their own dtype, and those same
    objects are returned -- cv2's observable behaviour, which the reference works around
    with a deepcopy (detect_pose.py:487-490) and which makes a float32 guess tvec
    (transform_helper.py:158-159) yield a float32 pose tvec."""
    _require_gpu()
    if flags != SOLVEPNP_ITERATIVE:
        raise error("solvePnP: only SOLVEPNP_ITERATIVE is built")
    obj = np.asarray(objectPoints); img = np.asarray(imagePoints)
    obj = obj.reshape(-1, 3); img = img.reshape(-1, 2)
    n = obj.shape[0]
    if img.shape[0] != n or not (n >= 4 or (n == 3 and useExtrinsicGuess)) or n > 256:
        raise error("solvePnP: need 4 <= N <= 256 matching object/image points")
    dt = np.float32 if (obj.dtype == np.float32 and img.dtype == np.float32) else np.float64
    ctx = _geom_context(n)
    if useExtrinsicGuess and (rvec is None or tvec is None or np.size(rvec) != 3 or np.size(tvec) != 3):
        raise error("solvePnP: useExtrinsicGuess needs 3-element rvec and tvec")
    with ctx.lock:
        ctx.use_current_stream()
        # ONE foreign call, host arrays in and out (agt_solve_pnp_host: no copy is enqueued -- the kernel reads a host-mapped staging
        # area and the call polls for its result; rounds 1-4: upload + launch + download + stream wait, 44 us per call)
        objc = np.ascontiguousarray(obj, dtype=dt); imgc = np.ascontiguousarray(img, dtype=dt)
        p = np.zeros(6, np.float64)
        if useExtrinsicGuess:
            p[:3] = np.asarray(rvec, np.float64).reshape(3); p[3:] = np.asarray(tvec, np.float64).reshape(3)
        inf = np.zeros(4, np.int32)
        Kh, _ = _host_f64(cameraMatrix); dh, nd = _host_f64(distCoeffs)
        rc = ctx.L.agt_solve_pnp_host(ctx.h, _addr(objc), _addr(imgc), H.F32 if dt == np.float32 else H.F64, n,
                                      _addr(Kh), _addr(dh) if nd else None, nd, _addr(p), 1 if useExtrinsicGuess else 0, _addr(inf), None)
        if rc:
            raise error(str(H.AgtError(rc, "agt_solve_pnp_host")))
        if not inf[H.INFO_OK]:
            raise error("solvePnP: not enough usable points (non-planar sets need 6 without a guess)")
        if useExtrinsicGuess and isinstance(rvec, np.ndarray) and isinstance(tvec, np.ndarray) \
                and rvec.dtype in (np.float32, np.float64) and tvec.dtype in (np.float32, np.float64):
            rvec.reshape(-1)[:] = p[:3]
            tvec.reshape(-1)[:] = p[3:]
            return True, rvec, tvec
        return True, p[:3].reshape(3, 1).copy(), p[3:].reshape(3, 1).copy()


def projectPoints(objectPoints, rvec, tvec, cameraMatrix, distCoeffs, jacobian=False):
    """cv2.projectPoints -> (imagePoints (N,1,2) in objectPoints' depth, jacobian (2N,6) f64 | None)."""
    _require_gpu()
    obj = np.asarray(objectPoints)
    dt = np.float32 if obj.dtype == np.float32 else np.float64
    obj = np.ascontiguousarray(obj.reshape(-1, 3), dtype=dt)
    n = obj.shape[0]
    ctx = _geom_context(min(n, 256))
    with ctx.lock:
        ctx.use_current_stream()
        if n <= 256:
            # ONE foreign call, host arrays in and out (agt_project_points_host; rounds 1-4: upload + launch + download + wait, 28 us)
            g = np.empty(6, np.float64)
            g[:3] = np.asarray(rvec, np.float64).reshape(3); g[3:] = np.asarray(tvec, np.float64).reshape(3)
            pts = np.empty((n, 1, 2), dt)
            jac = np.empty((2 * n, 6), np.float64) if jacobian else None
            Kh, _ = _host_f64(cameraMatrix); dh, nd = _host_f64(distCoeffs)
            rc = ctx.L.agt_project_points_host(ctx.h, _addr(obj), H.F32 if dt == np.float32 else H.F64, n, _addr(g), _addr(Kh),
                                               _addr(dh) if nd else None, nd, _addr(pts), _addr(jac) if jacobian else None)
            if rc:
                raise error(str(H.AgtError(rc, "agt_project_points_host")))
            return pts, jac
        dev = torch.device("cuda", ctx.device)
        g = np.concatenate([np.asarray(rvec, np.float64).reshape(3), np.asarray(tvec, np.float64).reshape(3)])
        pose = torch.from_numpy(g).to(dev).reshape(1, 6).contiguous()
        try:
            out, jac = ctx.project_points(torch.from_numpy(obj).to(dev), pose, cameraMatrix, distCoeffs, jacobian)
        except H.AgtError as e:
            raise error(str(e))
        return out.cpu().numpy().reshape(n, 1, 2), (jac.cpu().numpy().reshape(2 * n, 6) if jacobian else None)
