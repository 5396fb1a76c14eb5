"""Frame pre-processing (detect_pose.py:147-183 undistort_frame, :602 cvtColor).
CPU: oracle self-checks + the host-only getOptimalNewCameraMatrix of the C ABI vs the oracle.
GPU: maps, cv.undistort, and the fused undistort+gray+crop kernel, all BIT-EXACT vs the oracle."""
import numpy as np
import pytest

from accurate_aprilgroup_tracking_amd import synthetic as syn

DISTS = [np.array([[-0.25, 0.1, 1e-3, -5e-4, -0.02]]), syn.MILD_DIST, np.array([[0.1, -0.05, 0.0, 0.0]]), None]


# the full 14-coefficient model: rational radial, tangential, thin prism and a TILTED sensor (tau_x, tau_y in radians)
TILTED_DIST = np.array([[0.05, -0.02, 1e-3, 2e-3, 0.01, 0.02, -0.01, 0.005, 1e-3, -2e-3, 5e-4, 1e-3, 0.03, -0.02]])


def _bgr(h, w, seed):
    rng = np.random.default_rng(seed)
    from scipy.ndimage import gaussian_filter
    img = np.stack([gaussian_filter(rng.standard_normal((h, w)), 1.2) for _ in range(3)], axis=-1)
    img = (img - img.min()) / (img.max() - img.min()) * 255
    return img.astype(np.uint8)


def test_oracle_preproc_self_checks(oracle):
    K = syn.camera_matrix(640, 480)
    img = _bgr(480, 640, 0)
    assert np.array_equal(oracle.undistort(img, K, None, None, K), img)          # identity map
    g = oracle.cvtColor(img, oracle.COLOR_BGR2GRAY)
    ref = ((img[..., 0].astype(np.int64) * 1868 + img[..., 1].astype(np.int64) * 9617 + img[..., 2].astype(np.int64) * 4899 + 8192) >> 14)
    assert np.array_equal(g, ref.astype(np.uint8))
    for dist in DISTS[:3]:
        newK, roi = oracle.getOptimalNewCameraMatrix(K, dist, (640, 480), 1, (640, 480))
        m1, m2 = oracle.initUndistortRectifyMap(K, dist, newK, (640, 480))
        ys, xs = np.meshgrid(np.arange(0, 480, 37), np.arange(0, 640, 41), indexing="ij")
        xn = (xs - newK[0, 2]) / newK[0, 0]; yn = (ys - newK[1, 2]) / newK[1, 1]
        p = syn.project(np.stack([xn.ravel(), yn.ravel(), np.ones(xn.size)], 1), np.zeros(3), np.zeros(3), K, dist)
        u = m1[ys, xs, 0] + (m2[ys, xs] & 31) / 32.0; v = m1[ys, xs, 1] + (m2[ys, xs] >> 5) / 32.0
        assert np.abs(u.ravel() - p[:, 0]).max() <= 1 / 64 + 1e-9 and np.abs(v.ravel() - p[:, 1]).max() <= 1 / 64 + 1e-9
        x, y, w, h = roi
        assert 0 <= x and 0 <= y and x + w <= 640 and y + h <= 480 and w > 320 and h > 240
    # alpha = 0 keeps only valid pixels: the whole new image is the ROI (up to the 1-px rounding OpenCV has)
    nk0, roi0 = oracle.getOptimalNewCameraMatrix(K, DISTS[0], (640, 480), 0, (640, 480))
    assert roi0[2] >= 638 and roi0[3] >= 478


def _numpy_undistort_maps(K, dist, newK, w, h):
    """cv::initUndistortRectifyMap(K, dist, I, newK, (w, h), CV_16SC2), vectorised (written from OpenCV's documented model:
    inverse new camera matrix -> normalised point -> Brown-Conrady + thin prism -> pixel -> 1/32-pixel fixed point)"""
    k = np.zeros(14)
    if dist is not None:
        d = np.asarray(dist, np.float64).reshape(-1); k[:d.size] = d
    ir = np.linalg.inv(np.asarray(newK, np.float64))
    j, i = np.meshgrid(np.arange(w, dtype=np.float64), np.arange(h, dtype=np.float64))
    _x = j * ir[0, 0] + (i * ir[0, 1] + ir[0, 2]); _y = j * ir[1, 0] + (i * ir[1, 1] + ir[1, 2]); _w = j * ir[2, 0] + (i * ir[2, 1] + ir[2, 2])
    x, y = _x * (1.0 / _w), _y * (1.0 / _w)
    x2, y2 = x * x, y * y
    r2, _2xy = x2 + y2, 2 * x * y
    kr = (1 + ((k[4] * r2 + k[1]) * r2 + k[0]) * r2) / (1 + ((k[7] * r2 + k[6]) * r2 + k[5]) * r2)
    xd = x * kr + k[2] * _2xy + k[3] * (r2 + 2 * x2) + k[8] * r2 + k[9] * r2 * r2
    yd = y * kr + k[2] * (r2 + 2 * y2) + k[3] * _2xy + k[10] * r2 + k[11] * r2 * r2
    if k[12] != 0 or k[13] != 0:        # tilted sensor (tests/pnp_numpy.py tilt_matrix)
        from tests.pnp_numpy import tilt_matrix
        T = tilt_matrix(k[12], k[13])
        hx, hy, hw = (T[q, 0] * xd + T[q, 1] * yd + T[q, 2] for q in range(3))
        xd, yd = hx / hw, hy / hw
    iu = np.rint((K[0, 0] * xd + K[0, 2]) * 32).astype(np.int64); iv = np.rint((K[1, 1] * yd + K[1, 2]) * 32).astype(np.int64)
    m1 = np.stack([iu >> 5, iv >> 5], axis=-1).astype(np.int16)
    m2 = ((iv & 31) * 32 + (iu & 31)).astype(np.uint16)
    return m1, m2


def _numpy_remap(src, m1, m2):
    """cv::remap(INTER_LINEAR, BORDER_CONSTANT 0) with CV_16SC2 maps on 8-bit images: 15-bit weights from the 5-bit fractions"""
    h, w = src.shape[:2]
    sx, sy = m1[..., 0].astype(np.int64), m1[..., 1].astype(np.int64)
    fx, fy = (m2 & 31).astype(np.int64), (m2 >> 5).astype(np.int64)
    pad = np.zeros((h + 2, w + 2) + src.shape[2:], np.int64); pad[1:-1, 1:-1] = src

    def tap(dy, dx):
        yy, xx = sy + dy, sx + dx
        ok = (yy >= 0) & (yy < h) & (xx >= 0) & (xx < w)
        v = pad[np.clip(yy, -1, h) + 1, np.clip(xx, -1, w) + 1]
        return np.where(ok[..., None] if src.ndim == 3 else ok, v, 0)
    ws = [(32 - fx) * (32 - fy) * 32, fx * (32 - fy) * 32, (32 - fx) * fy * 32, fx * fy * 32]
    if src.ndim == 3:
        ws = [x[..., None] for x in ws]
    acc = tap(0, 0) * ws[0] + tap(0, 1) * ws[1] + tap(1, 0) * ws[2] + tap(1, 1) * ws[3]
    return ((acc + (1 << 14)) >> 15).astype(np.uint8)


def test_oracle_maps_and_remap_equal_numpy_statement(oracle):
    """the oracle's initUndistortRectifyMap and remap (cv.undistort) against a vectorised numpy statement, BIT-EXACT: maps for
    three distortion models (incl. strong pincushion whose map leaves the source image) and the remapped BGR / gray frames"""
    for (w, h), seed in (((333, 201), 2), ((640, 480), 3)):
        K = syn.camera_matrix(w, h)
        K[0, 2] += 2.3; K[1, 2] -= 1.1
        img = _bgr(h, w, seed)
        for dist in DISTS[:3] + [np.array([[0.4, 0.3, 0.01, -0.02, 0.1]]), np.array([[0.05, -0.02, 1e-3, 2e-3, 0.01, 0.02, -0.01, 0.005, 1e-3, -2e-3, 5e-4, 1e-3]]), TILTED_DIST]:
            newK, _ = oracle.getOptimalNewCameraMatrix(K, dist, (w, h), 1, (w, h))
            for nk in (newK, K):
                m1, m2 = oracle.initUndistortRectifyMap(K, dist, nk, (w, h))
                n1, n2 = _numpy_undistort_maps(K, dist, nk, w, h)
                assert np.array_equal(m1, n1) and np.array_equal(m2, n2)
                assert np.array_equal(oracle.undistort(img, K, dist, None, nk), _numpy_remap(img, n1, n2))
                assert np.array_equal(oracle.undistort(img[..., 1].copy(), K, dist, None, nk), _numpy_remap(img[..., 1], n1, n2))


@pytest.mark.parametrize("alpha", [0.0, 0.35, 1.0])
def test_host_get_optimal_new_camera_matrix_matches_oracle(oracle, alpha):
    from accurate_aprilgroup_tracking_amd import cv_hip
    for (w, h) in ((1280, 720), (640, 480), (333, 201)):
        K = syn.camera_matrix(w, h)
        K[0, 2] += 3.7; K[1, 2] -= 2.2
        for dist in DISTS:
            nk_o, roi_o = oracle.getOptimalNewCameraMatrix(K, dist, (w, h), alpha, (w, h))
            nk_g, roi_g = cv_hip.getOptimalNewCameraMatrix(K, dist, (w, h), alpha, (w, h))
            assert np.array_equal(nk_o, nk_g) and roi_o == roi_g
    with pytest.raises(ValueError):
        cv_hip.getOptimalNewCameraMatrix(np.eye(3), np.zeros(3), (10, 10), 1)


def test_pose_detector_process_frame_on_oracle_backend(tmp_path, oracle):
    import json, logging
    from oracle import cv2_shim
    from accurate_aprilgroup_tracking_amd.pose_detector import PoseDetector
    s = syn.Sequence(640, 480, n_frames=1, seed=3, dist=DISTS[0])
    (tmp_path / "april_group.json").write_text(json.dumps(s.group))

    class Det(PoseDetector):
        DIRPATH = str(tmp_path)
    log = logging.getLogger("t"); log.setLevel(logging.CRITICAL)
    det = Det(log, s.K, s.dist, True, cv=cv2_shim.make_cv2())
    frame = _bgr(480, 640, 5)
    out = det.process_frame(frame)
    newK, roi = oracle.getOptimalNewCameraMatrix(s.K, s.dist, (640, 480), 1, (640, 480))
    ref = oracle.undistort(frame, s.K, s.dist, None, newK)[roi[1]:roi[1] + roi[3], roi[0]:roi[0] + roi[2]]
    assert out.shape == ref.shape and np.array_equal(out, ref)
    det2 = Det(log, s.K, None, True, cv=cv2_shim.make_cv2())
    assert det2.process_frame(frame) is frame                       # dist None: no undistortion (detect_pose.py:616)
    assert det._to_gray(frame).shape == (480, 640)


# ------------------------------------------------------------------------------------------- GPU
@pytest.mark.gpu
@pytest.mark.parametrize("size", [(640, 480), (1280, 720), (333, 201)])
def test_maps_and_undistort_bit_exact(oracle, size):
    import torch
    from accurate_aprilgroup_tracking_amd import cv_hip
    w, h = size
    K = syn.camera_matrix(w, h)
    img = _bgr(h, w, w)
    for dist in DISTS[:3]:
        newK, roi = cv_hip.getOptimalNewCameraMatrix(K, dist, (w, h), 1, (w, h))
        ctx = cv_hip.Context(64, 64, max_level=0)
        ctx.undistort_init(K, dist, newK, w, h)
        m1, m2 = ctx.undistort_maps()
        o1, o2 = oracle.initUndistortRectifyMap(K, dist, newK, (w, h))
        assert np.array_equal(m1, o1) and np.array_equal(m2, o2)
        ref = oracle.undistort(img, K, dist, None, newK)
        out = cv_hip.undistort(img, K, dist, None, newK)
        assert np.array_equal(out, ref)
        # fused: undistort -> gray -> crop, batch of 3
        batch = torch.from_numpy(np.stack([img, img[::-1].copy(), img[:, ::-1].copy()])).cuda()
        g = ctx.preprocess_bgr(batch, roi).cpu().numpy()
        for b, src in enumerate((img, img[::-1].copy(), img[:, ::-1].copy())):
            r = oracle.cvtColor(oracle.undistort(src, K, dist, None, newK)[roi[1]:roi[1] + roi[3], roi[0]:roi[0] + roi[2]],
                                oracle.COLOR_BGR2GRAY)
            assert np.array_equal(g[b], r)


@pytest.mark.gpu
def test_cvtcolor_and_strong_distortion_borders(oracle):
    from accurate_aprilgroup_tracking_amd import cv_hip
    img = _bgr(201, 333, 9)
    assert np.array_equal(cv_hip.cvtColor(img, cv_hip.COLOR_BGR2GRAY), oracle.cvtColor(img, oracle.COLOR_BGR2GRAY))
    K = syn.camera_matrix(333, 201)
    dist = np.array([[0.4, 0.3, 0.01, -0.02, 0.1]])       # pincushion: the map leaves the source image -> BORDER_CONSTANT taps
    out = cv_hip.undistort(img, K, dist, None, K)
    ref = oracle.undistort(img, K, dist, None, K)
    assert (ref == 0).any() and np.array_equal(out, ref)
    # one channel (round 4: cv2.undistort takes gray frames too; the gray raw frames of PoseDetector.process_frame)
    g = np.ascontiguousarray(img[..., 1])
    assert np.array_equal(cv_hip.undistort(g, K, dist, None, K), oracle.undistort(g, K, dist, None, K))
    with pytest.raises(ValueError):
        cv_hip.undistort(img.astype(np.float32), K, dist)
