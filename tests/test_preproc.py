"""Frame pre-processing (detect_pose.py:147-183 undistort_frame, :602 cvtColor).
CPU: oracle self-checks + the host-only getOptimalNewCameraMatrix of the C ABI vs the oracle.
GPU: maps, cv.undistort, and the fused undistort+gray+crop kernel, all BIT-EXACT vs the oracle."""
import numpy as np
import pytest

from accurate_aprilgroup_tracking_amd import synthetic as syn

DISTS = [np.array([[-0.25, 0.1, 1e-3, -5e-4, -0.02]]), syn.MILD_DIST, np.array([[0.1, -0.05, 0.0, 0.0]]), None]


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
    with pytest.raises(ValueError):
        cv_hip.undistort(img[..., 0], K, dist)
