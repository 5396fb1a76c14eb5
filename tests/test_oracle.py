"""CPU: pin the oracle (C restatement of the cv2 algorithms) against independent
implementations available here -- scipy Rotation / least_squares / ndimage, numpy SVD and
the analytic projection of the synthetic generator.  cv2 itself is not installable, so
parity with real cv2 stays UNPINNED (oracle/cv_oracle.h)."""
import numpy as np
import pytest
from scipy.ndimage import correlate1d, gaussian_filter, shift
from scipy.optimize import least_squares
from scipy.spatial.transform import Rotation

from accurate_aprilgroup_tracking_amd import synthetic as syn


def test_rodrigues_matches_scipy(oracle):
    rng = np.random.default_rng(0)
    for _ in range(100):
        r = rng.normal(size=3) * rng.uniform(0, 3.1)
        R, J = oracle.Rodrigues(r)
        assert np.abs(R - Rotation.from_rotvec(r).as_matrix()).max() < 1e-14
        back, _ = oracle.Rodrigues(R)
        assert np.abs(back.ravel() - Rotation.from_matrix(R).as_rotvec()).max() < 1e-10
        num = np.stack([((oracle.Rodrigues(r + e)[0] - oracle.Rodrigues(r - e)[0]) / 2e-6).ravel()
                        for e in np.eye(3) * 1e-6])
        assert np.abs(J - num).max() < 1e-8
    R0, J0 = oracle.Rodrigues(np.zeros(3))
    assert np.array_equal(R0, np.eye(3)) and J0[0, 5] == -1 and J0[0, 7] == 1
    assert oracle.Rodrigues(np.zeros((3, 1), np.float32))[0].dtype == np.float32     # depth follows input
    # theta ~ pi branch
    for axis in np.eye(3):
        R = Rotation.from_rotvec(axis * np.pi).as_matrix()
        r, _ = oracle.Rodrigues(R)
        assert abs(np.linalg.norm(r) - np.pi) < 1e-7


def test_project_points_matches_analytic(oracle):
    s = syn.Sequence(640, 480, n_frames=2, seed=3, dist=syn.MILD_DIST)
    for k in range(2):
        out, jac = oracle.projectPoints(s.obj, s.rvecs[k], s.tvecs[k], s.K, s.dist, jacobian=True)
        assert np.abs(out.reshape(-1, 2) - syn.project(s.obj, s.rvecs[k], s.tvecs[k], s.K, s.dist)).max() < 1e-10
        p = np.concatenate([s.rvecs[k], s.tvecs[k]])
        for i in range(6):
            d = np.zeros(6); d[i] = 1e-7
            a, _ = oracle.projectPoints(s.obj, (p + d)[:3], (p + d)[3:], s.K, s.dist)
            b, _ = oracle.projectPoints(s.obj, (p - d)[:3], (p - d)[3:], s.K, s.dist)
            assert np.abs(jac[:, i] - ((a - b) / 2e-7).ravel()).max() < 1e-4 * max(1.0, np.abs(jac[:, i]).max())
    with pytest.raises(ValueError):
        oracle.projectPoints(s.obj, s.rvecs[0], s.tvecs[0], s.K, np.zeros(3))


def test_svd_and_solve(oracle):
    rng = np.random.default_rng(1)
    for (m, n) in ((3, 3), (6, 6), (12, 12), (20, 6)):
        A = rng.normal(size=(m, n))
        w, u, vt = oracle.svd(A)
        assert np.abs(w - np.linalg.svd(A, compute_uv=False)).max() < 1e-12
        assert np.abs((u * w) @ vt - A).max() < 1e-12
    A = rng.normal(size=(6, 6)); A = A @ A.T + np.eye(6); b = rng.normal(size=6)
    assert np.abs(oracle.solve_svd(A, b) - np.linalg.solve(A, b)).max() < 1e-11


@pytest.mark.parametrize("use_dist", [False, True])
def test_solve_pnp_matches_scipy_least_squares(oracle, use_dist):
    dist = syn.MILD_DIST if use_dist else None
    s = syn.Sequence(1280, 720, n_frames=3, seed=4, dist=dist)
    rng = np.random.default_rng(2)
    for k in range(3):
        img = s.corners(k).astype(np.float64) + rng.normal(0, 0.2, (48, 2))

        def resid(p):
            return (syn.project(s.obj, p[:3], p[3:], s.K, dist) - img).ravel()
        x0 = np.concatenate([s.rvecs[k], s.tvecs[k]])
        ref = least_squares(resid, x0, method="lm", xtol=1e-15, ftol=1e-15, gtol=1e-15).x
        ok, r, t, it = oracle.solvePnP(s.obj, img, s.K, dist, return_iters=True)            # DLT init
        assert ok and np.abs(np.concatenate([r.ravel(), t.ravel()]) - ref).max() < 2e-7
        g_r, g_t = x0[:3] + 0.02, x0[3:] - 0.004
        ok, r2, t2 = oracle.solvePnP(s.obj, img, s.K, dist, g_r.copy(), g_t.copy(), True)   # guess
        assert np.abs(np.concatenate([r2.ravel(), t2.ravel()]) - ref).max() < 2e-7
        assert 1 <= it <= 20
    # noise-free recovery (SURVEY 8c iii): <= 1e-8
    img = syn.project(s.obj, s.rvecs[0], s.tvecs[0], s.K, dist)
    ok, r, t = oracle.solvePnP(s.obj, img, s.K, dist)
    assert np.abs(r.ravel() - s.rvecs[0]).max() < 1e-8 and np.abs(t.ravel() - s.tvecs[0]).max() < 1e-8


def test_solve_pnp_guess_written_in_place(oracle):
    """cv2 writes the result into the guess arrays, in their dtype (detect_pose.py:487-490)"""
    s = syn.Sequence(640, 480, n_frames=1, seed=5)
    img = s.corners(0)
    g_r = (s.rvecs[0] + 0.01).reshape(3, 1); g_t = (s.tvecs[0] + 0.001).astype(np.float32).reshape(3, 1)
    ok, r, t = oracle.solvePnP(s.obj.astype(np.float32), img, s.K, None, g_r, g_t, True)
    assert r is g_r and t is g_t and t.dtype == np.float32
    assert np.abs(g_r.ravel() - s.rvecs[0]).max() < 1e-4


def test_planar_init_and_errors(oracle):
    K = syn.camera_matrix(640, 480)
    rng = np.random.default_rng(6)
    obj = np.concatenate([rng.uniform(-0.05, 0.05, (10, 2)), np.zeros((10, 1))], axis=1)
    r = np.array([0.3, -0.2, 0.1]); t = np.array([0.01, 0.02, 0.4])
    img = syn.project(obj, r, t, K)
    ok, rr, tt = oracle.solvePnP(obj, img, K, None)
    assert np.abs(rr.ravel() - r).max() < 1e-6 and np.abs(tt.ravel() - t).max() < 1e-6
    with pytest.raises(ValueError):
        oracle.solvePnP(obj[:3], img[:3], K, None)
    with pytest.raises(ValueError):
        oracle.solvePnP(obj, img[:5], K, None)


@pytest.mark.parametrize("shape", [(48, 64), (37, 53), (5, 7), (2, 9), (120, 160)])
def test_pyr_down_matches_scipy(oracle, shape):
    rng = np.random.default_rng(shape[0])
    img = rng.integers(0, 256, size=shape, dtype=np.uint8)
    k = np.array([1, 4, 6, 4, 1], dtype=np.int64)
    full = correlate1d(correlate1d(img.astype(np.int64), k, axis=1, mode="mirror"), k, axis=0, mode="mirror")
    ref = ((full[::2, ::2] + 128) >> 8).astype(np.uint8)
    assert np.array_equal(oracle.pyrDown(img), ref)


def test_scharr_matches_scipy(oracle):
    rng = np.random.default_rng(7)
    img = rng.integers(0, 256, size=(33, 41), dtype=np.uint8)
    a = img.astype(np.int64)
    sm, df = np.array([3, 10, 3]), np.array([-1, 0, 1])
    dx = correlate1d(correlate1d(a, sm, axis=0, mode="mirror"), df, axis=1, mode="mirror")
    dy = correlate1d(correlate1d(a, df, axis=0, mode="mirror"), sm, axis=1, mode="mirror")
    out = oracle.scharr(img)
    assert np.array_equal(out[..., 0], dx) and np.array_equal(out[..., 1], dy)


def test_pyramid_levels_and_early_stop(oracle):
    img = np.zeros((100, 90), np.uint8)
    p = oracle.Pyramid(img, win=21, max_level=5)
    assert p.levels == 2                      # 90x100 -> 45x50 -> 23x25; next would be 12x13 <= 21
    assert p.level(2).shape == (25, 23)


def test_lk_recovers_integer_shift(oracle):
    """pure integer translation of a band-limited texture, error <= 0.02 px"""
    rng = np.random.default_rng(8)
    base = gaussian_filter(rng.standard_normal((200, 260)), 2.0)
    base = ((base - base.min()) / (base.max() - base.min()) * 255).astype(np.uint8)
    a = base[20:180, 20:240]
    b = base[17:177, 25:245]          # content moves by (-5, +3)
    pts = rng.uniform([30, 30], [190, 130], size=(40, 2)).astype(np.float32)
    nx, st, er = oracle.calcOpticalFlowPyrLK(a, b, pts, maxLevel=2)
    good = st.ravel() == 1
    assert good.sum() >= 38
    d = nx.reshape(-1, 2)[good] - pts[good]
    assert np.abs(d - np.array([-5.0, 3.0])).max() < 0.02
    # exact-integer accumulation vs OpenCV's scalar float accumulation: same answer to ~1e-4 px
    nx2, st2, _ = oracle.calcOpticalFlowPyrLK(a, b, pts, maxLevel=2, acc_mode=oracle.ACC_FLOAT_SCALAR)
    assert np.array_equal(st, st2) and np.abs(nx - nx2).max() < 2e-3


def test_lk_status_and_flags(oracle, seq640):
    a, b = seq640.frame(0), seq640.frame(1)
    h, w = a.shape
    pts = np.array([[-30.0, 5.0], [w + 25.0, h / 2], [5.0, h + 40.0]], np.float32)
    nx, st, er = oracle.calcOpticalFlowPyrLK(a, b, pts, maxLevel=2)
    assert st.sum() == 0 and np.all(er == 0)
    flat = np.full((120, 160), 77, np.uint8)
    nx, st, er = oracle.calcOpticalFlowPyrLK(flat, flat, np.array([[80.0, 60.0]], np.float32), maxLevel=1)
    assert st.sum() == 0                       # minEig below threshold
    c = seq640.corners(0)
    nx, st, er = oracle.calcOpticalFlowPyrLK(a, b, c, maxLevel=2, flags=oracle.OPTFLOW_LK_GET_MIN_EIGENVALS)
    assert st.all() and (er > 1e-4).all()
    nx_i, st_i, _ = oracle.calcOpticalFlowPyrLK(a, b, c, seq640.corners(1), maxLevel=0, flags=oracle.OPTFLOW_USE_INITIAL_FLOW)
    assert st_i.all() and np.abs(nx_i.reshape(-1, 2) - seq640.corners(1)).max() < 0.5
    assert oracle.calcOpticalFlowPyrLK(a, b, np.zeros((0, 2), np.float32))[0].shape == (0, 1, 2)


def test_end_to_end_pose_from_rendered_frames(oracle, seq640):
    """render -> LK -> PnP lands within 1e-3 of the analytic pose (sanity of the whole chain)"""
    s = seq640
    pyr = oracle.Pyramid(s.frame(0))
    pts = s.corners(0)
    r, t = s.rvecs[0].copy(), s.tvecs[0].copy()
    for k in range(1, 4):
        pyr, pts, st, er, cnt, r, t = oracle.track_frame(pyr, s.frame(k), pts, s.obj, s.K, None, r, t)
        assert cnt == 48
        assert np.abs(r - s.rvecs[k]).max() < 2e-3 and np.abs(t - s.tvecs[k]).max() < 2e-3


def test_threaded_full_frame_passes_are_band_invariant(oracle):
    """the cpu_baseline threads pyrDown / Scharr in bands of rows: every band count must give the serial bytes"""
    rng = np.random.default_rng(3)
    img = rng.integers(0, 256, (243, 317), dtype=np.uint8)
    L = oracle.lib()
    ref_s = oracle.scharr(img)
    ref_p = [oracle.Pyramid(img, max_level=3).level(l) for l in range(4)]
    try:
        for nt in (2, 3, 8, 64):
            L.cvo_set_num_threads(nt)
            assert L.cvo_get_num_threads() == nt
            assert np.array_equal(oracle.scharr(img), ref_s)
            p = oracle.Pyramid(img, max_level=3)
            for l in range(4):
                assert np.array_equal(p.level(l), ref_p[l])
    finally:
        L.cvo_set_num_threads(1)


def test_find_homography_lm_refinement_vs_scipy(oracle):
    """cv::findHomography(method 0) = normalised DLT + LMSolver polish (fundam.cpp, levmarq.cpp) for > 4 points.  The
    restatement is pinned on its defining property: the polished H is the least-squares minimiser of the reprojection
    error (scipy least_squares 'lm' from the same DLT start), the un-polished DLT is measurably not."""
    from scipy.optimize import least_squares
    rng = np.random.default_rng(0)
    Ht = np.array([[1.1, 0.05, 0.02], [-0.03, 0.95, -0.01], [0.2, -0.1, 1.0]])
    for n, noise in ((5, 1e-3), (20, 2e-3), (60, 5e-4)):
        M = rng.uniform(-0.05, 0.05, (n, 2)).astype(np.float32).astype(np.float64)
        P = np.c_[M, np.ones(n)] @ Ht.T
        m = (P[:, :2] / P[:, 2:] + rng.normal(0, noise, (n, 2))).astype(np.float32).astype(np.float64)
        H0 = oracle.findHomography(M, m, refine=False)
        H1 = oracle.findHomography(M, m, refine=True)
        assert abs(H0[2, 2] - 1.0) < 1e-15 and H1[2, 2] == H0[2, 2]        # h33 = h33 * (1 / h33), not a parameter of the polish

        def res(h):
            Hm = np.r_[h, 1.0].reshape(3, 3)
            Q = np.c_[M, np.ones(n)] @ Hm.T
            return (Q[:, :2] / Q[:, 2:] - m).ravel()
        sol = least_squares(res, H0.ravel()[:8], method="lm", xtol=1e-15, ftol=1e-15, gtol=1e-15)
        c0, c1, cs = (res(H0.ravel()[:8]) ** 2).sum(), (res(H1.ravel()[:8]) ** 2).sum(), (sol.fun ** 2).sum()
        assert c1 <= c0 and abs(c1 - cs) <= 1e-9 * cs
        assert np.abs(H1.ravel()[:8] - sol.x).max() < 2e-5            # OpenCV stops at |step|_inf < FLT_EPSILON
        assert np.abs(H0.ravel()[:8] - sol.x).max() > 10 * np.abs(H1.ravel()[:8] - sol.x).max()
    # four points: exact fit, no polish (cv::findHomography refines only when npoints > 4)
    M4 = np.array([[-1.0, -1], [-1, 1], [1, 1], [1, -1]]); P4 = np.c_[M4, np.ones(4)] @ Ht.T; m4 = P4[:, :2] / P4[:, 2:]
    assert np.abs(oracle.findHomography(M4, m4, True) - oracle.findHomography(M4, m4, False)).max() == 0.0
