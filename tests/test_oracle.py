"""CPU: pin the oracle (C restatement of the cv2 algorithms) against independent
implementations available here -- scipy Rotation / least_squares / ndimage, numpy SVD and
the analytic projection of the synthetic generator.  cv2 itself is not installable, so
parity with real cv2 stays UNPINNED (oracle/cv_oracle.h)."""
import ctypes

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
    # exact-integer accumulation (deviation 1) vs OpenCV's scalar float accumulation, one call:
    # measured 1.4e-4 px on this low-gradient texture (6e-5 px on the tag scenes)
    nx2, st2, _ = oracle.calcOpticalFlowPyrLK(a, b, pts, maxLevel=2, acc_mode=oracle.ACC_FLOAT_SCALAR)
    assert np.array_equal(st, st2) and np.abs(nx - nx2).max() < 2.5e-4


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


# --------------------------------------------------------------------------- LK per-point loop, pinned independently
def _lk_scenes(seq640):
    """(name, prev, next, points, kwargs): tag corners, a band-limited texture with border / outside points, and sub-pixel
    positions that make the fourth bilinear weight 0 or negative"""
    rng = np.random.default_rng(21)
    a, b = seq640.frame(0), seq640.frame(2)
    h, w = a.shape
    yield "tags", a, b, seq640.corners(0), dict(maxLevel=2)
    base = gaussian_filter(rng.standard_normal((140, 190)), 1.7)
    base = (base - base.min()) / (base.max() - base.min()) * 255
    ta = base.astype(np.uint8)
    tb = np.clip(shift(base, (1.6, -2.3), order=3, mode="reflect"), 0, 255).astype(np.uint8)
    th, tw = ta.shape
    border = np.array([[0.0, 0.0], [tw - 1.0, th - 1.0], [-5.5, 10.25], [tw + 3.0, 7.0], [3.2, th + 8.9], [-40.0, -40.0],
                       [tw + 30.0, th + 30.0], [10.5, 10.5], [tw - 11.0, th - 11.0], [1.0, th / 2.0], [-20.9, 50.0],
                       [tw + 19.5, 60.0]], np.float32)
    pts = np.concatenate([border, rng.uniform([-12, -12], [tw + 12, th + 12], size=(28, 2)).astype(np.float32)])
    yield "texture_border", ta, tb, pts, dict(maxLevel=3)
    yield "texture_level0_count5", ta, tb, pts, dict(maxLevel=0, criteria=(1, 5, 0.0))
    yield "texture_mineig", ta, tb, pts, dict(maxLevel=2, flags=8, minEigThreshold=1e-2)
    tiny = np.float32(2.0 ** -11)             # a * b * 2^14 < 1.5: iw11 rounds to 0 or -1
    sub = np.concatenate([seq640.corners(0)[:12].round() + np.array([[tiny * (1 + i % 5), tiny * (1 + i % 3)] for i in range(12)], np.float32),
                          np.floor(seq640.corners(0)[12:18]) + np.float32(0.5),
                          seq640.corners(0)[18:24].round()]).astype(np.float32)
    yield "iw11_nonpositive", a, b, sub, dict(maxLevel=2)
    init = sub + rng.normal(0, 1.2, sub.shape).astype(np.float32)
    yield "initial_flow", a, b, sub, dict(maxLevel=1, flags=4, nextPts=init)


def test_lk_oracle_equals_numpy_statement(oracle, seq640):
    """The C oracle's per-point LK loop against a second, structurally different statement of SURVEY.md Appendix A
    (tests/lk_numpy.py: scipy-filtered whole images, 21x21 array windows, int64) -- nextPts (u32 view), status and err
    BIT-EXACT, in the exact-sum mode the HIP kernels are held to, in OpenCV's scalar float accumulation order and (round 5)
    in the order of its CV_SIMD128 loops (four float lanes, int32 pair sums: what the shipped x86 builds execute)."""
    from tests import lk_numpy
    seen_neg = False
    for name, a, b, pts, kw in _lk_scenes(seq640):
        for mode, forder in ((oracle.ACC_EXACT, False), (oracle.ACC_FLOAT_SCALAR, True), (oracle.ACC_FLOAT_SIMD, "simd")):
            if forder and name not in ("tags", "texture_border"):
                continue
            o = oracle.calcOpticalFlowPyrLK(a, b, pts, acc_mode=mode, **kw)
            m = lk_numpy.calc_optical_flow_pyr_lk(a, b, pts, next_pts=kw.get("nextPts"), max_level=kw["maxLevel"],
                                                  criteria=kw.get("criteria", (3, 30, 0.01)), flags=kw.get("flags", 0),
                                                  min_eig_threshold=kw.get("minEigThreshold", 1e-4), float_order=forder)
            assert np.array_equal(o[1], m[1]), "%s: status" % name
            assert np.array_equal(o[0].view(np.uint32), m[0].view(np.uint32)), \
                "%s: nextPts differ by %g" % (name, np.abs(o[0] - m[0]).max())
            assert np.array_equal(o[2].view(np.uint32), m[2].view(np.uint32)), "%s: err" % name
        if name == "iw11_nonpositive":
            for p in pts:
                q = p - np.float32(10.0)
                w = lk_numpy._weights(np.float32(q[0] - np.floor(q[0])), np.float32(q[1] - np.floor(q[1])))
                seen_neg |= w[3] <= 0
            assert o[1].sum() >= 20
    assert seen_neg, "the edge-case set no longer reaches iw11 <= 0"


def test_exact_vs_float_accumulation_bounded_at_pose_level(oracle, seq720_long):
    """DESIGN.md section 2 deviation 1 carried to the quantity north_star bounds (pose, <= 1e-4): the chained c2 stream
    (1280x720, 60 frames, raw LK chaining, no corner refresh) tracked twice by the oracle, with exact integer window sums
    (what the HIP kernels compute) and with OpenCV's scalar float accumulation.  Measured: corners 6e-5 px after one
    frame, pose gap 2e-6 after 60 frames; asserted: <= 1e-5 at every frame (tests/test_gpu_tracker.py repeats this with
    the HIP tracker on the exact side)."""
    s = seq720_long
    runs = {}
    for mode in (oracle.ACC_EXACT, oracle.ACC_FLOAT_SCALAR):
        pyr, pts = oracle.Pyramid(s.frame(0)), s.corners(0)
        r, t = s.rvecs[0].copy(), s.tvecs[0].copy()
        poses, first, counts = [], None, []
        for k in range(1, len(s)):
            pyr, pts, st, er, cnt, r, t = oracle.track_frame(pyr, s.frame(k), pts, s.obj, s.K, None, r, t, acc_mode=mode)
            counts.append(cnt)
            poses.append(np.concatenate([r, t]))
            first = pts.copy() if first is None else first
        runs[mode] = (np.array(poses), first, counts)
    assert runs[oracle.ACC_EXACT][2] == runs[oracle.ACC_FLOAT_SCALAR][2] and min(runs[oracle.ACC_EXACT][2]) >= 46
    gap = np.abs(runs[oracle.ACC_EXACT][0] - runs[oracle.ACC_FLOAT_SCALAR][0]).max(axis=1)
    assert gap.max() <= 1e-5, gap.max()
    assert np.abs(runs[oracle.ACC_EXACT][1] - runs[oracle.ACC_FLOAT_SCALAR][1]).max() < 2.5e-4


# --------------------------------------------------------------------------- the LM loop of solvePnP, pinned independently
def test_pnp_lm_oracle_equals_numpy_statement(oracle):
    """The C oracle's cv::solvePnP(ITERATIVE, useExtrinsicGuess) -- projectPoints with its Jacobian, the CvLevMarq state machine
    (accept / reject, lambda schedule, the two stop rules) -- against a second, structurally different statement written from
    SURVEY.md Appendices B and C (tests/pnp_numpy.py: vectorised numpy, numpy.linalg pseudo-inverse for the damped solves, an explicit
    two-state loop).  Equal iteration counts (4 .. 20, incl. runs that exhaust the limit) and poses to 2e-9 (measured 1e-16 typically) on guesses that converge at once, guesses that are far off (steps get
    rejected, lambda climbs), noisy points, lens distortion, few points; the projection Jacobian to 1e-9 relative."""
    from tests import pnp_numpy as P
    rng = np.random.default_rng(31)
    n_reject = n_cases = n_maxed = 0
    worst = 0.0
    for dist, seed in ((None, 11), (syn.MILD_DIST, 12)):
        s = syn.Sequence(1280, 720, n_frames=3, seed=seed, dist=dist)
        d = None if dist is None else np.asarray(dist, np.float64).reshape(-1)
        # Jacobian and projection
        img_o, jac_o = oracle.projectPoints(s.obj, s.rvecs[1], s.tvecs[1], s.K, d, jacobian=True)
        img_n, jac_n = P.project(s.obj, s.rvecs[1], s.tvecs[1], s.K, d, jacobian=True)
        assert np.abs(img_o.reshape(-1, 2) - img_n).max() < 1e-10
        assert np.abs(jac_o - jac_n).max() < 1e-9 * max(1.0, np.abs(jac_o).max())
        for k in range(3):
            for noise, dr, dt, npts in ((0.0, 0.01, 0.002, 48), (0.3, 0.05, 0.01, 48), (0.2, 0.4, 0.08, 48), (0.0, 0.9, 0.15, 48),
                                        (0.5, 0.02, 0.004, 5), (1.0, 0.6, 0.1, 12)):
                sel = rng.choice(48, npts, replace=False)
                obj = s.obj[sel]
                img = s.corners(k).astype(np.float64)[sel] + rng.normal(0, noise, (npts, 2))
                g_r = s.rvecs[k] + rng.normal(0, dr, 3); g_t = s.tvecs[k] + rng.normal(0, dt, 3)
                trace = []
                r_n, t_n, it_n = P.solve_pnp_guess(obj, img, s.K, d, g_r, g_t, trace=trace)
                ok, r_o, t_o, it_o = oracle.solvePnP(obj, img, s.K, d, g_r.copy(), g_t.copy(), True, return_iters=True)
                assert it_o == it_n, "iteration counts differ: oracle %d, numpy %d (noise %g, dr %g)" % (it_o, it_n, noise, dr)
                if it_n < 20:            # (a run that uses up its 20 iterations has not converged: chaotic, only the count is compared)
                    worst = max(worst, np.abs(r_o.ravel() - r_n).max(), np.abs(t_o.ravel() - t_n).max())
                else:
                    n_maxed += 1
                n_reject += sum(1 for e in trace if e[0] == "reject")
                n_cases += 1
    # (measured: 1e-16 in 29 of the 33 converged cases, 8e-10 at worst -- far-off guesses whose damped normal equations are
    # ill-conditioned, where the two pseudo-inverses differ in the last bits)
    assert worst < 2e-9, worst
    assert 1 <= n_maxed <= 4
    assert n_cases == 36 and n_reject >= 3, "the far-off guesses no longer exercise the rejection branch (%d rejections)" % n_reject


def test_pnp_noguess_init_oracle_equals_numpy_statement(oracle):
    """cv::solvePnP(ITERATIVE) WITHOUT a guess (detect_pose.py:509-515): the C oracle's initialisation --
    undistortPoints (5 fixed-point iterations), the 2N x 12 DLT, its smallest singular vector, sign, orthonormalisation and
    rescaling -- against tests/pnp_numpy.py (written from SURVEY.md Appendix B with numpy.linalg.svd; the oracle gets the vector by
    its own Jacobi eigen-solver of L^T L).  The initial pose to 1e-9 relative to its size (the singular vector of a 12 x 12 Gram
    matrix whose condition number is ~1e9), the refined pose to 2e-9, equal LM iteration counts; with and without lens distortion,
    noisy points, 6 .. 48 points."""
    from tests import pnp_numpy as P
    rng = np.random.default_rng(77)
    worst_init = worst_pose = 0.0
    n_cases = 0
    for dist, seed in ((None, 21), (syn.MILD_DIST, 22)):
        s = syn.Sequence(1280, 720, n_frames=3, seed=seed, dist=dist)
        d = None if dist is None else np.asarray(dist, np.float64).reshape(-1)
        pts = s.corners(1).astype(np.float64)
        und_o = oracle.undistortPoints(pts, s.K, d).reshape(-1, 2)
        assert np.abs(und_o - P.undistort_points(pts, s.K, d)).max() < 1e-14
        for k in range(3):
            for noise, npts in ((0.0, 48), (0.3, 48), (1.0, 24), (0.2, 8), (0.1, 6)):
                sel = rng.choice(48, npts, replace=False)
                obj = s.obj[sel]
                if P.is_planar(obj):
                    continue
                img = s.corners(k).astype(np.float64)[sel] + rng.normal(0, noise, (npts, 2))
                r0_o, t0_o = oracle.pnp_init(obj, img, s.K, d)
                r_n, t_n, it_n, (r0_n, t0_n) = P.solve_pnp_noguess(obj, img, s.K, d)
                scale = max(1.0, np.abs(r0_n).max(), np.abs(t0_n).max())
                worst_init = max(worst_init, np.abs(r0_o.ravel() - r0_n).max() / scale, np.abs(t0_o.ravel() - t0_n).max() / scale)
                ok, r_o, t_o, it_o = oracle.solvePnP(obj, img, s.K, d, return_iters=True)
                assert ok and it_o == it_n, (it_o, it_n, noise, npts)
                worst_pose = max(worst_pose, np.abs(r_o.ravel() - r_n).max(), np.abs(t_o.ravel() - t_n).max())
                n_cases += 1
    assert n_cases >= 24
    assert worst_init < 1e-9, worst_init
    assert worst_pose < 2e-9, worst_pose


def test_pnp_planar_init_oracle_equals_numpy_statement(oracle):
    """cv::solvePnP(ITERATIVE) WITHOUT a guess on a PLANAR object (first branch) -- the
    one branch of the restatement that had no second statement.  tests/pnp_numpy.py (round 4) states it
    independently: the plane rotation from numpy's SVD of the scatter, cv::findHomography as the normalised DLT on float32 points
    (numpy.linalg.eigh of the 9 x 9 Gram matrix) + the LMSolver polish of levmarq.cpp spelled out (gain-ratio lambda update,
    lambda * diag(J^T J) damping, pinv for the damped solves), the rotation from the homography's columns through a Rodrigues
    round trip -- against the C oracle's cvo_find_homography / cvo_pnp_init (Jacobi eigen-solver, own LM loop).  40 problems:
    4 .. 24 points (four points: no polish), planes z = const and tilted, with and without lens distortion, exact and noisy
    image points.  Homography entries to 1e-6 relative (two eigen-solvers on a Gram matrix), initial pose to 1e-6, refined pose to
    1e-9 with EQUAL LM iteration counts."""
    from scipy.spatial.transform import Rotation
    from tests import pnp_numpy as P
    rng = np.random.default_rng(0)
    K = np.array([[1000.0, 0, 640], [0, 1000.0, 360], [0, 0, 1]])
    worst_h = worst_init = worst_pose = 0.0
    for case in range(40):
        n = [4, 5, 8, 12, 24][case % 5]
        p2 = rng.uniform(-0.05, 0.05, (n, 2))
        if n == 4:
            p2 = np.array([[-.01, -.01], [-.01, .01], [.01, .01], [.01, -.01]]) * (1 + case * 0.1)      # one tag
        r_pl = Rotation.from_rotvec(rng.uniform(-1, 1, 3)).as_matrix() if case % 3 else np.eye(3)
        obj = (np.c_[p2, np.zeros(n)] @ r_pl.T + rng.uniform(-0.02, 0.02, 3)).astype(np.float32).astype(np.float64)
        assert P.is_planar(obj)
        rv = rng.uniform(-0.6, 0.6, 3)
        tv = np.array([rng.uniform(-0.05, 0.05), rng.uniform(-0.05, 0.05), rng.uniform(0.25, 0.6)])
        dist = None if case % 4 else np.array([0.05, -0.1, 1e-3, -1e-3, 0.02])
        img = P.project(obj, rv, tv, K, dist) + rng.normal(0, 0.3 if case % 2 else 0.0, (n, 2))
        img = img.astype(np.float32).astype(np.float64)
        # the homography on its own: plane coordinates -> normalised image points
        mn = P.undistort_points(img, K, dist)
        h_n = P.find_homography(p2, mn)
        h_o = oracle.findHomography(p2, mn, refine=True)
        worst_h = max(worst_h, np.abs(h_n - h_o).max() / np.abs(h_n).max())
        r0_o, t0_o = oracle.pnp_init(obj, img, K, dist)
        r_n, t_n, it_n, (r0_n, t0_n) = P.solve_pnp_noguess_planar(obj, img, K, dist)
        worst_init = max(worst_init, np.abs(r0_o.ravel() - r0_n).max(), np.abs(t0_o.ravel() - t0_n).max())
        ok, r_o, t_o, it_o = oracle.solvePnP(obj, img, K, dist, return_iters=True)
        assert ok and it_o == it_n, (case, it_o, it_n)
        worst_pose = max(worst_pose, np.abs(r_o.ravel() - r_n).max(), np.abs(t_o.ravel() - t_n).max())
    assert worst_h < 1e-6, worst_h
    assert worst_init < 1e-6, worst_init
    assert worst_pose < 1e-9, worst_pose


def test_tilted_sensor_model_oracle_equals_numpy_statement(oracle):
    """The 14-coefficient camera model with a TILTED sensor (tau_x, tau_y != 0; VERDICT r4 missing #6; what cv2.calibrateCamera returns under
    CALIB_TILTED_MODEL -- the reference calibrates 5 coefficients, calibrate_camera.py:178): the C oracle's projectPoints (+ Jacobian),
    undistortPoints and both solvePnP branches against tests/pnp_numpy.py, whose tilt is written from the model's geometry (rotate the
    distorted point by R_y R_x, project back along the rotated axis; numpy.linalg.inv for the way back)."""
    from tests import pnp_numpy as P
    rng = np.random.default_rng(77)
    # the matrices themselves: the oracle's pair are inverses of each other and equal the numpy statement
    L = oracle.lib()
    for tx, ty in ((0.03, -0.02), (-0.1, 0.07), (0.0, 0.05), (0.2, 0.0)):
        M = np.zeros(9); Mi = np.zeros(9)
        L.cvo_tilt_matrices(ctypes.c_double(tx), ctypes.c_double(ty), M.ctypes.data_as(ctypes.c_void_p), Mi.ctypes.data_as(ctypes.c_void_p))
        assert np.abs(M.reshape(3, 3) - P.tilt_matrix(tx, ty)).max() < 1e-15
        assert np.abs(M.reshape(3, 3) @ Mi.reshape(3, 3) - np.eye(3)).max() < 1e-15
    worst_pose = 0.0
    for case, dist in enumerate((np.array([0.05, -0.02, 1e-3, 2e-3, 0.01, 0.02, -0.01, 0.005, 1e-3, -2e-3, 5e-4, 1e-3, 0.03, -0.02]),
                                 np.array([0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, -0.05, 0.04]),
                                 np.array([-0.2, 0.1, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0.08, 0.0]))):
        s = syn.Sequence(1280, 720, n_frames=3, seed=40 + case)
        for k in range(3):
            r, t = s.rvecs[k], s.tvecs[k]
            img_o, jac_o = oracle.projectPoints(s.obj, r, t, s.K, dist, jacobian=True)
            img_n, jac_n = P.project(s.obj, r, t, s.K, dist, jacobian=True)
            assert np.abs(img_o.reshape(-1, 2) - img_n).max() < 1e-10
            assert np.abs(jac_o - jac_n).max() < 1e-9 * max(1.0, np.abs(jac_o).max())
            # the analytic Jacobian against central differences of the projection itself
            p0 = np.concatenate([r, t])
            for i in range(6):
                d = np.zeros(6); d[i] = 1e-7
                a = P.project(s.obj, (p0 + d)[:3], (p0 + d)[3:], s.K, dist); b = P.project(s.obj, (p0 - d)[:3], (p0 - d)[3:], s.K, dist)
                assert np.abs(jac_o[:, i] - ((a - b) / 2e-7).ravel()).max() < 2e-4 * max(1.0, np.abs(jac_o[:, i]).max())
            # the tilt is NOT a no-op: the same coefficients without it land elsewhere
            assert np.abs(img_n - P.project(s.obj, r, t, s.K, dist[:12])).max() > 0.5
            # undistortPoints: equal to the statement, and the inverse of the forward model (normalised coordinates of the rotated points)
            un_o = oracle.undistortPoints(img_n, s.K, dist)
            un_n = P.undistort_points(img_n, s.K, dist)
            assert np.abs(un_o.reshape(-1, 2) - un_n).max() < 1e-13
            Y = s.obj @ P.rodrigues(r)[0].T + t
            assert np.abs(un_n - Y[:, :2] / Y[:, 2:3]).max() < (1e-9 if case == 1 else 1e-5)
            # solvePnP with a guess: equal iteration counts, equal poses
            noisy = img_n + rng.normal(0, 0.2, img_n.shape)
            g_r = r + rng.normal(0, 0.05, 3); g_t = t + rng.normal(0, 0.01, 3)
            r_n, t_n, it_n = P.solve_pnp_guess(s.obj, noisy, s.K, dist, g_r, g_t)
            ok, r_o, t_o, it_o = oracle.solvePnP(s.obj, noisy, s.K, dist, g_r.copy(), g_t.copy(), True, return_iters=True)
            assert ok and it_o == it_n
            worst_pose = max(worst_pose, np.abs(r_o.ravel() - r_n).max(), np.abs(t_o.ravel() - t_n).max())
            # ... and without one (planar model: the homography branch; undistortPoints feeds it)
            ok, r_o, t_o = oracle.solvePnP(s.obj, img_n, s.K, dist)
            assert ok and np.abs(r_o.ravel() - r).max() < 1e-6 and np.abs(t_o.ravel() - t).max() < 1e-6
            r_n, t_n = P.solve_pnp_noguess(s.obj, noisy, s.K, dist)[:2]
            ok, r_o, t_o = oracle.solvePnP(s.obj, noisy, s.K, dist)
            assert ok
            worst_pose = max(worst_pose, np.abs(r_o.ravel() - np.ravel(r_n)).max(), np.abs(t_o.ravel() - np.ravel(t_n)).max())
    assert worst_pose < 2e-9, worst_pose
