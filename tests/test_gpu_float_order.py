"""-m gpu: DESIGN.md section 2, deviation 1 -- the HIP kernels form the LK window sums EXACTLY (integer sum, one rounding),
OpenCV accumulates them in float in a build-dependent order -- bounded where north_star bounds parity: at the POSE, <= 1e-4
"vs the reference's cv2.calcOpticalFlowPyrLK + cv2.solvePnP path".

Round 4 checked one scene (the c2 stream) against one order (OpenCV's scalar loop).  Here the device tracker
-- exact sums -- runs every scene ONCE and is compared with the CPU chain (oracle LK with the tracker's sticky status + the
reference-validated PoseDetector mirror on the oracle backend) run in BOTH float orders:
    CVO_ACC_FLOAT_SCALAR  sequential float adds in raster order (non-SIMD builds),
    CVO_ACC_FLOAT_SIMD    the CV_SIMD128 loops of the x86 builds the reference runs on [OpenCV-knowledge]: four float lanes per
                          covariance sum, int32 pair sums for the mismatch sums, tail columns scalar
on: the c2 stream (1280x720, 60 frames chained without a corner refresh), one 1920x1080 stream (configs[3] geometry), a 640x480
scene seen through a distorting lens, the c5 stream (60 tags / 240 corners, dense refinement and corner re-seed every frame) and
eight heterogeneous 1280x720 streams of one tracker (corners leaving the image, a stream below the gate, detector-fed frames).
Asserted: pose gap <= 1e-5 at every ACCEPTED frame (reprojection error below the reference's 2 px gate -- the only poses its
state machine uses) of every stream in both orders, an order of magnitude inside north_star's 1e-4; tracked-corner counts and
accept / reject decisions equal in all orders.  The heterogeneous set also holds REJECTED frames -- streams that jump two
frames at 2.4 x speed lose part of their track, the mean reprojection error is 4 .. 20 px -- where LK is chaotic: one iteration
more or less moves a corner on ambiguous texture by pixels, and OpenCV's own two orders differ from each other by up to 8e-4 in
the pose.  There the exact sums are held to that spread (2 x scalar-vs-SIMD + 1e-5; measured: 5.5e-4 / 3.5e-4 against 8.0e-4).
The measured worst gaps are printed (pytest -s) and tabulated in DESIGN.md section 2.
"""
import json
import logging

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
LOG = logging.getLogger("float_order"); LOG.setLevel(logging.CRITICAL)
BOUND = 1e-5


def _detector(tmp_path, tag, seq):
    from oracle import cv2_shim
    from accurate_aprilgroup_tracking_amd.pose_detector import PoseDetector
    d = tmp_path / tag
    d.mkdir(exist_ok=True)
    (d / "april_group.json").write_text(json.dumps(seq.group))

    class Det(PoseDetector):
        DIRPATH = str(d)
    return Det(LOG, seq.K, seq.dist, True, cv=cv2_shim.make_cv2())


def _chain(oracle, tmp_path, tag, seq, order, c0, det_tables, mode):
    """the stream's CPU chain with LK in accumulation order `mode` -> per frame (pose or None, reprojection error or None, tracked)"""
    det = _detector(tmp_path, "%s_m%d" % (tag, mode), seq)
    obj32 = seq.obj.astype(np.float32)
    n = obj32.shape[0]
    pyrs = {}

    def pyr_of(k):
        if k not in pyrs:
            pyrs[k] = oracle.Pyramid(seq.frame(k))
        return pyrs[k]
    pts = c0.astype(np.float32).copy(); alive = np.ones(n, bool); pyr = pyr_of(0)
    out = []
    for i, k in enumerate(order):
        npyr = pyr_of(k)
        if i in det_tables:
            tab, mask = det_tables[i]
            nx = tab.astype(np.float32).copy(); alive = mask.astype(bool).copy()
        else:
            nx, status, _ = oracle.calcOpticalFlowPyrLK(pyr, npyr, pts, maxLevel=2, acc_mode=mode)
            nx = nx.reshape(-1, 2); status = status.ravel().astype(bool)
            nx[~alive] = pts[~alive]
            alive = alive & status
        il = [nx[j].reshape(1, 1, 2) for j in range(n) if alive[j]]
        ol = [obj32[j].reshape(1, 3) for j in range(n) if alive[j]]
        det._estimate_pose(il if len(il) >= 8 else [], ol if len(il) >= 8 else [])
        pose = None
        if det.last_error is not None:
            pose = np.concatenate([det.last_pose[0].ravel(), det.last_pose[1].ravel()]).astype(np.float64)
        out.append((pose, det.last_error, int(alive.sum())))
        pts = nx.astype(np.float32); pyr = npyr
    return out


def _stream_gaps(oracle, tmp_path, tag, seq, order, c0, det_tables, rec, min_accepted=3):
    """Device records `rec` [steps, 16] (exact sums) against the stream's CPU chains in the exact, float-scalar and float-SIMD
    orders.  ACCEPTED frames (reprojection error below the reference's 2 px gate, detect_pose.py:539, in every chain -- the only
    poses the state machine ever uses) -> worst gap per order.  REJECTED frames (the track is partly lost: the error itself is
    several pixels, some corners sit on ambiguous texture and a one-iteration difference moves them by pixels) -> worst gap per
    order AND the gap between OpenCV's two own orders, which is what the exact sums are measured against there."""
    from accurate_aprilgroup_tracking_amd import hiplib as H
    ch = {m: _chain(oracle, tmp_path, tag, seq, order, c0, det_tables, m) for m in (oracle.ACC_EXACT, oracle.ACC_FLOAT_SCALAR, oracle.ACC_FLOAT_SIMD)}
    acc = {"exact": 0.0, "scalar": 0.0, "simd": 0.0}
    rej = {"scalar": 0.0, "simd": 0.0, "scalar_vs_simd": 0.0}
    n_acc = n_rej = 0
    for i in range(len(order)):
        ex, sc, si = ch[oracle.ACC_EXACT][i], ch[oracle.ACC_FLOAT_SCALAR][i], ch[oracle.ACC_FLOAT_SIMD][i]
        where = "%s frame %d" % (tag, i)
        assert int(rec[i, H.ST_NTRACK]) == ex[2] == sc[2] == si[2], where + ": tracked corners"
        if ex[0] is None:
            assert sc[0] is None and si[0] is None, where
            continue
        assert sc[0] is not None and si[0] is not None, where
        assert bool(rec[i, H.ST_OK]) == bool(ex[1] < 2), where + ": acceptance"
        g = lambda p: float(np.abs(rec[i, :6] - p).max())
        assert g(ex[0]) < 1e-8, where + ": exact-order chain %g" % g(ex[0])
        if ex[1] < 2 and sc[1] < 2 and si[1] < 2:
            n_acc += 1
            acc["exact"] = max(acc["exact"], g(ex[0])); acc["scalar"] = max(acc["scalar"], g(sc[0])); acc["simd"] = max(acc["simd"], g(si[0]))
        else:
            assert bool(ex[1] < 2) == bool(sc[1] < 2) == bool(si[1] < 2), where + ": the orders disagree about the gate"
            n_rej += 1
            rej["scalar"] = max(rej["scalar"], g(sc[0])); rej["simd"] = max(rej["simd"], g(si[0]))
            rej["scalar_vs_simd"] = max(rej["scalar_vs_simd"], float(np.abs(sc[0] - si[0]).max()))
    assert n_acc >= min_accepted, tag
    return dict(accepted=acc, rejected=rej, n_accepted=n_acc, n_rejected=n_rej)


def _device_stream(seq, depth, order=None, width=None, height=None):
    import torch
    from accurate_aprilgroup_tracking_amd.tracker import StreamTracker
    F = len(seq)
    order = list(range(1, F)) if order is None else order
    frames = torch.from_numpy(seq.frames()).cuda()
    trk = StreamTracker(seq.width, seq.height, seq.obj, seq.K, seq.dist, n_streams=1)
    trk.pipeline(depth)
    trk.reset(frames[0:1].contiguous(), torch.from_numpy(seq.corners(0)[None]).cuda().contiguous())
    so = trk.new_state_buffer(len(order))
    idx = torch.as_tensor(order, device="cuda")
    trk.step_many(frames[idx].unsqueeze(1).contiguous(), so)
    trk.join()
    torch.cuda.synchronize()
    return so.cpu().numpy()[:, 0], order


def _check(name, gaps):
    """accepted poses: both float orders within BOUND of the exact sums.  Rejected poses: the exact sums lie as close to either of
    OpenCV's orders as those lie to each other (x 2 + BOUND: not a theorem, a measurement -- all three are different roundings of
    the same sums)."""
    for tag, g in gaps.items():
        a, r = g["accepted"], g["rejected"]
        print("deviation-1 %-24s %-9s accepted frames %2d: exact %.1e scalar %.2e simd %.2e | rejected frames %2d: scalar %.2e simd %.2e (scalar vs simd %.2e)"
              % (name, tag, g["n_accepted"], a["exact"], a["scalar"], a["simd"], g["n_rejected"], r["scalar"], r["simd"], r["scalar_vs_simd"]))
    for tag, g in gaps.items():
        a, r = g["accepted"], g["rejected"]
        assert a["scalar"] <= BOUND and a["simd"] <= BOUND, "%s %s: accepted poses %g / %g" % (name, tag, a["scalar"], a["simd"])
        lim = 2.0 * r["scalar_vs_simd"] + BOUND
        assert r["scalar"] <= lim and r["simd"] <= lim, "%s %s: rejected poses %g / %g against OpenCV's own spread %g" % (name, tag, r["scalar"], r["simd"], r["scalar_vs_simd"])


SINGLE = {
    # id: (sequence factory, frames per launch)
    "c2_720p_60frames": (lambda syn: syn.Sequence(1280, 720, n_tags=12, n_frames=61, seed=1), 8),
    "c4_1080p_24frames": (lambda syn: syn.Sequence(1920, 1080, n_tags=12, n_frames=25, seed=4, supersample=2), 4),
    "640_distorted_16frames": (lambda syn: syn.Sequence(640, 480, n_tags=12, n_frames=17, seed=2, dist=syn.MILD_DIST), 4),
}


@pytest.mark.parametrize("scene", list(SINGLE), ids=list(SINGLE))
def test_exact_sums_vs_opencv_float_orders_single_stream(oracle, tmp_path, scene):
    from accurate_aprilgroup_tracking_amd import synthetic as syn
    make, depth = SINGLE[scene]
    seq = make(syn)
    rec, order = _device_stream(seq, depth)
    gaps = {"stream": _stream_gaps(oracle, tmp_path, scene, seq, order, seq.corners(0), {}, rec, min_accepted=len(order))}
    _check(scene, gaps)


def test_exact_sums_vs_opencv_float_orders_eight_heterogeneous_streams(oracle, tmp_path):
    """eight of the heterogeneous 1280x720 streams of tests/test_gpu_hetero.py on ONE tracker (lk_group split launch)"""
    from tests.test_gpu_hetero import make_streams, run_device
    steps, det_steps = 21, (18,)
    streams = make_streams(1280, 720, 8, steps, det_steps, supersample=2)
    rec, _, _ = run_device(streams, 4, 1280, 720)
    gaps = {}
    for b, st in enumerate(streams):
        gaps["stream %d" % b] = _stream_gaps(oracle, tmp_path, "het%d" % b, st.seq, st.order, st.c0, st.det, rec[:, b], min_accepted=2)
    _check("hetero8_720p", gaps)
    assert sum(g["n_rejected"] for g in gaps.values()) >= 10, "the set no longer holds partly lost tracks"


def test_exact_sums_vs_opencv_float_orders_c5_dense_reseed(oracle, tmp_path):
    """configs[4]: LK(240) -> PnP(240) -> dense refinement -> corner re-seed on the device; the chain's LK in both float orders.
    Both the PnP pose and the refined pose are bounded (the refinement starts at the PnP pose and sees the LK corners as its
    geometric rows)."""
    import torch
    from accurate_aprilgroup_tracking_amd import hiplib as H, synthetic as syn
    from accurate_aprilgroup_tracking_amd.tracker import StreamTracker
    s = syn.Sequence(1280, 720, n_tags=60, n_frames=9, seed=8, supersample=2)
    n = s.obj.shape[0]
    mx = syn.model_samples(s.group, 32)
    T = np.nan_to_num(syn.sample_bilinear(s.frame(0), syn.project(mx, s.rvecs[0], s.tvecs[0], s.K)), nan=128.0).astype(np.float32)
    iters, pw = 4, 0.05
    F = len(s)
    frames = torch.from_numpy(s.frames()).cuda()
    trk = StreamTracker(s.width, s.height, s.obj, s.K, None, n_streams=1)
    mxg, Tg = torch.from_numpy(mx).cuda(), torch.from_numpy(T).cuda()
    trk.dense_model(mxg, Tg, iters=iters, photo_weight=pw, reseed=True)
    trk.reset(frames[0:1].contiguous(), torch.from_numpy(s.corners(0)[None]).cuda().contiguous())
    so = trk.new_state_buffer(F - 1)
    do = torch.zeros((F - 1, 1, H.DENSE_STRIDE), dtype=torch.float64, device="cuda")
    trk.step_many_dense(frames[1:].unsqueeze(1).contiguous(), so, do)
    torch.cuda.synchronize()
    st, dn = so.cpu().numpy()[:, 0], do.cpu().numpy()[:, 0]
    obj32 = s.obj.astype(np.float32)
    out = []
    for mode in (oracle.ACC_EXACT, oracle.ACC_FLOAT_SCALAR, oracle.ACC_FLOAT_SIMD):
        det = _detector(tmp_path, "c5_m%d" % mode, s)
        pts = s.corners(0); alive = np.ones(n, bool); pyr = oracle.Pyramid(s.frame(0))
        w_pnp = w_ref = 0.0
        for k in range(1, F):
            npyr = oracle.Pyramid(s.frame(k))
            nx, status, _ = oracle.calcOpticalFlowPyrLK(pyr, npyr, pts, maxLevel=2, acc_mode=mode)
            nx = nx.reshape(-1, 2); status = status.ravel().astype(bool)
            nx[~alive] = pts[~alive]; alive &= status
            il = [nx[i].reshape(1, 1, 2) for i in range(n) if alive[i]]
            ol = [obj32[i].reshape(1, 3) for i in range(n) if alive[i]]
            det._estimate_pose(il, ol)
            assert det.last_error is not None and det.last_error < 2 and int(st[k - 1, H.ST_NTRACK]) == int(alive.sum()) == n
            r0 = det.last_pose[0].ravel().astype(np.float64); t0 = det.last_pose[1].ravel().astype(np.float64)
            w_pnp = max(w_pnp, np.abs(st[k - 1, :3] - r0).max(), np.abs(st[k - 1, 3:6] - t0).max())
            r, t, info = oracle.dense_refine(s.frame(k), mx, T, s.obj, nx.astype(np.float32), alive.astype(np.uint8), s.K, None, r0, t0,
                                             iters=iters, photo_weight=pw)
            assert dn[k - 1, H.DN_REFINED] == 1.0 and int(dn[k - 1, H.DN_ITERS]) == info["iters"]
            w_ref = max(w_ref, np.abs(dn[k - 1, :3] - r).max(), np.abs(dn[k - 1, 3:6] - t).max())
            pp, _ = oracle.projectPoints(s.obj, r, t, s.K, None)
            pts = pp.reshape(-1, 2).astype(np.float32); alive[:] = True
            pyr = npyr
        out.append((float(w_pnp), float(w_ref)))
    gaps = {"PnP pose": tuple(o[0] for o in out), "refined pose": tuple(o[1] for o in out)}
    for tag, (ex, sc, si) in gaps.items():
        print("deviation-1 %-24s %-12s exact %.2e  float-scalar %.2e  float-simd %.2e" % ("c5_dense_reseed", tag, ex, sc, si))
    for tag, (ex, sc, si) in gaps.items():
        assert ex < 1e-6, tag                    # re-seeded corners are rounded to float32 (tests/test_dense.py)
        assert sc <= BOUND and si <= BOUND, "%s: %g / %g" % (tag, sc, si)
