"""-m gpu: HETEROGENEOUS multi-stream parity of the device tracker.

Every other multi-stream test feeds ONE stream replicated B times, which cannot see a cross-stream indexing error
(stream b reading stream 0's corners, arrival counter, ring entry or tracker state returns the right bits).  Here every
stream of a batch carries its OWN sequence -- its own trajectory, background, speed and frame order, some with corners
that leave the image at the first step, one that falls below the 8-corner gate and stays there until a detector-fed
frame re-seeds it, one whose detector table holds a single tag -- and every stream's records are compared with ITS OWN
CPU chain: oracle LK (sticky status, as the tracker) + the reference-validated PoseDetector mirror on the oracle
backend (detect_pose.py:467-574 per stream, no cross-stream state: SURVEY 8e).

Bars: pose <= 1e-8 (north-star bound 1e-4), ST_NTRACK / ST_OK / ST_GUESS / PNP_TOO_FEW equal in every record, the final
corner set and status of every stream bit-exact, and a permutation property: shuffling the stream order permutes the
records bit for bit.  Launch modes covered (48 corners per stream): fused chained launch (B = 3, 5), lk_group split
(B = 8, 16), one-wave split with two half-batch chains (B = 24, 64 -- 64 x 1280x720 = BASELINE configs[2]), 1920x1080
(B = 2, configs[3] geometry per GPU) and the dense stage of configs[4] with B = 2.
"""
import ctypes as C
import json
import logging

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
POSE_TOL = 1e-8
LOG = logging.getLogger("hetero"); LOG.setLevel(logging.CRITICAL)


class Stream:
    """one stream of a batch: a rendered sequence, the order its frames are taken in, its initial corner set and the
    detector tables of the detector-fed steps"""

    def __init__(self, seq, order, c0, det):
        self.seq, self.order, self.c0, self.det = seq, order, c0, det


# a small sensor tilt on top of mild lens terms (the renderer has no tilt: the solver sees points that are sub-pixel inconsistent with the model)
TILT14 = np.array([[0.01, -0.005, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0.004, -0.003]])


def make_streams(width, height, B, steps, det_steps=(), n_tags=12, group_seed=0, n_frames=6, supersample=2, base_seed=100,
                 kinds=True, camera="pinhole"):
    """B distinct streams showing the SAME AprilGroup (the object points are shared by the streams of a tracker).
    Stream b: seed base_seed + b (trajectory phases, background), speed 0.6 ... 2.4, its own walk over its frames.
    kinds: stream 1 loses two corners at the first step, stream 2 (when B >= 3) keeps only 6 corners (below the gate
    until the detector speaks), the last stream's detector tables hold one tag only (no pose, guess cleared)."""
    from accurate_aprilgroup_tracking_amd import synthetic as syn
    rng = np.random.default_rng(base_seed)
    out = []
    for b in range(B):
        s = syn.Sequence(width, height, n_tags=n_tags, n_frames=n_frames, seed=base_seed + b, group_seed=group_seed,
                         supersample=supersample, speed=0.6 + 1.8 * ((b * 7) % 11) / 10.0, dist=syn.MILD_DIST if camera == "lens" else None)
        if camera == "tilt":
            s.dist = TILT14                                    # (camera model of the solver only: see TILT14)
        n = s.obj.shape[0]
        # a walk over the stream's frames with steps of -2 .. +2 frames (never 0: a zero velocity is an error in the reference)
        k, order = 0, []
        for i in range(steps):
            nk = k
            while nk == k:
                nk = int(np.clip(k + rng.integers(-2, 3), 0, n_frames - 1))
            order.append(nk); k = nk
        c0 = s.corners(0).copy()
        if kinds and b == 1 and B > 1:
            c0[5] = (-40.0, 100.0); c0[n - 3] = (width + 35.0, 50.0)
        if kinds and b == 2:
            c0[6:] = (-50.0, -50.0)                            # 6 corners left: below the gate
        det = {}
        for i in det_steps:
            mask = np.ones(n, np.uint8)
            if kinds and b == B - 1 and B > 1:
                mask[4:] = 0                                   # one tag only
            elif b % 3 == 0:
                mask[4 * (b % (n // 4)):4 * (b % (n // 4)) + 4] = 0    # the detector missed one tag
            det[i] = (s.corners(order[i]).copy(), mask)
        out.append(Stream(s, order, c0, det))
    return out


def cpu_chain(oracle, st, tmp_path, tag):
    """the stream's own CPU chain -> (records, final points, final status)"""
    from oracle import cv2_shim
    from accurate_aprilgroup_tracking_amd.pose_detector import PoseDetector
    s = st.seq
    d = tmp_path / ("g_%s" % tag)
    d.mkdir(exist_ok=True)
    (d / "april_group.json").write_text(json.dumps(s.group))

    class Det(PoseDetector):
        DIRPATH = str(d)
    det = Det(LOG, s.K, s.dist, True, cv=cv2_shim.make_cv2())
    obj32 = s.obj.astype(np.float32)
    n = obj32.shape[0]
    pyrs = {}

    def pyr_of(k):
        if k not in pyrs:
            pyrs[k] = oracle.Pyramid(s.frame(k))
        return pyrs[k]
    pts = st.c0.astype(np.float32).copy(); alive = np.ones(n, bool); pyr = pyr_of(0)
    recs = []
    for i, k in enumerate(st.order):
        npyr = pyr_of(k)
        if i in st.det:
            tab, mask = st.det[i]
            nx = tab.astype(np.float32).copy(); alive = mask.astype(bool).copy()
        else:
            nx, status, _ = oracle.calcOpticalFlowPyrLK(pyr, npyr, pts, maxLevel=2)
            nx = nx.reshape(-1, 2); status = status.ravel().astype(bool)
            nx[~alive] = pts[~alive]
            alive = alive & status
        il = [nx[j].reshape(1, 1, 2) for j in range(n) if alive[j]]
        ol = [obj32[j].reshape(1, 3) for j in range(n) if alive[j]]
        guided = det.extrinsic_guess[0] is not None
        det._estimate_pose(il if len(il) >= 8 else [], ol if len(il) >= 8 else [])
        solved = det.last_error is not None
        recs.append(dict(ntrack=int(alive.sum()), too_few=len(il) < 8, ok=bool(solved and det.last_error < 2),
                         guided=bool(guided and solved), err=det.last_error,
                         pose=None if not solved else np.concatenate([det.last_pose[0].ravel(), det.last_pose[1].ravel()]).astype(np.float64)))
        pts = nx.astype(np.float32); pyr = npyr
    return recs, pts, alive


def run_device(streams, depth, width, height, perm=None, reproject=False):
    """all streams through ONE StreamTracker -> (records [steps, B, 16], corners [B, n, 2], status [B, n])"""
    import torch
    from accurate_aprilgroup_tracking_amd import hiplib as H
    from accurate_aprilgroup_tracking_amd.tracker import StreamTracker
    B = len(streams)
    perm = list(range(B)) if perm is None else list(perm)
    ss = [streams[p] for p in perm]
    s0 = ss[0].seq
    n = s0.obj.shape[0]
    steps = len(ss[0].order)
    dev_frames = [torch.from_numpy(st.seq.frames()).cuda() for st in ss]          # per stream [F, H, W]
    trk = StreamTracker(width, height, s0.obj, s0.K, s0.dist, n_streams=B, reproject=reproject)
    trk.pipeline(depth)
    f0 = torch.stack([dev_frames[b][0] for b in range(B)]).contiguous()
    c0 = torch.from_numpy(np.stack([st.c0 for st in ss]).astype(np.float32)).cuda().contiguous()
    trk.reset(f0, c0)
    so = torch.zeros((steps, B, H.STATE_STRIDE), dtype=torch.float64, device="cuda")
    keep = [f0]
    for i in range(steps):
        f = torch.stack([dev_frames[b][ss[b].order[i]] for b in range(B)]).contiguous()
        keep.append(f)
        if i in ss[0].det:
            tab = torch.from_numpy(np.stack([st.det[i][0] for st in ss]).astype(np.float32)).cuda().contiguous()
            mask = torch.from_numpy(np.stack([st.det[i][1] for st in ss]).astype(np.uint8)).cuda().contiguous()
            keep += [tab, mask]
            trk.step_detected(f, tab, mask, so[i])
        else:
            trk.step(f, so[i])
    trk.join()
    torch.cuda.synchronize()
    rec = so.cpu().numpy()
    cp, sp = trk.corners()
    got_c = np.zeros((B, n, 2), np.float32); got_s = np.zeros((B, n), np.uint8)
    H.check(trk.ctx.L.agt_download(trk.ctx.h, got_c.ctypes.data_as(C.c_void_p), C.c_void_p(cp), got_c.nbytes), "agt_download")
    H.check(trk.ctx.L.agt_download(trk.ctx.h, got_s.ctypes.data_as(C.c_void_p), C.c_void_p(sp), got_s.nbytes), "agt_download")
    assert not (rec[:, :, H.ST_FLAGS].astype(int) & H.TRK_CHAIN_TIMEOUT).any()
    return rec, got_c, got_s


def compare(rec, got_c, got_s, chains, tol=POSE_TOL):
    from accurate_aprilgroup_tracking_amd import hiplib as H
    n_ok = 0
    for b, (recs, pts, alive) in enumerate(chains):
        for i, r in enumerate(recs):
            g = rec[i, b]
            where = "stream %d step %d" % (b, i)
            assert int(g[H.ST_NTRACK]) == r["ntrack"], where + ": tracked corners"
            assert bool(int(g[H.ST_FLAGS]) & H.PNP_TOO_FEW) == r["too_few"], where + ": too-few flag"
            assert bool(g[H.ST_OK]) == r["ok"], where + ": acceptance"
            if r["pose"] is not None:
                assert np.abs(g[:6] - r["pose"]).max() < tol, where + ": pose %g" % np.abs(g[:6] - r["pose"]).max()
                assert bool(g[H.ST_GUESS]) == r["guided"], where + ": guess use"
                assert abs(g[H.ST_ERR] - r["err"]) < 1e-4, where
            n_ok += r["ok"]
        assert np.array_equal(got_s[b].astype(bool), alive), "stream %d: final status" % b
        assert np.array_equal(got_c[b].view(np.uint32), pts.view(np.uint32)), "stream %d: final corners" % b
    return n_ok


CASES = [
    # (id, width, height, B, depth, steps, detector-fed steps, permutation check)
    ("fused_B3_d4", 640, 480, 3, 4, 14, (6,), True),
    ("fused_B5_d1", 640, 480, 5, 1, 10, (5,), False),
    ("fused_B2_serial", 640, 480, 2, 0, 8, (3,), False),
    ("group_B8_d4", 640, 480, 8, 4, 14, (6,), True),
    ("group_B16_d2", 640, 480, 16, 2, 11, (5,), False),
    ("halves_B24_d4", 640, 480, 24, 4, 14, (9,), True),
    ("c3_B64_720p_d16", 1280, 720, 64, 16, 21, (18,), False),
    ("c4_B2_1080p_d4", 1920, 1080, 2, 4, 9, (4,), False),
    # round 5: the split pipeline's pose kernels (pnp_group_kernel, compiled in agt_step.hip) had only ever seen distortion-free cameras
    ("group_B8_d4_lens", 640, 480, 8, 4, 14, (6,), False, "lens"),
    ("group_B8_d4_tilt", 640, 480, 8, 4, 14, (6,), False, "tilt"),
    ("fused_B3_d4_tilt", 640, 480, 3, 4, 14, (6,), False, "tilt"),
]


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_every_stream_matches_its_own_oracle_chain(oracle, tmp_path, case):
    name, w, h, B, depth, steps, det_steps, check_perm = case[:8]
    camera = case[8] if len(case) > 8 else "pinhole"
    streams = make_streams(w, h, B, steps, det_steps, supersample=2 if w < 1900 else 1, camera=camera)
    chains = [cpu_chain(oracle, st, tmp_path, "%s_%d" % (name, b)) for b, st in enumerate(streams)]
    # the scenario is what it claims to be: streams differ, somebody loses corners, somebody sits below the gate, most poses accepted
    assert len({tuple(st.order) for st in streams}) == B or B > 40
    if B > 1:
        assert chains[1][0][0]["ntrack"] == 46
        assert chains[B - 1][0][det_steps[0]]["too_few"]
    if B > 2:
        assert chains[2][0][0]["too_few"] and chains[2][0][0]["ntrack"] == 6
    if B > 3:
        assert not chains[2][0][det_steps[0]]["too_few"], "the detector-fed frame brings the stream back" 
    rec, got_c, got_s = run_device(streams, depth, w, h)
    n_ok = compare(rec, got_c, got_s, chains)
    assert n_ok >= 0.5 * B * steps, "most stream-frames end in an accepted pose (%d of %d)" % (n_ok, B * steps)
    if check_perm:
        perm = np.random.default_rng(B).permutation(B)
        assert (perm != np.arange(B)).any()
        rec2, c2, s2 = run_device(streams, depth, w, h, perm=perm)
        for j, p in enumerate(perm):
            assert np.array_equal(rec2[:, j].view(np.uint64), rec[:, p].view(np.uint64)), "slot %d holds stream %d" % (j, p)
            assert np.array_equal(c2[j].view(np.uint32), got_c[p].view(np.uint32)) and np.array_equal(s2[j], got_s[p])


def test_clip_submission_with_distinct_streams(oracle, tmp_path):
    """agt_track_frames (clips) with B distinct streams: batch_stride / frame_stride indexing of a [K, B, H, W] clip"""
    import torch
    from accurate_aprilgroup_tracking_amd import hiplib as H
    from accurate_aprilgroup_tracking_amd.tracker import StreamTracker
    for B, depth in ((3, 4), (24, 4)):
        streams = make_streams(640, 480, B, 12)
        chains = [cpu_chain(oracle, st, tmp_path, "clip%d_%d" % (B, b)) for b, st in enumerate(streams)]
        s0 = streams[0].seq
        dev_frames = [torch.from_numpy(st.seq.frames()).cuda() for st in streams]
        clip = torch.stack([torch.stack([dev_frames[b][streams[b].order[i]] for b in range(B)]) for i in range(12)]).contiguous()
        trk = StreamTracker(640, 480, s0.obj, s0.K, None, n_streams=B)
        trk.pipeline(depth)
        f0 = torch.stack([dev_frames[b][0] for b in range(B)]).contiguous()
        trk.reset(f0, torch.from_numpy(np.stack([st.c0 for st in streams]).astype(np.float32)).cuda().contiguous())
        so = torch.zeros((12, B, H.STATE_STRIDE), dtype=torch.float64, device="cuda")
        trk.step_many(clip[:5], so[:5]); trk.step_many(clip[5:], so[5:])
        trk.join(); torch.cuda.synchronize()
        rec = so.cpu().numpy()
        n = s0.obj.shape[0]
        cp, sp = trk.corners()
        got_c = np.zeros((B, n, 2), np.float32); got_s = np.zeros((B, n), np.uint8)
        H.check(trk.ctx.L.agt_download(trk.ctx.h, got_c.ctypes.data_as(C.c_void_p), C.c_void_p(cp), got_c.nbytes), "agt_download")
        H.check(trk.ctx.L.agt_download(trk.ctx.h, got_s.ctypes.data_as(C.c_void_p), C.c_void_p(sp), got_s.nbytes), "agt_download")
        compare(rec, got_c, got_s, chains)


def test_c5_dense_stage_two_distinct_streams(oracle, tmp_path):
    """BASELINE configs[4] with B = 2: two 60-tag streams (same model and template, different trajectories and
    backgrounds) through LK(240) -> PnP(240) -> dense refinement -> re-seed; each against its own CPU chain"""
    import torch
    from oracle import cv2_shim
    from accurate_aprilgroup_tracking_amd import hiplib as H, synthetic as syn
    from accurate_aprilgroup_tracking_amd.pose_detector import PoseDetector
    from accurate_aprilgroup_tracking_amd.tracker import StreamTracker
    seqs = [syn.Sequence(1280, 720, n_tags=60, n_frames=5, seed=8 + b, group_seed=8, supersample=2, speed=1.0 + 0.7 * b) for b in range(2)]
    n = seqs[0].obj.shape[0]
    assert n == 240 and np.array_equal(seqs[0].obj, seqs[1].obj)
    for s in seqs:
        for k in range(len(s)):
            c = s.corners(k)
            assert c[:, 0].min() > 12 and c[:, 0].max() < 1280 - 12 and c[:, 1].min() > 12 and c[:, 1].max() < 720 - 12
    mx = syn.model_samples(seqs[0].group, 32)
    s0 = seqs[0]
    T = np.nan_to_num(syn.sample_bilinear(s0.frame(0), syn.project(mx, s0.rvecs[0], s0.tvecs[0], s0.K)), nan=128.0).astype(np.float32)
    iters, pw, tol = 4, 0.05, 1e-6                 # re-seeded corners are rounded to float32 (as test_c5_stream_matches_oracle_chain)
    F = len(s0)
    fr = [torch.from_numpy(s.frames()).cuda() for s in seqs]
    trk = StreamTracker(1280, 720, s0.obj, s0.K, None, n_streams=2)
    mxg, Tg = torch.from_numpy(mx).cuda(), torch.from_numpy(T).cuda()
    trk.dense_model(mxg, Tg, iters=iters, photo_weight=pw, reseed=True)
    trk.reset(torch.stack([fr[0][0], fr[1][0]]).contiguous(),
              torch.from_numpy(np.stack([seqs[0].corners(0), seqs[1].corners(0)])).cuda().contiguous())
    so = trk.new_state_buffer(F - 1)
    do = torch.zeros((F - 1, 2, H.DENSE_STRIDE), dtype=torch.float64, device="cuda")
    keep = []
    for k in range(1, F):
        f = torch.stack([fr[0][k], fr[1][k]]).contiguous(); keep.append(f)
        trk.step_dense(f, so[k - 1], do[k - 1])
    torch.cuda.synchronize()
    st, dn = so.cpu().numpy(), do.cpu().numpy()
    obj32 = s0.obj.astype(np.float32)
    for b, s in enumerate(seqs):
        d = tmp_path / ("c5_%d" % b); d.mkdir()
        (d / "april_group.json").write_text(json.dumps(s.group))

        class Det(PoseDetector):
            DIRPATH = str(d)
        det = Det(LOG, s.K, None, True, cv=cv2_shim.make_cv2())
        pts = s.corners(0); alive = np.ones(n, bool); pyr = oracle.Pyramid(s.frame(0))
        for k in range(1, F):
            npyr = oracle.Pyramid(s.frame(k))
            nx, status, _ = oracle.calcOpticalFlowPyrLK(pyr, npyr, pts, maxLevel=2)
            nx = nx.reshape(-1, 2); status = status.ravel().astype(bool)
            nx[~alive] = pts[~alive]; alive &= status
            il = [nx[i].reshape(1, 1, 2) for i in range(n) if alive[i]]
            ol = [obj32[i].reshape(1, 3) for i in range(n) if alive[i]]
            det._estimate_pose(il if len(il) >= 8 else [], ol if len(il) >= 8 else [])
            accepted = det.last_error is not None and det.last_error < 2
            where = "stream %d frame %d" % (b, k)
            assert int(st[k - 1, b, H.ST_NTRACK]) == int(alive.sum()) and int(st[k - 1, b, H.ST_OK]) == int(accepted), where
            assert accepted, where
            r0 = det.last_pose[0].ravel().astype(np.float64); t0 = det.last_pose[1].ravel().astype(np.float64)
            assert np.abs(st[k - 1, b, :3] - r0).max() < tol and np.abs(st[k - 1, b, 3:6] - t0).max() < tol, where + " PnP pose"
            r, t, info = oracle.dense_refine(s.frame(k), mx, T, s.obj, nx.astype(np.float32), alive.astype(np.uint8), s.K, None, r0, t0,
                                             iters=iters, photo_weight=pw)
            assert dn[k - 1, b, H.DN_REFINED] == 1.0
            assert np.abs(dn[k - 1, b, :3] - r).max() < tol and np.abs(dn[k - 1, b, 3:6] - t).max() < tol, where + " refined pose"
            assert int(dn[k - 1, b, H.DN_VALID]) == info["valid"] and int(dn[k - 1, b, H.DN_ITERS]) == info["iters"]
            pp, _ = oracle.projectPoints(s.obj, r, t, s.K, None)
            pts = pp.reshape(-1, 2).astype(np.float32); alive[:] = True
            pyr = npyr
    # the two streams really differ
    assert np.abs(st[:, 0, :6] - st[:, 1, :6]).max() > 1e-3
