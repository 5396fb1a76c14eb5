"""CPU: on-disk / wire formats either side of the path: april_group.json
(detect_pose.py:122-130), CameraParams.npz (calibrate_camera.py:107-123), swatbotics-style detections
(detect_pose.py:389-437) and their replay through the PoseDetector mirror (oracle backend: tests may)."""
import json
import logging
import os

import numpy as np
import pytest

from accurate_aprilgroup_tracking_amd import formats, synthetic as syn
from accurate_aprilgroup_tracking_amd.pose_detector import PoseDetector

LOG = logging.getLogger("test"); LOG.setLevel(logging.CRITICAL)


def _detections(seq, k, drop=(), weak=(), extra_unknown=False):
    tag_ids = [int(t) for t in seq.group["tags"].keys()]
    c = seq.corners(k).reshape(-1, 4, 2)
    dets = []
    for i, t in enumerate(tag_ids):
        if t in drop:
            continue
        dets.append(formats.make_detection(t, c[i], decision_margin=30.0 if t in weak else 80.0 + i))
    if extra_unknown:
        dets.append(formats.make_detection(999, c[0] + 5.0, decision_margin=90.0))
    return dets


def test_april_group_round_trip_matches_mirror(tmp_path, oracle):
    from oracle import cv2_shim
    seq = syn.Sequence(640, 480, n_tags=12, n_frames=2, seed=3)
    path = tmp_path / "april_group.json"
    path.write_text(json.dumps(seq.group))
    ext = formats.load_april_group(path)

    class Det(PoseDetector):
        DIRPATH = str(tmp_path)
    det = Det(LOG, seq.K, None, True, cv=cv2_shim.make_cv2())
    assert list(ext) == list(det.extrinsics)
    for k in ext:
        assert ext[k][0] == det.extrinsics[k][0]
        for a, b in zip(ext[k][1:], det.extrinsics[k][1:]):
            assert a.dtype == np.float32 and a.shape == (3, 1) and np.array_equal(a, b)
    # writer -> reader is the identity on what the reference reads
    out = tmp_path / "copy.json"
    formats.save_april_group(out, ext)
    ext2 = formats.load_april_group(out)
    assert all(np.array_equal(ext[k][1], ext2[k][1]) and np.array_equal(ext[k][2], ext2[k][2]) and ext[k][0] == ext2[k][0] for k in ext)
    # from_files with an explicit model path gives the same object points
    cam = tmp_path / "CameraParams.npz"
    formats.save_camera_params(cam, seq.K, np.zeros(5))
    det2 = PoseDetector.from_files(LOG, cam, True, cv=cv2_shim.make_cv2(), april_group=out)
    assert np.array_equal(det2.all_objpts, det.all_objpts)


def test_april_group_errors(tmp_path):
    with pytest.raises(IOError):
        formats.load_april_group(tmp_path / "nope.json")
    p = tmp_path / "bad.json"
    p.write_text(json.dumps({"markers": {}}))
    with pytest.raises(ValueError):
        formats.load_april_group(p)
    p.write_text(json.dumps({"tags": {"3": {"size": 0.02, "extrinsics": [0, 0, 0]}}}))
    with pytest.raises(ValueError):
        formats.load_april_group(p)
    p.write_text(json.dumps({"tags": {"x": {"size": 0.02, "extrinsics": [0] * 6}}}))
    with pytest.raises(ValueError):
        formats.load_april_group(p)


def test_camera_params_round_trip_and_errors(tmp_path):
    K = syn.camera_matrix(1280, 720)
    d = np.array([[0.05, -0.1, 1e-3, -1e-3, 0.02]])
    p = tmp_path / "CameraParams.npz"
    formats.save_camera_params(p, K, d, rvecs=np.zeros((3, 3, 1)), tvecs=np.ones((3, 3, 1)))
    with np.load(p) as f:
        assert sorted(f.files) == ["dist", "mtx", "rvecs", "tvecs"]        # calibrate_camera.py:110-114
    mtx, dist, rv, tv = formats.load_camera_params(p)
    assert mtx.dtype == np.float64 and np.array_equal(mtx, K) and dist.shape == (1, 5) and np.array_equal(dist, d)
    assert rv.shape == (3, 3, 1) and tv.shape == (3, 3, 1)
    np.savez(tmp_path / "a.npz", mtx=K)
    with pytest.raises(ValueError):
        formats.load_camera_params(tmp_path / "a.npz")
    np.savez(tmp_path / "b.npz", mtx=np.eye(4), dist=d)
    with pytest.raises(ValueError):
        formats.load_camera_params(tmp_path / "b.npz")
    np.savez(tmp_path / "c.npz", mtx=K, dist=np.zeros(3))
    with pytest.raises(ValueError):
        formats.load_camera_params(tmp_path / "c.npz")


def test_detection_record_and_table():
    seq = syn.Sequence(640, 480, n_tags=12, n_frames=2, seed=3)
    tag_ids = [int(t) for t in seq.group["tags"].keys()]
    dets = _detections(seq, 0, drop={tag_ids[2]}, weak={tag_ids[5]}, extra_unknown=True)
    d0 = dets[0]
    assert d0._fields == ("tag_family", "tag_id", "hamming", "goodness", "decision_margin", "homography", "center", "corners")
    assert d0.corners.shape == (4, 2) and d0.center.shape == (2,) and "Tag Id" in d0.tostring(indent=2)
    corners, mask, n = formats.detections_to_corner_table(dets, tag_ids)
    assert corners.dtype == np.float32 and corners.shape == (48, 2) and mask.dtype == np.uint8 and n == 10
    m = mask.reshape(12, 4)
    assert not m[2].any() and not m[5].any() and m[[0, 1, 3, 4, 6, 7, 8, 9, 10, 11]].all()
    assert np.array_equal(corners[mask == 1], seq.corners(0)[mask == 1])
    assert not corners[mask == 0].any()


def test_replay_equals_direct_feed(tmp_path, oracle):
    from oracle import cv2_shim
    seq = syn.Sequence(640, 480, n_tags=12, n_frames=6, seed=3)
    tag_ids = [int(t) for t in seq.group["tags"].keys()]
    frames = [_detections(seq, k, drop={tag_ids[k % 12]}, weak={tag_ids[(k + 4) % 12]}) for k in range(6)]
    frames[3] = frames[3][:1]                                  # a frame with < 2 tags (detect_pose.py:494)
    rec = tmp_path / "detections.npz"
    formats.save_detections(rec, frames)
    back = formats.load_detections(rec)
    assert [len(f) for f in back] == [len(f) for f in frames]
    for fa, fb in zip(frames, back):
        for a, b in zip(fa, fb):
            assert a.tag_id == b.tag_id and a.decision_margin == b.decision_margin and np.array_equal(a.corners, b.corners)
    (tmp_path / "april_group.json").write_text(json.dumps(seq.group))
    cam = tmp_path / "CameraParams.npz"
    formats.save_camera_params(cam, seq.K, np.zeros(5))
    gray = np.zeros((480, 640), np.uint8)

    replay = PoseDetector.from_files(LOG, cam, True, cv=cv2_shim.make_cv2(), detector=rec, april_group=tmp_path / "april_group.json")
    direct = PoseDetector.from_files(LOG, cam, True, cv=cv2_shim.make_cv2(), april_group=tmp_path / "april_group.json")
    poses = 0
    for k in range(6):
        il, ol, ids = replay._obtain_detections(gray)
        assert ids == [d.tag_id for d in frames[k] if d.decision_margin >= 50]
        replay._estimate_pose(il, ol)
        # the same frame fed by hand
        il2 = [d.corners.reshape(1, 4, 2) for d in frames[k] if d.decision_margin >= 50]
        ol2 = [direct.transform_marker_corners(direct.get_initial_pts(direct.extrinsics[d.tag_id][0]),
                                               (direct.extrinsics[d.tag_id][2], direct.extrinsics[d.tag_id][1]))
               for d in frames[k] if d.decision_margin >= 50]
        direct._estimate_pose(il2, ol2)
        if k == 3:
            continue
        assert np.array_equal(replay.last_pose[0], direct.last_pose[0]) and np.array_equal(replay.last_pose[1], direct.last_pose[1])
        # noise-free corners: the pose is the generator's
        assert np.allclose(replay.last_pose[0].ravel(), seq.rvecs[k], atol=5e-4) and np.allclose(replay.last_pose[1].ravel(), seq.tvecs[k], atol=5e-4)
        poses += 1
    assert poses == 5
    with pytest.raises(IndexError):
        replay._obtain_detections(gray)
