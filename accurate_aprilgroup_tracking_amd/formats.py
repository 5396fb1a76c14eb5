"""On-disk / wire formats either side of the hot path.

* ``april_group.json``   -- the AprilGroup model the reference reads at
  aprilgroup_pose_estimation/detect_pose.py:105-145 (schema at :122-130):
  ``{"tags": {"<id>": {"size": s, "extrinsics": [tx, ty, tz, rx, ry, rz]}}}``.
* ``CameraParams.npz``   -- the calibration file written and read at
  calibration/calibrate_camera.py:107-123: keys ``mtx, dist, rvecs, tvecs``.
* detections             -- the swatbotics ``apriltag.Detection`` records the reference consumes at
  detect_pose.py:389-437 (``.corners`` 4x2, ``.tag_id``, ``.decision_margin`` >= 50) and a compact
  ``.npz`` recording of them so that a real session can be replayed without the detector.

Host-side only; nothing here touches the GPU.  ``detections_to_corner_table`` produces the dense
``[4T, 2]`` corner table + mask that the device state machine takes (``agt_estimate_pose``).
"""
import collections
import json
import os

import numpy as np

DECISION_MARGIN = 50.0            # detect_pose.py:391

_DetectionBase = collections.namedtuple(
    "Detection", "tag_family tag_id hamming goodness decision_margin homography center corners")


class Detection(_DetectionBase):
    """Field-compatible stand-in for swatbotics ``apriltag.Detection`` (same names, same order)."""
    __slots__ = ()

    def tostring(self, values=None, indent=0):
        """Same layout idea as the swatbotics record dump used by the reference's logger (detect_pose.py:399)."""
        pad = " " * indent
        rows = []
        for name in self._fields:
            text = str(getattr(self, name))
            if "\n" in text:
                text = text.replace("\n", "\n" + pad + " " * 18)
            rows.append("%s%-16s: %s" % (pad, name.replace("_", " ").title(), text))
        return "\n".join(rows)


def make_detection(tag_id, corners, decision_margin=100.0, tag_family=b"tag36h11", hamming=0, goodness=0.0,
                   homography=None, center=None):
    c = np.asarray(corners, dtype=np.float64).reshape(4, 2)
    return Detection(tag_family, int(tag_id), int(hamming), float(goodness), float(decision_margin),
                     np.eye(3) if homography is None else np.asarray(homography, np.float64).reshape(3, 3),
                     c.mean(axis=0) if center is None else np.asarray(center, np.float64).reshape(2), c)


# ------------------------------------------------------------------ april_group.json
def load_april_group(path):
    """-> {tag_id: [size, tvec (3,1) f32, rvec (3,1) f32]} exactly as PoseDetector.get_extrinsics builds it
    (detect_pose.py:122-137: translation = first three, rotation = last three, float32 columns)."""
    try:
        with open(path, "r") as f:
            data = json.load(f)
    except IOError as e:
        raise IOError("The filepath: {} does not exist.".format(path)) from e
    if not isinstance(data, dict) or "tags" not in data or not isinstance(data["tags"], dict):
        raise ValueError("%s: missing the top-level 'tags' object" % path)
    out = {}
    for key, tag in data["tags"].items():
        try:
            tag_id = int(key)
            size = tag["size"]
            ext = tag["extrinsics"]
        except (KeyError, TypeError, ValueError) as e:
            raise ValueError("%s: tag %r needs 'size' and 'extrinsics'" % (path, key)) from e
        if len(ext) < 6:
            raise ValueError("%s: tag %r: 'extrinsics' holds %d numbers, 6 expected" % (path, key, len(ext)))
        out[tag_id] = [size, np.array(ext[:3], dtype=np.float32).reshape(3, 1), np.array(ext[-3:], dtype=np.float32).reshape(3, 1)]
    return out


def save_april_group(path, extrinsics):
    """Inverse of load_april_group; accepts {id: [size, tvec, rvec]} or {id: {"size":, "extrinsics":}}."""
    tags = {}
    for tag_id, v in extrinsics.items():
        if isinstance(v, dict):
            size, ext = v["size"], list(v["extrinsics"])
        else:
            size, tvec, rvec = v[:3]
            ext = [float(x) for x in np.asarray(tvec).ravel()] + [float(x) for x in np.asarray(rvec).ravel()]
        if len(ext) != 6:
            raise ValueError("tag %r: 6 extrinsic values expected" % (tag_id,))
        tags[str(int(tag_id))] = {"size": float(size), "extrinsics": ext}
    with open(path, "w") as f:
        json.dump({"tags": tags}, f, indent=1)


# ------------------------------------------------------------------ CameraParams.npz
def load_camera_params(path):
    """-> (mtx (3,3) f64, dist (1,k) f64, rvecs, tvecs): the four keys of calibrate_camera.py:118-123.
    rvecs/tvecs (per calibration view) are optional here -- the pose path never reads them."""
    with np.load(path) as f:
        missing = [k for k in ("mtx", "dist") if k not in f.files]
        if missing:
            raise ValueError("%s: missing key(s) %s" % (path, ", ".join(missing)))
        mtx = np.asarray(f["mtx"], dtype=np.float64)
        dist = np.asarray(f["dist"], dtype=np.float64)
        rvecs = f["rvecs"] if "rvecs" in f.files else None
        tvecs = f["tvecs"] if "tvecs" in f.files else None
    if mtx.shape != (3, 3):
        raise ValueError("%s: mtx has shape %s, (3, 3) expected" % (path, mtx.shape))
    if dist.size not in (4, 5, 8, 12, 14):
        raise ValueError("%s: dist holds %d coefficients (4, 5, 8, 12 or 14 expected)" % (path, dist.size))
    return mtx, dist.reshape(1, -1), rvecs, tvecs


def save_camera_params(path, mtx, dist, rvecs=None, tvecs=None):
    """np.savez with the reference's key names (calibrate_camera.py:110-114)."""
    np.savez(path, mtx=np.asarray(mtx, np.float64), dist=np.asarray(dist, np.float64).reshape(1, -1),
             rvecs=np.zeros((0, 3, 1)) if rvecs is None else np.asarray(rvecs),
             tvecs=np.zeros((0, 3, 1)) if tvecs is None else np.asarray(tvecs))


# ------------------------------------------------------------------ detection recordings
def save_detections(path, frames):
    """frames: list (one entry per frame) of lists of Detection-like objects -> compact .npz."""
    fi, ids, dm, ham, cor, cen, hom = [], [], [], [], [], [], []
    for k, dets in enumerate(frames):
        for d in dets:
            fi.append(k); ids.append(int(d.tag_id)); dm.append(float(d.decision_margin))
            ham.append(int(getattr(d, "hamming", 0)))
            c = np.asarray(d.corners, np.float64).reshape(4, 2)
            cor.append(c)
            cen.append(np.asarray(getattr(d, "center", c.mean(axis=0)), np.float64).reshape(2))
            hom.append(np.asarray(getattr(d, "homography", np.eye(3)), np.float64).reshape(3, 3))
    np.savez(path, n_frames=np.int64(len(frames)), frame=np.asarray(fi, np.int64), tag_id=np.asarray(ids, np.int64),
             decision_margin=np.asarray(dm, np.float64), hamming=np.asarray(ham, np.int64),
             corners=np.asarray(cor, np.float64).reshape(-1, 4, 2), center=np.asarray(cen, np.float64).reshape(-1, 2),
             homography=np.asarray(hom, np.float64).reshape(-1, 3, 3))


def load_detections(path):
    """-> list (per frame) of lists of Detection, in recorded order."""
    with np.load(path) as f:
        n = int(f["n_frames"])
        frames = [[] for _ in range(n)]
        for k, tid, dm, ham, c, ce, hm in zip(f["frame"], f["tag_id"], f["decision_margin"], f["hamming"], f["corners"],
                                             f["center"], f["homography"]):
            if not 0 <= k < n:
                raise ValueError("%s: detection refers to frame %d of %d" % (path, k, n))
            frames[int(k)].append(make_detection(tid, c, dm, hamming=ham, center=ce, homography=hm))
    return frames


class ReplayDetector:
    """Callable with the detector contract PoseDetector expects (``detector(gray) -> detections``):
    hands out the recorded detections frame by frame, ignoring the image."""

    def __init__(self, frames):
        self.frames = load_detections(frames) if isinstance(frames, (str, os.PathLike)) else list(frames)
        self.cursor = 0

    def __call__(self, gray=None):
        if self.cursor >= len(self.frames):
            raise IndexError("recording exhausted after %d frames" % len(self.frames))
        dets = self.frames[self.cursor]
        self.cursor += 1
        return dets

    def rewind(self):
        self.cursor = 0


def detections_to_corner_table(detections, tag_ids, decision_margin=DECISION_MARGIN):
    """Dense input of the device state machine: (corners f32 [4T,2], mask u8 [4T], n_tags) with tag t of
    `tag_ids` (the order of the model's object points, PoseDetector.get_all_points) in rows 4t..4t+3.
    Detections under the decision margin (detect_pose.py:391) or of unknown tags are dropped; for a tag
    reported twice the last record wins."""
    row = {int(t): i for i, t in enumerate(tag_ids)}
    corners = np.zeros((4 * len(row), 2), np.float32)
    mask = np.zeros(4 * len(row), np.uint8)
    for d in detections:
        if d.decision_margin < decision_margin or int(d.tag_id) not in row:
            continue
        i = row[int(d.tag_id)]
        corners[4 * i:4 * i + 4] = np.asarray(d.corners, np.float64).reshape(4, 2)
        mask[4 * i:4 * i + 4] = 1
    return corners, mask, int(mask.sum()) // 4
