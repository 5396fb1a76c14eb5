"""Host-side mirror of the reference's `PoseDetector`
(/root/reference/aprilgroup_tracking/aprilgroup_pose_estimation/detect_pose.py:28-619),
restricted to the per-frame pose path: model loading (:105-145, :185-227), the
`_estimate_pose` state machine (:467-574) with its motion model (:229-349), and the
north-star LK corner tracking that fills the hole at :573-574.

Every cv2 call goes through `self.cv`, by default the HIP-backed `cv_hip` module, so the
numbers come from the gfx950 kernels; without the HIP library or a GPU construction fails
loudly.  Camera capture, drawing and the AprilTag detector itself are out of scope
(SURVEY.md section 2 rows 4-6): a detector is injected as a callable, drawing is a no-op that
records the projected points.

Two backends:

* ``backend="cv"`` (default): every cv2 call of the reference goes through `self.cv` (cv_hip), one
  synchronous host round trip per call -- the literal drop-in for the cv2 module.
* ``backend="stream"``: the reference's per-frame entry point on the DEVICE-RESIDENT path.
  `_detect_and_get_pose(frame)` = one pinned host-to-device copy of the frame -> (BGR: fused
  undistort / gray / crop kernel) -> `agt_track_frame` (LK + solvePnP + gate + motion model, state in HBM) or
  `agt_track_frame_detected` when the injected detector returned >= 2 tags -> one 128-byte device-to-host copy of
  the frame's record.  `extrinsic_guess`, `prev_transform` and the velocity buffers are materialised lazily from
  `agt_tracker_state_read` when somebody looks at them.  `step(raw_frame)` is `process_frame` +
  `_detect_and_get_pose` of the reference's loop (detect_pose.py:669-681) with the raw frame uploaded once.

For many independent streams use `tracker.StreamTracker`, which keeps this same state
machine resident on the device.
"""
import json
from copy import deepcopy
from pathlib import Path

import numpy as np

from .geometry import TransformHelper


class PoseDetector(TransformHelper):
    DIRPATH = 'aprilgroup_tracking/aprilgroup_pose_estimation'
    JSON_FILE = 'april_group.json'

    MIN_TAGS = 2            # detect_pose.py:494
    ERROR_GATE_PX = 2       # detect_pose.py:539
    DECISION_MARGIN = 50    # detect_pose.py:389

    def __init__(self, logger, mtx, dist, enhance_ape, cv=None, detector=None, backend="cv"):
        """`detector(gray) -> iterable of objects with .tag_id, .corners (4,2), .decision_margin`
        stands in for apriltag.Detector(...).detect (detect_pose.py:368-371)."""
        TransformHelper.__init__(self, logger, mtx, dist, cv=cv)
        if backend not in ("cv", "stream"):
            raise ValueError("backend must be 'cv' or 'stream'")
        self.backend = backend
        self._dev = None                # stream backend: _DeviceStream, created at the first frame (its size fixes the context)
        self.img = None
        self.draw_frame = None
        self.prev_transform = (None, None)
        self.extrinsic_guess = (None, None)
        self.rot_velocities = []
        self.tran_velocities = []
        self.enhance_ape = enhance_ape
        self.detector = detector
        self.extrinsics = self.get_extrinsics()
        self.all_objpts = self.get_all_points(self.extrinsics)
        # results of the latest frame (the reference only logs/draws them)
        self.last_pose = (None, None)
        self.last_error = None
        self._pp, self._pp_pending = None, None
        # LK tracking state (north-star): previous gray frame and the corners seen in it
        self._prev_gray = None
        self._prev_corners = None       # (N,2) float32, rows ordered as _prev_ids x 4
        self._prev_ids = None

    # ---- the four attributes _estimate_pose mutates (detect_pose.py:74-83).  backend "cv": plain attributes.  backend
    # "stream": they live in HBM (AgtTrackState) and are read back -- one synchronising 376-byte copy -- only when looked at.
    def _state(self, name):
        if self._dev is not None:
            self._dev.refresh(self)
        return self.__dict__["_st_" + name]

    extrinsic_guess = property(lambda self: self._state("extrinsic_guess"),
                               lambda self, v: self._set_state("extrinsic_guess", v))
    prev_transform = property(lambda self: self._state("prev_transform"),
                              lambda self, v: self._set_state("prev_transform", v))
    rot_velocities = property(lambda self: self._state("rot_velocities"),
                              lambda self, v: self._set_state("rot_velocities", v))
    tran_velocities = property(lambda self: self._state("tran_velocities"),
                               lambda self, v: self._set_state("tran_velocities", v))

    @property
    def projected_points(self):
        """projectPoints(all_objpts) at the latest ACCEPTED pose (detect_pose.py:441-465 minus the drawing).  backend
        "stream" computes it when somebody looks (the drawing is out of scope; the per-frame path does not pay for it)."""
        if self._pp_pending is not None:
            pose, self._pp_pending = self._pp_pending, None
            self._pp, _ = self.cv.projectPoints(self.all_objpts, pose[0], pose[1], self.mtx, self.dist)
        return self._pp

    @projected_points.setter
    def projected_points(self, value):
        self._pp, self._pp_pending = value, None

    def configure_stream(self, width, height):
        """backend "stream": fix the (processed) frame size before the first frame, e.g. to call `_estimate_pose` with
        detector-supplied correspondences only"""
        self._stream().configure(self, width, height)

    def _set_state(self, name, value):
        if self._dev is not None and self._dev.started:
            raise AttributeError("backend='stream': %s lives on the device; reset the detector instead of assigning it" % name)
        self.__dict__["_st_" + name] = value

    @classmethod
    def from_files(cls, logger, camera_params, enhance_ape=True, cv=None, detector=None, april_group=None, backend="cv"):
        """Build from the reference's on-disk files: `CameraParams.npz` (calibrate_camera.py:107-123) and,
        optionally, an `april_group.json` somewhere else than DIRPATH/JSON_FILE (detect_pose.py:54-55).
        `detector` may be a recorded-detections .npz (formats.ReplayDetector) to replay a session."""
        import os
        from . import formats
        mtx, dist, _, _ = formats.load_camera_params(camera_params)
        if isinstance(detector, (str, os.PathLike)):
            detector = formats.ReplayDetector(detector)
        if april_group is None:
            return cls(logger, mtx, dist, enhance_ape, cv=cv, detector=detector, backend=backend)
        folder, name = os.path.split(os.fspath(april_group))
        sub = type(cls.__name__, (cls,), {"DIRPATH": folder or ".", "JSON_FILE": name})
        return sub(logger, mtx, dist, enhance_ape, cv=cv, detector=detector, backend=backend)

    # ------------------------------------------------------------------ model (detect_pose.py:105-227)
    def get_extrinsics(self):
        filepath = Path(self.DIRPATH) / self.JSON_FILE
        try:
            with open(filepath, "r") as handle:
                data = json.load(handle)
        except IOError as file_error:
            raise IOError("The filepath: {} does not exist.".format(filepath)) from file_error
        extrinsics = {}
        for key, tag in data['tags'].items():
            tvec = np.array(tag['extrinsics'][:3], dtype=np.float32).reshape((3, 1))
            rvec = np.array(tag['extrinsics'][-3:], dtype=np.float32).reshape((3, 1))
            self.add_values_in_dict(extrinsics, int(key), [tag['size'], tvec, rvec])
        self.logger.info('Successfully Loaded AprilGroup Extrinsics!')
        return extrinsics

    def get_all_points(self, extrinsics):
        if not any(extrinsics):
            raise ValueError("The extrinsic matrix must be supplied.")
        corners = [self.transform_marker_corners(self.get_initial_pts(size), (rvec, tvec))
                   for size, tvec, rvec in (extrinsics[k][:3] for k in extrinsics)]
        self.logger.info('Successfully Obtained Aprilgroup Object Points!')
        return np.array(corners).reshape(-1, 3)

    # ------------------------------------------------------------------ motion model (:229-349)
    def _update_buffers(self, rot_vel, tran_vel, buf_size=2):
        if not np.all(rot_vel) or not np.all(tran_vel):
            raise ValueError("The rotational and translation velocities cannot be empty.")
        self.rot_velocities.append(rot_vel)
        self.tran_velocities.append(tran_vel)
        if len(self.rot_velocities) > buf_size:
            del self.rot_velocities[0]
            del self.tran_velocities[0]

    def get_pose_vel_acc(self, curr_transform, prev_transform):
        prev_rmat = self.cv.Rodrigues(prev_transform[0])[0]
        curr_rmat = self.cv.Rodrigues(curr_transform[0])[0]
        tran_vel = self.get_relative_trans(curr_rmat, curr_transform[1], prev_transform[1])
        rot_vel = self.get_relative_rot(prev_rmat, curr_rmat)
        self._update_buffers(rot_vel, tran_vel)
        if len(self.tran_velocities) < 2:
            return False, tran_vel, rot_vel, 0.0, 0.0
        tran_acc = self.get_relative_trans(self.rot_velocities[-1], self.tran_velocities[-1], self.tran_velocities[-2])
        rot_acc = self.get_relative_rot(self.rot_velocities[-2], self.rot_velocities[-1])
        return True, tran_vel, rot_vel, tran_acc, rot_acc

    def apply_vel_acc(self, transformation, tran_vel, tran_acc, rot_vel, rot_acc):
        half_acc = self.euler_angles_to_rotation_matrix(self.rotation_matrix_to_euler_angles(rot_acc) / 2)
        rmat = self.cv.Rodrigues(transformation[0])[0]
        pose_m = self.get_extrinsic_matrix(rmat, transformation[1])
        vel_m = self.get_extrinsic_matrix(rot_vel, tran_vel)
        acc_m = self.get_extrinsic_matrix(half_acc, 0.5 * tran_acc)
        rmat_pred, tvec_pred = self.get_rmat_tvec(acc_m @ vel_m @ pose_m)
        return self.cv.Rodrigues(rmat_pred)[0], tvec_pred

    # ------------------------------------------------------------------ detection front-end (:351-439)
    def _obtain_detections(self, gray):
        """Same output contract as the reference: lists of (1,4,2) image points, (4,3) object
        points and tag ids for detections with decision_margin >= 50."""
        if self.detector is None:
            raise RuntimeError("no AprilTag detector was injected (the swatbotics detector is out of scope)")
        img_list, obj_list, ids = [], [], []
        if self.mtx is None:
            return img_list, obj_list, ids
        for det in self.detector(gray):
            if det.decision_margin < self.DECISION_MARGIN:
                continue
            size, tvec, rvec = self.extrinsics[det.tag_id][:3]
            img_list.append(np.asarray(det.corners).reshape(1, 4, 2))
            obj_list.append(self.transform_marker_corners(self.get_initial_pts(size), (rvec, tvec)))
            ids.append(det.tag_id)
        return img_list, obj_list, ids

    def _project_draw_points(self, transformation):
        """detect_pose.py:441-465 minus the drawing: keeps the projected model corners."""
        self.projected_points, _ = self.cv.projectPoints(self.all_objpts, transformation[0], transformation[1],
                                                         self.mtx, self.dist)

    # ------------------------------------------------------------------ ★ the state machine (:467-574)
    def _estimate_pose(self, imgpoints_arr, objpoints_arr):
        if self.backend == "stream":
            return self._stream().estimate(self, imgpoints_arr, objpoints_arr)
        prev_snapshot = deepcopy(self.prev_transform)        # solvePnP overwrites aliased guess arrays
        self.last_pose, self.last_error = (None, None), None
        if not (imgpoints_arr and objpoints_arr and len(imgpoints_arr) >= self.MIN_TAGS):
            self.extrinsic_guess = (None, None)
            return
        obj = np.array(objpoints_arr, dtype=np.float32).reshape(-1, 3)
        img = np.array(imgpoints_arr, dtype=np.float32).reshape(-1, 2)
        guided = self.extrinsic_guess[0] is not None and self.enhance_ape
        if guided:
            ok, rvec, tvec = self.cv.solvePnP(obj, img, self.mtx, self.dist, self.extrinsic_guess[0],
                                              self.extrinsic_guess[1], True, flags=self.cv.SOLVEPNP_ITERATIVE)
        else:
            ok, rvec, tvec = self.cv.solvePnP(obj, img, self.mtx, self.dist, flags=self.cv.SOLVEPNP_ITERATIVE)
        if not ok:
            return
        pose = (rvec, tvec)
        self.last_pose = pose
        self.last_error = self.get_reprojection_error(obj, img, pose)
        if not self.last_error < self.ERROR_GATE_PX:
            self.extrinsic_guess = (None, None)
            return
        self._project_draw_points(pose)
        if not guided:
            self.extrinsic_guess = pose
        else:
            good, tran_vel, rot_vel, tran_acc, rot_acc = self.get_pose_vel_acc(pose, prev_snapshot)
            if good:
                self.extrinsic_guess = self.apply_vel_acc(prev_snapshot, tran_vel, tran_acc, rot_vel, rot_acc)
        self.prev_transform = pose

    # ------------------------------------------------------------------ per-frame entry (:576-619)
    def undistort_frame(self, frame):
        """detect_pose.py:147-183: optimal new camera matrix (alpha = 1), undistort, crop to the ROI.
        (As in the reference, the ORIGINAL mtx/dist keep being used by solvePnP afterwards.)"""
        height, width = frame.shape[:2]
        new_camera_matrix, roi = self.cv.getOptimalNewCameraMatrix(self.mtx, self.dist, (width, height), 1, (width, height))
        dst = self.cv.undistort(frame, self.mtx, self.dist, None, new_camera_matrix)
        x_val, y_val, width, height = roi
        return dst[y_val:y_val + height, x_val:x_val + width]

    def process_frame(self, frame):
        """detect_pose.py:611-619"""
        if self.dist is not None:
            frame = self.undistort_frame(frame)
        return frame

    def _to_gray(self, frame):
        if frame.ndim == 2:
            return frame
        return self.cv.cvtColor(np.ascontiguousarray(frame), self.cv.COLOR_BGR2GRAY)       # detect_pose.py:602

    def track_corners(self, gray):
        """North-star step: carry the previous frame's corners into `gray` with pyramidal LK
        (21x21 window, 3 levels, COUNT+EPS (30, 0.01)) and return them per tag."""
        nxt, status, _ = self.cv.calcOpticalFlowPyrLK(self._prev_gray, gray, self._prev_corners, None,
                                                      winSize=(21, 21), maxLevel=2)
        nxt = nxt.reshape(-1, 4, 2); ok = status.reshape(-1, 4).all(axis=1)
        img_list, obj_list, ids = [], [], []
        for t, tag_id in enumerate(self._prev_ids):
            if not ok[t]:
                continue
            size, tvec, rvec = self.extrinsics[tag_id][:3]
            img_list.append(nxt[t].reshape(1, 4, 2).astype(np.float64))
            obj_list.append(self.transform_marker_corners(self.get_initial_pts(size), (rvec, tvec)))
            ids.append(tag_id)
        return img_list, obj_list, ids

    def frame_buffer(self, shape):
        """backend "stream": a numpy view of PINNED host memory of the given frame shape ((H, W) gray or (H, W, 3) BGR).  A
        capture loop that reads into it (`cap.read(buf)`) and hands it to `_detect_and_get_pose` / `step` saves the host-side
        copy into the staging buffer that any other array costs (about 0.1 ms for a 1280x720 BGR frame)."""
        if self.backend != "stream":
            raise RuntimeError("frame_buffer() belongs to backend='stream'")
        return self._stream().frame_buffer(tuple(shape))

    def _stream(self):
        if self._dev is None:
            self._dev = _DeviceStream(self)
        return self._dev

    def step(self, raw_frame):
        """The body of the reference's capture loop (detect_pose.py:669-681): `frame = self.process_frame(frame)`
        followed by `self._detect_and_get_pose(frame)`.  backend "stream" uploads the RAW frame once and undistorts,
        converts and crops it on the device (no processed frame ever exists on the host; self.img stays the raw frame)."""
        if self.backend == "stream":
            self.img = raw_frame
            return self._stream().frame(self, raw_frame, raw=True)
        return self._detect_and_get_pose(self.process_frame(raw_frame))

    def reset_stream(self):
        """backend "stream": forget pose state and tracked corners (a new sequence); also the recovery from a chain fault"""
        if self._dev is not None:
            self._dev.reset(self)

    def _detect_and_get_pose(self, frame):
        if self.backend == "stream":
            self.img = frame
            return self._stream().frame(self, frame, raw=False)
        self.img = frame
        gray = self._to_gray(frame)
        img_list, obj_list, ids = ([], [], [])
        if self.detector is not None:
            img_list, obj_list, ids = self._obtain_detections(gray)
        if len(img_list) < self.MIN_TAGS and self._prev_gray is not None and self._prev_ids:
            img_list, obj_list, ids = self.track_corners(gray)      # fills detect_pose.py:573-574
        self._estimate_pose(img_list, obj_list)
        if ids:
            self._prev_gray = gray
            self._prev_corners = np.array(img_list, dtype=np.float32).reshape(-1, 2)
            self._prev_ids = list(ids)



class _DeviceStream:
    """backend "stream" of PoseDetector: one stream of tracker.StreamTracker plus the pinned staging buffers of the three
    copies a frame costs (frame up, corner table up when the detector spoke, 128-byte record down)."""

    def __init__(self, det):
        import ctypes as C
        import torch
        from . import hiplib as H
        from . import cv_hip
        cv_hip._require_gpu()
        self.C, self.torch, self.H = C, torch, H
        self.trk = None
        self.started = False
        self.n = det.all_objpts.shape[0]
        self.tag_ids = list(det.extrinsics)
        obj32 = det.all_objpts.astype(np.float32)
        # a detection's object points are the float32-identical rows of all_objpts: recognise the tag by them
        self.tag_of = {np.ascontiguousarray(obj32[4 * t:4 * t + 4]).tobytes(): t for t in range(self.n // 4)}
        self.dirty = False
        self.has_prev = False
        self.shape = None
        self._pin = {}                  # frame_buffer(): pinned arrays handed to the caller (the capture writes into them)
        self._stage = {}                # pinned staging of frames that live anywhere else (never the caller's buffers)
        self.roi = None

    # ---- set-up at the first frame
    def _setup(self, det, shape, raw):
        torch, H = self.torch, self.H
        from .tracker import StreamTracker
        h, w = shape[:2]
        self.src_hw = (h, w)
        self.undistort = bool(raw and det.dist is not None)
        if self.undistort:
            new_k, roi = det.cv.getOptimalNewCameraMatrix(det.mtx, det.dist, (w, h), 1, (w, h))
            self.roi = roi
            gw, gh = roi[2], roi[3]
        else:
            new_k, self.roi, gw, gh = None, (0, 0, w, h), w, h
        self.trk = StreamTracker(gw, gh, det.all_objpts, det.mtx, det.dist, n_streams=1, max_level=2, win=21,
                                 enhance_ape=det.enhance_ape, reproject=False, min_points=4 * det.MIN_TAGS,
                                 gate_px=float(det.ERROR_GATE_PX))
        self.trk.tag_gate(4)
        self.trk.pipeline(1)
        self.trk.reset()
        ctx = self.trk.ctx
        if self.undistort:
            ctx.undistort_init(det.mtx, det.dist, new_k, w, h)
        dev = self.trk.dev
        self.gpitch = (gw + 15) & ~15
        self.gray = [torch.zeros((1, gh, self.gpitch), dtype=torch.uint8, device=dev) for _ in range(4)]
        self.gi = 0
        self.bgr = None
        self.table_host = torch.zeros(self.n * 9, dtype=torch.uint8).pin_memory()
        self.table_dev = torch.zeros(self.n * 9, dtype=torch.uint8, device=dev)
        self.rec_host = torch.zeros(H.STATE_STRIDE, dtype=torch.float64).pin_memory()
        self.rec_np = self.rec_host.numpy()
        self.rec_dev = torch.zeros((1, H.STATE_STRIDE), dtype=torch.float64, device=dev)
        self.pts_np = self.table_host.numpy()[:self.n * 8].view(np.float32).reshape(self.n, 2)
        self.mask_np = self.table_host.numpy()[self.n * 8:]
        self.gw, self.gh = gw, gh
        self.shape = tuple(shape)
        self.raw = raw

    def frame_buffer(self, shape):
        buf = self._pin.get(shape)
        if buf is None:
            buf = self._pin[shape] = self.torch.zeros(shape, dtype=self.torch.uint8).pin_memory()
        return buf.numpy()

    def _pinned(self, frame):
        """-> pinned tensor holding `frame`: the caller's own frame_buffer() when the frame IS that buffer, otherwise a
        staging buffer of this object (a caller's capture buffer is never overwritten)"""
        t = self._pin.get(frame.shape)
        if t is not None and frame.ctypes.data == t.data_ptr() and frame.flags.c_contiguous:
            return t
        t = self._stage.get(frame.shape)
        if t is None:
            t = self._stage[frame.shape] = self.torch.zeros(frame.shape, dtype=self.torch.uint8).pin_memory()
        np.copyto(t.numpy(), frame)
        return t

    def _gray_as_bgr(self, frame):
        """a GRAY raw frame of a camera with lens distortion (process_frame undistorts whatever it gets, detect_pose.py:611-619):
        staged as three equal channels and sent down the BGR path -- remap acts per channel and BGR2GRAY of (g, g, g) is g
        ((1868 + 9617 + 4899) g + 2^13 >> 14), so the result is cv.undistort(gray) cropped to the ROI, byte for byte"""
        shape = frame.shape + (3,)
        t = self._stage.get(shape)
        if t is None:
            t = self._stage[shape] = self.torch.zeros(shape, dtype=self.torch.uint8).pin_memory()
        np.copyto(t.numpy(), frame[:, :, None])
        return t

    def _upload(self, dst, src_t, nbytes):
        ctx = self.trk.ctx
        self.H.check(ctx.L.agt_upload(ctx.h, self.C.c_void_p(dst.data_ptr()), self.C.c_void_p(src_t.data_ptr()), nbytes), "agt_upload")

    def _ingest(self, det, frame, raw):
        """frame (host) -> gray frame in HBM (one of four rotating buffers: the previous frame stays valid for LK)"""
        torch = self.torch
        frame = np.asarray(frame)
        if frame.dtype != np.uint8 or frame.ndim not in (2, 3) or (frame.ndim == 3 and frame.shape[2] != 3):
            raise ValueError("an 8-bit gray (H, W) or BGR (H, W, 3) frame is expected")
        if self.trk is None:
            self._setup(det, frame.shape, raw)
        elif tuple(frame.shape) != self.shape or raw != self.raw:
            raise ValueError("backend='stream': frame shape / kind changed from %r to %r; use a new detector" % (self.shape, frame.shape))
        ctx = self.trk.ctx
        ctx.use_current_stream()
        self.gi = (self.gi + 1) & 3
        g = self.gray[self.gi]
        if frame.ndim == 2 and not self.undistort:
            assert frame.shape == (self.gh, self.gw)
            pin = self._pinned(frame)
            if self.gpitch == self.gw:
                self._upload(g, pin, frame.size)
            else:       # pitch-padded rows: 2-D copy through torch (rare: widths that are not a multiple of 16)
                g[0, :, :self.gw].copy_(pin, non_blocking=True)
            return g[:, :, :self.gw]
        pin = self._gray_as_bgr(frame) if frame.ndim == 2 else self._pinned(frame)
        if self.bgr is None:
            self.bgr = torch.zeros((1,) + tuple(pin.shape), dtype=torch.uint8, device=self.trk.dev)
        self._upload(self.bgr, pin, pin.numel())
        ctx.preprocess_bgr(self.bgr, self.roi, undistort=self.undistort, out=g[:, :, :self.gw])
        return g[:, :, :self.gw]

    # ---- per-frame entry
    def frame(self, det, frame, raw):
        if det.detector is None and self.has_prev and self.trk is not None:
            return self._frame_tracked_one_call(det, frame, raw)
        g = self._ingest(det, frame, raw)
        img_list, obj_list, ids = [], [], []
        if det.detector is not None:
            gray_host = frame if (np.ndim(frame) == 2 and not self.undistort) else g[0].cpu().numpy()     # the PROCESSED frame
            img_list, obj_list, ids = det._obtain_detections(gray_host)
        if len(img_list) >= det.MIN_TAGS or not self.has_prev:
            self._table(img_list, obj_list)
            self._upload(self.table_dev, self.table_host, self.n * 9)
            pts = self.table_dev[:self.n * 8].view(self.torch.float32).view(1, self.n, 2)
            mask = self.table_dev[self.n * 8:].view(1, self.n)
            self.trk.step_detected(g, pts, mask, self.rec_dev)
            self.has_prev = self.has_prev or bool(ids)      # (the mirror's `if ids:`: a frame without any tag is no tracking source)
            self._finish(det)
        else:
            self.trk.step(g, self.rec_dev)        # LK from the previous frame's corners: fills detect_pose.py:573-574
            self.trk.join()
            self._finish(det, tracked=True)

    def _frame_tracked_one_call(self, det, frame, raw):
        """LK path without a detector: upload, pre-processing, agt_track_frame, join, record download and the wait in ONE
        foreign call (agt_track_host_frame)"""
        frame = np.asarray(frame)
        if tuple(frame.shape) != self.shape or raw != self.raw or frame.dtype != np.uint8:
            raise ValueError("backend='stream': frame shape / kind changed from %r to %r; use a new detector" % (self.shape, frame.shape))
        C, ctx = self.C, self.trk.ctx
        ctx.use_current_stream()
        self.gi = (self.gi + 1) & 3
        g = self.gray[self.gi]
        color = frame.ndim == 3 or self.undistort          # a gray raw frame that must be undistorted travels as (g, g, g)
        pin = self._gray_as_bgr(frame) if (color and frame.ndim == 2) else self._pinned(frame)
        if color and self.bgr is None:
            self.bgr = self.torch.zeros((1,) + tuple(pin.shape), dtype=self.torch.uint8, device=self.trk.dev)
        self.H.check(ctx.L.agt_track_host_frame(ctx.h, C.c_void_p(pin.data_ptr()), 3 if color else 1, self.src_hw[1], self.src_hw[0],
                                                C.c_void_p(self.bgr.data_ptr()) if color else None, int(self.undistort and color),
                                                self.roi[0] if color else 0, self.roi[1] if color else 0, C.c_void_p(g.data_ptr()), self.gpitch,
                                                None,      # (no device copy of the record: it arrives in host memory, polled)
                                                C.c_void_p(self.rec_host.data_ptr())), "agt_track_host_frame")
        self.trk._alive.append(g)
        if len(self.trk._alive) > self.trk._keep_frames:
            del self.trk._alive[0]
        self._finish(det, downloaded=True, tracked=True)

    def estimate(self, det, imgpoints_arr, objpoints_arr):
        """_estimate_pose(img_list, obj_list) with caller-supplied correspondences (no frame)"""
        if self.trk is None:
            raise RuntimeError("backend='stream': _estimate_pose needs the frame size; call configure(width, height) or feed a frame first")
        self.trk.ctx.use_current_stream()
        self._table(imgpoints_arr or [], objpoints_arr or [])
        self._upload(self.table_dev, self.table_host, self.n * 9)
        pts = self.table_dev[:self.n * 8].view(self.torch.float32).view(1, self.n, 2)
        mask = self.table_dev[self.n * 8:].view(1, self.n)
        self.trk.estimate_pose(pts, mask, self.rec_dev)
        self._finish(det)

    def configure(self, det, width, height):
        if self.trk is None:
            self._setup(det, (height, width), False)

    def _table(self, img_list, obj_list):
        self.pts_np[:] = 0.0
        self.mask_np[:] = 0
        for img, obj in zip(img_list, obj_list):
            t = self.tag_of.get(np.ascontiguousarray(np.asarray(obj, np.float32).reshape(4, 3)).tobytes())
            if t is None:
                raise ValueError("backend='stream': object points that are not a tag of april_group.json")
            self.pts_np[4 * t:4 * t + 4] = np.asarray(img, np.float32).reshape(4, 2)
            self.mask_np[4 * t:4 * t + 4] = 1

    def _finish(self, det, downloaded=False, tracked=False):
        H = self.H
        ctx = self.trk.ctx
        if not downloaded:
            H.check(ctx.L.agt_download(ctx.h, self.C.c_void_p(self.rec_host.data_ptr()), self.C.c_void_p(self.rec_dev.data_ptr()),
                                       8 * H.STATE_STRIDE), "agt_download")
        self.started = True
        self.dirty = True
        r = self.rec_np
        flags = int(r[H.ST_FLAGS])
        if tracked and int(r[H.ST_NTRACK]) == 0 and not (flags & H.TRK_CHAIN_TIMEOUT):
            # LK lost every tag: the reference (and the mirror's `if ids:`) keeps the OLDER frame as "previous" and tracks the next
            # frame from it (detect_pose.py:570-574).  The frame is taken back; its gray buffer is the next frame's.
            self.trk.rewind()
            self.gi = (self.gi - 1) & 3
        if flags & H.TRK_CHAIN_TIMEOUT:
            raise RuntimeError("backend='stream': the chained launch gave up waiting for this frame's corners; call reset_stream()")
        if flags & H.PNP_TOO_FEW:
            det.last_pose, det.last_error = (None, None), None
            return
        tdt = np.float32 if r[H.ST_TVEC_F32] else np.float64
        det.last_pose = (r[0:3].reshape(3, 1).copy(), r[3:6].astype(tdt).reshape(3, 1))
        det.last_error = float(r[H.ST_ERR])
        if r[H.ST_OK]:
            det._pp_pending = det.last_pose
        if flags & H.TRK_ZERO_VELOCITY:
            raise ValueError("The rotational and translation velocities cannot be empty.")      # detect_pose.py:236-237

    def refresh(self, det):
        """device tracker state -> the reference's attributes (only when somebody looks)"""
        if not self.dirty:
            return
        self.dirty = False
        s = self.trk.read_state()[0]
        d = det.__dict__
        if s.has_guess:
            g = np.array(s.guess[:])
            d["_st_extrinsic_guess"] = (g[:3].reshape(3, 1), g[3:].astype(np.float32 if s.guess_t_f32 else np.float64).reshape(3, 1))
        else:
            d["_st_extrinsic_guess"] = (None, None)
        if s.has_prev:
            p = np.array(s.prev[:])
            d["_st_prev_transform"] = (p[:3].reshape(3, 1), p[3:].astype(np.float32 if s.prev_t_f32 else np.float64).reshape(3, 1))
        else:
            d["_st_prev_transform"] = (None, None)
        d["_st_rot_velocities"] = [np.array(s.rot_vel[i][:]).reshape(3, 3) for i in range(s.n_vel)]
        d["_st_tran_velocities"] = [np.array(s.tran_vel[i][:]).reshape(3, 1) for i in range(s.n_vel)]

    def reset(self, det):
        if self.trk is not None:
            self.trk.reset()
        self.has_prev = False
        self.started = False
        d = det.__dict__
        d["_st_extrinsic_guess"] = (None, None); d["_st_prev_transform"] = (None, None)
        d["_st_rot_velocities"] = []; d["_st_tran_velocities"] = []
        self.dirty = False
