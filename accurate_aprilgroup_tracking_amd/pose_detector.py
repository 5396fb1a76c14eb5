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

    def __init__(self, logger, mtx, dist, enhance_ape, cv=None, detector=None):
        """`detector(gray) -> iterable of objects with .tag_id, .corners (4,2), .decision_margin`
        stands in for apriltag.Detector(...).detect (detect_pose.py:368-371)."""
        TransformHelper.__init__(self, logger, mtx, dist, cv=cv)
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
        self.projected_points = None
        # LK tracking state (north-star): previous gray frame and the corners seen in it
        self._prev_gray = None
        self._prev_corners = None       # (N,2) float32, rows ordered as _prev_ids x 4
        self._prev_ids = None

    @classmethod
    def from_files(cls, logger, camera_params, enhance_ape=True, cv=None, detector=None, april_group=None):
        """Build from the reference's on-disk files: `CameraParams.npz` (calibrate_camera.py:107-123) and,
        optionally, an `april_group.json` somewhere else than DIRPATH/JSON_FILE (detect_pose.py:54-55).
        `detector` may be a recorded-detections .npz (formats.ReplayDetector) to replay a session."""
        import os
        from . import formats
        mtx, dist, _, _ = formats.load_camera_params(camera_params)
        if isinstance(detector, (str, os.PathLike)):
            detector = formats.ReplayDetector(detector)
        if april_group is None:
            return cls(logger, mtx, dist, enhance_ape, cv=cv, detector=detector)
        folder, name = os.path.split(os.fspath(april_group))
        sub = type(cls.__name__, (cls,), {"DIRPATH": folder or ".", "JSON_FILE": name})
        return sub(logger, mtx, dist, enhance_ape, cv=cv, detector=detector)

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

    def _detect_and_get_pose(self, frame):
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
