"""Host-side mirror of the reference's `TransformHelper`
(/root/reference/aprilgroup_tracking/aprilgroup_pose_estimation/transform_helper.py:14-259).

Same class name, method names, argument meaning, return dtypes and ValueError behaviour, so
code written against the reference helper keeps working; the two cv2 calls it makes
(Rodrigues at :87, projectPoints at :106) go through the injected `cv` backend, which by
default is the HIP-backed `cv_hip` module (no CPU fallback).
"""
from math import atan2, cos, sin, sqrt

import numpy as np

from .host_math import Rodrigues


class TransformHelper:
    """Object-point construction, SE(3) pack/unpack, relative motion, Euler conversions."""

    def __init__(self, logger, mtx, dist, cv=None):
        self.logger = logger
        self.mtx = mtx
        self.dist = dist
        if cv is None:
            from . import cv_hip as cv          # raises without the HIP library / a GPU
        self.cv = cv

    # -- transform_helper.py:30-38
    @staticmethod
    def add_values_in_dict(sample_dict, key, list_of_values):
        sample_dict.setdefault(key, []).extend(list_of_values)
        return sample_dict

    # -- transform_helper.py:41-63: corner order (-,-), (-,+), (+,+), (+,-), z = 0, float64 (4,3)
    @staticmethod
    def get_initial_pts(tagsize):
        h = tagsize / 2.0
        return np.array([[-h, -h, 0.0], [-h, h, 0.0], [h, h, 0.0], [h, -h, 0.0]])

    # -- transform_helper.py:66-96: pts @ R(rvec)^T + tvec (Rodrigues keeps the rvec's depth: f32 tags)
    @staticmethod
    def transform_marker_corners(object_pts, transformation):
        rvec, tvec = transformation
        if rvec.size == 0 or tvec.size == 0:
            raise ValueError('The transform rotation or translation: {} entered is empty'.format(transformation))
        rmat = Rodrigues(rvec)[0]
        return object_pts @ rmat.T + tvec.reshape(-1, 3)

    # -- transform_helper.py:98-121: mean over points of the L2 reprojection residual
    def get_reprojection_error(self, obj_points, img_points, transformation):
        projected, _ = self.cv.projectPoints(obj_points, transformation[0], transformation[1], self.mtx, self.dist)
        projected = projected.reshape(-1, 2)
        total = sum(np.linalg.norm(img_points[i] - projected[i]) for i in range(len(projected)))
        return total / len(projected)

    # -- transform_helper.py:123-148
    def get_extrinsic_matrix(self, rmat, tvec):
        try:
            top = np.hstack((rmat, tvec))
            return np.vstack((top, np.array([0, 0, 0, 1])))
        except ValueError as err:
            raise ValueError('The rotation matrix: {} or translation vector: {} entered are not in the right '
                             'format (3x3 matrix and 3x1 vector) or are zero.'.format(rmat, tvec)) from err

    # -- transform_helper.py:151-164: NOTE the float32 cast of the translation
    @staticmethod
    def get_rmat_tvec(extrinsic_mat):
        try:
            rot = extrinsic_mat[0:3, 0:3]
            tvec = np.array(extrinsic_mat[0:3, 3], dtype=np.float32).reshape(3, -1)
        except (ValueError, IndexError) as err:
            raise ValueError('The extrinsic matrix entered: {} is not a 4x4 matrix or is zero.'.format(extrinsic_mat)) from err
        return rot, tvec

    # -- transform_helper.py:167-189: R^T (t0 - t1)
    @staticmethod
    def get_relative_trans(rot_mat, tvec1, tvec0):
        try:
            return rot_mat.T @ (tvec0 - tvec1)
        except ValueError as err:
            raise ValueError('The vectors entered, tvec0: {} and tvec1: {} are either not the same size '
                             'or zero.'.format(tvec0, tvec1)) from err

    # -- transform_helper.py:192-212: R1^T R0
    @staticmethod
    def get_relative_rot(rmat0, rmat1):
        try:
            return rmat1.T @ rmat0
        except ValueError as err:
            raise ValueError('The matrices entered, r0: {} and r1: {} are either not the same size '
                             'or zero.'.format(rmat0, rmat1)) from err

    # -- transform_helper.py:215-236: R = Rz Ry Rx (the reference's stray print is not reproduced)
    @staticmethod
    def euler_angles_to_rotation_matrix(theta):
        cx, sx = cos(theta[0]), sin(theta[0])
        cy, sy = cos(theta[1]), sin(theta[1])
        cz, sz = cos(theta[2]), sin(theta[2])
        r_x = np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]])
        r_y = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
        r_z = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]])
        return np.dot(r_z, np.dot(r_y, r_x))

    # -- transform_helper.py:239-259
    @staticmethod
    def rotation_matrix_to_euler_angles(rmat):
        s_y = sqrt(rmat[0, 0] * rmat[0, 0] + rmat[1, 0] * rmat[1, 0])
        if not s_y < 1e-6:
            angles = (atan2(rmat[2, 1], rmat[2, 2]), atan2(-rmat[2, 0], s_y), atan2(rmat[1, 0], rmat[0, 0]))
        else:
            angles = (atan2(-rmat[1, 2], rmat[1, 1]), atan2(-rmat[2, 0], s_y), 0)
        return np.array(angles)
