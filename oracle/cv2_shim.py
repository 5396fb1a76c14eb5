"""A module object that answers for `cv2` with the CPU oracle (TEST INFRASTRUCTURE).

Used only by tests/golden/make_reference_fixtures.py to run the reference's OWN Python
(PoseDetector._estimate_pose and its motion model) in this container, where the real cv2
cannot be installed: solvePnP / projectPoints / Rodrigues answer from oracle/libcvoracle.so
with cv2's in-place, depth-preserving behaviour.  PARITY UNPINNED against real cv2.
"""
import types
from . import cvoracle


def make_cv2():
    m = types.ModuleType("cv2")
    m.SOLVEPNP_ITERATIVE = cvoracle.SOLVEPNP_ITERATIVE
    m.solvePnP = cvoracle.solvePnP
    m.projectPoints = lambda o, r, t, K, d: cvoracle.projectPoints(o, r, t, K, d)
    m.Rodrigues = cvoracle.Rodrigues
    m.calcOpticalFlowPyrLK = cvoracle.calcOpticalFlowPyrLK
    m.getOptimalNewCameraMatrix = cvoracle.getOptimalNewCameraMatrix
    m.undistort = cvoracle.undistort
    m.cvtColor = cvoracle.cvtColor
    m.COLOR_BGR2GRAY = cvoracle.COLOR_BGR2GRAY
    m.error = ValueError
    return m


def make_apriltag():
    m = types.ModuleType("apriltag")

    class DetectorOptions:
        def __init__(self, **kw):
            self.__dict__.update(kw)

    class Detection:
        pass

    m.DetectorOptions = DetectorOptions
    m.Detection = Detection
    return m
