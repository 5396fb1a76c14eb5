"""ctypes binding of the gfx950 C-ABI library (include/agt_hip.h -> libagt_hip.so).

The library is the product: there is NO CPU fallback.  If it is missing or cannot be
loaded this module raises immediately (ImportError-like RuntimeError) instead of
silently degrading -- build it with `python -c "import __graft_entry__ as g; g.build()"`
or `make -C accurate_aprilgroup_tracking_amd/csrc`.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libagt_hip.so")

AGT_OK = 0
ERRORS = {-1: "ARG", -2: "ALLOC", -3: "DIST", -4: "NPOINTS", -5: "HIP", -6: "UNSUPPORTED", -7: "STATE", -8: "CHAIN"}
MAX_LEVELS = 6
LK_USE_INITIAL_FLOW, LK_GET_MIN_EIGENVALS = 4, 8
TERM_COUNT, TERM_EPS = 1, 2
F32, F64 = 0, 1
INFO_OK, INFO_ITERS, INFO_NUSED, INFO_FLAGS = 0, 1, 2, 3
PNP_SINGULAR, PNP_PLANAR, PNP_TOO_FEW = 1, 2, 4
STATE_STRIDE = 16
ST_RVEC, ST_TVEC, ST_OK, ST_ERR, ST_NTRACK, ST_ITERS, ST_GUESS, ST_FLAGS, ST_TVEC_F32 = 0, 3, 6, 7, 8, 9, 10, 11, 12
TRK_ZERO_VELOCITY = 256
TRK_CHAIN_TIMEOUT = 512
PROF_SPANS = 5
DENSE_STRIDE = 16
DN_RVEC, DN_TVEC, DN_REFINED, DN_PHOTO_RMS, DN_GEO_RMS, DN_VALID, DN_ITERS, DN_CORNERS = 0, 3, 6, 7, 8, 9, 10, 11

# every symbol include/agt_hip.h declares (tests check the .so exports all of them)
SYMBOLS = [
    "agt_version", "agt_error_string", "agt_create", "agt_destroy", "agt_set_stream",
    "agt_last_hip_error", "agt_synchronize", "agt_pyr_down_u8", "agt_pyramid_build", "agt_pyramid_build_pair",
    "agt_pyramid_level", "agt_pyramid_max_level", "agt_lk_track", "agt_solve_pnp",
    "agt_project_points", "agt_tracker_reset", "agt_tracker_options", "agt_estimate_pose",
    "agt_tracker_state_size", "agt_tracker_state_read", "agt_track_frame", "agt_track_frames", "agt_tracker_buffers",
    "agt_profile_begin", "agt_profile_end", "agt_tracker_pipeline", "agt_tracker_join",
    "agt_get_optimal_new_camera_matrix", "agt_undistort_init", "agt_undistort_maps", "agt_undistort_bgr",
    "agt_preprocess_bgr", "agt_dense_refine", "agt_tracker_dense", "agt_track_frame_dense", "agt_track_frames_dense", "agt_upload", "agt_download",
    "agt_tracker_tag_gate", "agt_track_frame_detected", "agt_track_host_frame", "agt_tracker_rewind",
    "agt_device_info", "agt_xcd_tile_order", "agt_lk_occupancy", "agt_lk_occupancy_cu", "agt_lk_lds_request",
    "agt_solve_pnp_host", "agt_project_points_host",
]


class Config(C.Structure):
    _fields_ = [("device", C.c_int), ("width", C.c_int), ("height", C.c_int), ("max_level", C.c_int),
                ("win", C.c_int), ("max_points", C.c_int), ("max_streams", C.c_int), ("reserved", C.c_int * 8)]


class AgtError(ValueError):
    """A C-ABI call returned a negative code (cv2 would raise cv2.error; the reference's
    own wrappers raise ValueError, transform_helper.py:82-84)."""

    def __init__(self, code, where):
        self.code = code
        super().__init__("%s failed: AGT_ERR_%s (%d)" % (where, ERRORS.get(code, "?"), code))


_lib = None


def lib():
    """Load libagt_hip.so (once).  Raises RuntimeError when the HIP extension is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            "HIP extension %s is missing: build it (python -c 'import __graft_entry__ as g; g.build()'). "
            "There is no CPU fallback." % LIB_PATH)
    # PyTorch-ROCm bundles its own libamdhip64.so.7; load it FIRST so that this library's
    # NEEDED libamdhip64.so.7 resolves to the same runtime instance (two HIP runtimes in one
    # process cannot both own the device).
    import torch  # noqa: F401
    L = C.CDLL(LIB_PATH)
    vp, i32, f64, sz = C.c_void_p, C.c_int, C.c_double, C.c_size_t
    L.agt_version.restype = i32
    L.agt_error_string.restype = C.c_char_p
    L.agt_error_string.argtypes = [i32]
    L.agt_create.argtypes = [C.POINTER(Config), vp, C.POINTER(vp)]
    L.agt_destroy.argtypes = [vp]
    L.agt_set_stream.argtypes = [vp, vp]
    L.agt_last_hip_error.argtypes = [vp]
    L.agt_synchronize.argtypes = [vp]
    L.agt_pyr_down_u8.argtypes = [vp, vp, i32, i32, sz, sz, vp, sz, sz, i32]
    L.agt_pyramid_build.argtypes = [vp, i32, vp, sz, sz, i32]
    L.agt_pyramid_build_pair.argtypes = [vp, vp, vp, sz, sz, i32]
    L.agt_pyramid_level.argtypes = [vp, i32, i32, C.POINTER(vp), C.POINTER(i32), C.POINTER(i32),
                                    C.POINTER(sz), C.POINTER(sz)]
    L.agt_pyramid_max_level.argtypes = [vp]
    L.agt_lk_track.argtypes = [vp, i32, i32, vp, vp, vp, vp, i32, i32, i32, i32, f64, i32, f64]
    L.agt_solve_pnp.argtypes = [vp, vp, sz, vp, i32, vp, i32, i32, vp, vp, i32, vp, i32, vp, vp]
    L.agt_project_points.argtypes = [vp, vp, sz, i32, i32, i32, vp, vp, vp, i32, vp, vp]
    L.agt_tracker_reset.argtypes = [vp, i32, vp, vp, i32, i32, vp, vp, i32, i32]
    L.agt_tracker_options.argtypes = [vp, i32, i32, f64]
    L.agt_estimate_pose.argtypes = [vp, vp, vp, i32, vp]
    L.agt_tracker_state_size.restype = i32
    L.agt_tracker_state_read.argtypes = [vp, vp, i32]
    L.agt_track_frame.argtypes = [vp, vp, sz, sz, i32, vp]
    L.agt_track_frames.argtypes = [vp, vp, sz, sz, sz, i32, i32, vp]
    L.agt_tracker_tag_gate.argtypes = [vp, i32]
    L.agt_track_frame_detected.argtypes = [vp, vp, sz, sz, i32, vp, vp, vp]
    L.agt_track_host_frame.argtypes = [vp, vp, i32, i32, i32, vp, i32, i32, i32, vp, sz, vp, vp]
    L.agt_tracker_rewind.argtypes = [vp]
    L.agt_device_info.argtypes = [vp, C.POINTER(i32), C.POINTER(i32), C.c_char_p, sz]
    L.agt_xcd_tile_order.argtypes = [i32, i32, i32]
    L.agt_xcd_tile_order.restype = i32
    L.agt_lk_occupancy.argtypes = [vp, i32]
    L.agt_lk_occupancy_cu.argtypes = [vp, i32]
    L.agt_lk_lds_request.argtypes = [i32]
    L.agt_solve_pnp_host.argtypes = [vp, vp, vp, i32, i32, vp, vp, i32, vp, i32, vp, vp]
    L.agt_project_points_host.argtypes = [vp, vp, i32, i32, vp, vp, vp, i32, vp, vp]
    L.agt_tracker_buffers.argtypes = [vp, C.POINTER(vp), C.POINTER(vp)]
    L.agt_tracker_pipeline.argtypes = [vp, i32]
    L.agt_tracker_join.argtypes = [vp]
    L.agt_get_optimal_new_camera_matrix.argtypes = [vp, vp, i32, i32, i32, f64, i32, i32, vp, vp]
    L.agt_undistort_init.argtypes = [vp, vp, vp, i32, vp, i32, i32]
    L.agt_undistort_maps.argtypes = [vp, C.POINTER(vp), C.POINTER(vp), C.POINTER(i32), C.POINTER(i32)]
    L.agt_undistort_bgr.argtypes = [vp, vp, sz, sz, vp, sz, sz, i32]
    L.agt_preprocess_bgr.argtypes = [vp, vp, sz, sz, i32, i32, i32, i32, i32, i32, i32, i32, vp, sz, sz]
    L.agt_dense_refine.argtypes = [vp, vp, sz, sz, i32, i32, vp, vp, i32, vp, vp, vp, i32, vp, vp, i32, vp, i32, i32, f64, vp]
    L.agt_tracker_dense.argtypes = [vp, vp, vp, i32, i32, f64, i32]
    L.agt_track_frame_dense.argtypes = [vp, vp, sz, sz, i32, vp, vp]
    L.agt_track_frames_dense.argtypes = [vp, vp, sz, sz, sz, i32, i32, vp, vp]
    L.agt_upload.argtypes = [vp, vp, vp, sz]
    L.agt_download.argtypes = [vp, vp, vp, sz]
    L.agt_profile_begin.argtypes = [vp, i32]
    L.agt_profile_end.argtypes = [vp, vp, C.POINTER(i32)]
    _lib = L
    return L


def check(rc, where):
    if rc != AGT_OK:
        raise AgtError(rc, where)
