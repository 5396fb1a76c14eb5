""""a CPU twin of each [C-ABI entry point] with identical signatures (host pointers)".

The product has no CPU path; the twin is TEST INFRASTRUCTURE in the oracle library (oracle/agt_cpu_twin.c:
agt_cpu_<name> on top of the restated OpenCV algorithms).  Here:
  * CPU: every twin has, token for token, the parameter list of its include/agt_hip.h counterpart (context type aside), and does
    what the oracle does;
  * GPU (-m gpu): ONE ctypes call sequence -- create, pyramid_build x 2, pyramid_level, lk_track, solve_pnp (guess and no guess,
    masked), project_points, pyr_down_u8 -- run through libagt_hip.so (device buffers) and through the twin (host buffers): pyramid
    levels and LK outputs bit-exact, poses / projections / Jacobians <= 1e-9, info words equal."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TWINS = ["create", "destroy", "pyr_down_u8", "pyramid_build", "pyramid_level", "pyramid_max_level", "lk_track", "solve_pnp", "project_points",
         "solve_pnp_host", "project_points_host"]


def _protos(text, prefix):
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    text = re.sub(r"//[^\n]*", "", text)
    out = {}
    for m in re.finditer(r"\bint\s+(%s[a-z0-9_]+)\s*\(([^;{]*?)\)\s*[;{]" % prefix, text, flags=re.S):
        params = " ".join(m.group(2).split())
        out[m.group(1)] = params
    return out


def test_twin_signatures_equal_the_header():
    hdr = _protos(open(os.path.join(ROOT, "include", "agt_hip.h")).read(), "agt_")
    twin = _protos(open(os.path.join(ROOT, "oracle", "agt_cpu_twin.c")).read(), "agt_cpu_")
    for name in TWINS:
        h, t = hdr["agt_" + name], twin["agt_cpu_" + name]
        norm = lambda p: re.sub(r"\bagt_cpu_ctx\b", "agt_ctx", re.sub(r"\bc\b", "ctx", p))
        assert norm(t) == norm(h), "agt_cpu_%s(%s) vs agt_%s(%s)" % (name, t, name, h)


from accurate_aprilgroup_tracking_amd.hiplib import Config as Cfg      # (agt_config: the ctypes mirror of the product binding, used for both)


def _twin_lib(oracle):
    L = oracle.lib()
    vp, i32, f64, sz = C.c_void_p, C.c_int, C.c_double, C.c_size_t
    L.agt_cpu_create.argtypes = [C.POINTER(Cfg), vp, C.POINTER(vp)]
    L.agt_cpu_destroy.argtypes = [vp]
    L.agt_cpu_pyr_down_u8.argtypes = [vp, vp, i32, i32, sz, sz, vp, sz, sz, i32]
    L.agt_cpu_pyramid_build.argtypes = [vp, i32, vp, sz, sz, i32]
    L.agt_cpu_pyramid_level.argtypes = [vp, i32, i32, C.POINTER(vp), C.POINTER(i32), C.POINTER(i32), C.POINTER(sz), C.POINTER(sz)]
    L.agt_cpu_pyramid_max_level.argtypes = [vp]
    L.agt_cpu_lk_track.argtypes = [vp, i32, i32, vp, vp, vp, vp, i32, i32, i32, i32, f64, i32, f64]
    L.agt_cpu_solve_pnp.argtypes = [vp, vp, sz, vp, i32, vp, i32, i32, vp, vp, i32, vp, i32, vp, vp]
    L.agt_cpu_project_points.argtypes = [vp, vp, sz, i32, i32, i32, vp, vp, vp, i32, vp, vp]
    L.agt_cpu_solve_pnp_host.argtypes = [vp, vp, vp, i32, i32, vp, vp, i32, vp, i32, vp, vp]
    L.agt_cpu_project_points_host.argtypes = [vp, vp, i32, i32, vp, vp, vp, i32, vp, vp]
    return L


def _sequence(call, mem, seq, B=3):
    """the call sequence, written once: `call(name, *args)` invokes agt_<name> / agt_cpu_<name>, `mem` turns host arrays into
    the library's buffers and back.  -> dict of host arrays"""
    from accurate_aprilgroup_tracking_amd import synthetic as syn
    W, H = seq.width, seq.height
    n = seq.obj.shape[0]
    out = {}
    cfg = Cfg(0, W, H, 2, 21, n, B, (C.c_int * 8)())
    h = C.c_void_p()
    assert call("create", C.byref(cfg), None, C.byref(h)) == 0
    fa = np.stack([seq.frame(0)] * B); fb = np.stack([seq.frame(1 + (b % 2)) for b in range(B)])
    da, db = mem.to(fa), mem.to(fb)
    assert call("pyramid_build", h, 0, mem.ptr(da), W, W * H, B) == 0
    assert call("pyramid_build", h, 1, mem.ptr(db), W, W * H, B) == 0
    assert call("pyramid_max_level", h) == 2
    for lvl in (1, 2):
        p, w, hh, pitch, bs = C.c_void_p(), C.c_int(), C.c_int(), C.c_size_t(), C.c_size_t()
        assert call("pyramid_level", h, 1, lvl, C.byref(p), C.byref(w), C.byref(hh), C.byref(pitch), C.byref(bs)) == 0
        raw = mem.read(p, bs.value * B).reshape(B, -1)[:, :hh.value * pitch.value].reshape(B, hh.value, pitch.value)
        out["level%d" % lvl] = raw[:, :, :w.value].copy()
    pts = np.stack([seq.corners(0)] * B).astype(np.float32)
    pts[1, 3] = (-30.0, 40.0)                                        # a corner outside the image in stream 1
    dp = mem.to(pts); dn = mem.to(np.zeros_like(pts)); ds = mem.to(np.zeros((B, n), np.uint8)); de = mem.to(np.zeros((B, n), np.float32))
    assert call("lk_track", h, 0, 1, mem.ptr(dp), mem.ptr(dn), mem.ptr(ds), mem.ptr(de), n, B, 3, 30, 0.01, 0, 1e-4) == 0
    out["next"], out["status"], out["err"] = mem.back(dn), mem.back(ds), mem.back(de)
    # solvePnP with a guess on the tracked corners (masked by the LK status), then without one
    Kh = np.ascontiguousarray(seq.K.reshape(-1)); Kp = Kh.ctypes.data_as(C.c_void_p)
    dist = np.array(syn.MILD_DIST, np.float64); dpx = dist.ctypes.data_as(C.c_void_p)
    obj = mem.to(seq.obj.astype(np.float32))
    for tag, guess, dd, nd in (("guess", 1, None, 0), ("noguess", 0, None, 0), ("guess_dist", 1, dpx, 5)):
        pose = np.stack([np.concatenate([seq.rvecs[0], seq.tvecs[0]])] * B).astype(np.float64)
        dpo = mem.to(pose); di = mem.to(np.zeros((B, 4), np.int32)); der = mem.to(np.zeros(B, np.float64))
        assert call("solve_pnp", h, mem.ptr(obj), 0, mem.ptr(dn), 0, mem.ptr(ds), n, B, Kp, dd, nd, mem.ptr(dpo), guess, mem.ptr(di), mem.ptr(der)) == 0
        out["pose_" + tag], out["info_" + tag], out["merr_" + tag] = mem.back(dpo), mem.back(di), mem.back(der)
    pose = np.stack([np.concatenate([seq.rvecs[b % 2], seq.tvecs[b % 2]]) for b in range(B)]).astype(np.float64)
    dpo = mem.to(pose); dimg = mem.to(np.zeros((B, n, 2), np.float32)); dj = mem.to(np.zeros((B, 2 * n, 6), np.float64))
    assert call("project_points", h, mem.ptr(obj), 0, 0, n, B, mem.ptr(dpo), Kp, dpx, 5, mem.ptr(dimg), mem.ptr(dj)) == 0
    out["proj"], out["jac"] = mem.back(dimg), mem.back(dj)
    # the synchronous host-array entry points (round 5): HOST pointers in both libraries
    hp = lambda a: a.ctypes.data_as(C.c_void_p)
    obj64 = np.ascontiguousarray(seq.obj, np.float64)
    img64 = np.ascontiguousarray(seq.corners(1), np.float64)
    for tag, guess in (("hguess", 1), ("hnoguess", 0)):
        pose_h = np.concatenate([seq.rvecs[0], seq.tvecs[0]]).astype(np.float64); info_h = np.zeros(4, np.int32); err_h = np.zeros(1)
        assert call("solve_pnp_host", h, hp(obj64), hp(img64), 1, n, Kp, dpx, 5, hp(pose_h), guess, hp(info_h), hp(err_h)) == 0
        out["pose_" + tag], out["info_" + tag], out["merr_" + tag] = pose_h[None].copy(), info_h[None].copy(), err_h.copy()
    pose_h = np.concatenate([seq.rvecs[1], seq.tvecs[1]]).astype(np.float64)
    proj_h = np.zeros((n, 2), np.float64); jac_h = np.zeros((2 * n, 6), np.float64)
    assert call("project_points_host", h, hp(obj64), 1, n, hp(pose_h), Kp, dpx, 5, hp(proj_h), hp(jac_h)) == 0
    out["hproj"], out["hjac"] = proj_h, jac_h
    assert call("project_points_host", h, hp(obj64), 1, 300, hp(pose_h), Kp, dpx, 5, hp(proj_h), None) == -4               # AGT_ERR_NPOINTS in both
    small = np.ascontiguousarray(fa[:, :100, :160])                  # pyrDown of a crop: 160 x 100 -> 80 x 50
    dsrc = mem.to(small); ddst = mem.to(np.zeros((B, 50, 80), np.uint8))
    assert call("pyr_down_u8", h, mem.ptr(dsrc), 160, 100, 160, 160 * 100, mem.ptr(ddst), 80, 80 * 50, B) == 0
    out["down"] = mem.back(ddst)
    assert call("lk_track", h, 0, 5, mem.ptr(dp), mem.ptr(dn), mem.ptr(ds), None, n, B, 3, 30, 0.01, 0, 1e-4) == -1       # bad slot: AGT_ERR_ARG in both
    assert call("destroy", h) == 0
    return out


class HostMem:
    def to(self, a):
        return np.ascontiguousarray(a).copy()

    def ptr(self, a):
        return a.ctypes.data_as(C.c_void_p)

    def back(self, a):
        return a.copy()

    def read(self, p, nbytes):
        return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), shape=(nbytes,)).copy()


def test_twin_does_what_the_oracle_does(oracle, seq640):
    L = _twin_lib(oracle)
    r = _sequence(lambda name, *a: getattr(L, "agt_cpu_" + name)(*a), HostMem(), seq640)
    for b in range(3):
        p0 = seq640.corners(0).copy()
        if b == 1:
            p0[3] = (-30.0, 40.0)
        o = oracle.calcOpticalFlowPyrLK(seq640.frame(0), seq640.frame(1 + (b % 2)), p0, maxLevel=2)
        assert np.array_equal(r["next"][b].view(np.uint32), o[0].reshape(-1, 2).view(np.uint32)) and np.array_equal(r["status"][b], o[1].ravel())
    assert r["status"][1, 3] == 0 and r["status"].sum() == 3 * 48 - 1
    m = r["status"][0] == 1
    ok, rv, tv = oracle.solvePnP(seq640.obj.astype(np.float32)[m], r["next"][0][m], seq640.K, None, seq640.rvecs[0].copy(), seq640.tvecs[0].copy(), True)
    assert np.abs(r["pose_guess"][0] - np.concatenate([rv.ravel(), tv.ravel()])).max() < 1e-12
    assert (r["info_guess"][:, 0] == 1).all() and (r["info_guess"][:, 2] == r["status"].sum(axis=1)).all()
    assert np.abs(r["pose_guess"] - r["pose_noguess"]).max() < 1e-6
    pp, _ = oracle.projectPoints(seq640.obj, seq640.rvecs[1], seq640.tvecs[1], seq640.K, np.array([0.05, -0.1, 1e-3, -1e-3, 0.02]))
    assert np.abs(r["proj"][1] - pp.reshape(-1, 2)).max() < 1e-3          # (float32 output)
    # the host-array twins are the same calls with B = 1
    from accurate_aprilgroup_tracking_amd import synthetic as syn
    pj, jj = oracle.projectPoints(seq640.obj, seq640.rvecs[1], seq640.tvecs[1], seq640.K, np.array(syn.MILD_DIST, np.float64).ravel(), jacobian=True)
    assert np.array_equal(r["hproj"], pj.reshape(-1, 2)) and r["hjac"].shape == (96, 6) and np.abs(r["hjac"] - jj).max() < 1e-9
    assert r["info_hguess"][0, 0] == 1 and r["info_hnoguess"][0, 0] == 1 and np.abs(r["pose_hguess"] - r["pose_hnoguess"]).max() < 1e-6


@pytest.mark.gpu
def test_same_call_sequence_on_the_hip_library_and_on_its_cpu_twin(oracle, seq640):
    import torch
    from accurate_aprilgroup_tracking_amd import hiplib as H
    LH = H.lib()
    LT = _twin_lib(oracle)

    class DevMem:
        def __init__(self):
            self.keep = []

        def to(self, a):
            t = torch.from_numpy(np.ascontiguousarray(a)).cuda().contiguous(); self.keep.append(t)
            return t

        def ptr(self, t):
            return C.c_void_p(t.data_ptr())

        def back(self, t):
            torch.cuda.synchronize()
            return t.cpu().numpy().copy()

    # (a raw device pointer handed out by agt_pyramid_level is read back through the library's own agt_download: dev.read below)
    dev = DevMem()
    state = {}

    def hip_call(name, *a):
        if name == "create":
            rc = LH.agt_create(a[0], a[1], a[2]); state["h"] = a[2]._obj if hasattr(a[2], "_obj") else None
            return rc
        return getattr(LH, "agt_" + name)(*a)

    def dev_read(p, nbytes):
        buf = np.zeros(nbytes, np.uint8)
        H.check(LH.agt_download(state["h"], buf.ctypes.data_as(C.c_void_p), p, nbytes), "agt_download")
        return buf
    dev.read = dev_read
    g = _sequence(hip_call, dev, seq640)
    c = _sequence(lambda name, *a: getattr(LT, "agt_cpu_" + name)(*a), HostMem(), seq640)
    for k in ("level1", "level2", "down", "status"):
        assert np.array_equal(g[k], c[k]), k
    for k in ("next", "err"):
        assert np.array_equal(g[k].view(np.uint32), c[k].view(np.uint32)), k
    for k in ("pose_guess", "pose_noguess", "pose_guess_dist", "jac"):
        assert np.abs(g[k] - c[k]).max() < (1e-9 if k != "jac" else 1e-6), "%s: %g" % (k, np.abs(g[k] - c[k]).max())
    for k in ("merr_guess", "merr_noguess", "merr_guess_dist"):
        assert np.abs(g[k] - c[k]).max() < 1e-9, k
    assert np.abs(g["proj"] - c["proj"]).max() < 1e-4                      # float32 outputs of values ~1e2: one ulp is 8e-6
    for k in ("info_guess", "info_noguess", "info_guess_dist", "info_hguess", "info_hnoguess"):
        assert np.array_equal(g[k][:, [0, 1, 2]], c[k][:, [0, 1, 2]]), k   # ok, LM iterations, points used
    # the host-array entry points (f64 in and out, distorting camera)
    for k in ("pose_hguess", "pose_hnoguess", "merr_hguess", "merr_hnoguess", "hproj"):
        assert np.abs(g[k] - c[k]).max() < 1e-9, "%s: %g" % (k, np.abs(g[k] - c[k]).max())
    assert np.abs(g["hjac"] - c["hjac"]).max() < 1e-6
