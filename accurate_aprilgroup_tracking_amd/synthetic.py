"""Deterministic synthetic AprilGroup sequences (SURVEY.md section 8d "Synthetic inputs").

The reference ships neither its `april_group.json` nor any recording
(/root/reference/.gitignore:133-142, README.md:43-45), so benchmarks and parity tests
render their own data: an AprilGroup of T tags on a camera-facing spherical cap, moved
along a smooth trajectory in front of a pinhole(+Brown-Conrady) camera.  Everything
is numpy/scipy, seeded, and independent of both the HIP library and the oracle.

The JSON written by :func:`make_april_group` follows the reference loader's schema
(detect_pose.py:122-130): {"tags": {"<id>": {"size": s, "extrinsics": [tx,ty,tz, rx,ry,rz]}}}.
"""
import json
import numpy as np
from scipy.ndimage import gaussian_filter
from scipy.spatial.transform import Rotation

MILD_DIST = np.array([[0.05, -0.1, 1e-3, -1e-3, 0.02]], dtype=np.float64)
DENSE_Z0 = 0.47          # object distance (m) of the 60-tag model


def camera_matrix(width, height):
    """fx = fy = 1000 * (W / 1280), principal point at the image centre."""
    f = 1000.0 * (width / 1280.0)
    return np.array([[f, 0.0, width / 2.0], [0.0, f, height / 2.0], [0.0, 0.0, 1.0]])


def _cap_directions(n, max_polar_rad):
    """n unit vectors on a cap around -z (towards the camera), Fibonacci spiral."""
    k = np.arange(n) + 0.5
    cos_max = np.cos(max_polar_rad)
    cos_t = 1.0 - (1.0 - cos_max) * k / n
    sin_t = np.sqrt(np.maximum(0.0, 1.0 - cos_t * cos_t))
    phi = k * np.pi * (3.0 - np.sqrt(5.0))
    return np.stack([sin_t * np.cos(phi), sin_t * np.sin(phi), -cos_t], axis=1)


def make_april_group(n_tags=12, tag_size=0.020, max_polar_deg=46.0, seed=0, sep=1.8):
    """AprilGroup on a spherical cap.  Returns the april_group.json dict.

    The sphere radius is the smallest one that keeps tag centres >= sep * tag_size apart,
    so the tags (with their 1-cell quiet zone, 1.25 * tag_size across) never overlap.
    """
    d = _cap_directions(n_tags, np.deg2rad(max_polar_deg))
    if n_tags > 1:
        g = d @ d.T
        np.fill_diagonal(g, -1.0)
        min_ang = np.arccos(np.clip(g.max(), -1.0, 1.0))
        radius = sep * tag_size / (2.0 * np.sin(min_ang / 2.0))
    else:
        radius = 0.0
    rng = np.random.default_rng(seed)
    tags = {}
    for i in range(n_tags):
        z = d[i]
        up = np.array([0.0, 1.0, 0.0]) if abs(z[1]) < 0.9 else np.array([1.0, 0.0, 0.0])
        x = np.cross(up, z); x /= np.linalg.norm(x)
        y = np.cross(z, x)
        spin = rng.uniform(-np.pi, np.pi)              # in-plane rotation of the tag
        c, s = np.cos(spin), np.sin(spin)
        R = np.stack([c * x + s * y, -s * x + c * y, z], axis=1)
        rvec = Rotation.from_matrix(R).as_rotvec()
        tvec = radius * z
        # the reference stores extrinsics as float32 (detect_pose.py:126-130)
        ext = [float(np.float32(v)) for v in (*tvec, *rvec)]
        tags[str(i)] = {"size": float(tag_size), "extrinsics": ext}
    return {"tags": tags}


def write_april_group(path, group):
    with open(path, "w") as f:
        json.dump(group, f, indent=1)


def group_object_points(group):
    """(4T, 3) float64 corner model, exactly as the reference builds it
    (transform_helper.py:41-96 get_initial_pts + transform_marker_corners; tag rvec/tvec are f32)."""
    pts = []
    for key in group["tags"]:
        tag = group["tags"][key]
        rad = tag["size"] / 2.0
        init = np.array([[-rad, -rad, 0.0], [-rad, rad, 0.0], [rad, rad, 0.0], [rad, -rad, 0.0]])
        tvec = np.array(tag["extrinsics"][:3], dtype=np.float32)
        rvec = np.array(tag["extrinsics"][-3:], dtype=np.float32)
        R = Rotation.from_rotvec(rvec.astype(np.float64)).as_matrix().astype(np.float32)
        pts.append(init @ R.T + tvec.reshape(-1, 3))
    return np.array(pts).reshape(-1, 3)


def model_samples(group, grid=32, extent=0.6):
    """(T*grid*grid, 3) float32 object-frame points on every tag plane, a grid x grid lattice over
    [-extent*size, +extent*size]^2 (covers the black border and its outer edge): the dense-alignment
    model of BASELINE config 5 (60 tags x 32 x 32 = 61,440 samples)."""
    pts = []
    lin = (np.arange(grid) + 0.5) / grid * 2.0 - 1.0
    for key in group["tags"]:
        tag = group["tags"][key]
        s = tag["size"] * extent
        uu, vv = np.meshgrid(lin * s, lin * s)
        local = np.stack([uu.ravel(), vv.ravel(), np.zeros(uu.size)], axis=1)
        tvec = np.array(tag["extrinsics"][:3], dtype=np.float32).astype(np.float64)
        rvec = np.array(tag["extrinsics"][-3:], dtype=np.float32).astype(np.float64)
        pts.append(local @ Rotation.from_rotvec(rvec).as_matrix().T + tvec)
    return np.concatenate(pts).astype(np.float32)


def sample_bilinear(img, uv):
    """bilinear samples of a gray image at (N,2) pixel coordinates (float64); NaN outside"""
    img = np.asarray(img, np.float64); h, w = img.shape
    x, y = uv[:, 0], uv[:, 1]
    x0 = np.floor(x).astype(np.int64); y0 = np.floor(y).astype(np.int64)
    ok = (x0 >= 0) & (x0 < w - 1) & (y0 >= 0) & (y0 < h - 1)
    x0c = np.clip(x0, 0, w - 2); y0c = np.clip(y0, 0, h - 2)
    a = x - x0; b = y - y0
    v = (1 - a) * (1 - b) * img[y0c, x0c] + a * (1 - b) * img[y0c, x0c + 1] + (1 - a) * b * img[y0c + 1, x0c] + a * b * img[y0c + 1, x0c + 1]
    return np.where(ok, v, np.nan)


def tag_bits(n_tags, seed=0):
    """(T, 6, 6) pseudo tag36 payloads (random bits; decoding is out of scope)."""
    return np.random.default_rng(seed + 7919).integers(0, 2, size=(n_tags, 6, 6)).astype(np.uint8)


def trajectory(n_frames, seed=0, t0=(0.01, -0.02, 0.30), r0=(0.2, -0.1, 0.3), speed=1.0):
    """Smooth sinusoidal pose trajectory, non-zero motion in every component
    (so detect_pose.py:236 `np.all` is not tripped).  Returns rvecs (F,3), tvecs (F,3)."""
    rng = np.random.default_rng(seed)
    k = np.arange(n_frames)[:, None].astype(np.float64)
    a_t = np.array([0.006, 0.005, 0.010]); a_r = np.array([0.04, 0.05, 0.06])
    w_t = 2 * np.pi / np.array([83.0, 71.0, 97.0]) * speed
    w_r = 2 * np.pi / np.array([101.0, 89.0, 113.0]) * speed
    p_t = rng.uniform(0, 2 * np.pi, 3); p_r = rng.uniform(0, 2 * np.pi, 3)
    tv = np.asarray(t0) + a_t * np.sin(w_t * k + p_t)
    rv = np.asarray(r0) + a_r * np.sin(w_r * k + p_r)
    return rv, tv


def project(obj, rvec, tvec, K, dist=None):
    """Pinhole + Brown-Conrady (k1 k2 p1 p2 k3) projection in float64 (analytic ground truth)."""
    R = Rotation.from_rotvec(np.asarray(rvec, float).reshape(3)).as_matrix()
    Y = np.asarray(obj, float).reshape(-1, 3) @ R.T + np.asarray(tvec, float).reshape(1, 3)
    x = Y[:, 0] / Y[:, 2]; y = Y[:, 1] / Y[:, 2]
    if dist is not None:
        k = np.zeros(5); d = np.asarray(dist, float).ravel(); k[:min(5, d.size)] = d[:5]
        r2 = x * x + y * y
        cd = 1 + k[0] * r2 + k[1] * r2 * r2 + k[4] * r2 ** 3
        xd = x * cd + 2 * k[2] * x * y + k[3] * (r2 + 2 * x * x)
        yd = y * cd + k[2] * (r2 + 2 * y * y) + 2 * k[3] * x * y
        x, y = xd, yd
    return np.stack([K[0, 0] * x + K[0, 2], K[1, 1] * y + K[1, 2]], axis=1)


def _undistort_norm(xd, yd, dist, iters=20):
    if dist is None:
        return xd, yd
    k = np.zeros(5); d = np.asarray(dist, float).ravel(); k[:min(5, d.size)] = d[:5]
    if not np.any(k):
        return xd, yd
    x, y = xd.copy(), yd.copy()
    for _ in range(iters):
        r2 = x * x + y * y
        icd = 1.0 / (1 + k[0] * r2 + k[1] * r2 * r2 + k[4] * r2 ** 3)
        dx = 2 * k[2] * x * y + k[3] * (r2 + 2 * x * x)
        dy = k[2] * (r2 + 2 * y * y) + 2 * k[3] * x * y
        x = (xd - dx) * icd; y = (yd - dy) * icd
    return x, y


def background(width, height, seed=0):
    """mid-gray + low-amplitude band-limited noise, float32 (H, W)."""
    rng = np.random.default_rng(seed + 104729)
    n = gaussian_filter(rng.standard_normal((height, width)), 2.0, mode="reflect")
    n *= 14.0 / max(np.std(n), 1e-9)
    return (128.0 + n).astype(np.float32)


def render_frame(group, bits, rvec, tvec, K, dist, width, height, bg, supersample=4, cover_out=None):
    """Render one u8 gray frame: every front-facing tag as a warped 8x8-cell pattern
    (1-cell black border, 1-cell white quiet zone), box-filtered over supersample^2 taps.
    cover_out: optional float32 (H, W) array that receives the SUM of the tags' coverage (a value above 1 marks
    pixels painted by two tags: the layout check of the 60-tag model)."""
    img = bg.astype(np.float32).copy()
    R_o = Rotation.from_rotvec(np.asarray(rvec, float).reshape(3)).as_matrix()
    t_o = np.asarray(tvec, float).reshape(3)
    ss = supersample
    offs = (np.arange(ss) + 0.5) / ss - 0.5
    for ti, key in enumerate(group["tags"]):
        tag = group["tags"][key]
        s = tag["size"]
        tv = np.array(tag["extrinsics"][:3], dtype=np.float32).astype(np.float64)
        rv = np.array(tag["extrinsics"][-3:], dtype=np.float32).astype(np.float64)
        R_t = Rotation.from_rotvec(rv).as_matrix()
        R = R_o @ R_t
        t = R_o @ tv + t_o
        if (R[:, 2] @ t) >= 0:          # back-facing (normal points away from the camera)
            continue
        q = 0.625 * s
        quad = np.array([[-q, -q, 0], [-q, q, 0], [q, q, 0], [q, -q, 0]]) @ R.T + t
        if np.any(quad[:, 2] <= 1e-6):
            continue
        pix = project(np.array([[-q, -q, 0], [-q, q, 0], [q, q, 0], [q, -q, 0]]), Rotation.from_matrix(R).as_rotvec(), t, K, dist)
        x0 = max(int(np.floor(pix[:, 0].min())) - 1, 0); x1 = min(int(np.ceil(pix[:, 0].max())) + 2, width)
        y0 = max(int(np.floor(pix[:, 1].min())) - 1, 0); y1 = min(int(np.ceil(pix[:, 1].max())) + 2, height)
        if x1 <= x0 or y1 <= y0:
            continue
        # plane -> normalised image homography: [r1 r2 t]
        Hn = np.stack([R[:, 0], R[:, 1], t], axis=1)
        Hinv = np.linalg.inv(Hn)
        xs = (np.arange(x0, x1)[None, :, None, None] + offs[None, None, None, :])
        ys = (np.arange(y0, y1)[:, None, None, None] + offs[None, None, :, None])
        xs, ys = np.broadcast_arrays(xs, ys)
        xn = (xs - K[0, 2]) / K[0, 0]; yn = (ys - K[1, 2]) / K[1, 1]
        xn, yn = _undistort_norm(xn, yn, dist)
        w = Hinv[2, 0] * xn + Hinv[2, 1] * yn + Hinv[2, 2]
        u = (Hinv[0, 0] * xn + Hinv[0, 1] * yn + Hinv[0, 2]) / w
        v = (Hinv[1, 0] * xn + Hinv[1, 1] * yn + Hinv[1, 2]) / w
        cu = (u / s + 0.5) * 8.0; cv = (v / s + 0.5) * 8.0
        inside = (cu >= -1) & (cu < 9) & (cv >= -1) & (cv < 9)
        iu = np.clip(np.floor(cu).astype(np.int64), -1, 8); iv = np.clip(np.floor(cv).astype(np.int64), -1, 8)
        pat = np.full((10, 10), 225.0, np.float32)      # quiet zone
        pat[1:9, 1:9] = 30.0                            # black border
        pat[2:8, 2:8] = np.where(bits[ti] > 0, 225.0, 30.0)
        val = pat[iv + 1, iu + 1]
        cover = inside.astype(np.float32)
        acc = (val * cover).mean(axis=(2, 3)); cov = cover.mean(axis=(2, 3))
        img[y0:y1, x0:x1] = img[y0:y1, x0:x1] * (1 - cov) + acc
        if cover_out is not None:
            cover_out[y0:y1, x0:x1] += cov
    return np.clip(np.rint(img), 0, 255).astype(np.uint8)


class Sequence:
    """A seeded synthetic stream: model, camera, trajectory, lazily rendered frames."""

    def __init__(self, width=1280, height=720, n_tags=12, n_frames=8, seed=0, dist=None,
                 supersample=4, speed=1.0, z0=None, group_seed=None, tag_size=None, sep=None):
        """seed drives trajectory and background; group_seed (default: seed) the AprilGroup model, so
        several streams can show the SAME object along different trajectories.
        Defaults: up to 12 tags of 20 mm at 0.30 m (the dodeca-like model of configs 1-4); more tags (the 60-tag
        model of BASELINE config 5) are 10 mm, 2.2 tag sizes apart, seen from DENSE_Z0 so that every tag of the cap
        is inside a 1280x720 frame and no two tags touch in the image (test_synthetic_60_tag_layout)."""
        self.width, self.height, self.seed = width, height, seed
        gs = seed if group_seed is None else group_seed
        dense = n_tags > 12
        tag_size = (0.010 if dense else 0.020) if tag_size is None else tag_size
        sep = (2.2 if dense else 1.8) if sep is None else sep
        z0 = (DENSE_Z0 if dense else 0.30) if z0 is None else z0
        self.group = make_april_group(n_tags=n_tags, tag_size=tag_size, seed=gs, sep=sep)
        self.bits = tag_bits(n_tags, gs)
        self.obj = group_object_points(self.group)                 # (4T, 3) f64
        self.K = camera_matrix(width, height)
        self.dist = None if dist is None else np.asarray(dist, np.float64).reshape(1, -1)
        self.rvecs, self.tvecs = trajectory(n_frames, seed, t0=(0.01, -0.02, z0), speed=speed)
        self.bg = background(width, height, seed)
        self.supersample = supersample
        self._frames = {}

    def __len__(self):
        return self.rvecs.shape[0]

    def corners(self, k):
        """exact projections of the model at frame k, (4T, 2) float32"""
        return project(self.obj, self.rvecs[k], self.tvecs[k], self.K, self.dist).astype(np.float32)

    def frame(self, k):
        if k not in self._frames:
            self._frames[k] = render_frame(self.group, self.bits, self.rvecs[k], self.tvecs[k], self.K,
                                           self.dist, self.width, self.height, self.bg, self.supersample)
        return self._frames[k]

    def frames(self):
        return np.stack([self.frame(k) for k in range(len(self))])

    def coverage(self, k):
        """(H, W) float32: sum of the tags' pixel coverage in frame k (> 1 where two tags overlap in the image)"""
        cov = np.zeros((self.height, self.width), np.float32)
        render_frame(self.group, self.bits, self.rvecs[k], self.tvecs[k], self.K, self.dist, self.width, self.height,
                     self.bg, 1, cover_out=cov)
        return cov
