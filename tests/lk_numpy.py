"""Independent, slow numpy / int64 statement of cv::calcOpticalFlowPyrLK (8-bit, single channel).

TEST INFRASTRUCTURE.  Written from the prose of SURVEY.md Appendix A (pyramid, criteria, Scharr, the per-point
tracker steps 1-7), NOT from oracle/cv_lk.c: whole images are filtered with scipy.ndimage.correlate1d, the
window is handled as 21x21 arrays indexed into edge-padded images (np.pad) instead of pointer walks over padded
buffers, and every integer quantity is int64.  It exists so that the C oracle's per-point loop -- bilinear weights,
DESCALE shifts, the 2x2 solve, the three stop rules, the level-0 error -- has a second, structurally different
implementation to be compared with BIT FOR BIT (tests/test_oracle.py::test_lk_oracle_equals_numpy_statement).

Inner sums are formed exactly and rounded once to float32 (the oracle's CVO_ACC_EXACT mode, DESIGN.md section 2
deviation 1); `float_order=True` instead accumulates in float32 in raster order, which is OpenCV's scalar loop;
`float_order="simd"` is the order of OpenCV's CV_SIMD128 loops on x86 [OpenCV-knowledge, 4.x lkpyramid.cpp; SURVEY.md
Appendix A step 4 "SIMD builds sum int32 pairs first"], stated here with float32 VECTORS of four lanes (numpy arrays)
where the C oracle (CVO_ACC_FLOAT_SIMD) walks the lanes one scalar at a time.
"""
import numpy as np
from scipy.ndimage import correlate1d

F = np.float32
W_BITS = 14
OPTFLOW_USE_INITIAL_FLOW, OPTFLOW_LK_GET_MIN_EIGENVALS = 4, 8
TERM_COUNT, TERM_EPS = 1, 2


def _descale(x, n):
    return (x + (1 << (n - 1))) >> n


def pyr_down(img):
    k = np.array([1, 4, 6, 4, 1], np.int64)
    full = correlate1d(correlate1d(img.astype(np.int64), k, axis=1, mode="mirror"), k, axis=0, mode="mirror")
    return ((full[::2, ::2] + 128) >> 8).astype(np.uint8)


def build_pyramid(img, win, max_level):
    levels = [np.asarray(img, np.uint8)]
    for _ in range(max_level):
        h, w = levels[-1].shape
        if (w + 1) // 2 <= win or (h + 1) // 2 <= win:
            break
        levels.append(pyr_down(levels[-1]))
    return levels


def scharr(img):
    """-> (dx, dy) int64 arrays; reflect-101 inside the image"""
    a = img.astype(np.int64)
    smooth, diff = np.array([3, 10, 3], np.int64), np.array([-1, 0, 1], np.int64)
    dx = correlate1d(correlate1d(a, smooth, axis=0, mode="mirror"), diff, axis=1, mode="mirror")
    dy = correlate1d(correlate1d(a, diff, axis=0, mode="mirror"), smooth, axis=1, mode="mirror")
    return dx, dy


def _weights(fx, fy):
    """14-bit bilinear weights from the float32 fractions (cvRound = round half to even = np.rint)"""
    one, s = F(1.0), F(1 << W_BITS)
    w00 = int(np.rint((one - fx) * (one - fy) * s))
    w01 = int(np.rint(fx * (one - fy) * s))
    w10 = int(np.rint((one - fx) * fy * s))
    return w00, w01, w10, (1 << W_BITS) - w00 - w01 - w10


def _bilinear(pad, off, ix, iy, win, w):
    """sum of the four taps of every window pixel (int64 [win, win]); `pad` is the image padded by `off`"""
    y0, x0 = iy + off, ix + off
    p = pad[y0:y0 + win + 1, x0:x0 + win + 1]
    return p[:-1, :-1] * w[0] + p[:-1, 1:] * w[1] + p[1:, :-1] * w[2] + p[1:, 1:] * w[3]


def _reduce4(q):
    """v_reduce_sum(v_float32x4) on SSE: the high half is added onto the low half, then lane 1 onto lane 0"""
    half = (q[:2] + q[2:]).astype(F)
    return F(half[0] + half[1])


def _sum_f32_simd(prod, kind):
    """float32 sum of the integer window `prod` ([win, win]) in the vector loops' order.  The first win // 8 * 8 columns of
    every row go through the vector loop in steps of eight, the other columns through the scalar loop (its own accumulator,
    raster order); the two meet at the end.  kind "cov": two groups of four pixels per step, pixel 4 g + l into lane l
    (products rounded to float, then added: no FMA).  kind "mis": the step's int32 PAIR sums p[k] + p[k + 4], k = 0..3 --
    k = 0, 1 are lanes of one register, k = 2, 3 of a second one; the registers are added, then the two lanes."""
    win = prod.shape[1]
    vw = win // 8 * 8
    tail = F(0.0)
    for v in prod[:, vw:].ravel():
        tail = F(tail + F(int(v)))
    if vw == 0 or kind == "err":
        return _sum_f32(prod, True)
    if kind == "cov":
        q = np.zeros(4, F)
        for row in prod[:, :vw].reshape(prod.shape[0], vw // 4, 4):
            for four in row:
                q = (q + four.astype(F)).astype(F)
        return F(tail + _reduce4(q))
    qa, qb = np.zeros(2, F), np.zeros(2, F)                              # this sum's two lanes of qb0 (k = 0, 1) and of qb1 (k = 2, 3)
    for row in prod[:, :vw].reshape(prod.shape[0], vw // 8, 2, 4):
        for step in row:
            pairs = step[0] + step[1]                                    # exact integers: p[k] + p[k + 4]
            qa = (qa + pairs[:2].astype(F)).astype(F)
            qb = (qb + pairs[2:].astype(F)).astype(F)
    q = (qa + qb).astype(F)
    return F(tail + _reduce4(np.array([q[0], q[1], 0, 0], F)))


def _sum_f32(prod, float_order, kind="cov"):
    if float_order == "simd":
        return _sum_f32_simd(prod, kind)
    if float_order:
        acc = F(0.0)
        for v in prod.ravel():
            acc = F(acc + F(int(v)))
        return acc
    return F(np.float64(int(prod.sum())))


def calc_optical_flow_pyr_lk(prev, nxt, prev_pts, next_pts=None, win=21, max_level=3, criteria=(3, 30, 0.01), flags=0,
                             min_eig_threshold=1e-4, float_order=False):
    pts = np.asarray(prev_pts, F).reshape(-1, 2)
    n = pts.shape[0]
    out = np.zeros((n, 2), F)
    if next_pts is not None and (flags & OPTFLOW_USE_INITIAL_FLOW):
        out[:] = np.asarray(next_pts, F).reshape(-1, 2)
    status = np.ones(n, np.uint8)
    err = np.zeros(n, F)
    max_count = min(max(int(criteria[1]), 0), 100) if criteria[0] & TERM_COUNT else 30
    eps = min(max(float(criteria[2]), 0.0), 10.0) if criteria[0] & TERM_EPS else 0.01
    eps *= eps
    pyr_i, pyr_j = build_pyramid(prev, win, max_level), build_pyramid(nxt, win, max_level)
    top = min(len(pyr_i), len(pyr_j)) - 1
    off = win + 2
    half = F((win - 1) * 0.5)
    flt_scale = F(1.0) / F(1 << 20)
    area2 = F(2 * win * win)
    for level in range(top, -1, -1):
        img_i, img_j = pyr_i[level], pyr_j[level]
        rows, cols = img_i.shape
        pad_i = np.pad(img_i.astype(np.int64), off, mode="reflect")
        pad_j = np.pad(img_j.astype(np.int64), off, mode="reflect")
        dx, dy = scharr(img_i)
        pad_dx, pad_dy = np.pad(dx, off), np.pad(dy, off)            # derivative border: zeros
        scale = F(1.0 / (1 << level))
        for i in range(n):
            p = pts[i] * scale
            if level == top:
                q = out[i] * scale if (flags & OPTFLOW_USE_INITIAL_FLOW) else p.copy()
            else:
                q = out[i] * F(2.0)
            out[i] = q
            p = p - half
            ix, iy = int(np.floor(p[0])), int(np.floor(p[1]))
            if ix < -win or ix >= cols or iy < -win or iy >= rows:
                if level == 0:
                    status[i] = 0
                    err[i] = 0
                continue
            w = _weights(F(p[0] - F(ix)), F(p[1] - F(iy)))
            patch = _descale(_bilinear(pad_i, off, ix, iy, win, w), W_BITS - 5)
            gx = _descale(_bilinear(pad_dx, off, ix, iy, win, w), W_BITS)
            gy = _descale(_bilinear(pad_dy, off, ix, iy, win, w), W_BITS)
            a11 = _sum_f32(gx * gx, float_order) * flt_scale
            a12 = _sum_f32(gx * gy, float_order) * flt_scale
            a22 = _sum_f32(gy * gy, float_order) * flt_scale
            det = F(F(a11 * a22) - F(a12 * a12))
            d = F(a11 - a22)
            min_eig = F(F(F(a22 + a11) - np.sqrt(F(F(d * d) + F(F(F(4.0) * a12) * a12)))) / area2)
            if flags & OPTFLOW_LK_GET_MIN_EIGENVALS:
                err[i] = min_eig
            if float(min_eig) < min_eig_threshold or det < np.finfo(F).eps:
                if level == 0:
                    status[i] = 0
                continue
            inv = F(F(1.0) / det)
            q = q - half
            prev_delta = np.zeros(2, F)
            for j in range(max_count):
                jx, jy = int(np.floor(q[0])), int(np.floor(q[1]))
                if jx < -win or jx >= cols or jy < -win or jy >= rows:
                    if level == 0:
                        status[i] = 0
                    break
                w = _weights(F(q[0] - F(jx)), F(q[1] - F(jy)))
                diff = _descale(_bilinear(pad_j, off, jx, jy, win, w), W_BITS - 5) - patch
                b1 = _sum_f32(diff * gx, float_order, "mis") * flt_scale
                b2 = _sum_f32(diff * gy, float_order, "mis") * flt_scale
                delta = np.array([F(F(F(a12 * b2) - F(a22 * b1)) * inv), F(F(F(a12 * b1) - F(a11 * b2)) * inv)], F)
                q = q + delta
                out[i] = q + half
                if float(delta[0]) * float(delta[0]) + float(delta[1]) * float(delta[1]) <= eps:
                    break
                if j > 0 and abs(float(F(delta[0] + prev_delta[0]))) < 0.01 and abs(float(F(delta[1] + prev_delta[1]))) < 0.01:
                    out[i] = out[i] - delta * F(0.5)
                    break
                prev_delta = delta
            if status[i] and level == 0 and not (flags & OPTFLOW_LK_GET_MIN_EIGENVALS):
                e = out[i] - half
                ex, ey = int(np.floor(e[0])), int(np.floor(e[1]))
                if ex < -win or ex >= cols or ey < -win or ey >= rows:
                    status[i] = 0
                    continue
                w = _weights(F(e[0] - F(ex)), F(e[1] - F(ey)))
                diff = np.abs(_descale(_bilinear(pad_j, off, ex, ey, win, w), W_BITS - 5) - patch)
                err[i] = F(_sum_f32(diff, float_order, "err") * F(1.0)) / F(32 * win * win)
    return out.reshape(-1, 1, 2), status.reshape(-1, 1), err.reshape(-1, 1)
