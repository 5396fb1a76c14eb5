/*
 * cv_track.c -- CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE).  See cv_oracle.h.
 *
 * One whole per-frame step of the north-star path on the CPU, in the reference's
 * call order: build the new frame's pyramid, calcOpticalFlowPyrLK(prev -> next) for the
 * AprilGroup corners, then solvePnP(SOLVEPNP_ITERATIVE) with the extrinsic guess over the
 * tracked corners (reference call site for the PnP: detect_pose.py:517-526; the LK step
 * fills the hole at detect_pose.py:573-574).  Used for parity of the fused HIP step and
 * as bench.py's timed cpu_baseline ("port").
 */
#include "cv_oracle.h"
#include <stdlib.h>

static int track_frame_impl(const cvo_pyramid* prev_pyr, const uint8_t* next_img, int w, int h, int stride,
                    const float* prev_pts, float* next_pts, uint8_t* status, float* err, int npoints,
                    const double* obj, const double K[9], const double* dist, int ndist,
                    double rvec[3], double tvec[3], int use_guess,
                    int win, int max_level, int crit_max_count, double crit_eps,
                    int acc_mode, int nthreads, cvo_pyramid** next_pyr_out)
{
    if (!prev_pyr || !next_img || !obj || npoints <= 0) return -1;
    cvo_pyramid* N = cvo_pyramid_build(next_img, w, h, stride, win, win, max_level);
    if (!N) return -2;
    int rc = cvo_lk_on_pyramids(prev_pyr, N, prev_pts, next_pts, status, err, npoints, win, win, max_level,
                                CVO_TERM_COUNT | CVO_TERM_EPS, crit_max_count, crit_eps, 0, 1e-4,
                                acc_mode, nthreads);
    if (rc) { cvo_pyramid_free(N); return rc; }
    double* o = (double*)malloc((size_t)npoints * 5 * sizeof(double));
    if (!o) { cvo_pyramid_free(N); return -2; }
    double* m = o + (size_t)npoints * 3;
    int cnt = 0;
    for (int i = 0; i < npoints; i++) {
        if (!status[i]) continue;
        o[cnt * 3] = obj[i * 3]; o[cnt * 3 + 1] = obj[i * 3 + 1]; o[cnt * 3 + 2] = obj[i * 3 + 2];
        cnt++;
    }
    /* image points packed after the object points (m starts at o + 3*npoints) */
    cnt = 0;
    for (int i = 0; i < npoints; i++) {
        if (!status[i]) continue;
        m[cnt * 2] = (double)next_pts[i * 2]; m[cnt * 2 + 1] = (double)next_pts[i * 2 + 1];
        cnt++;
    }
    if (cnt >= 6 || (cnt >= 4 && use_guess)) {
        rc = cvo_solve_pnp_iterative(o, m, cnt, K, dist, ndist, rvec, tvec, use_guess, NULL);
        if (rc) cnt = rc;
    } else cnt = 0;
    free(o);
    if (next_pyr_out) *next_pyr_out = N; else cvo_pyramid_free(N);
    return cnt;
}

int cvo_track_frame(const cvo_pyramid* prev_pyr, const uint8_t* next_img, int w, int h, int stride,
                    const float* prev_pts, float* next_pts, uint8_t* status, float* err, int npoints,
                    const double* obj, const double K[9], const double* dist, int ndist,
                    double rvec[3], double tvec[3], int use_guess,
                    int win, int max_level, int crit_max_count, double crit_eps,
                    int acc_mode, int nthreads, cvo_pyramid** next_pyr_out)
{
    /* the full-frame passes (pyrDown, Scharr) run in `nthreads` bands of rows, the per-point loop over points */
    const int old = cvo_get_num_threads();
    cvo_set_num_threads(nthreads);
    const int rc = track_frame_impl(prev_pyr, next_img, w, h, stride, prev_pts, next_pts, status, err, npoints, obj, K, dist, ndist,
                                    rvec, tvec, use_guess, win, max_level, crit_max_count, crit_eps, acc_mode, nthreads, next_pyr_out);
    cvo_set_num_threads(old);
    return rc;
}
