/*
 * cv_oracle.h -- CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE)
 *
 * Plain-C restatement of the OpenCV algorithms that the reference's per-frame
 * hot path calls (reference: aprilgroup_pose_estimation/detect_pose.py:509-526
 * cv.solvePnP(SOLVEPNP_ITERATIVE); transform_helper.py:87 cv.Rodrigues;
 * transform_helper.py:106-111 cv.projectPoints) plus the north-star
 * cv.calcOpticalFlowPyrLK step that has no call site in the reference
 * (hole at detect_pose.py:573-574).
 *
 * The arithmetic lives in a third-party dependency that is absent from
 * /root/reference: OpenCV (unpinned; requirements.txt:1 pins only numpy;
 * README.md:104 implies Ubuntu 20.04 python3-opencv = 4.2.0).  This file
 * restates its published algorithms:
 *   modules/video/src/lkpyramid.cpp   (calcSharrDeriv, LKTrackerInvoker,
 *                                      buildOpticalFlowPyramid)
 *   modules/imgproc/src/pyramids.cpp  (pyrDown, u8, 5x5 [1 4 6 4 1])
 *   modules/calib3d/src/calibration.cpp (cvRodrigues2, cvProjectPoints2Internal,
 *                                      cvFindExtrinsicCameraParams2)
 *   modules/calib3d/src/compat_ptsetreg.cpp (CvLevMarq)
 *   modules/imgproc/src/undistort.dispatch.cpp (cvUndistortPointsInternal)
 *
 * PARITY UNPINNED: cv2 cannot be imported or installed in this pipeline and
 * the reference ships no tests, golden vectors or fixtures for this path
 * (SURVEY.md section 8c).  What pins this oracle instead: analytic ground truth from
 * the synthetic generator, scipy cross-checks (Rotation, least_squares,
 * ndimage.correlate1d) and the fixtures under tests/golden/.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library.  The product (accurate_aprilgroup_tracking_amd/) never does.
 */
#ifndef CV_ORACLE_H
#define CV_ORACLE_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CVO_MAX_LEVELS 8

/* LK flags (cv::OPTFLOW_*) */
#define CVO_LK_USE_INITIAL_FLOW   4
#define CVO_LK_GET_MIN_EIGENVALS  8
/* TermCriteria type bits */
#define CVO_TERM_COUNT 1
#define CVO_TERM_EPS   2

/* accumulation mode for the LK inner sums */
#define CVO_ACC_EXACT        0  /* exact integer sums, rounded once (every OpenCV build approximates this) */
#define CVO_ACC_FLOAT_SCALAR 1  /* OpenCV's non-SIMD order: sequential float adds in (y,x) order */
#define CVO_ACC_FLOAT_SIMD   2  /* OpenCV's CV_SIMD128 order on x86 (SSE baseline, no FMA) [OpenCV-knowledge, 4.x lkpyramid.cpp]: four float
                                   lanes per covariance sum over 8-pixel steps, int32 pair sums (v_dotprod) for the mismatch sums, the
                                   window's last ww % 8 columns through the scalar loop, horizontal sums (l0 + l2) + (l1 + l3) at the end */

typedef struct cvo_pyramid cvo_pyramid;

/* ---- image pyramid (pyramids.cpp pyrDown_, lkpyramid.cpp buildOpticalFlowPyramid) ---- */
/* dst is ((sw+1)/2) x ((sh+1)/2); strides in bytes */
int cvo_pyr_down_u8(const uint8_t* src, int sw, int sh, int sstride,
                    uint8_t* dst, int dstride);
/* Builds levels 0..max_level (stops early as OpenCV does); every level is
 * stored with a (win_w, win_h) REFLECT_101 border. */
cvo_pyramid* cvo_pyramid_build(const uint8_t* img, int w, int h, int stride,
                               int win_w, int win_h, int max_level);
void cvo_pyramid_free(cvo_pyramid* p);
int  cvo_pyramid_levels(const cvo_pyramid* p);              /* = max usable level index */
int  cvo_pyramid_level_size(const cvo_pyramid* p, int level, int* w, int* h);
/* copies the un-padded level into dst (tight, dstride bytes per row) */
int  cvo_pyramid_level_copy(const cvo_pyramid* p, int level, uint8_t* dst, int dstride);

/* ---- Scharr derivative image (lkpyramid.cpp calcSharrDeriv), s16 interleaved (dx,dy) ---- */
/* dstride in int16 elements per row (>= 2*w) */
int cvo_scharr_deriv(const uint8_t* src, int w, int h, int sstride,
                     int16_t* dst, int dstride);

/* ---- cv::calcOpticalFlowPyrLK ---- */
/* next_pts is in/out (read only with CVO_LK_USE_INITIAL_FLOW). err may be NULL. */
int cvo_calc_optical_flow_pyr_lk(const uint8_t* prev_img, const uint8_t* next_img,
                                 int w, int h, int stride,
                                 const float* prev_pts, float* next_pts,
                                 uint8_t* status, float* err, int npoints,
                                 int win_w, int win_h, int max_level,
                                 int crit_type, int crit_max_count, double crit_eps,
                                 int flags, double min_eig_threshold,
                                 int acc_mode, int nthreads);
/* same, on pre-built pyramids (the streaming case: prev pyramid is cached) */
int cvo_lk_on_pyramids(const cvo_pyramid* prev_pyr, const cvo_pyramid* next_pyr,
                       const float* prev_pts, float* next_pts,
                       uint8_t* status, float* err, int npoints,
                       int win_w, int win_h, int max_level,
                       int crit_type, int crit_max_count, double crit_eps,
                       int flags, double min_eig_threshold,
                       int acc_mode, int nthreads);

/* ---- cv::Rodrigues (calibration.cpp cvRodrigues2) ---- */
/* jac (may be NULL) is 3x9 row-major: jac[i*9+k] = dR[k]/dr[i] */
void cvo_rodrigues_vec2mat(const double r[3], double R[9], double* jac);
/* jac (may be NULL) is 9x3 row-major as OpenCV: jac[k*3+i] = dr[i]/dR[k] */
int  cvo_rodrigues_mat2vec(const double R[9], double r[3], double* jac);

/* ---- cv::projectPoints (cvProjectPoints2Internal), distortion k1 k2 p1 p2 [k3 [k4 k5 k6 [s1..s4]]] ---- */
/* obj: n x 3, img_out: n x 2, dpdr/dpdt (may be NULL): (2n) x 3 row-major */
int cvo_project_points(const double* obj, int n, const double rvec[3], const double tvec[3],
                       const double K[9], const double* dist, int ndist,
                       double* img_out, double* dpdr, double* dpdt);

/* ---- cv::undistortPoints as used by solvePnP (5 fixed iterations, R=I, no P) ---- */
/* tilted-sensor matrices of the 14-coefficient model (k[12] = tau_x, k[13] = tau_y): identity when both are zero */
void cvo_tilt_matrices(double tauX, double tauY, double matTilt[9], double invMatTilt[9]);
void cvo_tilt_of(const double k[14], double matTilt[9], double invMatTilt[9]);
int cvo_undistort_points(const double* img, int n, const double K[9],
                         const double* dist, int ndist, double* out);

/* ---- cv::solvePnP(..., SOLVEPNP_ITERATIVE) (cvFindExtrinsicCameraParams2 + CvLevMarq) ---- */
/* returns 0 on success (<0 on argument error).  rvec/tvec are in/out when use_guess.
 * iters_out (may be NULL) receives CvLevMarq's iteration count. */
int cvo_solve_pnp_iterative(const double* obj, const double* img, int n,
                            const double K[9], const double* dist, int ndist,
                            double rvec[3], double tvec[3], int use_guess,
                            int* iters_out);
/* initialisation only (DLT / planar homography), exposed for tests */
int cvo_pnp_init(const double* obj, const double* img, int n,
                 const double K[9], const double* dist, int ndist,
                 double rvec[3], double tvec[3]);

/* mean reprojection error as transform_helper.py:98-121 (mean of per-point L2 norms) */
double cvo_mean_reproj_error(const double* obj, const double* img, int n,
                             const double rvec[3], const double tvec[3],
                             const double K[9], const double* dist, int ndist);

/* ---- frame pre-processing (cv_imgproc.c): detect_pose.py:147-183 undistort_frame, :602 cvtColor ---- */
int cvo_cvt_bgr2gray(const uint8_t* src, int w, int h, int sstride, uint8_t* dst, int dstride);
int cvo_get_optimal_new_camera_matrix(const double K[9], const double* dist, int ndist, int w, int h, double alpha,
                                      int new_w, int new_h, double newK[9], int roi[4]);
int cvo_init_undistort_rectify_map(const double K[9], const double* dist, int ndist, const double newK[9],
                                   int w, int h, int16_t* map1, uint16_t* map2);
int cvo_remap_bilinear_u8(const uint8_t* src, int sw, int sh, int sstride, int cn,
                          const int16_t* map1, const uint16_t* map2, int dw, int dh, uint8_t* dst, int dstride);
int cvo_undistort_u8(const uint8_t* src, int w, int h, int sstride, int cn, const double K[9], const double* dist, int ndist,
                     const double newK[9], uint8_t* dst, int dstride);

/* ---- dense photometric pose refinement (cv_dense.c; semantics DEFINED by this build, see that file) ---- */
int cvo_dense_sample(const uint8_t* img, int w, int h, int stride, double u, double v, const double ju[6], const double jv[6],
                     double t, double* r, double jrow[6]);
int cvo_dense_refine(const uint8_t* img, int w, int h, int stride,
                     const float* model_xyz, const float* model_t, int M,
                     const float* obj, const float* img_pts, const uint8_t* mask, int N,
                     const double K[9], const double* dist, int ndist,
                     double pose[6], int iters, double photo_weight, double mu, double stats[8]);

/* ---- small dense linear algebra (one-sided Jacobi SVD, as OpenCV's JacobiSVDImpl_) ---- */
/* A: m x n row-major (m >= n). w: n, u: m x n (columns = left vectors), vt: n x n. sorted descending. */
int cvo_svd(const double* A, int m, int n, double* w, double* u, double* vt);
/* cv::findHomography(M -> m, method 0): points rounded to float32, normalised DLT, and (refine != 0, count > 4) the
 * LMSolver polish of fundam.cpp / levmarq.cpp (<= 10 iterations).  M, m: count x 2 doubles; H row-major, H[8] = 1. */
int cvo_find_homography(const double* M, const double* m, int count, int refine, double H[9]);
/* solve A x = b (n x n) through the SVD pseudo-inverse (cv::solve DECOMP_SVD) */
int cvo_solve_svd(const double* A, const double* b, int n, double* x);

/* ---- whole reference CPU frame step, for the timed baseline ---- */
/* threads used by the full-frame passes (pyrDown, Scharr) -- bands of rows, results independent of the count */
void cvo_set_num_threads(int n);
int cvo_get_num_threads(void);

/* pyramid(next) + LK(prev_pyr -> next) + solvePnP(guess) on status==1 points.
 * Returns number of tracked points used (<0 on error). next_pyr_out receives the new pyramid. */
int cvo_track_frame(const cvo_pyramid* prev_pyr, const uint8_t* next_img, int w, int h, int stride,
                    const float* prev_pts, float* next_pts, uint8_t* status, float* err, int npoints,
                    const double* obj, const double K[9], const double* dist, int ndist,
                    double rvec[3], double tvec[3], int use_guess,
                    int win, int max_level, int crit_max_count, double crit_eps,
                    int acc_mode, int nthreads, cvo_pyramid** next_pyr_out);

#ifdef __cplusplus
}
#endif
#endif
