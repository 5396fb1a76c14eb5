/*
 * agt_hip.h -- C ABI of the MI355X (gfx950) AprilGroup tracking hot path.
 *
 * The reference (Virtana/accurate-aprilgroup-tracking) has no FFI layer: its hot path
 * is Python calling OpenCV.  A drop-in therefore replaces the four
 * cv2 entry points the path uses, plus a fused per-frame step, behind plain C
 * functions that a Python maintainer binds with ctypes:
 *
 *   agt_solve_pnp        <- cv.solvePnP(..., flags=SOLVEPNP_ITERATIVE)
 *                           /root/reference/aprilgroup_tracking/aprilgroup_pose_estimation/detect_pose.py:509-515 (no guess)
 *                           detect_pose.py:517-526 (useExtrinsicGuess=True)
 *   agt_project_points   <- cv.projectPoints      transform_helper.py:106-111, detect_pose.py:455-461
 *   (d_err of agt_solve_pnp) <- TransformHelper.get_reprojection_error   transform_helper.py:98-121
 *   agt_pyramid_build /
 *   agt_lk_track         <- cv.calcOpticalFlowPyrLK (north-star step; no call site in the
 *                           reference, belongs at the hole detect_pose.py:573-574)
 *   agt_track_frame      <- one PoseDetector._estimate_pose step (detect_pose.py:467-574)
 *                           with LK-tracked corners, state kept on the device
 *
 * Conventions: every function returns 0 (AGT_OK) or a negative AGT_ERR_*; nothing
 * throws across the ABI.  Pointers named d_* are raw device addresses (e.g.
 * torch.Tensor.data_ptr()); the library never frees or retains caller memory beyond
 * what each function documents.  All work is enqueued on the context's HIP stream;
 * nothing synchronises unless stated.  One context per host thread / stream; a context
 * is not re-entrant.  B = number of independent streams (frames/sequences) per call.
 */
#ifndef AGT_HIP_H
#define AGT_HIP_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AGT_VERSION 505

#define AGT_OK               0
#define AGT_ERR_ARG         (-1)   /* NULL pointer / bad size / bad shape (cv2 would raise cv2.error) */
#define AGT_ERR_ALLOC       (-2)
#define AGT_ERR_DIST        (-3)   /* distortion count not in {0,4,5,8,12,14} */
#define AGT_ERR_NPOINTS     (-4)   /* too few / too many points */
#define AGT_ERR_HIP         (-5)   /* a HIP runtime call failed; see agt_last_hip_error */
#define AGT_ERR_UNSUPPORTED (-6)   /* e.g. LK window outside 3..63, a device other than gfx950 */
#define AGT_ERR_STATE       (-7)   /* pyramid slot not built, context mismatch */
#define AGT_ERR_CHAIN       (-8)   /* a chained launch gave up waiting for a stream's corners (records flagged AGT_TRK_CHAIN_TIMEOUT);
                                      reported by agt_synchronize / agt_tracker_join until agt_tracker_reset */

#define AGT_MAX_LEVELS 6

/* cv::OPTFLOW_* / cv::TermCriteria bits (same numeric values as OpenCV) */
#define AGT_LK_USE_INITIAL_FLOW   4
#define AGT_LK_GET_MIN_EIGENVALS  8
#define AGT_TERM_COUNT 1
#define AGT_TERM_EPS   2

/* element type of object / image point arrays */
#define AGT_F32 0
#define AGT_F64 1

/* agt_solve_pnp info[b*4 + ...] */
#define AGT_INFO_OK     0   /* 1 = a pose was produced */
#define AGT_INFO_ITERS  1   /* CvLevMarq iteration count */
#define AGT_INFO_NUSED  2   /* points with mask != 0 */
#define AGT_INFO_FLAGS  3   /* AGT_PNP_* bits */
#define AGT_PNP_SINGULAR  1 /* a damped normal-equation solve hit a non-positive pivot */
#define AGT_PNP_PLANAR    2 /* un-guessed solve initialised through the planar (homography) branch */
#define AGT_PNP_TOO_FEW   4 /* fewer usable points than the solve needs; pose untouched */

/* agt_track_frame state flags, state_out[b*AGT_STATE_STRIDE + ...] (doubles) */
#define AGT_STATE_STRIDE 16
#define AGT_ST_RVEC    0    /* 0..2 pose rvec of this frame (valid if AGT_ST_OK) */
#define AGT_ST_TVEC    3    /* 3..5 pose tvec */
#define AGT_ST_OK      6    /* 1.0 = pose accepted (mean reprojection error < gate) */
#define AGT_ST_ERR     7    /* mean reprojection error, px (transform_helper.py:98-121) */
#define AGT_ST_NTRACK  8    /* corners with LK status 1 */
#define AGT_ST_ITERS   9    /* LM iterations */
#define AGT_ST_GUESS   10   /* 1.0 = an extrinsic guess was used for this frame */
#define AGT_ST_FLAGS   11   /* AGT_PNP_* bits | AGT_TRK_* bits */
#define AGT_ST_TVEC_F32 12  /* 1.0 = tvec carries float32 precision (cv2 wrote it into the f32 guess array) */
#define AGT_TRK_ZERO_VELOCITY 256  /* a velocity element was exactly 0: reference raises ValueError (detect_pose.py:236-237) */
#define AGT_TRK_CHAIN_TIMEOUT 512  /* pipelined tracker: the pose solve gave up waiting for the frame's corners (or did so for an earlier frame of
                                      the stream): nothing was solved, the record is invalid, the stream's state is frozen until agt_tracker_reset */

typedef struct agt_ctx agt_ctx;

#define AGT_WIN_RECT(w, h) ((w) | ((h) << 8))     /* agt_config::win of a w x h window (w = h: the side itself) */

typedef struct agt_config {
    int device;        /* HIP device ordinal */
    int width, height; /* level-0 frame size in pixels */
    int max_level;     /* LK maxLevel: pyramid levels 0..max_level ("3-level" = 2) */
    int win;           /* LK window (cv2 winSize): the side of a square window, or AGT_WIN_RECT(w, h); 3 <= w, h <= 63.  21 x 21 -- the
                          north-star's -- has the specialised bodies and is the only window of the pipelined tracker; 15 and 31 have compiled-in
                          bodies; every other size runs the general body (ABI 504), same results, slower */
    int max_points;    /* correspondences per stream (<= 256) */
    int max_streams;   /* B upper bound */
    int reserved[8];   /* must be 0 */
} agt_config;

/* ---- lifetime ---- */
int  agt_version(void);
/* What the context's device reported at agt_create (round 5): CU count (hipDeviceProp_t::multiProcessorCount), the XCD count the
 * XCD-aware block orders are laid out for (gfx950: one XCD per 32 CUs -- 8 on a whole MI355X, 4 / 2 / 1 on its DPX / QPX / CPX
 * partitions; 1 = plain order when the CU count is not 32 x a power of two) and the architecture name.  agt_create refuses
 * devices other than gfx950 with AGT_ERR_UNSUPPORTED (the library holds gfx950 code objects only).  The launch rules that used to
 * say "256 CUs" (fused step while one LK workgroup per CU is co-resident, chained dense launch) read this CU count. */
int  agt_device_info(const agt_ctx* ctx, int* cus, int* xcds, char* arch, size_t arch_cap);
/* The XCD-aware block order as a pure index map (host function, no GPU needed): block `block` of a launch of `nblocks` blocks
 * (a multiple of xcds; xcds in {1, 2, 4, 8}) works on item (block mod xcds) * (nblocks / xcds) + block / xcds.  For every xcds
 * a permutation of 0 .. nblocks - 1 in which the blocks of one XCD (equal block mod xcds) take a contiguous run of items.
 * Returns the item, or AGT_ERR_ARG. */
int  agt_xcd_tile_order(int block, int nblocks, int xcds);
const char* agt_error_string(int code);
/* hip_stream: a hipStream_t (NULL = default stream).  Allocates the context's pyramid
 * storage (levels >= 1 for two slots), tracker state and scratch. */
int  agt_create(const agt_config* cfg, void* hip_stream, agt_ctx** out);
int  agt_destroy(agt_ctx* ctx);
int  agt_set_stream(agt_ctx* ctx, void* hip_stream);
int  agt_last_hip_error(const agt_ctx* ctx);   /* hipError_t of the last failing HIP call */
int  agt_synchronize(agt_ctx* ctx);            /* joins the tracker pipeline, then hipStreamSynchronize(ctx stream); AGT_ERR_CHAIN after a chain give-up */

/* ---- staging copies for host-array callers (the cv2-shaped Python functions): both on the context's stream;
 * agt_upload enqueues host -> device (pin the host buffer for it to be asynchronous), agt_download enqueues device -> host
 * and then WAITS for the stream, so the bytes are valid on return. ---- */
int agt_upload(agt_ctx* ctx, void* d_dst, const void* h_src, size_t bytes);
int agt_download(agt_ctx* ctx, void* h_dst, const void* d_src, size_t bytes);

/* ---- image pyramid (cv::pyrDown / buildOpticalFlowPyramid) ---- */
/* One pyrDown: dst is ((sw+1)/2) x ((sh+1)/2).  Pitches and batch strides in bytes;
 * base pointers and pitches must be multiples of 4. */
int agt_pyr_down_u8(agt_ctx* ctx, const uint8_t* d_src, int sw, int sh, size_t spitch, size_t sbatch,
                    uint8_t* d_dst, size_t dpitch, size_t dbatch, int B);
/* Build slot (0/1) of the context's pyramid from B frames of cfg.width x cfg.height.
 * Level 0 aliases d_frames: the caller keeps those frames valid and unmodified until
 * the last agt_lk_track / agt_track_frame that reads the slot has completed.  The two slots are ring
 * entries of the tracker as well: the call first joins a running tracker (agt_tracker_join), so the build
 * is ordered behind every frame in flight; building a slot invalidates the tracker's own use of that entry
 * until the next agt_tracker_reset. */
int agt_pyramid_build(agt_ctx* ctx, int slot, const uint8_t* d_frames, size_t pitch, size_t batch_stride, int B);
/* Both slots of a frame pair in one call: slot 0 <- d_prev, slot 1 <- d_next (same pitch / batch stride), i.e. the two pyramids
 * cv2.calcOpticalFlowPyrLK(prev, next, ...) builds internally -- levels 1 and 2 of all 2 B images by ONE launch of the
 * two-level pass where it applies (ABI 502).  Same ownership and ordering rules as agt_pyramid_build. */
int agt_pyramid_build_pair(agt_ctx* ctx, const uint8_t* d_prev, const uint8_t* d_next, size_t pitch, size_t batch_stride, int B);
/* Introspection (tests): device pointer and geometry of one level of a built slot. */
int agt_pyramid_level(const agt_ctx* ctx, int slot, int level, const uint8_t** d_ptr,
                      int* w, int* h, size_t* pitch, size_t* batch_stride);
/* number of usable levels - 1 (OpenCV stops early when a level would be <= win) */
int agt_pyramid_max_level(const agt_ctx* ctx);

/* ---- cv::calcOpticalFlowPyrLK on two built slots ---- */
/* d_prev_pts, d_next_pts: [B][n][2] f32.  d_next_pts is read only with
 * AGT_LK_USE_INITIAL_FLOW.  d_status: [B][n] u8.  d_err: [B][n] f32 or NULL. */
int agt_lk_track(agt_ctx* ctx, int prev_slot, int next_slot,
                 const float* d_prev_pts, float* d_next_pts, uint8_t* d_status, float* d_err,
                 int n, int B, int crit_type, int crit_max_count, double crit_eps,
                 int flags, double min_eig_threshold);

/* ---- cv::solvePnP(SOLVEPNP_ITERATIVE), batched ---- */
/* d_obj: n x 3 (obj_batch_stride = 0: shared by all B) or [B][n][3] (stride in elements).
 * d_img: [B][n][2].  dtype: AGT_F32 / AGT_F64 for both arrays.
 * d_mask: [B][n] u8 (0 = skip the point, e.g. LK status) or NULL.
 * K: 9 host doubles row-major.  dist: ndist host doubles (k1 k2 p1 p2 [k3 [k4 k5 k6 [s1..s4 [tau_x tau_y]]]]) or NULL -- cv2's
 * 4 / 5 / 8 / 12 / 14-coefficient models; tau_x, tau_y: the tilted-sensor term (what calibrateCamera returns under CALIB_TILTED_MODEL),
 * applied wherever cv2 applies it: projectPoints and its Jacobian, undistortPoints, the undistortion maps, getOptimalNewCameraMatrix.
 * d_pose: [B][6] f64 (rvec, tvec): read when use_guess, always written on success
 * (cv2 overwrites the guess arrays in place too, detect_pose.py:487-490).
 * d_info: [B][4] i32 (AGT_INFO_*), d_err: [B] f64 mean reprojection error; either may be NULL. */
int agt_solve_pnp(agt_ctx* ctx, const void* d_obj, size_t obj_batch_stride, const void* d_img, int dtype,
                  const uint8_t* d_mask, int n, int B,
                  const double* K, const double* dist, int ndist,
                  double* d_pose, int use_guess, int32_t* d_info, double* d_err);

/* ---- cv::projectPoints, batched ---- */
/* d_pose: [B][6] f64.  d_img_out: [B][n][2] in dtype.  d_jac: [B][2n][6] f64 (d/dr | d/dt) or NULL. */
int agt_project_points(agt_ctx* ctx, const void* d_obj, size_t obj_batch_stride, int dtype, int n, int B,
                       const double* d_pose, const double* K, const double* dist, int ndist,
                       void* d_img_out, double* d_jac);

/* ---- the same two calls SYNCHRONOUS, host arrays in and out: what the reference does once per frame (cv2.solvePnP at
 * detect_pose.py:509-526, cv2.projectPoints at :441-465).  One launch each and no copy: the arguments are placed in a host-mapped
 * staging area of the context, the kernel reads them and writes its results there, the calling thread polls a sequence word the
 * kernel stores behind them.  h_obj: n x 3, h_img: n x 2 (dtype AGT_F32 / AGT_F64 for both), 3 <= n <= 256 (4 without a guess);
 * h_pose: 6 doubles (rvec, tvec), read when use_guess, written on return; h_info: 4 int32 (AGT_INFO_*) or NULL; h_err: mean
 * reprojection error or NULL.  agt_project_points_host: 1 <= n <= 256, h_img_out n x 2 of dtype, h_jac [2n][6] doubles or NULL.
 * The calls are ordered on the context's stream like every other launch and return when THEIR result is there. */
int agt_solve_pnp_host(agt_ctx* ctx, const void* h_obj, const void* h_img, int dtype, int n,
                       const double* K, const double* dist, int ndist,
                       double* h_pose, int use_guess, int32_t* h_info, double* h_err);
int agt_project_points_host(agt_ctx* ctx, const void* h_obj, int dtype, int n, const double* h_pose,
                            const double* K, const double* dist, int ndist, void* h_img_out, double* h_jac);

/* ---- fused per-frame step (PoseDetector._estimate_pose with LK-tracked corners) ---- */
/* Tracker state lives in the context, one record per stream: current corners, the
 * extrinsic guess, prev_transform and the two-deep velocity buffers of
 * detect_pose.py:74-83, 229-349. */
/* (Re)initialise streams [0,B): corners [B][n][2] f32 seen in the frames of `slot`
 * (already built; NULL when only agt_estimate_pose will be used), shared object points
 * n x 3 f32, camera; clears guess, prev_transform and the velocity buffers. */
int agt_tracker_reset(agt_ctx* ctx, int slot, const float* d_corners, const float* d_obj, int n, int B,
                      const double* K, const double* dist, int ndist, int enhance_ape);
/* Tracker options: reproject != 0 -> after an accepted pose the corner set is refreshed with
 * projectPoints(all object points) (the projection the reference draws, detect_pose.py:441-465),
 * which revives lost corners and stops LK drift; 0 (default) chains raw LK outputs.
 * min_points: corners needed to attempt a pose (default 8 = the reference's >= 2 tags, detect_pose.py:494-496).
 * gate_px: reprojection gate (default 2.0, detect_pose.py:539). */
int agt_tracker_options(agt_ctx* ctx, int reproject, int min_points, double gate_px);
/* Residency cap of the big-batch LK kernel (one wave per corner, > 1024 corners per launch): at most workgroups_per_cu of its waves
 * resident per CU (four SIMDs); 0 = no cap; -1 (default) = the library's choice: none for a launch of its own (agt_lk_track), 10 for
 * the half-batch launches of the multi-stream tracker, which run beside the pyramid launch of the frames ahead.  For launches that
 * share the device with other kernels (several batches software-pipelined over contexts / streams, bench.py --workload c3pairs): an
 * uncapped launch fills every SIMD with three trackers for as long as its slowest corner iterates and the HBM-bound pyramid passes
 * beside it starve; capped at 8 the cold-pair step of BASELINE configs[2] ran 48.4 -> 45.3 us (round 5) while the LK launch alone
 * takes 45 instead of 36 us.  A context that sets a count also runs its tracker waves at issue priority 1 (round 6: cold pairs 43.1 -> 42.3 us).
 * Results do not depend on it.  No reference counterpart (cv2 has no notion of co-tenancy).
 * agt_lk_occupancy: the same in waves per SIMD (4 x waves_per_simd per CU; ABI 500). */
int agt_lk_occupancy_cu(agt_ctx* ctx, int workgroups_per_cu);
/* The LDS a one-wave LK workgroup asks for under that cap (host function, no GPU needed; 0 = no cap, or a cap the 64-KB limit of a
 * workgroup cannot express): workgroups_per_cu + 1 workgroups do not fit in a CU's 160 KB whatever the allocation granule ("at most"),
 * workgroups_per_cu do for granules up to 512 bytes (one fewer at worst for 1,024 / 1,280; exactly as many at 8 and 10 per CU, the counts
 * the library uses itself, for those too). */
int agt_lk_lds_request(int workgroups_per_cu);
int agt_lk_occupancy(agt_ctx* ctx, int waves_per_simd);
/* corners_per_tag = 4: the pose solve of the tracker uses a corner only while all four corners of its tag (corners 4t..4t+3)
 * are usable, and min_points = 8 then means the reference's ">= 2 tags" (detect_pose.py:494-496; its detections are whole
 * tags, :400-437).  0 (default): every usable corner counts. */
int agt_tracker_tag_gate(agt_ctx* ctx, int corners_per_tag);
/* Software pipelining across frames.  depth 0: separate launches per stage, the record of frame t is complete
 * in stream order after its call.  depth F in 1..32 (default 1; needs reproject == 0, otherwise the call falls back to
 * depth 0 behaviour): agt_track_frame registers the frame and, every F calls, issues ONE fused launch that advances every
 * pipeline stage by F frames, stages of different frames side by side in disjoint workgroup ranges: the pyramid of
 * frames t-F+1..t, the LK of the F frames before those, and the pose solves.  Up to 256 corners in flight the launch is
 * CHAINED: its PnP workgroups follow its LK workgroups frame by frame through per-frame arrival counters in device
 * memory (the LK role counts each corner in after a write-through store of its result; the PnP wave of the stream polls
 * the frame's counter, then acquires) -- one frame behind the LK role while frames keep coming, so neither role ever
 * stalls, and right behind it in the launches agt_tracker_join issues, so the last record of a clip is complete one LK
 * + one PnP latency after its pyramid.  A wait that is not satisfied within 2^16 polls (>= 30 ms of the waiting wave
 * executing; queue preemption does not count) gives up FAIL-STOP: the frame is not solved, its record is zeroed and flagged
 * AGT_TRK_CHAIN_TIMEOUT, the stream's tracker state is frozen, every later record of that stream is flagged too and no
 * later frame waits again, agt_synchronize / agt_tracker_join return AGT_ERR_CHAIN -- until agt_tracker_reset.  (The LK
 * workgroups have lower indices than the waiting ones, are dispatched first and wait for nothing themselves, so a
 * give-up needs a fault elsewhere; the design depends on in-index-order workgroup dispatch)  A step costs max(stage) instead of their sum and the launch boundary is paid once
 * per F frames; results are bit-identical to the serial order.  Frame t's state record is written up to (L+2)*F calls
 * later; agt_tracker_join enqueues the remaining stages of all supplied frames (no host synchronisation) and
 * agt_synchronize joins and waits.  Frames handed to agt_track_frame must stay valid and unmodified until their pose
 * has been produced ((L+2)*F + F frames are in flight at most), and the frames of one group must share pitch and batch
 * stride.  Changing the depth joins first.
 * With more than 256 corners in flight (depth >= 1) the same pipeline runs as one launch per role and
 * group: the pyramid role on the context's stream, the LK and PnP roles on two library-owned streams, ordered by events
 * (PnP one group behind LK); agt_tracker_join makes the context's stream wait for the library's.  Frame lifetime as above. */
int agt_tracker_pipeline(agt_ctx* ctx, int depth);
int agt_tracker_join(agt_ctx* ctx);
/* PoseDetector._estimate_pose (detect_pose.py:467-574) for B streams with device-resident
 * state: d_img [B][n][2] f32 corners (detector- or LK-supplied), d_mask [B][n] u8 or NULL.
 * Needs agt_tracker_reset first (its corners argument may be NULL when only this entry is used). */
int agt_estimate_pose(agt_ctx* ctx, const float* d_img, const uint8_t* d_mask, int B, double* d_state_out);
/* copy of the raw per-stream state records (AgtTrackState, see csrc/agt_kernels.h) for tests */
int agt_tracker_state_size(void);
int agt_tracker_state_read(agt_ctx* ctx, void* host_dst, int B);   /* synchronises the stream */
/* One frame for B streams: pyramid(new frames) -> LK(prev corners) -> solvePnP(guess) ->
 * reprojection gate -> motion-model guess update.  LK status is sticky: a corner whose status dropped to 0
 * (left the image, flat patch) is not tracked again -- its position is carried and it stays masked out of the
 * PnP -- until agt_tracker_reset re-seeds the corner set (or, with the reproject option, an accepted pose does).  d_state_out: [B][AGT_STATE_STRIDE] f64 or NULL
 * (device memory; read it back whenever convenient).  No host synchronisation. */
int agt_track_frame(agt_ctx* ctx, const uint8_t* d_frames, size_t pitch, size_t batch_stride, int B,
                    double* d_state_out);
/* One detector-fed frame for B streams (detect_pose.py:576-609 when the detector returned >= 2 tags): the frame becomes frame t
 * of the stream -- its pyramid is built, so the next agt_track_frame tracks FROM it --, d_corners [B][n][2] f32 / d_mask [B][n]
 * u8 (NULL = all) become its corner set and LK status (a corner the detector did not deliver is not trackable until the next
 * detector-fed frame), and PoseDetector._estimate_pose runs on the table (exactly agt_estimate_pose).  Works after an
 * agt_tracker_reset without corners.  Joins the pipeline first; pyramid pass + one PnP launch in stream order. */
int agt_track_frame_detected(agt_ctx* ctx, const uint8_t* d_frames, size_t pitch, size_t batch_stride, int B,
                             const float* d_corners, const uint8_t* d_mask, double* d_state_out);
/* Take the newest frame back as the tracking source (the reference keeps the older frame as "previous" when a frame yields no
 * tag at all: detect_pose.py:570-574, the hole the LK step fills, and the host mirror's `if ids:`): joins the pipeline, then the
 * next agt_track_frame tracks FROM frame t - 1 again (its pyramid, corner set and LK status; the caller's frame t - 1 must still
 * be valid) and re-uses frame t's ring entry.  The pose state machine is not touched (the lost frame cleared the guess, as
 * _estimate_pose does with < 2 tags).  A caller that reads every record calls this when AGT_ST_NTRACK came back 0 after an
 * agt_track_frame; PoseDetector(backend="stream") does.  All B streams go back together (use it with B = 1). */
int agt_tracker_rewind(agt_ctx* ctx);
/* The body of the reference's live loop (detect_pose.py:669-681) for ONE stream whose frames live on the HOST, in one call:
 * h_frame -> d_gray (context-size gray frame in device memory, row pitch gpitch; must stay valid while the NEXT frame is
 * tracked: alternate at least two buffers) -> agt_track_frame -> agt_tracker_join -> the frame's record in h_state
 * (AGT_STATE_STRIDE doubles, valid on return).  Needs B = 1 and a seeded tracker (agt_tracker_reset with corners or
 * agt_track_frame_detected).
 *   h_frame   channels = 1: gray W x H of the context (undistort, roi_x, roi_y must be 0); channels = 3: BGR src_w x src_h, passed
 *             through agt_preprocess_bgr with the given undistort flag and ROI origin.  PINNED host memory (hipHostMalloc /
 *             hipHostRegister -- PoseDetector.frame_buffer) is read by the device directly: a gray frame by the pyramid pass
 *             itself (which writes d_gray on the way: no separate upload), a BGR frame without undistortion by the gray
 *             conversion.  Pageable memory, BGR with undistortion, profiling spans and the stage-by-stage mode copy first.
 *   d_staging device buffer of src_w * src_h * 3 bytes for the copied BGR frame; may be NULL for channels = 1 (required for
 *             channels = 3 whenever the frame has to be copied: AGT_ERR_ARG without it)
 *   d_state   optional device copy of the record (stream-ordered, complete at the next synchronisation); NULL = none
 * The record reaches h_state without a copy command: the pose solver writes it to host-mapped memory of the context and
 * stores a sequence number behind it, which this call polls (falls back to waiting for the stream after 2 s and reports the
 * runtime's error).  A chain fault of the step is returned as AGT_ERR_CHAIN.
 * THE STREAM IS NOT IDLE ON RETURN: the call comes back when the record is in h_state, while the tail of the pose kernel and the
 * optional d_state copy may still be running on the context's stream.  Work the caller issues on that stream is ordered behind
 * them as usual; a caller that touches d_staging, d_gray or d_state from ANOTHER stream (or from the host) must order itself
 * behind the context's stream first (agt_synchronize or an event) -- the implicit stream wait of ABI <= 300 is gone. */
int agt_track_host_frame(agt_ctx* ctx, const uint8_t* h_frame, int channels, int src_w, int src_h, uint8_t* d_staging,
                         int undistort, int roi_x, int roi_y, uint8_t* d_gray, size_t gpitch, double* d_state, double* h_state);
/* A clip: `count` consecutive frames of the B streams in one call, frame k at d_frames + k * frame_stride (bytes), its
 * record at d_state_out + k * B * AGT_STATE_STRIDE (or NULL).  Exactly `count` calls of agt_track_frame, made without the
 * per-call host cost (at ~16 us of device time per 720p frame a Python caller's ~5 us per call is a third of the budget):
 * same pipeline, same launches, same records. */
int agt_track_frames(agt_ctx* ctx, const uint8_t* d_frames, size_t pitch, size_t batch_stride, size_t frame_stride, int B, int count,
                     double* d_state_out);
/* ---- frame pre-processing (the step right before the path) ---- */
/* cv.getOptimalNewCameraMatrix(K, dist, (w,h), alpha, (new_w,new_h)) -- host arithmetic only
 * (detect_pose.py:167-173).  newK: 9 doubles out, roi: {x, y, w, h} out (may be NULL). */
int agt_get_optimal_new_camera_matrix(const double* K, const double* dist, int ndist, int w, int h, double alpha,
                                      int new_w, int new_h, double* newK, int* roi);
/* cv.initUndistortRectifyMap(K, dist, I, newK, (w,h), CV_16SC2) into context-owned device maps
 * (6 B/pixel, built once per camera; newK NULL = K). */
int agt_undistort_init(agt_ctx* ctx, const double* K, const double* dist, int ndist, const double* newK, int w, int h);
int agt_undistort_maps(const agt_ctx* ctx, const int16_t** d_map1, const uint16_t** d_map2, int* w, int* h);
/* cv.undistort(frame, K, dist, None, newK) on B BGR u8 frames of the map size (detect_pose.py:176-177). */
int agt_undistort_bgr(agt_ctx* ctx, const uint8_t* d_src, size_t spitch, size_t sbatch,
                      uint8_t* d_dst, size_t dpitch, size_t dbatch, int B);
/* Fused undistort (optional) -> cv.cvtColor(BGR2GRAY) -> ROI crop (detect_pose.py:176-181, :602):
 * B BGR u8 frames (src_w x src_h) in, gray u8 (roi_w x roi_h) out; no intermediate image in HBM. */
int agt_preprocess_bgr(agt_ctx* ctx, const uint8_t* d_bgr, size_t spitch, size_t sbatch, int src_w, int src_h, int B,
                       int undistort, int roi_x, int roi_y, int roi_w, int roi_h,
                       uint8_t* d_gray, size_t gpitch, size_t gbatch);

/* ---- dense photometric + geometric pose refinement (BASELINE configs[4]; no reference code exists:
 * the semantics are specified in oracle/cv_dense.c and implemented by csrc/agt_dense.hip) ----
 * E(p) = sum |project(obj_j; p) - img_j|^2 + photo_weight * sum (I~(project(X_i; p)) - T_i)^2, damped
 * Gauss-Newton (mu = 1e-3), `iters` iterations at most, early stop at relative step < FLT_EPSILON.
 * d_img: B gray u8 frames (w x h).  d_model_xyz [M][3] f32 / d_model_t [M] f32: model samples and
 * template intensities (shared by the B streams).  d_obj [N][3] f32, d_img_pts [B][N][2] f32,
 * d_mask [B][N] u8 or NULL: corner term (N may be 0).  d_pose [B][6] f64 in/out.
 * d_stats [B][8] f64: photometric RMS, geometric RMS, valid samples, iterations run, corners used. */
int agt_dense_refine(agt_ctx* ctx, const uint8_t* d_img, size_t pitch, size_t batch_stride, int w, int h,
                     const float* d_model_xyz, const float* d_model_t, int M,
                     const float* d_obj, const float* d_img_pts, const uint8_t* d_mask, int N,
                     const double* K, const double* dist, int ndist,
                     double* d_pose, int B, int iters, double photo_weight, double* d_stats);

/* ---- dense refinement as a stage of the per-frame step (BASELINE configs[4]: 60 tags / 240 corners + dense photometric
 * residuals) ----
 * agt_tracker_dense registers the dense model with the tracker (the two arrays are RETAINED by pointer until the stage is
 * disabled with M = 0 or the context is destroyed).  agt_track_frame_dense then runs, per frame and without any host round
 * trip: pyramid -> LK -> solvePnP(guess) + gate + motion model (exactly agt_track_frame, stage kernels in stream order) ->
 * `iters` damped Gauss-Newton iterations of agt_dense_refine started from the frame's ACCEPTED PnP pose, over the frame, the
 * model samples and the LK corners (mask = LK status) -> with reseed != 0, the corner set of the frame is replaced by
 * projectPoints(object points; refined pose) and every corner made trackable again, so LK of the next frame starts from the
 * refined geometry (raw LK chaining drifts).  The tracker's state machine (guess, prev_transform, velocities) is the
 * reference's and is NOT altered by the refinement.  d_dense_out: [B][AGT_DENSE_STRIDE] f64 per frame. */
#define AGT_DENSE_STRIDE 16
#define AGT_DN_RVEC      0    /* 0..2 refined rvec (the PnP pose when AGT_DN_REFINED is 0) */
#define AGT_DN_TVEC      3    /* 3..5 */
#define AGT_DN_REFINED   6    /* 1.0 = the PnP pose was accepted and at least one GN step was applied */
#define AGT_DN_PHOTO_RMS 7    /* photometric RMS at the last linearisation point */
#define AGT_DN_GEO_RMS   8    /* geometric RMS there (px) */
#define AGT_DN_VALID     9    /* valid model samples there */
#define AGT_DN_ITERS     10   /* GN iterations executed */
#define AGT_DN_CORNERS   11   /* corners used */
int agt_tracker_dense(agt_ctx* ctx, const float* d_model_xyz, const float* d_model_t, int M, int iters, double photo_weight, int reseed);
int agt_track_frame_dense(agt_ctx* ctx, const uint8_t* d_frames, size_t pitch, size_t batch_stride, int B,
                          double* d_state_out, double* d_dense_out);
/* A clip of `count` consecutive frames (frame k at d_frames + k * frame_stride; all with the same pitch and stream stride):
 * agt_track_frame_dense for each, in order, d_state_out [count][B][AGT_STATE_STRIDE] (or NULL), d_dense_out
 * [count][B][AGT_DENSE_STRIDE].  Same records as `count` single calls, bit for bit; knowing the next frame, the library
 * shortens every frame's serial chain: the next frame's pyramid pass rides in one of the current frame's launches, the dense
 * stage's last step (final Gauss-Newton update + corner re-seed) of frame k runs at the head of frame k + 1's LK launch, and with
 * more than 64 corners on one stream LK and PnP of a frame are one chained launch.  All of the clip's work is enqueued when the
 * call returns, in stream order: frame k's dense record is complete when the work enqueued by this call is (the record of every
 * frame but the last is finished by the launch that tracks the NEXT frame). */
int agt_track_frames_dense(agt_ctx* ctx, const uint8_t* d_frames, size_t pitch, size_t batch_stride, size_t frame_stride, int B, int count,
                           double* d_state_out, double* d_dense_out);

/* ---- per-kernel timing of agt_track_frame with HIP events on the context's stream ---- */
/* After agt_profile_begin every agt_track_frame records AGT_PROF_EVENTS events around its
 * four launches (pyrDown L0->L1, pyrDown L1->L2.., LK, PnP) into the next of max_frames
 * slots.  agt_profile_end synchronises, writes ms[frame][AGT_PROF_SPANS] and the number of
 * frames recorded, and releases the events.  With the fused software-pipelined step the frame is
 * ONE launch: spans 0 and 1 are ~0 and span 2 is the step_kernel launch.  Recording perturbs
 * timing: never leave it on in a throughput measurement. */
#define AGT_PROF_DENSE_MAX 16      /* (kept for the event-slot layout: the dense stage uses two of its 2 x 16 slots) */
#define AGT_PROF_EVENTS (4 + 2 * AGT_PROF_DENSE_MAX)
#define AGT_PROF_SPANS  5          /* 0: pyramid (all pyrDown launches), 1: LK, 2: PnP+state machine,
                                      3: the dense Gauss-Newton launches of the frame, one span over all of them (launch i: update of
                                         iteration i - 1, then the accumulate of iteration i), 4: its final launch (last update + corner
                                         re-seed); 0 without the stage */
int agt_profile_begin(agt_ctx* ctx, int max_frames);
int agt_profile_end(agt_ctx* ctx, float* ms_out, int* n_frames);

/* device pointers to the live tracker buffers (corners [B][n][2] f32, status [B][n] u8) */
int agt_tracker_buffers(const agt_ctx* ctx, const float** d_corners, const uint8_t** d_status);

#ifdef __cplusplus
}
#endif
#endif
