// agt_kernels.h -- host-visible launch interface of the gfx950 kernels (internal to the library).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/agt_hip.h"

struct AgtLevel {
    const uint8_t* ptr;   // level base for stream 0
    long pitch;           // bytes per row (multiple of 4)
    long bstride;         // bytes between streams
    int w, h;
};

// ---- the chip a context runs on (round 5: no literal 8 XCDs / 256 CUs in kernels or launch rules).
// Workgroups of a launch are dealt round-robin to the XCDs, each with its own L2; the kernels that re-use lines between
// neighbouring work items (pyramid tiles / strips, LK corners of a stream, undistortion bands) re-order their block index so
// that every XCD walks a CONTIGUOUS run of items:  item = (b mod X) * (n / X) + b / X  for a grid of n blocks, n a multiple of
// X = 2^xshift.  A pure index map (a permutation of 0 .. n - 1 for every X): correct on any device, tuned when X is the device's
// XCD count.  gfx950: one XCD = 32 CUs (MI355X: 256 CUs = 8 XCDs; its CPX / QPX / DPX partitions: 32 / 64 / 128 CUs = 1 / 2 / 4).
struct AgtChip {
    int cus;                  // hipDeviceProp_t::multiProcessorCount
    int xcds;                 // 1, 2, 4 or 8 (cus / 32 when that is a power of two, else 1: plain order)
    int xshift;               // log2(xcds)
    char arch[32];            // gcnArchName up to the first ':'
};
__host__ __device__ inline int agt_xcd_order(int b, int nblk, int xshift)
{
    return (b & ((1 << xshift) - 1)) * (nblk >> xshift) + (b >> xshift);
}
inline unsigned agt_xcd_grid(long items, int xshift) { const long x = 1L << xshift; return (unsigned)((items + x - 1) / x * x); }
const AgtChip* agt_chip_of(int device);        // cached device query (agt_api.hip); null when the query fails
const AgtChip& agt_chip_current(void);         // of the calling thread's current device; MI355X's figures if the query fails

struct AgtPyrArgs {
    const uint8_t* src; uint8_t* dst;
    long spitch, sbatch, dpitch, dbatch;
    int sw, sh, dw, dh;
    int gx, gy, B;            // tile grid (x, y) and images
    int pad;
    int xshift;               // log2 of the XCD count the block order is laid out for (agt_xcd_order); set by the launchers
    int rsv_;                 // two-level rolling pass: 1 = every strip top-down (diagnostic A/B; 0 = alternating directions)
};

#define AGT_MAX_GROUP 32         // frames one fused launch may advance each pipeline stage by (the per-frame tables are kernel arguments: 7.4 KB with the parameters)

#define AGT_LK_FLAG_COTENANT 0x20000     // internal launch flag: the context declared co-tenancy (agt_lk_occupancy_cu): tracker waves at issue priority 1
struct AgtLkParams {
    AgtLevel prev[AGT_MAX_LEVELS];
    AgtLevel next[AGT_MAX_LEVELS];
    int max_level;            // effective (after OpenCV's early stop)
    int n;                    // points per stream
    int max_count;            // criteria, already clamped
    double eps2;              // epsilon^2
    int flags;                // AGT_LK_* of the ABI in the low bits; internal: 0x10000 = general body for every corner (diagnostic), AGT_LK_FLAG_COTENANT
    double min_eig_threshold;
    const float* prev_pts;    // [B][n][2]
    const uint8_t* prev_status;   // [B][n] or null: tracker mode, a corner lost in an earlier frame stays lost (position carried)
    float* next_pts;          // [B][n][2]
    uint8_t* status;          // [B][n]
    float* err;               // [B][n] or null
    int xshift;               // XCD-aware corner order (agt_xcd_order); set by the launchers
    int lds_min;              // host side only: least dynamic LDS a one-wave workgroup asks for = a residency cap (agt_lk_occupancy_cu); 0 = none
    // HYBRID launch of big batches (round 5, agt_lk.hip lk_hybrid_kernel): a corner that took >= slow_thr iterations (all levels) in the
    // PREVIOUS frame is tracked by FOUR waves (0.5 us per iteration, 5.8 us fixed), the others by one wave each (0.8 / 6.8 us) -- a
    // per-frame launch lasts as long as its slowest corner, and the corners that iterate long do so in every frame.  The two bodies
    // compute the same bits, so the choice never shows in the results.
    uint8_t* iters_out;       // [B][n] iterations of this frame per corner (saturated at 255), or null
    const uint8_t* iters_prev;    // [B][n] of the previous frame, or null (= every corner on one wave)
    int slow_thr;             // 0 = no hybrid launch
    int rsv2_;
};

struct AgtCameraHost {
    double fx, fy, cx, cy;
    double k[12];
    // tilted sensor (coefficients 13, 14 = tau_x, tau_y; round 5): DEVICE pointer to matTilt | invMatTilt (2 x 9 doubles, row-major, as
    // detail::computeTiltProjectionMatrix builds them), null without tilt.  In global memory, not in this by-value struct: loads from the
    // kernel-argument segment are speculated in front of the `if (tilt)` branches and the 36 scalar registers they fill cost the fused
    // step kernel 86 more spilled SGPRs (measured); a load through a pointer that may be null stays inside its branch.
    const double* tilt;
};
// the same matrices on the host (agt_api.hip fill_camera): m = matTilt | invMatTilt, on = 0: identities
struct AgtTiltHost { double m[18]; int on; };

struct AgtPnpParams {
    const void* obj;          // n x 3, f32 or f64
    long obj_bstride;         // elements between streams (0 = shared)
    const void* img;          // [B][n][2]
    const uint8_t* mask;      // [B][n] or null
    int dtype;                // AGT_F32 / AGT_F64 (img and obj)
    int img_f32_obj_f32;      // unused, keeps layout explicit
    int n;
    int use_guess;
    AgtCameraHost cam;
    double* pose;             // [B][6] in/out
    int32_t* info;            // [B][4] or null
    double* err;              // [B] or null
    // tracker mode (null = plain solvePnP): PoseDetector._estimate_pose on device state
    struct AgtTrackState* track;   // [B]
    double* state_out;             // [B][AGT_STATE_STRIDE] or null
    float* corners_rw;             // [B][n][2] corner set to refresh by reprojection, or null
    uint8_t* status_rw;            // [B][n] LK status revived together with corners_rw, or null
    // dense stage hand-over (agt_track_frame_dense): start pose, done word (1 = pose not accepted: skip) and the frame's record
    double* dense_pose;            // [B][6] or null
    int* dense_done;               // [B]
    double* dense_rec;             // [B][AGT_DENSE_STRIDE]
    int enhance_ape;
    int reproject;
    int min_points;                // corners needed to attempt a pose (8 = two tags)
    double gate_px;                // reprojection gate (2.0, detect_pose.py:539)
    int tag_gate;                  // 4: a corner counts only while all four corners of its tag are usable (the reference solves on
                                   // whole tags, detect_pose.py:400-437, :494-496); 0: every usable corner counts
    int pad_;
    int* fault;                    // chained launch: host-mapped word set to 1 when a wait gave up (agt_synchronize reports it), or null
    float* seed_pts;               // agt_track_frame_detected: the detector's corner table (img, mask) becomes the frame's corner set
    uint8_t* seed_status;          // / LK status ([B][n][2], [B][n]: the tracker's ring entry of the frame), or null
    // agt_track_host_frame: host-mapped sequence word the solver of stream 0 stores (system scope) behind the frame's record -- whose
    // destination is host-mapped too -- so that the calling thread can poll for the record instead of waiting for the stream;
    // frame k of a launch stores host_seq_base + k.  null: off.
    unsigned long long* host_seq;
    unsigned long long host_seq_base;
};

// the frame's record (written by this wave a moment ago, to host-mapped memory) is complete: tell the polling host thread
__device__ __forceinline__ void agt_host_seq_store(unsigned long long* host_seq, unsigned long long value, bool writer_lane)
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");                  // system scope: the record's stores are performed first
    if (writer_lane) __hip_atomic_store(host_seq, value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// device-resident per-stream tracker state: the attributes of PoseDetector
// (detect_pose.py:74-83) that _estimate_pose mutates
struct AgtTrackState {
    double guess[6];        // extrinsic_guess (rvec, tvec)
    double prev[6];         // prev_transform
    double rot_vel[2][9];   // rot_velocities (oldest first)
    double tran_vel[2][3];  // tran_velocities
    double prev_R[9];       // Rodrigues(prev rvec), kept from the solve that produced prev (device cache, not reference state)
    int has_guess;          // extrinsic_guess[0] is not None
    int has_prev;           // prev_transform[0] is not None
    int n_vel;              // len(rot_velocities)
    int frame;
    int guess_t_f32;        // dtype of extrinsic_guess[1] is float32 (from get_rmat_tvec, transform_helper.py:158-159)
    int prev_t_f32;         // dtype of prev_transform[1] is float32
    int chain_fault;        // sticky: a chained launch gave up waiting for this stream's corners (AGT_TRK_CHAIN_TIMEOUT); the stream's
                            // state is frozen and every later record is flagged invalid until agt_tracker_reset
    int pad;
};

// one fused launch (agt_step.hip): block ranges [LK | PnP | pyr stage 0 | stage 1 | ..]; every role advances
// by up to AGT_MAX_GROUP consecutive frames (the LK and PnP roles loop over them inside the launch: their
// chains are serial across frames; the pyramid stages treat the frames as a batch)
struct AgtStepParams {
    AgtPyrArgs pyr[AGT_MAX_LEVELS - 1];                        // geometry; src / dst per frame in AgtStepTables
    int pyr_fused;                    // != 0: stage 0 builds levels 1 AND 2 in one pass (pyr[0].gx / gy = its tile grid, pyr[1] =
    int pyr_pad;                      // the level 1 -> 2 geometry, pyr_dst[1] = the level-2 buffers); stage 1 then has no blocks
    int pyr_nf[AGT_MAX_LEVELS - 1];
    int n_pyr[AGT_MAX_LEVELS - 1];    // blocks of each pyramid stage = tiles x streams x frames (0 = stage idle)
    AgtLkParams lk;                   // geometry of prev[] / next[], criteria; images per frame in AgtStepTables
    int lk_nf;
    int n_lk;                         // != 0: LK role active (block count is derived at launch)
    int lk_B;                         // streams of the LK role
    AgtPnpParams pnp;
    int pnp_nf;
    int n_pnp;                        // blocks of the PnP role (= streams) or 0
    int xshift;                       // XCD-aware block orders of the LK and pyramid roles (agt_xcd_order); set by the launchers
    int rsv_;
};

// per-frame pointer tables of the fused launch: its SECOND kernel argument.  They are indexed with run-time
// frame numbers and therefore read from the kernel-argument segment directly (never through a by-value copy).
struct AgtLkTables {
    const uint8_t* img[AGT_MAX_GROUP + 1][AGT_MAX_LEVELS];     // image k of the group per level (k = 0: the frame before it)
    float* next[AGT_MAX_GROUP];                                // frame k+1's corners / status
    uint8_t* status[AGT_MAX_GROUP];
    unsigned* done[AGT_MAX_GROUP];                             // chained launch: arrival counters of frame k+1 ([B], one count per corner) or null
};
struct AgtPnpTables {
    const float* img[AGT_MAX_GROUP];                           // per frame: corners, LK status, caller's state record
    const uint8_t* mask[AGT_MAX_GROUP];
    double* so[AGT_MAX_GROUP];
    const unsigned* wait[AGT_MAX_GROUP];                       // chained launch: the frame's LK arrival counters ([B]) or null = complete before the launch
    unsigned long long target[AGT_MAX_GROUP];                  // counter value that means "all corners of the frame are written"
};
struct AgtStepTables {
    const uint8_t* pyr_src[AGT_MAX_LEVELS - 1][AGT_MAX_GROUP];
    uint8_t* pyr_dst[AGT_MAX_LEVELS - 1][AGT_MAX_GROUP];
    AgtLkTables lk;
    AgtPnpTables pnp;
};

struct AgtProjParams {
    const void* obj; long obj_bstride; int dtype; int n;
    const double* pose;       // [B][6]
    AgtCameraHost cam;
    void* img_out;            // [B][n][2]
    double* jac;              // [B][2n][6] or null
    // agt_project_points_host (one block, B = 1): host-mapped sequence word stored behind the outputs (see AgtPnpParams::host_seq); null: off
    unsigned long long* host_seq;
    unsigned long long host_seq_base;
};

void agt_pyr_grid(int dw, int dh, int* gx, int* gy);
void agt_pyr_plan(AgtPyrArgs* A, uintptr_t src_align, uintptr_t dst_align, int frames);
hipError_t agt_launch_pyr_upload2(hipStream_t stream, const uint8_t* src, int sw, int sh, long spitch, uint8_t* copy, long cpitch,
                                  uint8_t* dst1, long dpitch1, uint8_t* dst2, long dpitch2);     // fused upload + two-level pyramid of one frame (agt_pyramid.hip)
void agt_pyr2_plan(AgtPyrArgs* A0, AgtPyrArgs* A1, uintptr_t src_align, uintptr_t dst_align, int frames, int oh_cap = 16);     // the same for the two-level pass      // tiled or register-rolling form of one pyrDown pass (agt_pyramid.hip)
void agt_pyr2_grid(int w2, int h2, int* gx, int* gy);           // tile grid of the two-level pass (64 x 16 tiles of L2)
int agt_pyr2_lds_bytes(void);
hipError_t agt_launch_pyr_down2(hipStream_t stream, const uint8_t* src, int sw, int sh, long spitch, long sbatch,
                                uint8_t* dst1, long dpitch1, long dbatch1, uint8_t* dst2, long dpitch2, long dbatch2, int B);
hipError_t agt_launch_pyr_down(hipStream_t stream, const uint8_t* src, int sw, int sh, long spitch, long sbatch,
                               uint8_t* dst, long dpitch, long dbatch, int B);
// waves: 0 = by batch size (agt_lk_wide), 1 / 4 = that many waves per corner (win 21 only)
hipError_t agt_launch_lk(hipStream_t stream, const AgtLkParams& p, int win, int B, int waves = 0);
hipError_t agt_launch_lk_hybrid(hipStream_t stream, const AgtLkParams& p, int B);      // win 21, <= 3 levels, p.iters_prev / slow_thr set (agt_lk.hip)
// ride != null (two argument blocks of agt_pyr2_args): the two-level pyramid pass of another frame as extra workgroups of the
// launch; only where agt_pnp_can_ride(n) (the four-wave kernel of n > 64)
hipError_t agt_launch_pnp(hipStream_t stream, const AgtPnpParams& p, int B, const AgtPyrArgs* ride = nullptr);
bool agt_pnp_can_ride(int n);
hipError_t agt_launch_project(hipStream_t stream, const AgtProjParams& p, int B);
hipError_t agt_launch_undistort_map(hipStream_t stream, const double* K, const AgtCameraHost& cam, const AgtTiltHost& tilt, const double* ir,
                                    int w, int h, short2* map1, unsigned short* map2);
hipError_t agt_launch_preprocess(hipStream_t stream, const uint8_t* src, long spitch, long sbatch, int sw, int sh,
                                 const short2* map1, const unsigned short* map2, int mw,
                                 int rx, int ry, int rw, int rh, uint8_t* dst, long dpitch, long dbatch,
                                 int undistort, int gray, int B);
// The dense stage's last step (final Gauss-Newton update + corner re-seed) handed on to the LK launch of the NEXT frame instead of
// being launched: the stage's parameter block as agt_dense.hip fills it (opaque here; agt_step.hip agt_launch_lk_reseed reads it)
struct AgtDenseFinal { alignas(8) unsigned char bytes[776]; };
hipError_t agt_launch_lk_reseed(hipStream_t stream, const struct AgtStepParams& S, const struct AgtStepTables& T, int win, const AgtDenseFinal* F, const struct AgtPyrArgs* ride = nullptr);
hipError_t agt_launch_dense_final(hipStream_t stream, const AgtDenseFinal& F, int B);      // the deferred step as its own launch after all
hipError_t agt_launch_dense(hipStream_t stream, const uint8_t* img, long pitch, long ibatch, int w, int h,
                            const float* mxyz, const float* mt, int M,
                            const float* obj, const float* ipts, const uint8_t* mask, int N,
                            const AgtCameraHost& cam, double* pose, double* partials, double* stats, int* done,
                            int B, int iters, double photo_weight, double mu, double* rec, float* seed_pts, uint8_t* seed_status,
                            hipEvent_t* ev = nullptr, int n_ev = 0, const AgtPyrArgs* next_pyr = nullptr, AgtDenseFinal* defer_final = nullptr);
// geometry of the two-level pyramid pass (agt_pyramid.hip) as the pair of argument blocks its body takes
void agt_pyr2_args(const uint8_t* src, int sw, int sh, long spitch, long sbatch, uint8_t* dst1, long dpitch1, long dbatch1,
                   uint8_t* dst2, long dpitch2, long dbatch2, int B, AgtPyrArgs* A0, AgtPyrArgs* A1);
int agt_dense_blocks(int M);
size_t agt_dense_doubles(int M, int B);
bool agt_lk_window_supported(int win);
void agt_lk_window_size(int win, int* ww, int* wh);       // AgtConfig::win -> (width, height): a side, or AGT_WIN_RECT(w, h)
bool agt_lk_wide(int n, int B);
bool agt_step_supported(int win);
bool agt_step_fits(int n, int B);   // the fused launch (all roles in one kernel) is used while its LK workgroups fit one per CU (256 corners on a whole MI355X)
// role subsets of one pipeline group (split mode launches them separately, each with its own LDS size and register budget)
#define AGT_STEP_PYR 1
#define AGT_STEP_LK  2
#define AGT_STEP_PNP 4
#define AGT_STEP_ALL 7
hipError_t agt_launch_step(hipStream_t stream, const AgtStepParams& S, const AgtStepTables& T, int win, int roles);
