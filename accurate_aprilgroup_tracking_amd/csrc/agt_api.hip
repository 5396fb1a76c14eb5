// agt_api.hip -- C ABI (include/agt_hip.h) over the gfx950 kernels: context, pyramid slots,
// argument checking, launches.  No host<->device copies and no synchronisation on the
// per-frame path; everything is enqueued on the context's stream.
#include "agt_kernels.h"
#include <new>
#include <string.h>
#include <math.h>
#include <float.h>
#include <time.h>
#include <stdio.h>
#include <stdlib.h>
#include <atomic>
#include <mutex>
#include <vector>

#define AGT_SLOTS 4              // ring entries every context owns (slots 0 / 1 are also the public pyramid slots)
#define AGT_RING_MAX 224         // (levels + 1) * AGT_MAX_GROUP frames in flight at the deepest pipeline
#define AGT_EV_SLOTS 8           // events of the split pipeline, per kind: a launch waits for events at most three launches old (l_ev_hist), slots are re-recorded modulo 8
// The two-level pyramid pass saves a launch / pipeline stage and 16 % of the pyramid's HBM bytes, but its 41 KB workgroups
// (3 per CU, eight barriers per tile) stream at 2.5 TB/s against 3.9 + 3.4 TB/s for two single-level passes (8 per CU): it is
// used where the stage count matters (few streams), the two passes where throughput does (measured at 64 x 720p: 30.5 vs 25 us).
#define AGT_PYR2_MAX_B 8
#define AGT_TILT_SLOTS 8          // device table of tilted-sensor matrices: slot 0 the tracker's camera, 1.. the stateless calls'
#define AGT_SPLIT_SLACK 2        // split mode: groups of extra ring entries (pyramid launches run that far ahead of LK)

struct agt_ctx {
    agt_config cfg;
    AgtChip chip;                            // the device the context was created on (CU / XCD counts: launch rules and block orders)
    int lk_cap_cu;                           // agt_lk_occupancy_cu: resident one-wave LK workgroups per CU (0 = no cap, -1 = the library's choice)
    hipStream_t stream;
    int last_hip;
    int eff_max_level;                       // after OpenCV's early stop
    int lw[AGT_MAX_LEVELS], lh[AGT_MAX_LEVELS];
    long lpitch[AGT_MAX_LEVELS];             // levels >= 1 (context-owned)
    char* hcall_host; char* hcall_dev; unsigned long long hcall_n;      // host-mapped staging of the synchronous host-array calls (agt_solve_pnp_host)
    double* d_tilt; double tilt_host[AGT_TILT_SLOTS][18]; int tilt_valid[AGT_TILT_SLOTS]; int tilt_next;   // tilted-sensor matrices (camera_on)
    uint8_t* lmem[AGT_RING_MAX][AGT_MAX_LEVELS];
    const uint8_t* l0_ptr[AGT_RING_MAX];
    long l0_pitch[AGT_RING_MAX], l0_bstride[AGT_RING_MAX];
    int built_B[AGT_RING_MAX];
    // tracker: rings (frame t lives in entry t % ring) so that one fused launch can work on pyramid
    // stage s of frames t-sF.., LK of frames t-LF.. and PnP of frames t-(L+1)F.. at once (F = group)
    int ring;                                // allocated ring entries: >= (L + 2) * group
    float* corners[AGT_RING_MAX];            // [B][n][2]
    uint8_t* status[AGT_RING_MAX];           // [B][n]
    uint8_t* lk_iters[AGT_RING_MAX];         // [B][n] iterations every corner took in the frame of the ring entry (hybrid LK launch: a hint, never a result)
    int lk_slow_thr;                         // agt_lk_hybrid: corners at or above it in the previous frame are tracked by four waves (0 = off)
    double* so_ring[AGT_RING_MAX];           // caller's state_out of the frames in flight
    int pipeline;                            // 1 = software-pipelined fused step (agt_step.hip)
    int group;                               // frames per fused launch (1..AGT_MAX_GROUP)
    int ramp;                                // split pipeline: frames per group while the pipeline fills (agt_step.hip launch_group: ramp_group)
    int live_ring;                           // ring modulus in use (<= ring): (L + 2) * group, at least AGT_SLOTS
    // big batches: the three stages of a step run on three library-owned streams (stage kernels of different frames
    // overlap: 57 us against 93 us back to back at 64 streams); events carry the exact dependencies
    hipStream_t ms_stream[3];                // pyramid, LK, PnP
    hipEvent_t ms_ev[5][AGT_EV_SLOTS];       // per launch (modulo AGT_EV_SLOTS): pyramid done, LK done, [2]: join / hand-over events, PnP done, LK done (second half of the streams)
    int ms_pool_slot;                        // which set of the process's library streams the context holds (-1: none)
    int ms_ready, ms_active;                 // streams / events exist; frames are in flight on them
    // split mode (more corners in flight than the fused launch takes): the pipeline's groups go out as three launches,
    // pyramid on the caller's stream, LK and PnP on library streams (ms_stream[1], [2])
    long split_seq;                          // groups issued in split mode
    int last_p_ev;                           // event slot of the most recent pyramid launch (-1 = none)
    int l_ev_hist[3];                        // event slots of the three most recent LK launches (-1 = none)
    int y_ev_hist[2];                        // event slots of the two most recent PnP launches (-1 = none)
    long trk_frame;                          // frames supplied since reset (0 = only the reset frame)
    long prebuilt_t = -1;                    // serial step, clip submission: frame whose pyramid the previous frame's dense launch built (-1 = none)
    // clip submission of the dense stage: the previous frame's last step (final update + re-seed) waits for this frame's LK launch
    // (agt_step.hip lk_reseed_kernel); only ever set between two frames of one agt_track_frames_dense call
    int dense_pending = 0;
    AgtDenseFinal dense_final;
    long n_stage[AGT_MAX_LEVELS];            // frames whose pyramid stage s (level s -> s+1) is done
    long n_lk, n_pnp;                        // frames whose LK / PnP is done (enqueued)
    // chained launches (fused step): per ring entry, [max_streams] arrival counters the LK role counts corners into and
    // the PnP role of the same launch waits on; lk_target = the value the entry's counters reach once every corner
    // of its current frame is written (counters only ever grow: no reset, no reuse hazard)
    unsigned* lk_done;
    unsigned lk_target[AGT_RING_MAX];
    float* lkerr;                            // [B][n]
    float* obj;                              // [n][3]
    double* pose;                            // [B][6]
    AgtTrackState* tstate;                   // [B]
    AgtCameraHost cam;
    int trk_n, trk_B, enhance_ape, trk_ready;
    int reproject, min_points, tag_gate;
    double gate_px;
    int* fault_host; int* fault_dev;         // host-mapped word a chained launch sets when a wait gave up (agt_synchronize reports it)
    // agt_track_host_frame: the frame's record and a sequence word in host-mapped memory (same allocation as the fault word: +64 the
    // record, +192 the word); seq(frame t) = hseq_off + t, monotonic across resets and rewinds
    double* hrec_host; double* hrec_dev; unsigned long long* hseq_host; unsigned long long* hseq_dev;
    unsigned long long hseq_off, hseq_last; int host_seq_on;
    // LK parameters of the fused step (COUNT+EPS (30, 0.01), minEig 1e-4, flags 0)
    int lk_max_count; double lk_eps; double lk_min_eig;
    // undistortion maps of the pre-processing stage (built once per camera)
    short2* map1; unsigned short* map2; int map_w, map_h;
    // scratch of the dense refinement: per-block partial sums and the per-stream done words
    double* dense_partials; int* dense_done; size_t dense_cap; int dense_done_B;   // capacities: doubles / streams
    // dense stage of the tracker (agt_tracker_dense): model retained by pointer
    const float* dn_xyz; const float* dn_t; int dn_M, dn_iters, dn_reseed; double dn_weight;
    // optional per-kernel timing (agt_profile_begin/end)
    hipEvent_t* prof_ev;
    int prof_cap, prof_n;
    int* prof_dense;                         // per recorded frame: dense iterations whose launches carry events
};

static int ms_join(agt_ctx* c);
static int join_pipeline(agt_ctx* c);
static int ms_init(agt_ctx* c);
static int ms_pool_acquire(int device, hipStream_t out[3]);
static void ms_pool_release(int slot);

namespace {

int g_last_hip = 0;      // last failing HIP call made without a context (agt_create)

int hip_fail(agt_ctx* c, hipError_t e)
{
    if (c) c->last_hip = (int)e;
    g_last_hip = (int)e;
    (void)hipGetLastError();   // clear the sticky error so later launches are not blamed
    return AGT_ERR_HIP;
}

// matTilt | invMatTilt of a coefficient vector (detail::computeTiltProjectionMatrix<double>, distortion_model.hpp; Matx products
// accumulate s = 0; s += a * b): 14 coefficients = the 12 + the tilted-sensor angles (tau_x, tau_y), which cv2.calibrateCamera returns
// non-zero only under CALIB_TILTED_MODEL -- the reference calibrates 5 coefficients, calibrate_camera.py:178
void tilt_matrices(const double* dist, int ndist, AgtTiltHost* t)
{
    static const double I3[9] = { 1, 0, 0, 0, 1, 0, 0, 0, 1 };
    memcpy(t->m, I3, sizeof(I3)); memcpy(t->m + 9, I3, sizeof(I3));
    t->on = 0;
    if (ndist != 14 || !dist || (dist[12] == 0.0 && dist[13] == 0.0)) return;
    const double cX = cos(dist[12]), sX = sin(dist[12]), cY = cos(dist[13]), sY = sin(dist[13]);
    const double rotX[9] = { 1, 0, 0, 0, cX, sX, 0, -sX, cX }, rotY[9] = { cY, 0, -sY, 0, 1, 0, sY, 0, cY };
    auto mul = [](const double* A, const double* B, double* C) {
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) { double a = 0; for (int q = 0; q < 3; q++) a += A[i * 3 + q] * B[q * 3 + j]; C[i * 3 + j] = a; }
    };
    double rotXY[9];
    mul(rotY, rotX, rotXY);
    const double projZ[9] = { rotXY[8], 0, -rotXY[2], 0, rotXY[8], -rotXY[5], 0, 0, 1 };
    mul(projZ, rotXY, t->m);
    const double inv = 1. / rotXY[8];
    const double invProjZ[9] = { inv, 0, inv * rotXY[2], 0, inv, inv * rotXY[5], 0, 0, 1 };
    const double rt[9] = { rotXY[0], rotXY[3], rotXY[6], rotXY[1], rotXY[4], rotXY[7], rotXY[2], rotXY[5], rotXY[8] };
    mul(rt, invProjZ, t->m + 9);
    t->on = 1;
}

// camera of a call: intrinsics + the 12 polynomial coefficients by value; `tilt` (optional) receives the tilted-sensor matrices, the
// struct's device pointer stays null (camera_on: the form with the matrices in the context's device table)
int fill_camera(const double* K, const double* dist, int ndist, AgtCameraHost* cam, AgtTiltHost* tilt = nullptr)
{
    if (!K) return AGT_ERR_ARG;
    if (!(ndist == 0 || ndist == 4 || ndist == 5 || ndist == 8 || ndist == 12 || ndist == 14)) return AGT_ERR_DIST;
    if (ndist > 0 && !dist) return AGT_ERR_ARG;
    cam->fx = K[0]; cam->fy = K[4]; cam->cx = K[2]; cam->cy = K[5];
    for (int i = 0; i < 12; i++) cam->k[i] = i < ndist ? dist[i] : 0.0;
    cam->tilt = nullptr;
    if (tilt) tilt_matrices(dist, ndist, tilt);
    return AGT_OK;
}

// ... for a launch of context c.  A tilted camera's matrices go to a slot of the context's device table (AGT_TILT_SLOTS x 18 doubles):
// slot 0 belongs to the tracker (written at agt_tracker_init, behind a stream synchronisation: the tracker's internal streams read
// it), slots 1.. serve the stateless calls round-robin -- a slot that already holds the same matrices is re-used without a copy, a new
// camera's copy is ordered on the context's stream behind every launch that read the slot's previous content.
int camera_on(agt_ctx* c, const double* K, const double* dist, int ndist, AgtCameraHost* cam, bool tracker = false)
{
    AgtTiltHost t;
    int rc = fill_camera(K, dist, ndist, cam, &t);
    if (rc || !t.on) return rc;
    if (!c->d_tilt) {
        if (hipMalloc((void**)&c->d_tilt, sizeof(double) * 18 * AGT_TILT_SLOTS) != hipSuccess) { (void)hipGetLastError(); return AGT_ERR_ALLOC; }
        memset(c->tilt_valid, 0, sizeof(c->tilt_valid));
    }
    int slot = 0;
    if (!tracker) {
        slot = -1;
        for (int i = 1; i < AGT_TILT_SLOTS; i++) if (c->tilt_valid[i] && !memcmp(c->tilt_host[i], t.m, sizeof(t.m))) { slot = i; break; }
        if (slot < 0) {
            c->tilt_next = c->tilt_next % (AGT_TILT_SLOTS - 1) + 1; slot = c->tilt_next;
            // an eighth distinct tilted camera recycles a slot: launches that read its old content may sit on ANY stream the context was
            // bound to (agt_set_stream) -- wait for the device once (a camera change of this kind is not a per-frame event)
            if (c->tilt_valid[slot] && hipDeviceSynchronize() != hipSuccess) return hip_fail(c, hipGetLastError());
            c->tilt_valid[slot] = 0;
        }
    }
    if (!c->tilt_valid[slot] || memcmp(c->tilt_host[slot], t.m, sizeof(t.m))) {
        memcpy(c->tilt_host[slot], t.m, sizeof(t.m));
        // (the copy is waited for, tracker slot or not: the slot may be re-used WITHOUT a copy by a later call on another stream -- agt_set_stream --
        // and by the tracker's internal streams, none of which is ordered behind this stream.  A new tilted camera is not a per-frame event.  ADVICE r5)
        hipError_t e = hipMemcpyAsync(c->d_tilt + 18 * slot, c->tilt_host[slot], sizeof(t.m), hipMemcpyHostToDevice, c->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        if (e != hipSuccess) return hip_fail(c, e);
        c->tilt_valid[slot] = 1;
    }
    cam->tilt = c->d_tilt + 18 * slot;
    return AGT_OK;
}

// grow the rings to `want` entries (pyramid levels >= 1, corners, status)
int ensure_ring(agt_ctx* c, int want)
{
    if (want > AGT_RING_MAX) return AGT_ERR_UNSUPPORTED;
    const size_t B = (size_t)c->cfg.max_streams, N = (size_t)c->cfg.max_points;
    for (int s = c->ring; s < want; s++) {
        bool ok = true;
        for (int l = 1; l <= c->eff_max_level && ok; l++)
            ok = hipMalloc((void**)&c->lmem[s][l], B * (size_t)c->lh[l] * (size_t)c->lpitch[l]) == hipSuccess;
        ok = ok && hipMalloc((void**)&c->corners[s], B * N * 2 * sizeof(float)) == hipSuccess;
        ok = ok && hipMalloc((void**)&c->status[s], B * N) == hipSuccess;
        ok = ok && hipMemsetAsync(c->status[s], 1, B * N, c->stream) == hipSuccess;
#ifdef AGT_DEBUG_KNOBS      // (the hybrid LK launch that reads / writes it exists only in the knobs build: ADVICE r5)
        ok = ok && hipMalloc((void**)&c->lk_iters[s], B * N) == hipSuccess;
        ok = ok && hipMemsetAsync(c->lk_iters[s], 0, B * N, c->stream) == hipSuccess;
#endif
        if (!ok) { hip_fail(c, hipGetLastError()); return AGT_ERR_ALLOC; }      // partial entry is freed by agt_destroy
        c->ring = s + 1;
    }
    return AGT_OK;
}

// scratch of the dense refinement: `need` doubles of block partials and one done word per stream; the two
// capacities are tracked separately (a later call may bring more streams with fewer samples)
int dense_scratch(agt_ctx* c, size_t need, int B)
{
    if (need <= c->dense_cap && B <= c->dense_done_B) return AGT_OK;
    hipError_t e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess) return hip_fail(c, e);
    if (need > c->dense_cap) {
        if (c->dense_partials) (void)hipFree(c->dense_partials);
        c->dense_partials = nullptr; c->dense_cap = 0;
        if (hipMalloc((void**)&c->dense_partials, need * sizeof(double)) != hipSuccess) { hip_fail(c, hipGetLastError()); return AGT_ERR_ALLOC; }
        c->dense_cap = need;
    }
    if (B > c->dense_done_B) {
        if (c->dense_done) (void)hipFree(c->dense_done);
        c->dense_done = nullptr; c->dense_done_B = 0;
        if (hipMalloc((void**)&c->dense_done, (size_t)B * sizeof(int) + 64) != hipSuccess) { hip_fail(c, hipGetLastError()); return AGT_ERR_ALLOC; }
        c->dense_done_B = B;
    }
    return AGT_OK;
}

}  // namespace

// ---- the device's geometry (agt_kernels.h AgtChip), queried once per device.  Contexts may be created from several threads (one context per
// thread is the library's threading model): the table is filled under a mutex, read through an acquire load of the entry's state, and
// sized by the runtime's device count (16 fixed entries turned a valid device 16+ into hipErrorInvalidDevice).
#include <atomic>
#include <mutex>
#include <vector>
namespace {
constexpr int CHIP_MAX = 256;
AgtChip g_chip[CHIP_MAX];
std::atomic<int> g_chip_state[CHIP_MAX];       // 0 = not queried, 1 = valid, -1 = query failed
std::mutex g_chip_mutex;
}
static const AgtChip g_chip_default = { 256, 8, 3, "gfx950" };       // a whole MI355X

void agt_chip_from_props(int cus, const char* gcn_arch, AgtChip* out)
{
    memset(out, 0, sizeof(*out));
    out->cus = cus;
    size_t i = 0;
    for (; gcn_arch && gcn_arch[i] && gcn_arch[i] != ':' && i + 1 < sizeof(out->arch); i++) out->arch[i] = gcn_arch[i];
    out->arch[i] = 0;
    // gfx950: 32 CUs per XCD; workgroups are dealt round-robin to the XCDs of the partition.  A CU count that is not 32 x a power of
    // two (a CU mask, an unknown part) gets the plain block order: still correct, just not L2-aware.
    int x = cus / 32;
    out->xcds = (cus % 32 == 0 && (x == 1 || x == 2 || x == 4 || x == 8)) ? x : 1;
    out->xshift = out->xcds == 8 ? 3 : out->xcds == 4 ? 2 : out->xcds == 2 ? 1 : 0;
}

const AgtChip* agt_chip_of(int device)
{
    if (device < 0 || device >= CHIP_MAX) return nullptr;
    int st = g_chip_state[device].load(std::memory_order_acquire);
    if (st == 0) {
        std::lock_guard<std::mutex> lock(g_chip_mutex);
        st = g_chip_state[device].load(std::memory_order_relaxed);
        if (st == 0) {
            int ndev = 0;
            hipDeviceProp_t p;
            if (hipGetDeviceCount(&ndev) == hipSuccess && device < ndev && hipGetDeviceProperties(&p, device) == hipSuccess) {
                agt_chip_from_props(p.multiProcessorCount, p.gcnArchName, &g_chip[device]); st = 1;
            } else { (void)hipGetLastError(); st = -1; }
            g_chip_state[device].store(st, std::memory_order_release);
        }
    }
    return st == 1 ? &g_chip[device] : nullptr;
}

// (every launcher asks this once or twice per launch: while the process has only ever created contexts on ONE device -- the usual
// case -- the answer is that device's entry, without a runtime call.  g_chip_only: null = no context yet, a table entry = the one device so
// far, &g_chip_default used as the "several devices" mark; set with a compare-exchange in agt_create, read with one acquire load)
static std::atomic<const AgtChip*> g_chip_only{nullptr};

const AgtChip& agt_chip_current(void)
{
    const AgtChip* only = g_chip_only.load(std::memory_order_acquire);
    if (only && only != &g_chip_default) return *only;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); return g_chip_default; }
    const AgtChip* c = agt_chip_of(dev);
    return c ? *c : g_chip_default;
}

extern "C" {

int agt_version(void) { return AGT_VERSION; }

const char* agt_error_string(int code)
{
    switch (code) {
    case AGT_OK: return "ok";
    case AGT_ERR_ARG: return "invalid argument";
    case AGT_ERR_ALLOC: return "allocation failed";
    case AGT_ERR_DIST: return "unsupported distortion coefficient count (0, 4, 5, 8, 12 or 14)";
    case AGT_ERR_NPOINTS: return "bad point count";
    case AGT_ERR_HIP: return "HIP runtime error";
    case AGT_ERR_UNSUPPORTED: return "unsupported configuration";
    case AGT_ERR_STATE: return "bad context state";
    case AGT_ERR_CHAIN: return "a chained launch gave up waiting for corners (stream state frozen until agt_tracker_reset)";
    default: return "unknown error";
    }
}

int agt_xcd_tile_order(int block, int nblocks, int xcds)
{
    const int xs = xcds == 8 ? 3 : xcds == 4 ? 2 : xcds == 2 ? 1 : xcds == 1 ? 0 : -1;
    if (xs < 0 || nblocks <= 0 || (nblocks & (xcds - 1)) || block < 0 || block >= nblocks) return AGT_ERR_ARG;
    return agt_xcd_order(block, nblocks, xs);
}

int agt_device_info(const agt_ctx* c, int* cus, int* xcds, char* arch, size_t arch_cap)
{
    if (!c) return AGT_ERR_ARG;
    if (cus) *cus = c->chip.cus;
    if (xcds) *xcds = c->chip.xcds;
    if (arch && arch_cap) { strncpy(arch, c->chip.arch, arch_cap - 1); arch[arch_cap - 1] = 0; }
    return AGT_OK;
}

int agt_create(const agt_config* cfg, void* hip_stream, agt_ctx** out)
{
    if (!cfg || !out) return AGT_ERR_ARG;
    *out = nullptr;
    if (cfg->width <= 0 || cfg->height <= 0 || cfg->max_level < 0 || cfg->max_level >= AGT_MAX_LEVELS) return AGT_ERR_ARG;
    if (cfg->max_points <= 0 || cfg->max_points > 256) return AGT_ERR_NPOINTS;
    if (cfg->max_streams <= 0) return AGT_ERR_ARG;
    if (!agt_lk_window_supported(cfg->win)) return AGT_ERR_UNSUPPORTED;
    for (int i = 0; i < 8; i++) if (cfg->reserved[i] != 0) return AGT_ERR_ARG;
    hipError_t e = hipSetDevice(cfg->device);
    if (e != hipSuccess) return hip_fail(nullptr, e);
    // the library holds gfx950 code objects only: refuse any other device here, with a code, instead of failing at the first launch
    const AgtChip* chip = agt_chip_of(cfg->device);
    if (!chip) return hip_fail(nullptr, hipErrorInvalidDevice);
    if (strcmp(chip->arch, "gfx950") != 0) return AGT_ERR_UNSUPPORTED;
    agt_ctx* c = new (std::nothrow) agt_ctx;
    if (!c) return AGT_ERR_ALLOC;
    memset(c, 0, sizeof(*c));
    c->ms_pool_slot = -1;
    c->cfg = *cfg;
    c->chip = *chip;
    {   // first device of the process: remember it; a second one: from now on the launchers ask the runtime which device is current
        const AgtChip* seen = nullptr;
        if (!g_chip_only.compare_exchange_strong(seen, chip, std::memory_order_acq_rel) && seen != chip) g_chip_only.store(&g_chip_default, std::memory_order_release);
    }
    c->stream = (hipStream_t)hip_stream;
    // buildOpticalFlowPyramid level geometry + early stop
    int w = cfg->width, h = cfg->height;
    int win_w, win_h;
    agt_lk_window_size(cfg->win, &win_w, &win_h);
    for (int l = 0; l <= cfg->max_level; l++) {
        c->lw[l] = w; c->lh[l] = h;
        c->lpitch[l] = ((long)w + 63) & ~63L;
        c->eff_max_level = l;
        w = (w + 1) / 2; h = (h + 1) / 2;
        if (w <= win_w || h <= win_h) break;
    }
    const size_t B = (size_t)cfg->max_streams, N = (size_t)cfg->max_points;
    c->group = 1;
    const int ring0 = c->eff_max_level + 2 > AGT_SLOTS ? c->eff_max_level + 2 : AGT_SLOTS;
    bool ok = ensure_ring(c, ring0) == AGT_OK;
    c->live_ring = ring0;
    ok = ok && hipMalloc((void**)&c->lkerr, B * N * sizeof(float)) == hipSuccess;
    ok = ok && hipMalloc((void**)&c->obj, N * 3 * sizeof(float)) == hipSuccess;
    ok = ok && hipMalloc((void**)&c->pose, B * 6 * sizeof(double)) == hipSuccess;
    ok = ok && hipMalloc((void**)&c->tstate, B * sizeof(AgtTrackState)) == hipSuccess;
    ok = ok && hipMalloc((void**)&c->lk_done, (size_t)AGT_RING_MAX * B * sizeof(unsigned)) == hipSuccess;
    ok = ok && hipMemset(c->lk_done, 0, (size_t)AGT_RING_MAX * B * sizeof(unsigned)) == hipSuccess;
    ok = ok && hipHostMalloc((void**)&c->fault_host, 256, hipHostMallocMapped) == hipSuccess;
    if (ok) { memset(c->fault_host, 0, 256); ok = hipHostGetDevicePointer((void**)&c->fault_dev, c->fault_host, 0) == hipSuccess; }
    if (ok) {
        c->hrec_host = (double*)((char*)c->fault_host + 64); c->hrec_dev = (double*)((char*)c->fault_dev + 64);
        c->hseq_host = (unsigned long long*)((char*)c->fault_host + 192); c->hseq_dev = (unsigned long long*)((char*)c->fault_dev + 192);
    }
    if (!ok) { hip_fail(nullptr, hipGetLastError()); agt_destroy(c); return AGT_ERR_ALLOC; }
    c->last_p_ev = -1; c->l_ev_hist[0] = c->l_ev_hist[1] = c->l_ev_hist[2] = -1; c->y_ev_hist[0] = c->y_ev_hist[1] = -1;
    c->pipeline = agt_step_supported(cfg->win) ? 1 : 0;
    c->reproject = 0; c->min_points = 8; c->gate_px = 2.0;
    c->lk_max_count = 30; c->lk_eps = 0.01; c->lk_min_eig = 1e-4;
    c->lk_cap_cu = -1;                        // (the library's choice: agt_lk_occupancy_cu)
#ifdef AGT_DEBUG_KNOBS
    { const char* e = getenv("AGT_LK_SPLIT_CU"); if (e) c->lk_cap_cu = atoi(e); }     // (residency cap of every one-wave LK launch of the context)
    { const char* e = getenv("AGT_LK_HYBRID"); if (e) c->lk_slow_thr = atoi(e); }
#endif
#ifdef AGT_DEBUG_KNOBS      // diagnostic library only: fixed iteration counts (AGT_LK_MAX_COUNT=n AGT_LK_EPS=0) separate the per-iteration
    // cost of the LK role from its per-frame cost
    { const char* e = getenv("AGT_LK_MAX_COUNT"); if (e) c->lk_max_count = atoi(e); }
    { const char* e = getenv("AGT_LK_EPS"); if (e) c->lk_eps = atof(e); }
#endif
    *out = c;
    return AGT_OK;
}

int agt_destroy(agt_ctx* c)
{
    if (!c) return AGT_OK;
    (void)hipStreamSynchronize(c->stream);
    for (int s = 0; s < AGT_RING_MAX; s++) {
        for (int l = 1; l < AGT_MAX_LEVELS; l++) if (c->lmem[s][l]) (void)hipFree(c->lmem[s][l]);
        if (c->corners[s]) (void)hipFree(c->corners[s]);
        if (c->status[s]) (void)hipFree(c->status[s]);
        if (c->lk_iters[s]) (void)hipFree(c->lk_iters[s]);
    }
    if (c->ms_pool_slot >= 0) {                      // (also after an ms_init that stopped half-way: the slot and the events it did create)
        for (int i = 0; i < 3; i++) (void)hipStreamSynchronize(c->ms_stream[i]);
        for (int k = 0; k < 5; k++) for (int i = 0; i < AGT_EV_SLOTS; i++) if (c->ms_ev[k][i]) (void)hipEventDestroy(c->ms_ev[k][i]);
        ms_pool_release(c->ms_pool_slot);            // (the streams go back to the process's pool: see ms_pool_acquire)
    }
    if (c->lkerr) (void)hipFree(c->lkerr);
    if (c->obj) (void)hipFree(c->obj);
    if (c->pose) (void)hipFree(c->pose);
    if (c->tstate) (void)hipFree(c->tstate);
    if (c->lk_done) (void)hipFree(c->lk_done);
    if (c->fault_host) (void)hipHostFree(c->fault_host);
    if (c->dense_partials) (void)hipFree(c->dense_partials);
    if (c->dense_done) (void)hipFree(c->dense_done);
    if (c->d_tilt) (void)hipFree(c->d_tilt);
    if (c->hcall_host) (void)hipHostFree(c->hcall_host);
    if (c->map1) (void)hipFree(c->map1);
    if (c->map2) (void)hipFree(c->map2);
    if (c->prof_ev) {
        for (size_t i = 0; i < (size_t)c->prof_cap * AGT_PROF_EVENTS; i++) (void)hipEventDestroy(c->prof_ev[i]);
        delete[] c->prof_ev;
        delete[] c->prof_dense;
    }
    delete c;
    return AGT_OK;
}

int agt_set_stream(agt_ctx* c, void* hip_stream)
{
    if (!c) return AGT_ERR_ARG;
    c->stream = (hipStream_t)hip_stream;
    return AGT_OK;
}

int agt_last_hip_error(const agt_ctx* c) { return c ? c->last_hip : g_last_hip; }

int agt_synchronize(agt_ctx* c)
{
    if (!c) return AGT_ERR_ARG;
    int rc = join_pipeline(c);            // flush frames still inside the software pipeline
    if (rc) return rc;
    hipError_t e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess) return hip_fail(c, e);
    return *(volatile int*)c->fault_host ? AGT_ERR_CHAIN : AGT_OK;
}

int agt_upload(agt_ctx* c, void* d_dst, const void* h_src, size_t bytes)
{
    if (!c || !d_dst || !h_src) return AGT_ERR_ARG;
    hipError_t e = hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice, c->stream);
    return e == hipSuccess ? AGT_OK : hip_fail(c, e);
}

int agt_download(agt_ctx* c, void* h_dst, const void* d_src, size_t bytes)
{
    if (!c || !h_dst || !d_src) return AGT_ERR_ARG;
    hipError_t e = hipMemcpyAsync(h_dst, d_src, bytes, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    return e == hipSuccess ? AGT_OK : hip_fail(c, e);
}

int agt_pyr_down_u8(agt_ctx* c, const uint8_t* d_src, int sw, int sh, size_t spitch, size_t sbatch,
                    uint8_t* d_dst, size_t dpitch, size_t dbatch, int B)
{
    if (!c || !d_src || !d_dst || sw <= 0 || sh <= 0 || B <= 0) return AGT_ERR_ARG;
    if ((spitch & 3) || (dpitch & 3) || ((uintptr_t)d_src & 3) || ((uintptr_t)d_dst & 3) || (sbatch & 3) || (dbatch & 3)) return AGT_ERR_ARG;
    if (spitch < (size_t)sw || dpitch < (size_t)((sw + 1) / 2)) return AGT_ERR_ARG;
    hipError_t e = agt_launch_pyr_down(c->stream, d_src, sw, sh, (long)spitch, (long)sbatch, d_dst, (long)dpitch, (long)dbatch, B);
    return e == hipSuccess ? AGT_OK : hip_fail(c, e);
}

static int pyramid_build_on(agt_ctx* c, hipStream_t stream, int slot, const uint8_t* d_frames, size_t pitch, size_t batch_stride, int B)
{
    if (!c || !d_frames || slot < 0 || slot >= c->ring || B <= 0 || B > c->cfg.max_streams) return AGT_ERR_ARG;
    if ((pitch & 3) || ((uintptr_t)d_frames & 3) || (batch_stride & 3) || pitch < (size_t)c->cfg.width) return AGT_ERR_ARG;
    c->l0_ptr[slot] = d_frames; c->l0_pitch[slot] = (long)pitch; c->l0_bstride[slot] = (long)batch_stride;
    const uint8_t* src = d_frames; long sp = (long)pitch, sb = (long)batch_stride;
    int l0 = 1;
    bool two_level = c->eff_max_level >= 2 && B <= AGT_PYR2_MAX_B;
    if (c->eff_max_level >= 2 && !two_level) {
        // big batches: the two-level pass only in its register-rolling form (the tiled one loses to two single passes there)
        AgtPyrArgs A0, A1;
        agt_pyr2_args(src, c->lw[0], c->lh[0], sp, sb, c->lmem[slot][1], c->lpitch[1], (long)c->lh[1] * c->lpitch[1],
                      c->lmem[slot][2], c->lpitch[2], (long)c->lh[2] * c->lpitch[2], B, &A0, &A1);
        agt_pyr2_plan(&A0, &A1, (uintptr_t)src, (uintptr_t)c->lmem[slot][1] | (uintptr_t)c->lmem[slot][2], 1);
        two_level = A0.pad != 0;
    }
    if (two_level) {
        // levels 1 and 2 in one pass: level 0 is read once, level 1 is never re-read from HBM
        const long db1 = (long)c->lh[1] * c->lpitch[1], db2 = (long)c->lh[2] * c->lpitch[2];
        hipError_t e = agt_launch_pyr_down2(stream, src, c->lw[0], c->lh[0], sp, sb, c->lmem[slot][1], c->lpitch[1], db1,
                                            c->lmem[slot][2], c->lpitch[2], db2, B);
        if (e != hipSuccess) return hip_fail(c, e);
        src = c->lmem[slot][2]; sp = c->lpitch[2]; sb = db2;
        l0 = 3;
    }
    for (int l = l0; l <= c->eff_max_level; l++) {
        const long db = (long)c->lh[l] * c->lpitch[l];
        hipError_t e = agt_launch_pyr_down(stream, src, c->lw[l - 1], c->lh[l - 1], sp, sb, c->lmem[slot][l], c->lpitch[l], db, B);
        if (e != hipSuccess) return hip_fail(c, e);
        src = c->lmem[slot][l]; sp = c->lpitch[l]; sb = db;
    }
    c->built_B[slot] = B;
    return AGT_OK;
}

int agt_pyramid_build(agt_ctx* c, int slot, const uint8_t* d_frames, size_t pitch, size_t batch_stride, int B)
{
    if (!c || slot < 0 || slot > 1) return AGT_ERR_ARG;
    // slots 0 / 1 are ring entries of the tracker too: frames still in flight (fused pipeline groups not yet
    // launched, stage kernels on the library's streams) are enqueued / ordered in front of this build first
    int rc = join_pipeline(c);
    if (rc) return rc;
    c->prebuilt_t = -1;
    return pyramid_build_on(c, c->stream, slot, d_frames, pitch, batch_stride, B);
}

// Both pyramids of a frame pair (slot 0 <- d_prev, slot 1 <- d_next) with ONE launch for levels 1 and 2 (round 6).  A batch of cold
// pairs pays two pyramid launches per step otherwise, each a stream of its own length with a ramp and a tail and a kernel boundary
// between them (~2 us between two streaming kernels on one stream): the two-level rolling pass of 2 B images is the same pass, once.
// The geometry of both frames is one (pitch, batch_stride); deeper levels, other windows and small batches fall back to two builds.
int agt_pyramid_build_pair(agt_ctx* c, const uint8_t* d_prev, const uint8_t* d_next, size_t pitch, size_t batch_stride, int B)
{
    if (!c || !d_prev || !d_next) return AGT_ERR_ARG;
    int rc = join_pipeline(c);
    if (rc) return rc;
    c->prebuilt_t = -1;
    const int L = c->eff_max_level;
    bool one = L >= 2 && agt_step_supported(c->cfg.win) && B > 0 && B <= c->cfg.max_streams &&
               !((pitch & 3) || ((uintptr_t)d_prev & 3) || ((uintptr_t)d_next & 3) || (batch_stride & 3) || pitch < (size_t)c->cfg.width);
    AgtStepParams S;
    AgtStepTables T;
    if (one) {
        memset(&S, 0, sizeof(S));
        memset(&T, 0, sizeof(T));
        S.pnp.fault = c->fault_dev;
        AgtPyrArgs& A = S.pyr[0];
        A.sw = c->lw[0]; A.sh = c->lh[0]; A.dw = c->lw[1]; A.dh = c->lh[1];
        A.spitch = (long)pitch; A.sbatch = (long)batch_stride;
        A.dpitch = c->lpitch[1]; A.dbatch = (long)c->lh[1] * c->lpitch[1];
        A.B = B;
        uintptr_t src_align = 0, dst_align = 0;
        const uint8_t* src[2] = { d_prev, d_next };
        for (int k = 0; k < 2; k++) {
            T.pyr_src[0][k] = src[k]; T.pyr_dst[0][k] = c->lmem[k][1]; T.pyr_dst[1][k] = c->lmem[k][2];
            src_align |= (uintptr_t)src[k]; dst_align |= (uintptr_t)c->lmem[k][1] | (uintptr_t)c->lmem[k][2];
        }
        A.src = T.pyr_src[0][0]; A.dst = T.pyr_dst[0][0];
        AgtPyrArgs A0 = A, A1 = S.pyr[1];
        A1.sw = c->lw[1]; A1.sh = c->lh[1]; A1.dw = c->lw[2]; A1.dh = c->lh[2];
        A1.spitch = c->lpitch[1]; A1.sbatch = (long)c->lh[1] * c->lpitch[1];
        A1.dpitch = c->lpitch[2]; A1.dbatch = (long)c->lh[2] * c->lpitch[2];
        A1.B = B; A1.src = nullptr; A1.dst = nullptr;
        agt_pyr2_plan(&A0, &A1, src_align, dst_align, 2);
        one = A0.pad != 0 || B <= AGT_PYR2_MAX_B;             // (rolling form for big batches; the tiled two-level pass for small ones)
        if (one) {
            S.pyr[0] = A0; S.pyr[1] = A1;
            S.pyr_fused = 1;
            S.pyr_nf[0] = 2;
            S.n_pyr[0] = A0.gx * A0.gy * B * 2;
        }
    }
    if (!one) {
        rc = pyramid_build_on(c, c->stream, 0, d_prev, pitch, batch_stride, B);
        return rc ? rc : pyramid_build_on(c, c->stream, 1, d_next, pitch, batch_stride, B);
    }
    hipError_t e = agt_launch_step(c->stream, S, T, c->cfg.win, AGT_STEP_PYR);
    if (e != hipSuccess) return hip_fail(c, e);
    const uint8_t* src[2] = { d_prev, d_next };
    for (int k = 0; k < 2; k++) {
        c->l0_ptr[k] = src[k]; c->l0_pitch[k] = (long)pitch; c->l0_bstride[k] = (long)batch_stride;
        const uint8_t* sp_ = c->lmem[k][2]; long spitch = c->lpitch[2], sb = (long)c->lh[2] * c->lpitch[2];
        for (int l = 3; l <= L; l++) {
            const long db = (long)c->lh[l] * c->lpitch[l];
            e = agt_launch_pyr_down(c->stream, sp_, c->lw[l - 1], c->lh[l - 1], spitch, sb, c->lmem[k][l], c->lpitch[l], db, B);
            if (e != hipSuccess) return hip_fail(c, e);
            sp_ = c->lmem[k][l]; spitch = c->lpitch[l]; sb = db;
        }
        c->built_B[k] = B;
    }
    return AGT_OK;
}

int agt_pyramid_max_level(const agt_ctx* c) { return c ? c->eff_max_level : AGT_ERR_ARG; }

int agt_pyramid_level(const agt_ctx* c, int slot, int level, const uint8_t** d_ptr,
                      int* w, int* h, size_t* pitch, size_t* batch_stride)
{
    if (!c || slot < 0 || slot >= c->ring || level < 0 || level > c->eff_max_level) return AGT_ERR_ARG;
    if (c->built_B[slot] <= 0) return AGT_ERR_STATE;
    if (d_ptr) *d_ptr = level == 0 ? c->l0_ptr[slot] : c->lmem[slot][level];
    if (w) *w = c->lw[level];
    if (h) *h = c->lh[level];
    if (pitch) *pitch = (size_t)(level == 0 ? c->l0_pitch[slot] : c->lpitch[level]);
    if (batch_stride) *batch_stride = (size_t)(level == 0 ? c->l0_bstride[slot] : (long)c->lh[level] * c->lpitch[level]);
    return AGT_OK;
}

static void fill_levels(const agt_ctx* c, int slot, AgtLevel* L)
{
    for (int l = 0; l <= c->eff_max_level; l++) {
        L[l].w = c->lw[l]; L[l].h = c->lh[l];
        if (l == 0) { L[l].ptr = c->l0_ptr[slot]; L[l].pitch = c->l0_pitch[slot]; L[l].bstride = c->l0_bstride[slot]; }
        else { L[l].ptr = c->lmem[slot][l]; L[l].pitch = c->lpitch[l]; L[l].bstride = (long)c->lh[l] * c->lpitch[l]; }
    }
}

#define AGT_SPLIT_LK_CU 10       // (see agt_lk_occupancy_cu)
#define AGT_SPLIT_PYR_OH 6       // strip height (level-2 rows) of the split pipeline's pyramid role (agt_pyramid.hip agt_pyr2_plan) ...
#define AGT_SPLIT_PYR_OH_B0 12   // ... for batches of 12 .. 96 streams: measured (session r6ar, us per step, strips of 6 / 16) 8 streams 20.1 / 19.8,
#define AGT_SPLIT_PYR_OH_B1 96   // 16: 27.5 / 28.3, 32: 29.4 / 29.9, 64: 35.1-35.9 / 36.7-37.6, 128: 63.7 / 62.4
static int lk_lds_min(int per_cu);
static int lk_track_on(agt_ctx* c, hipStream_t stream, int prev_slot, int next_slot,
                       const float* d_prev_pts, const uint8_t* d_prev_status, float* d_next_pts, uint8_t* d_status, float* d_err,
                       int n, int B, int crit_type, int crit_max_count, double crit_eps,
                       int flags, double min_eig_threshold, int b0 = 0, int waves = 0, bool hybrid = false)
{
    // b0: first stream of the launch (streams b0 .. b0 + B - 1 of the slots and of the point arrays); waves: see agt_launch_lk
    if (!c || !d_prev_pts || !d_next_pts || !d_status) return AGT_ERR_ARG;
    if (prev_slot < 0 || prev_slot >= c->ring || next_slot < 0 || next_slot >= c->ring) return AGT_ERR_ARG;
    if (n < 0 || B <= 0 || b0 < 0) return AGT_ERR_ARG;
    if (n == 0) return AGT_OK;
    if (c->built_B[prev_slot] < b0 + B || c->built_B[next_slot] < b0 + B) return AGT_ERR_STATE;
    AgtLkParams p;
    memset(&p, 0, sizeof(p));
    fill_levels(c, prev_slot, p.prev);
    fill_levels(c, next_slot, p.next);
    p.max_level = c->eff_max_level;
    p.n = n;
    // SparsePyrLKOpticalFlowImpl::calc criteria normalisation
    p.max_count = (crit_type & AGT_TERM_COUNT) ? (crit_max_count < 0 ? 0 : crit_max_count > 100 ? 100 : crit_max_count) : 30;
    double eps = (crit_type & AGT_TERM_EPS) ? (crit_eps < 0. ? 0. : crit_eps > 10. ? 10. : crit_eps) : 0.01;
    p.eps2 = eps * eps;
    p.flags = flags & 0xffff;            // (the bits above are internal launch flags)
    // (waves == 1: a half-batch launch of the split pipeline -- beside the other half's launch and the pyramid launch of the frames ahead)
    p.lds_min = lk_lds_min(c->lk_cap_cu >= 0 ? c->lk_cap_cu : (waves == 1 ? AGT_SPLIT_LK_CU : 0));
    if (c->lk_cap_cu > 0) p.flags |= AGT_LK_FLAG_COTENANT;        // (declared by the caller: agt_lk_occupancy_cu; see agt_lk.hip lk_kernel)
#ifdef AGT_DEBUG_KNOBS      // AGT_LK_COTENANT_PRIO=0: the cap without the issue priority (A/B of the two)
    { static const int on = [] { const char* e = getenv("AGT_LK_COTENANT_PRIO"); return e ? atoi(e) : 1; }(); if (!on) p.flags &= ~AGT_LK_FLAG_COTENANT; }
#endif
    p.min_eig_threshold = min_eig_threshold;
    p.prev_pts = d_prev_pts; p.prev_status = d_prev_status; p.next_pts = d_next_pts; p.status = d_status; p.err = d_err;
    if (b0) {
        for (int l = 0; l <= c->eff_max_level; l++) { p.prev[l].ptr += (long)b0 * p.prev[l].bstride; p.next[l].ptr += (long)b0 * p.next[l].bstride; }
        p.prev_pts += (size_t)b0 * n * 2; p.next_pts += (size_t)b0 * n * 2; p.status += (size_t)b0 * n;
        if (p.prev_status) p.prev_status += (size_t)b0 * n;
        if (p.err) p.err += (size_t)b0 * n;
    }
    // HYBRID launch, MEASURED AND NOT SHIPPED (round 5; knobs build: AGT_LK_HYBRID=<iterations>): corners that took >= that
    // many iterations in the previous frame tracked by four waves, the rest by one, in one launch of two workgroup roles
    // (agt_lk.hip lk_hybrid_kernel).  Bit-identical records; 64 streams: 40.2 us per step without, 49.5 with the launch form alone
    // (threshold 200: nobody on four waves), 49.8 / 51.9 / 61.6 / 75.2 at thresholds 14 / 12 / 10 / 8 -- the chip is as much
    // throughput- as latency-bound there, and a four-wave corner costs 3 x the wave-time of a one-wave corner (profiles/r05_experiments.md).
    if (hybrid && c->lk_slow_thr > 0 && c->cfg.win == 21 && c->eff_max_level < 3 && !p.err && flags == 0 &&
        (size_t)(b0 + B) * n <= (size_t)c->cfg.max_streams * c->cfg.max_points) {
        p.iters_prev = c->lk_iters[prev_slot] + (size_t)b0 * n; p.iters_out = c->lk_iters[next_slot] + (size_t)b0 * n;
        p.slow_thr = c->lk_slow_thr;
        hipError_t e = agt_launch_lk_hybrid(stream, p, B);
        return e == hipSuccess ? AGT_OK : hip_fail(c, e);
    }
    hipError_t e = agt_launch_lk(stream, p, c->cfg.win, B, c->cfg.win == 21 ? waves : 0);
    return e == hipSuccess ? AGT_OK : hip_fail(c, e);
}

int agt_lk_track(agt_ctx* c, int prev_slot, int next_slot,
                 const float* d_prev_pts, float* d_next_pts, uint8_t* d_status, float* d_err,
                 int n, int B, int crit_type, int crit_max_count, double crit_eps,
                 int flags, double min_eig_threshold)
{
    if (!c || prev_slot < 0 || prev_slot > 1 || next_slot < 0 || next_slot > 1) return AGT_ERR_ARG;
    return lk_track_on(c, c->stream, prev_slot, next_slot, d_prev_pts, nullptr, d_next_pts, d_status, d_err, n, B,
                       crit_type, crit_max_count, crit_eps, flags, min_eig_threshold);
}

int agt_solve_pnp(agt_ctx* c, const void* d_obj, size_t obj_batch_stride, const void* d_img, int dtype,
                  const uint8_t* d_mask, int n, int B,
                  const double* K, const double* dist, int ndist,
                  double* d_pose, int use_guess, int32_t* d_info, double* d_err)
{
    if (!c || !d_obj || !d_img || !d_pose || B <= 0) return AGT_ERR_ARG;
    if (dtype != AGT_F32 && dtype != AGT_F64) return AGT_ERR_ARG;
    if (n < 3 || n > 256) return AGT_ERR_NPOINTS;
    if (!use_guess && n < 4) return AGT_ERR_NPOINTS;
    AgtPnpParams p;
    memset(&p, 0, sizeof(p));
    int rc = camera_on(c, K, dist, ndist, &p.cam);
    if (rc) return rc;
    p.obj = d_obj; p.obj_bstride = (long)obj_batch_stride; p.img = d_img; p.mask = d_mask; p.dtype = dtype;
    p.n = n; p.use_guess = use_guess ? 1 : 0; p.pose = d_pose; p.info = d_info; p.err = d_err;
    p.gate_px = 2.0;
    hipError_t e = agt_launch_pnp(c->stream, p, B);
    return e == hipSuccess ? AGT_OK : hip_fail(c, e);
}

int agt_project_points(agt_ctx* c, const void* d_obj, size_t obj_batch_stride, int dtype, int n, int B,
                       const double* d_pose, const double* K, const double* dist, int ndist,
                       void* d_img_out, double* d_jac)
{
    if (!c || !d_obj || !d_pose || !d_img_out || n <= 0 || B <= 0) return AGT_ERR_ARG;
    if (dtype != AGT_F32 && dtype != AGT_F64) return AGT_ERR_ARG;
    AgtProjParams p;
    memset(&p, 0, sizeof(p));
    int rc = camera_on(c, K, dist, ndist, &p.cam);
    if (rc) return rc;
    p.obj = d_obj; p.obj_bstride = (long)obj_batch_stride; p.dtype = dtype; p.n = n; p.pose = d_pose;
    p.img_out = d_img_out; p.jac = d_jac;
    hipError_t e = agt_launch_project(c->stream, p, B);
    return e == hipSuccess ? AGT_OK : hip_fail(c, e);
}

// ---- the reference's per-frame calls as ONE synchronous call each, host arrays in and out (detect_pose.py:509-526 solvePnP,
// :441-465 projectPoints).  No copies are enqueued: the arguments go into a host-mapped staging area of the
// context, the kernel reads them and writes its results there over PCIe and stores a sequence word behind them (system scope), the
// calling thread polls the word.  Against upload + launch + download + stream wait (44 / 28 us per call): see DESIGN.md section 6.
namespace {
constexpr size_t HC_SEQ = 0, HC_POSE = 64, HC_INFO = 128, HC_ERR = 144, HC_OBJ = 256, HC_IMG = HC_OBJ + 256 * 3 * 8,
                 HC_JAC = HC_IMG + 256 * 2 * 8, HC_SIZE = HC_JAC + 2 * 256 * 6 * 8;

int hcall_ready(agt_ctx* c)
{
    if (c->hcall_host) return AGT_OK;
    if (hipHostMalloc((void**)&c->hcall_host, HC_SIZE, hipHostMallocMapped) != hipSuccess) { (void)hipGetLastError(); c->hcall_host = nullptr; return AGT_ERR_ALLOC; }
    memset(c->hcall_host, 0, HC_SIZE);
    if (hipHostGetDevicePointer((void**)&c->hcall_dev, c->hcall_host, 0) != hipSuccess) {
        (void)hipGetLastError(); (void)hipHostFree(c->hcall_host); c->hcall_host = nullptr; return AGT_ERR_HIP;
    }
    c->hcall_n = 0;
    return AGT_OK;
}

// poll the staging area's sequence word for `want`; after 2 s without it the stream is asked what happened
int hcall_wait(agt_ctx* c, unsigned long long want)
{
    const volatile unsigned long long* seq = (const volatile unsigned long long*)(c->hcall_host + HC_SEQ);
    timespec t0; clock_gettime(CLOCK_MONOTONIC, &t0);
    for (unsigned long spins = 1; *seq < want; spins++) {
#if defined(__x86_64__) || defined(__i386__)
        __builtin_ia32_pause();
#else
        __asm__ __volatile__("" ::: "memory");
#endif
        if ((spins & 0xffff) == 0) {
            timespec t1; clock_gettime(CLOCK_MONOTONIC, &t1);
            if ((t1.tv_sec - t0.tv_sec) + (t1.tv_nsec - t0.tv_nsec) * 1e-9 > 2.0) {
                hipError_t e = hipStreamSynchronize(c->stream);
                if (e != hipSuccess) return hip_fail(c, e);
                if (*seq < want) return AGT_ERR_STATE;
            }
        }
    }
    __atomic_thread_fence(__ATOMIC_ACQUIRE);
    return AGT_OK;
}
}  // namespace

int agt_solve_pnp_host(agt_ctx* c, const void* h_obj, const void* h_img, int dtype, int n,
                       const double* K, const double* dist, int ndist,
                       double* h_pose, int use_guess, int32_t* h_info, double* h_err)
{
    if (!c || !h_obj || !h_img || !h_pose) return AGT_ERR_ARG;
    if (dtype != AGT_F32 && dtype != AGT_F64) return AGT_ERR_ARG;
    if (n < 3 || n > 256) return AGT_ERR_NPOINTS;
    if (!use_guess && n < 4) return AGT_ERR_NPOINTS;
    int rc = hcall_ready(c);
    if (rc) return rc;
    AgtPnpParams p;
    memset(&p, 0, sizeof(p));
    rc = camera_on(c, K, dist, ndist, &p.cam);
    if (rc) return rc;
    const size_t es = dtype == AGT_F32 ? 4 : 8;
    memcpy(c->hcall_host + HC_OBJ, h_obj, (size_t)n * 3 * es);
    memcpy(c->hcall_host + HC_IMG, h_img, (size_t)n * 2 * es);
    double* pose = (double*)(c->hcall_host + HC_POSE);
    if (use_guess) memcpy(pose, h_pose, 6 * sizeof(double)); else memset(pose, 0, 6 * sizeof(double));
    p.obj = c->hcall_dev + HC_OBJ; p.obj_bstride = 0; p.img = c->hcall_dev + HC_IMG; p.mask = nullptr; p.dtype = dtype;
    p.n = n; p.use_guess = use_guess ? 1 : 0; p.pose = (double*)(c->hcall_dev + HC_POSE);
    p.info = (int32_t*)(c->hcall_dev + HC_INFO); p.err = (double*)(c->hcall_dev + HC_ERR);
    p.gate_px = 2.0;
    const unsigned long long want = ++c->hcall_n;
    p.host_seq = (unsigned long long*)(c->hcall_dev + HC_SEQ); p.host_seq_base = want;
    hipError_t e = agt_launch_pnp(c->stream, p, 1);
    if (e != hipSuccess) return hip_fail(c, e);
    rc = hcall_wait(c, want);
    if (rc) return rc;
    memcpy(h_pose, pose, 6 * sizeof(double));
    if (h_info) memcpy(h_info, c->hcall_host + HC_INFO, 4 * sizeof(int32_t));
    if (h_err) *h_err = *(const double*)(c->hcall_host + HC_ERR);
    return AGT_OK;
}

int agt_project_points_host(agt_ctx* c, const void* h_obj, int dtype, int n, const double* h_pose,
                            const double* K, const double* dist, int ndist, void* h_img_out, double* h_jac)
{
    if (!c || !h_obj || !h_pose || !h_img_out) return AGT_ERR_ARG;
    if (dtype != AGT_F32 && dtype != AGT_F64) return AGT_ERR_ARG;
    if (n <= 0 || n > 256) return AGT_ERR_NPOINTS;          // (one workgroup: the reference projects its 48 .. 240 model corners)
    int rc = hcall_ready(c);
    if (rc) return rc;
    AgtProjParams p;
    memset(&p, 0, sizeof(p));
    rc = camera_on(c, K, dist, ndist, &p.cam);
    if (rc) return rc;
    const size_t es = dtype == AGT_F32 ? 4 : 8;
    memcpy(c->hcall_host + HC_OBJ, h_obj, (size_t)n * 3 * es);
    memcpy(c->hcall_host + HC_POSE, h_pose, 6 * sizeof(double));
    p.obj = c->hcall_dev + HC_OBJ; p.obj_bstride = 0; p.dtype = dtype; p.n = n; p.pose = (const double*)(c->hcall_dev + HC_POSE);
    p.img_out = c->hcall_dev + HC_IMG; p.jac = h_jac ? (double*)(c->hcall_dev + HC_JAC) : nullptr;
    const unsigned long long want = ++c->hcall_n;
    p.host_seq = (unsigned long long*)(c->hcall_dev + HC_SEQ); p.host_seq_base = want;
    hipError_t e = agt_launch_project(c->stream, p, 1);
    if (e != hipSuccess) return hip_fail(c, e);
    rc = hcall_wait(c, want);
    if (rc) return rc;
    memcpy(h_img_out, c->hcall_host + HC_IMG, (size_t)n * 2 * es);
    if (h_jac) memcpy(h_jac, c->hcall_host + HC_JAC, (size_t)n * 12 * sizeof(double));
    return AGT_OK;
}

int agt_tracker_reset(agt_ctx* c, int slot, const float* d_corners, const float* d_obj, int n, int B,
                      const double* K, const double* dist, int ndist, int enhance_ape)
{
    if (!c || !d_obj || slot < 0 || slot > 1) return AGT_ERR_ARG;
    if (n < 4 || n > c->cfg.max_points) return AGT_ERR_NPOINTS;
    if (c->tag_gate && n % c->tag_gate) return AGT_ERR_NPOINTS;      // whole tags: corners 4t .. 4t + 3
    if (B <= 0 || B > c->cfg.max_streams) return AGT_ERR_ARG;
    if (d_corners && c->built_B[slot] < B) return AGT_ERR_STATE;
    int rc = join_pipeline(c);            // frames of an earlier run still in flight (fused pipeline or library streams)
    if (rc) return rc;
    if (*(volatile int*)c->fault_host) {
        // recovery from a chained-wait give-up: everything in flight has to be over before the fault word is cleared
        hipError_t es = hipStreamSynchronize(c->stream);
        if (es != hipSuccess) return hip_fail(c, es);
        *(volatile int*)c->fault_host = 0;
    }
    rc = camera_on(c, K, dist, ndist, &c->cam, true);
    if (rc) return rc;
    hipError_t e = hipSuccess;
    // the tracker's frame 0 lives in ring entry 0: adopt the caller's slot as pyramid slot 0
    if (d_corners && slot != 0) {
        for (int l = 1; l <= c->eff_max_level; l++) { uint8_t* t = c->lmem[0][l]; c->lmem[0][l] = c->lmem[slot][l]; c->lmem[slot][l] = t; }
        c->l0_ptr[0] = c->l0_ptr[slot]; c->l0_pitch[0] = c->l0_pitch[slot]; c->l0_bstride[0] = c->l0_bstride[slot];
        c->built_B[0] = c->built_B[slot]; c->built_B[slot] = 0;
    }
    if (d_corners) e = hipMemcpyAsync(c->corners[0], d_corners, (size_t)B * n * 2 * sizeof(float), hipMemcpyDeviceToDevice, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(c->obj, d_obj, (size_t)n * 3 * sizeof(float), hipMemcpyDeviceToDevice, c->stream);
    if (e == hipSuccess) e = hipMemsetAsync(c->tstate, 0, (size_t)B * sizeof(AgtTrackState), c->stream);
    for (int s = 0; s < c->ring && e == hipSuccess; s++) e = hipMemsetAsync(c->status[s], 1, (size_t)B * n, c->stream);
#ifdef AGT_DEBUG_KNOBS
    for (int s = 0; s < c->ring && e == hipSuccess; s++) e = hipMemsetAsync(c->lk_iters[s], 0, (size_t)B * n, c->stream);
#endif
    // arrival counters of the chained launches: a run counts n corners per stream-frame, the next run may have other n / B
    if (e == hipSuccess) e = hipMemsetAsync(c->lk_done, 0, (size_t)AGT_RING_MAX * c->cfg.max_streams * sizeof(unsigned), c->stream);
    memset(c->lk_target, 0, sizeof(c->lk_target));
    if (e != hipSuccess) return hip_fail(c, e);
    c->prebuilt_t = -1;
    c->hseq_off = c->hseq_last + 1;          // (sequence numbers of agt_track_host_frame stay monotonic across runs)
    c->trk_n = n; c->trk_B = B; c->trk_frame = 0; c->n_lk = c->n_pnp = 0; c->enhance_ape = enhance_ape ? 1 : 0;
    for (int s = 0; s < AGT_MAX_LEVELS; s++) c->n_stage[s] = 0;
    c->trk_ready = d_corners ? 2 : 1;
    return AGT_OK;
}

int agt_tracker_options(agt_ctx* c, int reproject, int min_points, double gate_px)
{
    if (!c || min_points < 6 || min_points > 256 || !(gate_px > 0.0)) return AGT_ERR_ARG;
    int rc = join_pipeline(c);
    if (rc) return rc;
    c->reproject = reproject ? 1 : 0; c->min_points = min_points; c->gate_px = gate_px;
    return AGT_OK;
}

// Residency cap of the one-wave-per-corner LK kernel (big batches): at most `workgroups_per_cu` of its waves resident on a CU (0 = no
// cap: as many as registers and LDS allow, 14-16; -1 = the library's choice, the default).  For launches that SHARE the device with
// other kernels: a tracker wave holds 112 registers and ~7.5 KB of LDS, twelve of them on a CU leave one wave slot of registers per SIMD
// to anybody else, and a launch lasts as long as its slowest corner -- the HBM-bound pyramid passes beside it starve.  The cap is
// enforced with LDS: every workgroup (one wave) asks for at least LDS_per_CU / workgroups_per_cu bytes (more than its tiles need).
// The library's choice: none for a launch of its own (agt_lk_track, the fused step); AGT_SPLIT_LK_CU for the half-batch launches of the
// split pipeline, which run beside the pyramid launch of the frames ahead (round 6, 64 streams of 48 corners: 12 per CU -- all 3,072
// corners resident, the LK launch alone 30.8 us -- step 39.5-40.2 us; 11: 39.1; 10: 38.4; 9: 38.7-39.5; 8: 40.1 -- with 10 the launch alone
// takes 41 us, the step is the shortest; gpurun_out/r6s, profiles/r06_experiments.md).
static int lk_lds_min(int per_cu)
{
    if (per_cu <= 0) return 0;
    // gfx950: 160 KB of LDS per CU (MI355X_MICROARCH.md; hipDeviceProp_t reports the 64 KB a workgroup may ask for, not this).  The
    // allocation granule is not documented (measured: 12,800 B -> 12 per CU, 13,824 B -> 11: consistent with 512, 1,024 and 1,280 B), so
    // the request is the smallest multiple of 256 B above LDS_per_CU / (per_cu + 1): per_cu + 1 workgroups do not fit WHATEVER the granule
    // ("at most per_cu"), and per_cu do fit for granules up to 512 B (for 1,024 / 1,280 B too at the counts the library itself uses, 8 and
    // 10; at some other counts one fewer would: tests/test_chip.py)
    const long lds_cu = 160L * 1024;
    const long v = (lds_cu / (per_cu + 1) / 256 + 1) * 256;
    return v > 64L * 1024 ? 0 : (int)v;                 // (a workgroup may ask for 64 KB in all: one per CU cannot be expressed -- no cap then)
}

int agt_lk_lds_request(int workgroups_per_cu) { return lk_lds_min(workgroups_per_cu); }

int agt_lk_occupancy_cu(agt_ctx* c, int workgroups_per_cu)
{
    if (!c || workgroups_per_cu < -1 || workgroups_per_cu > 32) return AGT_ERR_ARG;
    int rc = join_pipeline(c);
    if (rc) return rc;
    c->lk_cap_cu = workgroups_per_cu;
    return AGT_OK;
}

// the same in waves per SIMD (round 5's entry point): 4 x waves_per_simd workgroups per CU, 0 = no cap
int agt_lk_occupancy(agt_ctx* c, int waves_per_simd)
{
    if (!c || waves_per_simd < 0 || waves_per_simd > 8) return AGT_ERR_ARG;
    return agt_lk_occupancy_cu(c, 4 * waves_per_simd);
}

int agt_tracker_tag_gate(agt_ctx* c, int corners_per_tag)
{
    if (!c || (corners_per_tag != 0 && corners_per_tag != 4)) return AGT_ERR_ARG;
    if (corners_per_tag && c->trk_ready && c->trk_n % corners_per_tag) return AGT_ERR_NPOINTS;
    int rc = join_pipeline(c);
    if (rc) return rc;
    c->tag_gate = corners_per_tag;
    return AGT_OK;
}

// move frame `f`'s ring entry when the ring size changes (only the newest frame is live after a join)
static void ring_move(agt_ctx* c, long f, int old_ring, int new_ring)
{
    const int a = (int)(f % old_ring), b = (int)(f % new_ring);
    if (a == b) return;
    for (int l = 1; l < AGT_MAX_LEVELS; l++) { uint8_t* t = c->lmem[a][l]; c->lmem[a][l] = c->lmem[b][l]; c->lmem[b][l] = t; }
    { float* t = c->corners[a]; c->corners[a] = c->corners[b]; c->corners[b] = t; }
    { uint8_t* t = c->status[a]; c->status[a] = c->status[b]; c->status[b] = t; }
    { uint8_t* t = c->lk_iters[a]; c->lk_iters[a] = c->lk_iters[b]; c->lk_iters[b] = t; }
    c->l0_ptr[b] = c->l0_ptr[a]; c->l0_pitch[b] = c->l0_pitch[a]; c->l0_bstride[b] = c->l0_bstride[a];
    c->built_B[b] = c->built_B[a]; c->built_B[a] = 0;
}

// depth 0: separate launches per stage, pose complete in stream order.  depth F >= 1: fused software-pipelined
// step that advances every stage by F frames per launch (one launch every F calls of agt_track_frame).
int agt_tracker_pipeline(agt_ctx* c, int depth)
{
    if (!c || depth < 0 || depth > AGT_MAX_GROUP) return AGT_ERR_ARG;
    if (depth && !agt_step_supported(c->cfg.win)) return AGT_ERR_UNSUPPORTED;
    int rc = join_pipeline(c);
    if (rc) return rc;
    c->pipeline = depth ? 1 : 0;
    const int group = depth ? depth : 1;
    if (group != c->group) {
        // every frame <= trk_frame is complete; re-seat the newest one in a ring that fits the new depth
        const int want = (c->eff_max_level + 2) * group > AGT_SLOTS ? (c->eff_max_level + 2) * group : AGT_SLOTS;
        const int live_ring = c->live_ring;
        rc = ensure_ring(c, want);
        if (rc) return rc;
        if (c->trk_ready == 2) ring_move(c, c->trk_frame, live_ring, want);
        c->live_ring = want;
        c->group = group;
    }
    return AGT_OK;
}

static void fill_estimate(const agt_ctx* c, AgtPnpParams* p, const float* d_img, const uint8_t* d_mask,
                          double* d_state_out, float* corners_rw, uint8_t* status_rw = nullptr)
{
    memset(p, 0, sizeof(*p));
    p->obj = c->obj; p->obj_bstride = 0; p->img = d_img; p->mask = d_mask; p->dtype = AGT_F32;
    p->n = c->trk_n; p->cam = c->cam; p->pose = c->pose;
    p->track = c->tstate; p->state_out = d_state_out; p->corners_rw = corners_rw; p->status_rw = status_rw;
    p->enhance_ape = c->enhance_ape; p->reproject = c->reproject; p->min_points = c->min_points; p->gate_px = c->gate_px;
    p->tag_gate = c->tag_gate; p->fault = c->fault_dev;
}

// the solver of the launch's first frame `frame` (tracker frame index) reports to the polling host thread (agt_track_host_frame)
static void arm_host_seq(agt_ctx* c, AgtPnpParams* p, long frame)
{
    if (!c->host_seq_on) return;
    p->host_seq = c->hseq_dev;
    p->host_seq_base = c->hseq_off + (unsigned long long)frame;
}

static int fill_lk(const agt_ctx* c, AgtLkParams* p, int prev_slot, int next_slot, const float* d_prev, const uint8_t* d_prev_status, float* d_next,
                   uint8_t* d_status, float* d_err, int n, int crit_type, int crit_max_count, double crit_eps,
                   int flags, double min_eig_threshold)
{
    memset(p, 0, sizeof(*p));
    fill_levels(c, prev_slot, p->prev);
    fill_levels(c, next_slot, p->next);
    p->max_level = c->eff_max_level;
    p->n = n;
    // SparsePyrLKOpticalFlowImpl::calc criteria normalisation
    p->max_count = (crit_type & AGT_TERM_COUNT) ? (crit_max_count < 0 ? 0 : crit_max_count > 100 ? 100 : crit_max_count) : 30;
    double eps = (crit_type & AGT_TERM_EPS) ? (crit_eps < 0. ? 0. : crit_eps > 10. ? 10. : crit_eps) : 0.01;
    p->eps2 = eps * eps;
    p->flags = flags;
    p->min_eig_threshold = min_eig_threshold;
    p->prev_pts = d_prev; p->prev_status = d_prev_status; p->next_pts = d_next; p->status = d_status; p->err = d_err;
    p->lds_min = lk_lds_min(c->lk_cap_cu > 0 ? c->lk_cap_cu : 0);
    return AGT_OK;
}

// One fused launch: every pipeline stage advances by up to `group` frames whose input is complete.
// fmax: the most frames a stage advances by in this call (0 = c->group).  The split pipeline (big batches: three launches per group)
// runs groups of c->group frames in the steady state but SMALLER ones while it fills and drains (round 6, tools/trace_blocks.py: of a
// 256-frame block of 64 streams 8 % was the first pyramid launch with nothing beside it and the last pose launch with nothing beside it --
// a stage's launch covers a whole group, and the next stage waits for all of it): split_ramp() frames at first, doubling per launch.
static int split_ramp(const agt_ctx* c) { const int r = c->group / 4; return r < 2 ? (c->group < 2 ? c->group : 2) : r; }
static int launch_group(agt_ctx* c, int B, int fmax = 0)
{
    const int L = c->eff_max_level;              // pyramid stages 0..L-1 (stage s: level s -> s+1)
    const int F = (fmax > 0 && fmax < c->group) ? fmax : c->group, M = c->live_ring;
    AgtStepParams S;
    AgtStepTables T;
    memset(&S, 0, sizeof(S));
    memset(&T, 0, sizeof(T));
    S.pnp.fault = c->fault_dev;                  // (also where the launch has no pose role: the LK role reports a table it cannot use through it)
    long done_before[AGT_MAX_LEVELS + 1];        // [s] = frames available as input of stage s (s = L: input of LK)
    done_before[0] = c->trk_frame;
    for (int s = 0; s < L; s++) done_before[s + 1] = c->n_stage[s];
    const long lk_before = c->n_lk;
    long lk_f0 = 0;                              // LK role: the frame before its group
    bool any = false;
    // Chained launch (fused step only): the PnP role follows the LK role of the SAME launch frame by frame through arrival
    // counters (agt_step.hip pnp_role), instead of one launch behind it.
    bool chain = agt_step_fits(c->trk_n, B);
#ifdef AGT_DEBUG_KNOBS      // diagnostic library only: AGT_CHAIN=0 keeps the PnP role one launch behind the LK role
    { static const int on = [] { const char* e = getenv("AGT_CHAIN"); return e ? atoi(e) : 1; }(); if (!on) chain = false; }
#endif
    // stage 0 builds levels 1 and 2 in one pass (stage 1 then only keeps the books): always while the batch is small (tiled or
    // register-rolling two-level pass), for big batches where the rolling form applies to every frame of the launch (decided below,
    // in the s = 0 trip) and stage 1 has no backlog from an earlier launch that ran the levels as two passes
    bool fused = L >= 2 && B <= AGT_PYR2_MAX_B;
    // a launch that will carry pyramid work ONLY (the first one of a run: no frame is ready for the LK role, none for the pose role)
    // goes out as the pyramid kernel proper (below), which has the register-rolling two-level pass -- the fused step kernel has not
    const bool pyr_only = c->n_lk >= done_before[L] && c->n_pnp >= c->n_lk;
    for (int s = 0; s < L; s++) {
        if (fused && s == 1) continue;
        long cnt = done_before[s] - c->n_stage[s];
        if (cnt <= 0) continue;
        if (cnt > F) cnt = F;
        AgtPyrArgs& A = S.pyr[s];
        A.sw = c->lw[s]; A.sh = c->lh[s];
        A.dw = c->lw[s + 1]; A.dh = c->lh[s + 1];
        A.dpitch = c->lpitch[s + 1]; A.dbatch = (long)c->lh[s + 1] * c->lpitch[s + 1];
        A.B = B;
        uintptr_t src_align = 0, dst_align = 0;
        for (long k = 0; k < cnt; k++) {
            const int slot = (int)((c->n_stage[s] + 1 + k) % M);
            if (s == 0) {
                // level 0 is the caller's frame: the frames of a group must share pitch and stream stride
                if (k == 0) { A.spitch = c->l0_pitch[slot]; A.sbatch = c->l0_bstride[slot]; }
                else if (A.spitch != c->l0_pitch[slot] || A.sbatch != c->l0_bstride[slot]) { cnt = k; break; }
                T.pyr_src[s][k] = c->l0_ptr[slot];
            } else {
                A.spitch = c->lpitch[s]; A.sbatch = (long)c->lh[s] * c->lpitch[s];
                T.pyr_src[s][k] = c->lmem[slot][s];
            }
            T.pyr_dst[s][k] = c->lmem[slot][s + 1];
            src_align |= (uintptr_t)T.pyr_src[s][k]; dst_align |= (uintptr_t)T.pyr_dst[s][k];
        }
        A.src = T.pyr_src[s][0]; A.dst = T.pyr_dst[s][0];
        agt_pyr_plan(&A, src_align, dst_align, (int)cnt);          // tiled or register-rolling form (A.pad), workgroups per image in A.gx * A.gy
        if (s == 0 && L >= 2) {
            // the level 1 -> 2 geometry and buffers ride in stage 1's slots; the pass's grid in A: 64 x 16 tiles of level 2 (tiled
            // form) or workgroups per image with the strip height in A.pad (register-rolling form, agt_pyramid4_body.h)
            AgtPyrArgs A0 = A, A1 = S.pyr[1];
            A1.sw = c->lw[1]; A1.sh = c->lh[1]; A1.dw = c->lw[2]; A1.dh = c->lh[2];
            A1.spitch = c->lpitch[1]; A1.sbatch = (long)c->lh[1] * c->lpitch[1];
            A1.dpitch = c->lpitch[2]; A1.dbatch = (long)c->lh[2] * c->lpitch[2];
            A1.B = B; A1.src = nullptr; A1.dst = nullptr;
            uintptr_t d2_align = 0;
            for (long k = 0; k < cnt; k++) d2_align |= (uintptr_t)c->lmem[(int)((c->n_stage[0] + 1 + k) % M)][2];
            // (fused step: agt_step_fits, <= 256 corners in flight -- its kernel carries the tiled two-level pass only: plan as ONE frame, which
            // keeps the launch below the rolling form's 16 images.  Round 5: except the pyramid-only launch at the head of a run, which is
            // pyr_group_kernel -- 20 frames of 1280x720, the driver's block: 15.1 us tiled, see DESIGN.md section 6 for the rolling figure)
            agt_pyr2_plan(&A0, &A1, src_align, dst_align | d2_align, (agt_step_fits(c->trk_n, B) && !pyr_only) ? 1 : (int)cnt,
                          (agt_step_fits(c->trk_n, B) || B < AGT_SPLIT_PYR_OH_B0 || B > AGT_SPLIT_PYR_OH_B1) ? 16 : AGT_SPLIT_PYR_OH);
            if (!fused && A0.pad != 0 && c->n_stage[1] == c->n_stage[0]) fused = true;       // big batch, rolling form, no backlog
            if (fused) {
                A = A0; S.pyr[1] = A1;
                for (long k = 0; k < cnt; k++) T.pyr_dst[1][k] = c->lmem[(int)((c->n_stage[0] + 1 + k) % M)][2];
                c->n_stage[1] += cnt;            // (same frames: level 2 is complete when level 1 is)
            }
        }
        S.pyr_nf[s] = (int)cnt;
        S.n_pyr[s] = A.gx * A.gy * B * (int)cnt;
        c->n_stage[s] += cnt;
        any = true;
    }
    S.pyr_fused = fused ? 1 : 0;
    if (c->n_lk < done_before[L]) {
        long cnt = done_before[L] - c->n_lk;
        if (cnt > F) cnt = F;
        const long f0 = c->n_lk;                 // the frame before the group
        lk_f0 = f0;
        // level 0 geometry comes from the frame's own registration; a group needs it uniform
        for (long k = 1; k <= cnt; k++) {
            const int a = (int)(f0 % M), q = (int)((f0 + k) % M);
            if (c->l0_pitch[q] != c->l0_pitch[a] || c->l0_bstride[q] != c->l0_bstride[a]) { cnt = k > 1 ? k - 1 : 1; break; }
        }
        const int pslot = (int)(f0 % M), slot = (int)((f0 + 1) % M);
        fill_lk(c, &S.lk, pslot, slot, c->corners[pslot], c->status[pslot], c->corners[slot], c->status[slot], nullptr, c->trk_n,
                AGT_TERM_COUNT | AGT_TERM_EPS, c->lk_max_count, c->lk_eps, 0, c->lk_min_eig);
        S.lk_nf = (int)cnt;
        for (long k = 0; k <= cnt; k++) {
            const int q = (int)((f0 + k) % M);
            for (int l = 0; l <= L; l++) T.lk.img[k][l] = l == 0 ? c->l0_ptr[q] : c->lmem[q][l];
            if (k) {
                T.lk.next[k - 1] = c->corners[q]; T.lk.status[k - 1] = c->status[q];
                (void)chain;          // (arrival counters are attached below, to the frames the PnP role of THIS launch waits for)
            }
        }
        S.n_lk = 1; S.lk_B = B;
        c->n_lk += cnt;
        any = true;
    }
    // frames the PnP role may take: everything tracked before this launch; chained, also the frames tracked BY it, waiting
    // for each one's counter -- all of them when nothing more is registered (a drain: the last frames' latency counts), all
    // but the newest while frames keep coming (the role then never stalls: the LK role works one frame ahead of it)
    long pnp_avail = lk_before;
    if (chain) pnp_avail = c->n_lk - (c->n_lk < c->trk_frame && c->n_lk > lk_before ? 1 : 0);
    if (c->n_pnp < pnp_avail) {
        long cnt = pnp_avail - c->n_pnp;
        if (cnt > F) cnt = F;
        const int slot = (int)((c->n_pnp + 1) % M);
        fill_estimate(c, &S.pnp, c->corners[slot], c->status[slot], c->so_ring[slot], nullptr);
        arm_host_seq(c, &S.pnp, c->n_pnp + 1);
        for (long k = 0; k < cnt; k++) {
            const long f = c->n_pnp + 1 + k;
            const int q = (int)(f % M);
            T.pnp.img[k] = c->corners[q]; T.pnp.mask[k] = c->status[q]; T.pnp.so[k] = c->so_ring[q];
            if (f > lk_before) {
                // tracked by this launch: the LK role counts the frame's corners in (device-scope stores + acknowledgement, only
                // for the frames somebody waits for), the PnP wave waits for the count
                c->lk_target[q] += (unsigned)c->trk_n;
                T.lk.done[f - lk_f0 - 1] = c->lk_done + (size_t)q * c->cfg.max_streams;
                T.pnp.wait[k] = T.lk.done[f - lk_f0 - 1]; T.pnp.target[k] = c->lk_target[q];
            }
        }
        S.pnp_nf = (int)cnt;
        S.n_pnp = B;
        c->n_pnp += cnt;
        any = true;
#ifdef AGT_DEBUG_KNOBS      // diagnostic library only: AGT_CHAIN_WITHHOLD=k makes the k-th launch with chained waits expect one arrival more than
        // its LK role delivers for the SECOND waited frame (the first if there is only one): tests/test_gpu_tracker.py drives the give-up path with it
        { static const int wh = [] { const char* e = getenv("AGT_CHAIN_WITHHOLD"); return e ? atoi(e) : 0; }();
          static int seen = 0;
          int waited[2] = { -1, -1 }, nw = 0;
          for (long k = 0; k < cnt && nw < 2; k++) if (T.pnp.wait[k]) waited[nw++] = (int)k;
          if (wh > 0 && nw > 0 && ++seen == wh) T.pnp.target[waited[nw - 1]] += 1; }
#endif
    }
    if (!any) return AGT_OK;
#ifdef AGT_DEBUG_KNOBS      // diagnostic library only: AGT_TABLE_POISON_PNP=k / AGT_TABLE_POISON_LK=k put a word that cannot be an address (the bits of a
    // NaN) into the second frame's entry of the k-th launch whose pose / LK role runs more than one frame -- what a stale or clobbered LDS
    // table would hand the role.  tests/test_gpu_tracker.py: the launch must drain, flag and report, not fault.
    { static const int pk = [] { const char* e = getenv("AGT_TABLE_POISON_PNP"); return e ? atoi(e) : 0; }();
      static const int lk = [] { const char* e = getenv("AGT_TABLE_POISON_LK"); return e ? atoi(e) : 0; }();
      static int seen_p = 0, seen_l = 0;
      if (pk > 0 && S.pnp_nf > 1 && ++seen_p == pk) T.pnp.img[1] = (const float*)0x7ff8dead00000000ull;
      if (lk > 0 && S.lk_nf > 1 && ++seen_l == lk) T.lk.img[2][0] = (const uint8_t*)0x7ff8dead00000000ull; }
#endif
    if (agt_step_fits(c->trk_n, B)) {
#ifdef AGT_DEBUG_KNOBS      // diagnostic library only: AGT_PYR_SEPARATE=1 issues the pyramid role as its own launch, ahead of LK | PnP
        { static const int sep = [] { const char* e = getenv("AGT_PYR_SEPARATE"); return e ? atoi(e) : 0; }();
          if (sep) {
              hipError_t e = agt_launch_step(c->stream, S, T, c->cfg.win, AGT_STEP_PYR);
              if (e == hipSuccess && (S.n_lk > 0 || S.n_pnp > 0)) e = agt_launch_step(c->stream, S, T, c->cfg.win, AGT_STEP_LK | AGT_STEP_PNP);
              return e == hipSuccess ? AGT_OK : hip_fail(c, e);
          } }
#endif
        // a launch with pyramid work only (the first one of a run) goes out as the pyramid kernel proper: eight workgroups per
        // CU instead of the one the fused kernel's register budget leaves (16 frames of 720p: ~7 us instead of ~12)
        const int roles = (S.n_lk > 0 || S.n_pnp > 0) ? AGT_STEP_ALL : AGT_STEP_PYR;
        hipError_t e = agt_launch_step(c->stream, S, T, c->cfg.win, roles);
        return e == hipSuccess ? AGT_OK : hip_fail(c, e);
    }
    // ---- split mode: the same group as three launches, one per role, each with its own LDS size and register budget:
    // pyramid on the caller's stream (the frames come from there), LK and PnP on two library streams.  The roles of one
    // group are independent (the pipeline skews them across frames), so the three launches overlap; ordering is by events:
    //   LK(k)  waits for pyramid(k-1)   (its images)          and for PnP(k-2)  (ring reuse of corner / status entries)
    //   PnP(k) waits for LK(k-1)        (its corners)
    //   pyramid(k) waits for LK(k-3)    (ring reuse of level entries: split mode keeps AGT_SPLIT_SLACK = 2 groups of extra
    //                                    ring entries, so the HBM-bound pyramid launches run AHEAD of the LK launches they share
    //                                    the chip with and never sit on the critical path; the wait also keeps the caller's
    //                                    stream behind the readers of frames handed over (L + 6) groups ago, so stream-ordered
    //                                    allocators may recycle those)
    // Ten HIP calls per GROUP of F frames (round 1 issued thirteen per frame).
    int rc = ms_init(c);
    if (rc) return rc;
    hipStream_t sL = c->ms_stream[1], sL2 = c->ms_stream[0], sY = c->ms_stream[2];
    hipEvent_t *evP = c->ms_ev[0], *evL = c->ms_ev[1], *evY = c->ms_ev[3], *evL2 = c->ms_ev[4];
    // The per-frame LK launches of a group depend on each other (a corner starts where it ended), leave ~4 us between one
    // kernel's end and the next one's start on their stream, and each lasts as long as its slowest corner.  A batch of the
    // one-wave-per-corner kernel whose halves still hold >= 512 corners goes out as two launches per frame on two streams,
    // streams [0, B1) and [B1, B), both with the one-wave kernel: each half's gaps and tails are filled by the other half's
    // kernel.  Every wait on / record of the LK role below is done for both streams.  (Round 3: the halves used to be taken
    // only from 44 streams of 48 corners on; 22 .. 43 streams gain 12-20 % -- 42 streams 43.9 -> 36.6 us per step.  Four
    // quarters on four streams lose badly -- 64 streams 160 us, 86 with GPU_MAX_HW_QUEUES=8 against 49.7: more streams than
    // hardware queues, and four kernels of <= 1 wave per SIMD each stretch one another.)
    const int B1 = (!agt_lk_wide(c->trk_n, B) && (long)c->trk_n * (B / 2) >= 512) ? B / 2 : B;
    const bool two = B1 < B;
    const int slot_ev = (int)(c->split_seq % AGT_EV_SLOTS);
    bool p_work = false;
    for (int s = 0; s < AGT_MAX_LEVELS - 1; s++) p_work |= S.n_pyr[s] > 0;
    const int p_before = c->last_p_ev, l_before = c->l_ev_hist[0];
    hipError_t e = hipSuccess;
    if (p_work) {
        if (c->l_ev_hist[AGT_SPLIT_SLACK] >= 0) {
            e = hipStreamWaitEvent(c->stream, evL[c->l_ev_hist[AGT_SPLIT_SLACK]], 0);
            if (e == hipSuccess) e = hipStreamWaitEvent(c->stream, evL2[c->l_ev_hist[AGT_SPLIT_SLACK]], 0);
        }
        if (e == hipSuccess) e = agt_launch_step(c->stream, S, T, c->cfg.win, AGT_STEP_PYR);
        if (e == hipSuccess) e = hipEventRecord(evP[slot_ev], c->stream);
        if (e != hipSuccess) return hip_fail(c, e);
        c->last_p_ev = slot_ev;
    }
    if (S.n_lk > 0) {
        hipEvent_t first = nullptr;
        if (p_before >= 0) first = evP[p_before];
        else {
            // no pyramid launch since the last join (single-level pyramids, or the first LK of a run): order the LK streams
            // behind the caller's stream explicitly (reset's corner copy, earlier modes' pyramids)
            e = hipEventRecord(c->ms_ev[2][2], c->stream);
            first = c->ms_ev[2][2];
        }
        for (int h = 0; h < (two ? 2 : 1) && e == hipSuccess; h++) {
            hipStream_t sh = h ? sL2 : sL;
            e = hipStreamWaitEvent(sh, first, 0);
            if (e == hipSuccess && c->y_ev_hist[1] >= 0) e = hipStreamWaitEvent(sh, evY[c->y_ev_hist[1]], 0);
        }
        if (e != hipSuccess) return hip_fail(c, e);
        // Up to 1024 corners in flight (the four-waves-per-corner body): the LK role of the WHOLE group as one launch --
        // lk_group_kernel, the frame-chained body of the fused step with its own register budget (153 VGPRs, three workgroups
        // per CU) -- instead of one ~17 us launch + ~5 us gap per frame (round 3: 16 streams 33.9 -> see profiles/r03_stream_sweep.txt).
        bool lk_group = agt_lk_wide(c->trk_n, B);
#ifdef AGT_DEBUG_KNOBS
        { static const int on = [] { const char* e = getenv("AGT_SPLIT_LK_GROUP"); return e ? atoi(e) : 1; }(); if (!on) lk_group = false; if (on == 2) lk_group = true; }   // (2: also for the one-wave kernel)
#endif
        if (lk_group) {
            e = agt_launch_step(sL, S, T, c->cfg.win, AGT_STEP_LK);
            if (e != hipSuccess) return hip_fail(c, e);
        }
        // otherwise one stand-alone LK launch per frame of the group (and half), back to back on its stream (as a role with an in-kernel
        // frame loop the one-wave-per-corner kernel needs 240 B of scratch per lane at its 128-register budget and runs at half speed)
        for (int k = 1; k <= S.lk_nf && !lk_group; k++) {
            const int ps = (int)((lk_f0 + k - 1) % M), sl = (int)((lk_f0 + k) % M);
            // (waves = 1: a half is sized by the whole batch's kernel choice, not by its own corner count)
            // (hybrid: only the one-wave launches of big batches -- a batch that the four-wave kernel takes whole has nothing to gain)
            const bool hyb = !agt_lk_wide(c->trk_n, B);
            rc = lk_track_on(c, sL, ps, sl, c->corners[ps], c->status[ps], c->corners[sl], c->status[sl], nullptr, c->trk_n, B1,
                             AGT_TERM_COUNT | AGT_TERM_EPS, c->lk_max_count, c->lk_eps, 0, c->lk_min_eig, 0, two ? 1 : 0, hyb);
            if (rc == AGT_OK && two)
                rc = lk_track_on(c, sL2, ps, sl, c->corners[ps], c->status[ps], c->corners[sl], c->status[sl], nullptr, c->trk_n, B - B1,
                                 AGT_TERM_COUNT | AGT_TERM_EPS, c->lk_max_count, c->lk_eps, 0, c->lk_min_eig, B1, 1, hyb);
            if (rc) return rc;
        }
        e = hipEventRecord(evL[slot_ev], sL);
        // (recorded on the second stream in any case: an idle stream's event is complete at once, the waits stay uniform)
        if (e == hipSuccess) e = hipEventRecord(evL2[slot_ev], two ? sL2 : sL);
        if (e != hipSuccess) return hip_fail(c, e);
        c->l_ev_hist[2] = c->l_ev_hist[1]; c->l_ev_hist[1] = c->l_ev_hist[0]; c->l_ev_hist[0] = slot_ev;
        c->ms_active = 1;
    }
    if (S.n_pnp > 0) {
        if (l_before >= 0) {
            e = hipStreamWaitEvent(sY, evL[l_before], 0);
            if (e == hipSuccess) e = hipStreamWaitEvent(sY, evL2[l_before], 0);
        }
        if (e == hipSuccess) e = agt_launch_step(sY, S, T, c->cfg.win, AGT_STEP_PNP);
        if (e == hipSuccess) e = hipEventRecord(evY[slot_ev], sY);
        if (e != hipSuccess) return hip_fail(c, e);
        c->y_ev_hist[1] = c->y_ev_hist[0]; c->y_ev_hist[0] = slot_ev;
        c->ms_active = 1;
    }
    c->split_seq++;
    return AGT_OK;
}

// The part of registering frame T+1 that comes before the registration itself: ring modulus of the mode, a free ring entry.
static int pipelined_entry(agt_ctx* c, int B)
{
    c->prebuilt_t = -1;
    // ring modulus of this mode: (L + 2) groups, plus the split mode's slack when the entries are available
    {
        const bool split = !agt_step_fits(c->trk_n, B);
        int groups = c->eff_max_level + 2 + (split ? AGT_SPLIT_SLACK : 0);
        if (groups * c->group > AGT_RING_MAX) groups = c->eff_max_level + 2;
        const int want = groups * c->group > AGT_SLOTS ? groups * c->group : AGT_SLOTS;
        if (want != c->live_ring) {
            int rc = join_pipeline(c);            // only the newest frame is live afterwards
            if (rc) return rc;
            rc = ensure_ring(c, want);
            if (rc) return rc;
            ring_move(c, c->trk_frame, c->live_ring, want);
            c->live_ring = want;
        }
    }
    const long t = c->trk_frame + 1;
    // The ring entry frame t takes must be free: its previous tenant, frame t - M, has to be past its pose solve and past the
    // LK of the frame after it (which reads it as the previous image).  The ring size follows from every stage advancing F
    // frames per launch; groups cut short (frames of differing pitch) let stages fall behind that rhythm.
    {
        const long M = c->live_ring;
        while (c->n_pnp < t - M || c->n_lk < t - M + 1) {
            const long before = c->n_pnp + c->n_lk + c->n_stage[0];
            int rc = launch_group(c, B);
            if (rc) return rc;
            if (c->n_pnp + c->n_lk + c->n_stage[0] == before) return AGT_ERR_STATE;      // (cannot happen: work was pending)
        }
    }
    return AGT_OK;
}

// Register frame T+1 of the fused pipeline; a launch goes out once `group` frames wait for their first stage.
static int step_pipelined(agt_ctx* c, const uint8_t* d_frames, size_t pitch, size_t batch_stride, int B, double* d_state_out)
{
    int rc = pipelined_entry(c, B);
    if (rc) return rc;
    const long t = c->trk_frame + 1;
    const int slot = (int)(t % c->live_ring);
    c->l0_ptr[slot] = d_frames; c->l0_pitch[slot] = (long)pitch; c->l0_bstride[slot] = (long)batch_stride;
    c->built_B[slot] = B;
    c->so_ring[slot] = d_state_out;
    c->trk_frame = t;
    const long first_done = c->eff_max_level > 0 ? c->n_stage[0] : c->n_lk;
    if (agt_step_fits(c->trk_n, B)) {
        if (t - first_done < c->group) return AGT_OK;
        return launch_group(c, B);
    }
    // split pipeline: groups grow from split_ramp() to c->group frames while the pipeline fills (see launch_group)
    if (c->ramp <= 0 || c->ramp > c->group) c->ramp = split_ramp(c);
    if (t - first_done < c->ramp) return AGT_OK;
    rc = launch_group(c, B, c->ramp);
    c->ramp = c->ramp * 2 > c->group ? c->group : c->ramp * 2;
    return rc;
}

// Frame T+1 of ONE stream arrives in pinned HOST memory (h_dev: its device address): upload and pyramid in one launch -- the
// two-level register-rolling pass reads the frame over PCIe, stores its level 0 to d_gray and levels 1 / 2 to the frame's ring
// entry (agt_pyramid.hip agt_launch_pyr_upload2) -- and the frame is registered with its pyramid stages done; the LK | PnP launch
// follows at the join as ever.  Returns 1 when the form does not apply (geometry, pyramid work of earlier frames still pending):
// nothing has been enqueued for the frame then and the caller copies and calls agt_track_frame.
static int step_pipelined_uploaded(agt_ctx* c, const uint8_t* h_dev, uint8_t* d_gray, size_t gpitch, double* d_state_out)
{
    if (c->eff_max_level != 2 || c->trk_B != 1) return 1;
    int rc = pipelined_entry(c, 1);
    if (rc) return rc;
    const long t = c->trk_frame + 1;
    if (c->n_stage[0] != t - 1 || c->n_stage[1] != t - 1) return 1;
    const int slot = (int)(t % c->live_ring);
    const int W = c->cfg.width, H = c->cfg.height;
    hipError_t e = agt_launch_pyr_upload2(c->stream, h_dev, W, H, (long)W, d_gray, (long)gpitch, c->lmem[slot][1], c->lpitch[1], c->lmem[slot][2], c->lpitch[2]);
    if (e == hipErrorInvalidValue) { (void)hipGetLastError(); return 1; }
    if (e != hipSuccess) return hip_fail(c, e);
    c->l0_ptr[slot] = d_gray; c->l0_pitch[slot] = (long)gpitch; c->l0_bstride[slot] = (long)gpitch * H;
    c->built_B[slot] = 1;
    c->so_ring[slot] = d_state_out;
    c->trk_frame = t;
    c->n_stage[0] = c->n_stage[1] = t;
    return AGT_OK;
}

// Drain the software pipeline: enqueue the remaining stages of every frame supplied so far.
// (Enqueue only; the results are ordered on the context's stream like any other work.)
static int join_pipeline(agt_ctx* c)
{
    if (c->trk_ready != 2) return AGT_OK;
    // (split pipeline: the drain runs in groups of split_ramp() frames -- the last pose launch then trails the last LK launch by a
    // quarter of a group instead of a whole one; afterwards the pipeline is empty and fills again in small groups)
    const int fmax = agt_step_fits(c->trk_n, c->trk_B) ? 0 : split_ramp(c);
    while (c->n_pnp < c->trk_frame) {
        int rc = launch_group(c, c->trk_B, fmax);
        if (rc) return rc;
    }
    c->ramp = 0;
    return c->ms_active ? ms_join(c) : AGT_OK;
}

int agt_tracker_join(agt_ctx* c)
{
    if (!c) return AGT_ERR_ARG;
    int rc = join_pipeline(c);
    // (enqueue only: a give-up of a chained wait is reported once the device has written the word -- by the next
    // agt_synchronize at the latest; the flagged records say which frames)
    if (rc == AGT_OK && *(volatile int*)c->fault_host) rc = AGT_ERR_CHAIN;
    return rc;
}

int agt_estimate_pose(agt_ctx* c, const float* d_img, const uint8_t* d_mask, int B, double* d_state_out)
{
    if (!c || !d_img) return AGT_ERR_ARG;
    if (!c->trk_ready) return AGT_ERR_STATE;
    if (B <= 0 || B > c->trk_B) return AGT_ERR_ARG;
    int rc = join_pipeline(c);            // state updates of frames in flight come first
    if (rc) return rc;
    AgtPnpParams p;
    fill_estimate(c, &p, d_img, d_mask, d_state_out, nullptr);
    hipError_t e = agt_launch_pnp(c->stream, p, B);
    return e == hipSuccess ? AGT_OK : hip_fail(c, e);
}

// Big batches: pyramid, LK and PnP of one frame as separate kernels on three library-owned streams.  The host runs
// ahead, so the pyramid of frame t+1 overlaps the LK of frame t and the PnP of frame t-1; every true dependency
// (and every buffer reuse of the rings) is an event wait, nothing is assumed about timing.
// The library's streams are a PROCESS resource (round 6).  A process has four hardware queues and the runtime deals
// every new stream onto them round-robin; it does not hand a destroyed stream's place to the next one.  Contexts that each created and
// destroyed their own three streams therefore left the next context's streams wherever the deal happened to stand -- two of them on
// one queue sooner or later, and the split pipeline, which needs its pyramid / LK / LK / pose launches on four different queues, ran
// 1.5-1.8 x slower from then on.  Now a set of three highest-priority streams per device is created once, lent
// to the context that runs a split pipeline and taken back at agt_destroy; a second live context on the device gets a second set.
namespace {
struct MsPoolEntry { int device; hipStream_t s[3]; bool busy; };
std::vector<MsPoolEntry> g_ms_pool;
std::mutex g_ms_pool_mutex;
}
static int ms_pool_acquire(int device, hipStream_t out[3])
{
    std::lock_guard<std::mutex> lock(g_ms_pool_mutex);
    for (size_t i = 0; i < g_ms_pool.size(); i++)
        if (!g_ms_pool[i].busy && g_ms_pool[i].device == device) {
            g_ms_pool[i].busy = true;
            for (int k = 0; k < 3; k++) out[k] = g_ms_pool[i].s[k];
            return (int)i;
        }
    // the LK launches are the long pole of a split-mode step and share the chip with the pyramid launches of the caller's stream, whose
    // short workgroups otherwise take every wave slot first: the library's streams get the highest priority
    MsPoolEntry e; e.device = device; e.busy = true;
    // (a stream belongs to the device that is current when it is created: the context's, whatever the calling thread has selected)
    int cur = device;
    if (hipGetDevice(&cur) != hipSuccess) return -1;
    if (cur != device && hipSetDevice(device) != hipSuccess) return -1;
    int pr_lo = 0, pr_hi = 0;
    (void)hipDeviceGetStreamPriorityRange(&pr_lo, &pr_hi);
#ifdef AGT_DEBUG_KNOBS      // AGT_MS_PRIO=low|mid: the library's streams at the lowest / the default priority instead of the highest (A/B)
    { const char* e = getenv("AGT_MS_PRIO"); if (e && e[0] == 'l') pr_hi = pr_lo; else if (e && e[0] == 'm') pr_hi = 0; }
#endif
    bool ok = true;
    for (int k = 0; k < 3 && ok; k++) {
        if (hipStreamCreateWithPriority(&e.s[k], hipStreamNonBlocking, pr_hi) != hipSuccess) {
            for (int q = 0; q < k; q++) (void)hipStreamDestroy(e.s[q]);
            ok = false;
        } else out[k] = e.s[k];
    }
    if (cur != device) (void)hipSetDevice(cur);
    if (!ok) return -1;
    g_ms_pool.push_back(e);
    return (int)g_ms_pool.size() - 1;
}
static void ms_pool_release(int slot)
{
    std::lock_guard<std::mutex> lock(g_ms_pool_mutex);
    if (slot >= 0 && slot < (int)g_ms_pool.size()) g_ms_pool[slot].busy = false;
}

static int ms_init(agt_ctx* c)
{
    if (c->ms_ready) return AGT_OK;
    if (c->ms_pool_slot < 0) c->ms_pool_slot = ms_pool_acquire(c->cfg.device, c->ms_stream);        // (a retry after a failed event creation keeps its slot)
    if (c->ms_pool_slot < 0) return hip_fail(c, hipGetLastError());
    for (int k = 0; k < 5; k++)
        for (int i = 0; i < AGT_EV_SLOTS; i++)
            if (!c->ms_ev[k][i] && hipEventCreateWithFlags(&c->ms_ev[k][i], hipEventDisableTiming) != hipSuccess) { c->ms_ev[k][i] = nullptr; return hip_fail(c, hipGetLastError()); }
    c->ms_ready = 1;
    return AGT_OK;
}

// the caller's stream waits for everything in flight on the library's streams (the LK and PnP launches of split mode)
static int ms_join(agt_ctx* c)
{
    if (!c->ms_active) return AGT_OK;
    hipEvent_t ev1 = c->ms_ev[2][0], ev2 = c->ms_ev[2][1], ev0 = c->ms_ev[2][3];
    hipError_t e = hipEventRecord(ev1, c->ms_stream[1]);
    if (e == hipSuccess) e = hipStreamWaitEvent(c->stream, ev1, 0);
    if (e == hipSuccess) e = hipEventRecord(ev0, c->ms_stream[0]);
    if (e == hipSuccess) e = hipStreamWaitEvent(c->stream, ev0, 0);
    if (e == hipSuccess) e = hipEventRecord(ev2, c->ms_stream[2]);
    if (e == hipSuccess) e = hipStreamWaitEvent(c->stream, ev2, 0);
    if (e != hipSuccess) return hip_fail(c, e);
    c->ms_active = 0;
    c->last_p_ev = -1; c->l_ev_hist[0] = c->l_ev_hist[1] = c->l_ev_hist[2] = -1; c->y_ev_hist[0] = c->y_ev_hist[1] = -1;     // everything issued so far is ordered before the caller's next work
    return AGT_OK;
}

// Dense clips (step_serial below): which of the round-4 forms are on.
static int dense_defer_on()
{
#ifdef AGT_DEBUG_KNOBS      // diagnostic library only: AGT_DENSE_DEFER=0 keeps dense_final_kernel a launch of its own in clips, 1 keeps LK and the pose solve separate launches
    static const int on = [] { const char* e = getenv("AGT_DENSE_DEFER"); return e ? atoi(e) : 2; }();
    return on;
#else
    return 2;
#endif
}

// pyrDown.. -> LK -> PnP (+ dense stage) as launches on the context's stream: pose complete in stream order.
// next_frame != null (clip submission with the dense stage): the caller will hand that frame in next, with the same pitch and
// stream stride -- its two-level pyramid pass rides in one of this frame's launches (the chained LK | PnP launch, the four-wave PnP
// launch, or the second dense launch) instead of being the first launch of its own chain, and this frame's last dense step is left
// to that frame's LK launch (c->dense_pending).
static int step_serial(agt_ctx* c, const uint8_t* d_frames, size_t pitch, size_t batch_stride, int B,
                       double* d_state_out, double* d_dense_out, hipEvent_t* pev, const uint8_t* next_frame = nullptr)
{
    int rc = join_pipeline(c);            // a mode switch drains the pipeline first
    if (rc) return rc;
    const long t = c->trk_frame + 1;
    const int slot = (int)(t % c->live_ring), pslot = (int)((t - 1) % c->live_ring);
    hipStream_t M = c->stream;
    if (pev) (void)hipEventRecord(pev[0], M);
    const bool prebuilt = c->prebuilt_t == t && c->l0_ptr[slot] == d_frames && c->l0_pitch[slot] == (long)pitch &&
                          c->l0_bstride[slot] == (long)batch_stride && c->built_B[slot] == B;
    c->prebuilt_t = -1;
    rc = prebuilt ? AGT_OK : pyramid_build_on(c, M, slot, d_frames, pitch, batch_stride, B);
    if (rc) return rc;
    if (pev) (void)hipEventRecord(pev[1], M);
    // Clip submission: the next frame's two-level pyramid pass rides in one of this frame's launches -- the PnP launch where that
    // is the four-wave kernel (n > 64: one workgroup per stream, the chip idles beside it), else the dense stage's second launch.
    // (Measured on configs[4], us per frame: no ride 89.0; in the PnP launch 84.0-84.5; in the dense launch 84.0-85.3; in the LK
    // launch 84.5-84.8 as tiles behind the LK role launch's own workgroups, 86.5 as the pyramid role of the fused step kernel.)
    AgtPyrArgs npyr[2];
    const int nslot = (int)((t + 1) % c->live_ring);
    const bool lk_role_launch = c->cfg.win == 21 && agt_lk_wide(c->trk_n, B) && c->l0_pitch[pslot] == (long)pitch && c->l0_bstride[pslot] == (long)batch_stride;
    const bool ride = next_frame && c->eff_max_level == 2 && B <= AGT_PYR2_MAX_B && nslot != slot && nslot < c->ring &&
                      ((uintptr_t)next_frame & 3) == 0 && (agt_pnp_can_ride(c->trk_n) || (d_dense_out && c->dn_iters > 0));
    // Dense clips with the cooperative solver (64 < n <= 256): LK and PnP of the frame in ONE launch, the solver waiting for the
    // frame's arrival count (agt_step.hip lk_pnp_coop_kernel); the next frame's pyramid pass then rides in that launch as well
    // (while trackers + solvers are co-resident at the one workgroup per CU the solver's registers leave: more streams keep the LK
    // launch of its own, whose 78 registers put several workgroups on a CU)
    // (the LK role's grid is rounded up to a multiple of 8; the device's CU count, not a literal: ADVICE r4)
    const bool chain_pnp = lk_role_launch && d_dense_out && c->dn_iters > 0 && !pev && agt_pnp_can_ride(c->trk_n) && dense_defer_on() > 1 &&
                           (long)agt_xcd_grid((long)c->trk_n * B, 3) + B <= c->chip.cus;
    const bool ride_pnp = ride && agt_pnp_can_ride(c->trk_n) && !chain_pnp;
    if (ride) {
        // (what pyramid_build_on registers for a frame, for frame t + 1 in its ring entry)
        c->l0_ptr[nslot] = next_frame; c->l0_pitch[nslot] = (long)pitch; c->l0_bstride[nslot] = (long)batch_stride;
        const long db1 = (long)c->lh[1] * c->lpitch[1], db2 = (long)c->lh[2] * c->lpitch[2];
        agt_pyr2_args(next_frame, c->lw[0], c->lh[0], (long)pitch, (long)batch_stride, c->lmem[nslot][1], c->lpitch[1], db1,
                      c->lmem[nslot][2], c->lpitch[2], db2, B, &npyr[0], &npyr[1]);
        c->built_B[nslot] = B;
        c->prebuilt_t = t + 1;
    }
    AgtPnpParams p;
    fill_estimate(c, &p, c->corners[slot], c->status[slot], d_state_out, c->corners[slot], c->status[slot]);
    arm_host_seq(c, &p, t);
    if (d_dense_out) {
        rc = dense_scratch(c, agt_dense_doubles(c->dn_M, B), B);
        if (rc) return rc;
        p.dense_pose = c->pose; p.dense_done = c->dense_done; p.dense_rec = d_dense_out;
    }
    if (lk_role_launch) {
        // four waves per corner: the LK role of the step as a one-frame group (the frame-chained body; see agt_step.hip lk_role)
        AgtStepParams S; AgtStepTables T;
        memset(&S, 0, sizeof(S)); memset(&T, 0, sizeof(T));
        fill_lk(c, &S.lk, pslot, slot, c->corners[pslot], c->status[pslot], c->corners[slot], c->status[slot], nullptr, c->trk_n,
                AGT_TERM_COUNT | AGT_TERM_EPS, c->lk_max_count, c->lk_eps, 0, c->lk_min_eig);
        S.lk_nf = 1; S.n_lk = 1; S.lk_B = B;
        for (int k = 0; k < 2; k++) {
            const int q = k ? slot : pslot;
            for (int l = 0; l <= c->eff_max_level; l++) T.lk.img[k][l] = l == 0 ? c->l0_ptr[q] : c->lmem[q][l];
        }
        T.lk.next[0] = c->corners[slot]; T.lk.status[0] = c->status[slot];
        hipError_t e;
        if (chain_pnp) {
            S.pnp = p; S.pnp_nf = 1; S.n_pnp = B;
            // (the slot's arrival target advances only once the launch is known to be out: ADVICE r4 -- a failed launch must leave
            // the counters' bookkeeping and the deferred dense step as they were)
            T.lk.done[0] = c->lk_done + (size_t)slot * c->cfg.max_streams;
            T.pnp.img[0] = c->corners[slot]; T.pnp.mask[0] = c->status[slot]; T.pnp.so[0] = d_state_out;
            T.pnp.wait[0] = T.lk.done[0]; T.pnp.target[0] = c->lk_target[slot] + (unsigned)c->trk_n;
#ifdef AGT_DEBUG_KNOBS      // diagnostic library only: AGT_CHAIN_WITHHOLD_DENSE=k makes the k-th chained LK | PnP launch of the dense tracker expect one
            // arrival more than its LK role delivers (tests/test_dense.py drives the solver's give-up path with it)
            { static const int wh = [] { const char* e = getenv("AGT_CHAIN_WITHHOLD_DENSE"); return e ? atoi(e) : 0; }();
              static int seen = 0;
              if (wh > 0 && ++seen == wh) T.pnp.target[0] += 1; }
#endif
        }
        if (c->dense_pending || chain_pnp) {
            e = agt_launch_lk_reseed(M, S, T, c->cfg.win, c->dense_pending ? &c->dense_final : nullptr, (ride && chain_pnp) ? npyr : nullptr);
            if (e == hipSuccess) {                  // on failure dense_pending stays set: the caller's error path flushes it (agt_launch_dense_final)
                c->dense_pending = 0;
                if (chain_pnp) c->lk_target[slot] += (unsigned)c->trk_n;
            }
        } else e = agt_launch_step(M, S, T, c->cfg.win, AGT_STEP_LK);
        if (e != hipSuccess) return hip_fail(c, e);
    } else {
    if (c->dense_pending) {          // (cannot happen: the deferral is decided with this frame's launch form known)
        c->dense_pending = 0;
        hipError_t e = agt_launch_dense_final(M, c->dense_final, B);
        if (e != hipSuccess) return hip_fail(c, e);
    }
    rc = lk_track_on(c, M, pslot, slot, c->corners[pslot], c->status[pslot], c->corners[slot], c->status[slot], nullptr,
                     c->trk_n, B, AGT_TERM_COUNT | AGT_TERM_EPS, c->lk_max_count, c->lk_eps, 0, c->lk_min_eig);
    if (rc) return rc;
    }
    if (pev) (void)hipEventRecord(pev[2], M);
    hipError_t e = chain_pnp ? hipSuccess : agt_launch_pnp(M, p, B, ride_pnp ? npyr : nullptr);
    if (e != hipSuccess) return hip_fail(c, e);
    if (pev) { (void)hipEventRecord(pev[3], M); c->prof_n++; }
    if (d_dense_out) {
        // Clip submission: the stage's last step (final update + re-seed, a one-workgroup launch of 5.5 us) is left to the NEXT
        // frame's LK launch, whose workgroups do it as their prologue (agt_step.hip lk_reseed_kernel) -- when that launch will be the
        // four-waves-per-corner LK role launch (same geometry as this frame's) and no stage spans are being recorded
        const bool defer = next_frame && !pev && c->dn_iters > 0 && c->cfg.win == 21 && agt_lk_wide(c->trk_n, B) && dense_defer_on();
        // dense photometric + geometric refinement of the frame's accepted pose, then (reseed) the corner set from it
        e = agt_launch_dense(M, d_frames, (long)pitch, (long)batch_stride, c->cfg.width, c->cfg.height, c->dn_xyz, c->dn_t, c->dn_M,
                             c->obj, c->corners[slot], c->status[slot], c->trk_n, c->cam, c->pose, c->dense_partials, nullptr,
                             c->dense_done, B, c->dn_iters, c->dn_weight, 1e-3, d_dense_out, c->dn_reseed ? c->corners[slot] : nullptr,
                             c->dn_reseed ? c->status[slot] : nullptr, pev ? pev + 4 : nullptr, 2 * AGT_PROF_DENSE_MAX, (ride && !ride_pnp && !chain_pnp) ? npyr : nullptr,
                             defer ? &c->dense_final : nullptr);
        if (e == hipSuccess && defer) c->dense_pending = 1;
        if (pev) c->prof_dense[c->prof_n - 1] = c->dn_iters < AGT_PROF_DENSE_MAX ? c->dn_iters : AGT_PROF_DENSE_MAX;
        if (e != hipSuccess) return hip_fail(c, e);
    }
    c->trk_frame = t; c->n_lk = c->n_pnp = t;
    for (int s = 0; s < AGT_MAX_LEVELS; s++) c->n_stage[s] = t;
    return AGT_OK;
}

// One frame for B streams.
//   pipeline on  (default, needs reproject == 0): ONE fused launch; frame t's pose is produced
//                 L+1 launches later (or by agt_tracker_join / agt_synchronize).
//   pipeline off : pyrDown.. -> LK -> PnP as separate launches, pose complete in stream order.
int agt_track_frame(agt_ctx* c, const uint8_t* d_frames, size_t pitch, size_t batch_stride, int B,
                    double* d_state_out)
{
    if (!c || !d_frames) return AGT_ERR_ARG;
    if (c->trk_ready != 2) return AGT_ERR_STATE;
    if (B <= 0 || B != c->trk_B) return AGT_ERR_ARG;
    if ((pitch & 3) || ((uintptr_t)d_frames & 3) || (batch_stride & 3) || pitch < (size_t)c->cfg.width) return AGT_ERR_ARG;
    hipEvent_t* pev = (c->prof_ev && c->prof_n < c->prof_cap) ? c->prof_ev + (size_t)c->prof_n * AGT_PROF_EVENTS : nullptr;
    // the fused launch pays off while the stages are latency-bound (few streams); the biggest batches fill
    // the chip per stage and run faster as separate launches with their own register budgets
    if (c->pipeline && !c->reproject && (agt_step_fits(c->trk_n, B) || !pev)) {
        // fused launch: the three spans collapse into one (span 2 = the whole step_kernel launch)
        if (pev) { (void)hipEventRecord(pev[0], c->stream); (void)hipEventRecord(pev[1], c->stream); (void)hipEventRecord(pev[2], c->stream); }
        int rc = step_pipelined(c, d_frames, pitch, batch_stride, B, d_state_out);
        if (pev && rc == AGT_OK) { (void)hipEventRecord(pev[3], c->stream); c->prof_n++; }
        return rc;
    }

    return step_serial(c, d_frames, pitch, batch_stride, B, d_state_out, nullptr, pev);
}

int agt_track_frames(agt_ctx* c, const uint8_t* d_frames, size_t pitch, size_t batch_stride, size_t frame_stride, int B, int count,
                     double* d_state_out)
{
    if (count < 0 || (frame_stride & 3)) return AGT_ERR_ARG;
    for (int k = 0; k < count; k++) {
        int rc = agt_track_frame(c, d_frames + (size_t)k * frame_stride, pitch, batch_stride, B,
                                 d_state_out ? d_state_out + (size_t)k * B * AGT_STATE_STRIDE : nullptr);
        if (rc) return rc;
    }
    return AGT_OK;
}

// A frame whose corners come from the detector (detect_pose.py:389-437 found >= 2 tags, so the reference does not track):
// the frame joins the stream as frame t (its pyramid is built: the next agt_track_frame tracks FROM it), the supplied table
// becomes the frame's corner set and LK status, and _estimate_pose runs on it -- pyramid pass + one PnP launch, in stream order.
int agt_track_frame_detected(agt_ctx* c, const uint8_t* d_frames, size_t pitch, size_t batch_stride, int B,
                             const float* d_corners, const uint8_t* d_mask, double* d_state_out)
{
    if (!c || !d_frames || !d_corners) return AGT_ERR_ARG;
    if (!c->trk_ready) return AGT_ERR_STATE;
    if (B <= 0 || B != c->trk_B) return AGT_ERR_ARG;
    if ((pitch & 3) || ((uintptr_t)d_frames & 3) || (batch_stride & 3) || pitch < (size_t)c->cfg.width) return AGT_ERR_ARG;
    int rc = join_pipeline(c);
    if (rc) return rc;
    c->prebuilt_t = -1;
    const long t = c->trk_frame + 1;
    const int slot = (int)(t % c->live_ring);
    rc = pyramid_build_on(c, c->stream, slot, d_frames, pitch, batch_stride, B);
    if (rc) return rc;
    AgtPnpParams p;
    fill_estimate(c, &p, d_corners, d_mask, d_state_out, c->corners[slot], c->status[slot]);
    p.seed_pts = c->corners[slot]; p.seed_status = c->status[slot];
    hipError_t e = agt_launch_pnp(c->stream, p, B);
    if (e != hipSuccess) return hip_fail(c, e);
    c->trk_frame = t; c->n_lk = c->n_pnp = t;
    for (int s = 0; s < AGT_MAX_LEVELS; s++) c->n_stage[s] = t;
    c->trk_ready = 2;
    return AGT_OK;
}

// The live camera loop in ONE call (detect_pose.py:669-681, LK path): host frame -> HBM -> [undistort / gray / crop] -> agt_track_frame
// -> join -> record back -> wait.  Everything a Python caller would otherwise issue as five separate foreign calls.
int agt_track_host_frame(agt_ctx* c, const uint8_t* h_frame, int channels, int src_w, int src_h, uint8_t* d_staging,
                         int undistort, int roi_x, int roi_y, uint8_t* d_gray, size_t gpitch, double* d_state, double* h_state)
{
    if (!c || !h_frame || !d_gray || !h_state || (channels != 1 && channels != 3)) return AGT_ERR_ARG;
    if (c->trk_ready != 2 || c->trk_B != 1) return AGT_ERR_STATE;
    const int W = c->cfg.width, H = c->cfg.height;
    hipError_t e;
#ifdef AGT_DEBUG_KNOBS      // diagnostic library only: AGT_HOST_FRAME_TIMES=1 prints where the host time of this call goes (us, mean) at exit
    struct Acc { double t[6]; long n; ~Acc() { if (n) fprintf(stderr, "agt_track_host_frame host us: upload-call %.2f track-call %.2f join %.2f d2h-call %.2f sync %.2f total %.2f (n=%ld)\n",
                                                              t[0] / n, t[1] / n, t[2] / n, t[3] / n, t[4] / n, t[5] / n, n); } };
    static Acc acc = {};
    static const int timing = [] { const char* v = getenv("AGT_HOST_FRAME_TIMES"); return v ? atoi(v) : 0; }();
    auto now = [] { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec * 1e6 + ts.tv_nsec * 1e-3; };
    double tq[6] = { 0, 0, 0, 0, 0, 0 };
    if (timing) tq[0] = now();
#define AGT_TQ(i) do { if (timing) tq[i] = now(); } while (0)
#else
#define AGT_TQ(i)
#endif
    // Polled record: the frame's record goes straight to host-mapped memory of the context and the solver stores a sequence word behind
    // it (system scope); this thread polls the word -- no 128-byte copy (a blit KERNEL of ~4.5 us behind the step's launch,
    // rocprofv3), no completion signal, no stream wait.  (With profiling spans armed the call waits for the stream as before.)
    bool polled = !(c->prof_ev && c->prof_n < c->prof_cap);
#ifdef AGT_DEBUG_KNOBS
    { static const int on = [] { const char* v = getenv("AGT_HOST_POLL"); return v ? atoi(v) : 1; }(); if (!on) polled = false; }
#endif
    double* const d_rec = (polled || !d_state) ? c->hrec_dev : d_state;       // where the pose solver writes the frame's record
    bool registered = false;
    // (the device address of the caller's buffer when it is pinned host memory, asked for on EVERY call: a cached answer would outlive the buffer)
    const uint8_t* h_dev = nullptr;
    {
        hipPointerAttribute_t at;
        if (hipPointerGetAttributes(&at, h_frame) == hipSuccess && at.type == hipMemoryTypeHost) h_dev = (const uint8_t*)at.devicePointer;
        else (void)hipGetLastError();
    }
    if (channels == 1) {
        if (src_w != W || src_h != H || roi_x || roi_y || undistort) return AGT_ERR_ARG;
        // Round 4: a gray frame in PINNED host memory is not copied first: the two-level pyramid pass reads it over PCIe and writes its
        // level 0 to d_gray on the way (step_pipelined_uploaded) -- one launch instead of a copy-engine transfer (19 us + ~8 us of
        // submission and hand-over) and a pyramid launch (6 us).  Pageable memory, frame sizes the rolling pass does not take, the
        // stage-by-stage mode and pending pyramid work of earlier frames keep the copy.
        if (c->pipeline && !c->reproject && agt_step_fits(c->trk_n, 1) && !(c->prof_ev && c->prof_n < c->prof_cap)) {
            if (h_dev) {
                int rcu = step_pipelined_uploaded(c, h_dev, d_gray, gpitch, d_rec);
                if (rcu < 0) return rcu;
                registered = rcu == AGT_OK;
            }
        }
        if (!registered) {
            if (gpitch == (size_t)W) e = hipMemcpyAsync(d_gray, h_frame, (size_t)W * H, hipMemcpyHostToDevice, c->stream);
            else e = hipMemcpy2DAsync(d_gray, gpitch, h_frame, (size_t)W, (size_t)W, (size_t)H, hipMemcpyHostToDevice, c->stream);
            if (e != hipSuccess) return hip_fail(c, e);
        }
    } else {
        // BGR: without undistortion the gray conversion streams through the frame once -- from pinned host memory it reads the frame
        // over PCIe itself (no 2.8 MB copy first); the undistortion's gather keeps the frame in HBM (d_staging)
        const uint8_t* src = d_staging;
        if (h_dev && !undistort) src = h_dev;
        else {
            if (!d_staging) return AGT_ERR_ARG;
            e = hipMemcpyAsync(d_staging, h_frame, (size_t)src_w * src_h * 3, hipMemcpyHostToDevice, c->stream);
            if (e != hipSuccess) return hip_fail(c, e);
        }
        int rc = agt_preprocess_bgr(c, src, (size_t)src_w * 3, (size_t)src_w * src_h * 3, src_w, src_h, 1, undistort, roi_x, roi_y, W, H,
                                    d_gray, gpitch, gpitch * (size_t)H);
        if (rc) return rc;
    }
    AGT_TQ(1);
    int rc = AGT_OK;
    unsigned long long want = 0;
    if (polled) c->host_seq_on = 1;            // (every pose launch issued from here on reports the frames it solves)
    if (!registered) rc = agt_track_frame(c, d_gray, gpitch, gpitch * (size_t)H, 1, d_rec);
    if (polled && rc == AGT_OK) { want = c->hseq_off + (unsigned long long)c->trk_frame; c->hseq_last = want; }
    if (rc == AGT_OK) { AGT_TQ(2); rc = join_pipeline(c); }
    c->host_seq_on = 0;
    if (rc) return rc;
    AGT_TQ(3);
    if (polled) {
        if (d_state) {          // the device copy of the record, for callers that keep one (stream-ordered; nobody waits for it here)
            e = hipMemcpyAsync(d_state, c->hrec_dev, AGT_STATE_STRIDE * sizeof(double), hipMemcpyDeviceToDevice, c->stream);
            if (e != hipSuccess) return hip_fail(c, e);
        }
        AGT_TQ(4);
        const volatile unsigned long long* seq = c->hseq_host;
        timespec t0; clock_gettime(CLOCK_MONOTONIC, &t0);
        for (unsigned long spins = 1; *seq < want; spins++) {
#if defined(__x86_64__) || defined(__i386__)
            __builtin_ia32_pause();
#else
            __asm__ __volatile__("" ::: "memory");       // (other hosts: a compiler barrier; the volatile load above is the poll)
#endif
            if ((spins & 0xffff) == 0) {
                // nothing after 2 s: a launch failed or the device is gone -- let the runtime say which
                timespec t1; clock_gettime(CLOCK_MONOTONIC, &t1);
                if ((t1.tv_sec - t0.tv_sec) + (t1.tv_nsec - t0.tv_nsec) * 1e-9 > 2.0) {
                    e = hipStreamSynchronize(c->stream);
                    if (e != hipSuccess) return hip_fail(c, e);
                    if (*seq < want) return AGT_ERR_STATE;
                }
            }
        }
        __atomic_thread_fence(__ATOMIC_ACQUIRE);
        memcpy(h_state, c->hrec_host, AGT_STATE_STRIDE * sizeof(double));
#ifdef AGT_DEBUG_KNOBS
        if (timing) { const double t5 = now(); for (int i = 0; i < 4; i++) acc.t[i] += tq[i + 1] - tq[i]; acc.t[4] += t5 - tq[4]; acc.t[5] += t5 - tq[0]; acc.n++; }
#endif
        return *(volatile int*)c->fault_host ? AGT_ERR_CHAIN : AGT_OK;
    }
    // (the waited form: profiling spans armed, or the knobs build's AGT_HOST_POLL=0.  HSA_ENABLE_SDMA=0 -- blit-kernel copies -- made
    // the copy-first form of round 3 slower still: 107 us.)
    if (d_state) {
        e = hipMemcpyAsync(h_state, d_state, AGT_STATE_STRIDE * sizeof(double), hipMemcpyDeviceToHost, c->stream);
        AGT_TQ(4);
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        if (e != hipSuccess) return hip_fail(c, e);
    } else {                                   // (no device record asked for: the record is in the context's host-mapped memory once the stream is done)
        AGT_TQ(4);
        e = hipStreamSynchronize(c->stream);
        if (e != hipSuccess) return hip_fail(c, e);
        memcpy(h_state, c->hrec_host, AGT_STATE_STRIDE * sizeof(double));
    }
#ifdef AGT_DEBUG_KNOBS
    if (timing) { const double t5 = now(); for (int i = 0; i < 4; i++) acc.t[i] += tq[i + 1] - tq[i]; acc.t[4] += t5 - tq[4]; acc.t[5] += t5 - tq[0]; acc.n++; }
#endif
    return *(volatile int*)c->fault_host ? AGT_ERR_CHAIN : AGT_OK;
}

// detect_pose.py:570-574 / the mirror's `if ids:`: a frame in which LK lost EVERY tag does not become the "previous" frame --
// the next frame is tracked from the frame before it, with that frame's corner set and status.  The host sees the record
// (AGT_ST_NTRACK == 0) and takes the frame back: its ring entry is re-used by the next frame, entry t - 1 (never written since)
// is the LK source again.  The pose state machine keeps what the lost frame did to it (guess cleared), as in the reference.
int agt_tracker_rewind(agt_ctx* c)
{
    if (!c) return AGT_ERR_ARG;
    if (c->trk_ready != 2 || c->trk_frame < 1) return AGT_ERR_STATE;
    int rc = join_pipeline(c);
    if (rc) return rc;
    c->prebuilt_t = -1;
    c->trk_frame -= 1;
    c->hseq_off += 1;                         // (the next frame re-uses the frame index: its sequence number must not)
    c->n_lk = c->n_pnp = c->trk_frame;
    for (int s = 0; s < AGT_MAX_LEVELS; s++)
        if (c->n_stage[s] > c->trk_frame) c->n_stage[s] = c->trk_frame;
    return AGT_OK;
}

int agt_tracker_state_size(void) { return (int)sizeof(AgtTrackState); }

int agt_tracker_state_read(agt_ctx* c, void* host_dst, int B)
{
    if (!c || !host_dst || !c->trk_ready || B <= 0 || B > c->trk_B) return AGT_ERR_ARG;
    int rc = join_pipeline(c);
    if (rc) return rc;
    hipError_t e = hipMemcpyAsync(host_dst, c->tstate, (size_t)B * sizeof(AgtTrackState), hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    return e == hipSuccess ? AGT_OK : hip_fail(c, e);
}

// valid after agt_tracker_join: the newest frame's corner set and LK status
int agt_tracker_buffers(const agt_ctx* c, const float** d_corners, const uint8_t** d_status)
{
    if (!c || !c->trk_ready) return AGT_ERR_STATE;
    if (d_corners) *d_corners = c->corners[c->trk_frame % c->live_ring];
    if (d_status) *d_status = c->status[c->trk_frame % c->live_ring];
    return AGT_OK;
}

// ---------------------------------------------------------------------------------------------
// frame pre-processing: detect_pose.py:147-183 undistort_frame, :602 cvtColor

namespace {

// cvUndistortPointsInternal for one point, R = I, optional new camera matrix P, 5 fixed iterations
void undistort_point_host(double u, double v, const double* K, const AgtCameraHost& cam, const AgtTiltHost& tilt, bool has_dist, const double* P,
                          double* ox, double* oy)
{
    const double* k = cam.k;
    const double ifx = 1. / K[0], ify = 1. / K[4], cx = K[2], cy = K[5];
    double x = (u - cx) * ifx, y = (v - cy) * ify;
    if (has_dist) {
        if (tilt.on) {          // compensate tilt distortion: invMatTilt (x, y, 1), dehomogenised
            const double* M = tilt.m + 9;
            double t[3];
            for (int r = 0; r < 3; r++) { double a = 0; a += M[r * 3] * x; a += M[r * 3 + 1] * y; a += M[r * 3 + 2] * 1; t[r] = a; }
            const double ip = t[2] ? 1. / t[2] : 1;
            x = ip * t[0]; y = ip * t[1];
        }
        const double x0 = x, y0 = y;
        for (int it = 0; it < 5; it++) {
            const double r2 = x * x + y * y;
            const double icdist = (1 + ((k[7] * r2 + k[6]) * r2 + k[5]) * r2) / (1 + ((k[4] * r2 + k[1]) * r2 + k[0]) * r2);
            if (icdist < 0) { x = (u - cx) * ifx; y = (v - cy) * ify; break; }
            const double dX = 2 * k[2] * x * y + k[3] * (r2 + 2 * x * x) + k[8] * r2 + k[9] * r2 * r2;
            const double dY = k[2] * (r2 + 2 * y * y) + 2 * k[3] * x * y + k[10] * r2 + k[11] * r2 * r2;
            x = (x0 - dX) * icdist;
            y = (y0 - dY) * icdist;
        }
    }
    if (P) {
        const double xx = P[0] * x + P[1] * y + P[2], yy = P[3] * x + P[4] * y + P[5], ww = 1. / (P[6] * x + P[7] * y + P[8]);
        x = xx * ww; y = yy * ww;
    }
    *ox = x; *oy = y;
}

// calibration.cpp icvGetRectangles: inscribed / circumscribed rectangles of the undistorted 9x9 grid (float)
struct RectF { float x, y, w, h; };
void grid_rectangles(const double* K, const AgtCameraHost& cam, const AgtTiltHost& tilt, bool has_dist, const double* P, int w, int h, RectF* inner, RectF* outer)
{
    const int N = 9;
    float in_l = -FLT_MAX, in_r = FLT_MAX, in_t = -FLT_MAX, in_b = FLT_MAX;
    float out_l = FLT_MAX, out_r = -FLT_MAX, out_t = FLT_MAX, out_b = -FLT_MAX;
    for (int gy = 0; gy < N; gy++)
        for (int gx = 0; gx < N; gx++) {
            const float px = (float)gx * w / (N - 1), py = (float)gy * h / (N - 1);
            double ux, uy;
            undistort_point_host((double)px, (double)py, K, cam, tilt, has_dist, P, &ux, &uy);
            const float qx = (float)ux, qy = (float)uy;
            out_l = qx < out_l ? qx : out_l; out_r = qx > out_r ? qx : out_r;
            out_t = qy < out_t ? qy : out_t; out_b = qy > out_b ? qy : out_b;
            if (gx == 0 && qx > in_l) in_l = qx;
            if (gx == N - 1 && qx < in_r) in_r = qx;
            if (gy == 0 && qy > in_t) in_t = qy;
            if (gy == N - 1 && qy < in_b) in_b = qy;
        }
    *inner = RectF{ in_l, in_t, in_r - in_l, in_b - in_t };
    *outer = RectF{ out_l, out_t, out_r - out_l, out_b - out_t };
}

bool invert3(const double* A, double* B)
{
    const double c00 = A[4] * A[8] - A[5] * A[7], c01 = A[5] * A[6] - A[3] * A[8], c02 = A[3] * A[7] - A[4] * A[6];
    const double det = A[0] * c00 + A[1] * c01 + A[2] * c02;
    if (det == 0) return false;
    const double id = 1. / det;
    B[0] = c00 * id; B[1] = (A[2] * A[7] - A[1] * A[8]) * id; B[2] = (A[1] * A[5] - A[2] * A[4]) * id;
    B[3] = c01 * id; B[4] = (A[0] * A[8] - A[2] * A[6]) * id; B[5] = (A[2] * A[3] - A[0] * A[5]) * id;
    B[6] = c02 * id; B[7] = (A[1] * A[6] - A[0] * A[7]) * id; B[8] = (A[0] * A[4] - A[1] * A[3]) * id;
    return true;
}

}  // namespace

// cv::getOptimalNewCameraMatrix(K, dist, (w,h), alpha, (new_w,new_h), centerPrincipalPoint=false); host only
int agt_get_optimal_new_camera_matrix(const double* K, const double* dist, int ndist, int w, int h, double alpha,
                                      int new_w, int new_h, double* newK, int* roi)
{
    if (!K || !newK || w <= 0 || h <= 0) return AGT_ERR_ARG;
    AgtCameraHost cam;
    AgtTiltHost tilt;
    int rc = fill_camera(K, dist, ndist, &cam, &tilt);
    if (rc) return rc;
    const bool has_dist = dist != nullptr && ndist > 0;
    if ((long)new_w * new_h == 0) { new_w = w; new_h = h; }
    alpha = alpha < 0. ? 0. : alpha > 1. ? 1. : alpha;
    RectF inner, outer;
    grid_rectangles(K, cam, tilt, has_dist, nullptr, w, h, &inner, &outer);
    const double fx0 = (new_w - 1) / inner.w, fy0 = (new_h - 1) / inner.h;     // int / float, as OpenCV
    const double cx0 = -fx0 * inner.x, cy0 = -fy0 * inner.y;
    const double fx1 = (new_w - 1) / outer.w, fy1 = (new_h - 1) / outer.h;
    const double cx1 = -fx1 * outer.x, cy1 = -fy1 * outer.y;
    for (int i = 0; i < 9; i++) newK[i] = K[i];
    newK[0] = fx0 * (1 - alpha) + fx1 * alpha;
    newK[4] = fy0 * (1 - alpha) + fy1 * alpha;
    newK[2] = cx0 * (1 - alpha) + cx1 * alpha;
    newK[5] = cy0 * (1 - alpha) + cy1 * alpha;
    if (roi) {
        grid_rectangles(K, cam, tilt, has_dist, newK, w, h, &inner, &outer);
        const int rx = (int)lrintf(inner.x), ry = (int)lrintf(inner.y), rw = (int)lrintf(inner.w), rh = (int)lrintf(inner.h);
        const int x1 = rx > 0 ? rx : 0, y1 = ry > 0 ? ry : 0;
        const int x2 = rx + rw < new_w ? rx + rw : new_w, y2 = ry + rh < new_h ? ry + rh : new_h;
        if (x2 <= x1 || y2 <= y1) roi[0] = roi[1] = roi[2] = roi[3] = 0;
        else { roi[0] = x1; roi[1] = y1; roi[2] = x2 - x1; roi[3] = y2 - y1; }
    }
    return AGT_OK;
}

// cv::initUndistortRectifyMap(K, dist, I, newK, (w,h), CV_16SC2) into context-owned device maps
int agt_undistort_init(agt_ctx* c, const double* K, const double* dist, int ndist, const double* newK, int w, int h)
{
    if (!c || !K || w <= 0 || h <= 0 || w > 32767 || h > 32767) return AGT_ERR_ARG;
    AgtCameraHost cam;
    AgtTiltHost tilt;
    int rc = fill_camera(K, dist, ndist, &cam, &tilt);
    if (rc) return rc;
    double ir[9];
    if (!invert3(newK ? newK : K, ir)) return AGT_ERR_ARG;
    if (c->map_w != w || c->map_h != h) {
        hipError_t e = hipStreamSynchronize(c->stream);
        if (e != hipSuccess) return hip_fail(c, e);
        if (c->map1) (void)hipFree(c->map1);
        if (c->map2) (void)hipFree(c->map2);
        c->map1 = nullptr; c->map2 = nullptr; c->map_w = c->map_h = 0;
        if (hipMalloc((void**)&c->map1, (size_t)w * h * sizeof(short2)) != hipSuccess ||
            hipMalloc((void**)&c->map2, (size_t)w * h * sizeof(unsigned short)) != hipSuccess) {
            if (c->map1) { (void)hipFree(c->map1); c->map1 = nullptr; }
            return AGT_ERR_ALLOC;
        }
        c->map_w = w; c->map_h = h;
    }
    hipError_t e = agt_launch_undistort_map(c->stream, K, cam, tilt, ir, w, h, c->map1, c->map2);
    return e == hipSuccess ? AGT_OK : hip_fail(c, e);
}

int agt_undistort_maps(const agt_ctx* c, const int16_t** d_map1, const uint16_t** d_map2, int* w, int* h)
{
    if (!c || !c->map1) return AGT_ERR_STATE;
    if (d_map1) *d_map1 = reinterpret_cast<const int16_t*>(c->map1);
    if (d_map2) *d_map2 = c->map2;
    if (w) *w = c->map_w;
    if (h) *h = c->map_h;
    return AGT_OK;
}

// cv::undistort on B BGR frames of the map size (same-size output)
int agt_undistort_bgr(agt_ctx* c, const uint8_t* d_src, size_t spitch, size_t sbatch,
                      uint8_t* d_dst, size_t dpitch, size_t dbatch, int B)
{
    if (!c || !d_src || !d_dst || B <= 0) return AGT_ERR_ARG;
    if (!c->map1) return AGT_ERR_STATE;
    if (spitch < (size_t)c->map_w * 3 || dpitch < (size_t)c->map_w * 3) return AGT_ERR_ARG;
    hipError_t e = agt_launch_preprocess(c->stream, d_src, (long)spitch, (long)sbatch, c->map_w, c->map_h, c->map1, c->map2, c->map_w,
                                         0, 0, c->map_w, c->map_h, d_dst, (long)dpitch, (long)dbatch, 1, 0, B);
    return e == hipSuccess ? AGT_OK : hip_fail(c, e);
}

// undistort (optional) + BGR2GRAY + ROI crop in one pass: BGR (src_w x src_h) -> gray (roi_w x roi_h)
int agt_preprocess_bgr(agt_ctx* c, const uint8_t* d_bgr, size_t spitch, size_t sbatch, int src_w, int src_h, int B,
                       int undistort, int roi_x, int roi_y, int roi_w, int roi_h,
                       uint8_t* d_gray, size_t gpitch, size_t gbatch)
{
    if (!c || !d_bgr || !d_gray || B <= 0 || src_w <= 0 || src_h <= 0) return AGT_ERR_ARG;
    if (roi_x < 0 || roi_y < 0 || roi_w <= 0 || roi_h <= 0 || roi_x + roi_w > src_w || roi_y + roi_h > src_h) return AGT_ERR_ARG;
    if (spitch < (size_t)src_w * 3 || gpitch < (size_t)roi_w) return AGT_ERR_ARG;
    if (undistort && (!c->map1 || c->map_w != src_w || c->map_h != src_h)) return AGT_ERR_STATE;
    hipError_t e = agt_launch_preprocess(c->stream, d_bgr, (long)spitch, (long)sbatch, src_w, src_h, c->map1, c->map2, src_w,
                                         roi_x, roi_y, roi_w, roi_h, d_gray, (long)gpitch, (long)gbatch, undistort ? 1 : 0, 1, B);
    return e == hipSuccess ? AGT_OK : hip_fail(c, e);
}

// dense photometric + geometric pose refinement (semantics: oracle/cv_dense.c / csrc/agt_dense.hip)
int agt_dense_refine(agt_ctx* c, const uint8_t* d_img, size_t pitch, size_t batch_stride, int w, int h,
                     const float* d_model_xyz, const float* d_model_t, int M,
                     const float* d_obj, const float* d_img_pts, const uint8_t* d_mask, int N,
                     const double* K, const double* dist, int ndist,
                     double* d_pose, int B, int iters, double photo_weight, double* d_stats)
{
    if (!c || !d_img || !d_pose || !d_stats || B <= 0 || w < 4 || h < 4 || iters < 0 || iters > 1000) return AGT_ERR_ARG;
    if (M < 0 || N < 0 || (M > 0 && (!d_model_xyz || !d_model_t)) || (N > 0 && (!d_obj || !d_img_pts))) return AGT_ERR_ARG;
    if (M + N == 0 || pitch < (size_t)w || !(photo_weight >= 0.0)) return AGT_ERR_ARG;
    AgtCameraHost cam;
    int rc = camera_on(c, K, dist, ndist, &cam);
    if (rc) return rc;
    rc = dense_scratch(c, agt_dense_doubles(M, B), B);
    if (rc) return rc;
    hipError_t e = agt_launch_dense(c->stream, d_img, (long)pitch, (long)batch_stride, w, h, d_model_xyz, d_model_t, M,
                                    d_obj, d_img_pts, d_mask, N, cam, d_pose, c->dense_partials, d_stats, c->dense_done,
                                    B, iters, photo_weight, 1e-3, nullptr, nullptr, nullptr);
    return e == hipSuccess ? AGT_OK : hip_fail(c, e);
}

// dense stage of the per-frame step (BASELINE configs[4]); semantics in include/agt_hip.h
int agt_tracker_dense(agt_ctx* c, const float* d_model_xyz, const float* d_model_t, int M, int iters, double photo_weight, int reseed)
{
    if (!c || M < 0 || iters < 0 || iters > 1000 || !(photo_weight >= 0.0)) return AGT_ERR_ARG;
    if (M > 0 && (!d_model_xyz || !d_model_t || iters == 0)) return AGT_ERR_ARG;
    int rc = join_pipeline(c);
    if (rc) return rc;
    c->dn_xyz = M ? d_model_xyz : nullptr; c->dn_t = M ? d_model_t : nullptr; c->dn_M = M;
    c->dn_iters = iters; c->dn_weight = photo_weight; c->dn_reseed = reseed ? 1 : 0;
    return AGT_OK;
}

int agt_track_frame_dense(agt_ctx* c, const uint8_t* d_frames, size_t pitch, size_t batch_stride, int B,
                          double* d_state_out, double* d_dense_out)
{
    if (!c || !d_frames || !d_dense_out) return AGT_ERR_ARG;
    if (c->trk_ready != 2 || c->dn_M <= 0) return AGT_ERR_STATE;
    if (B <= 0 || B != c->trk_B) return AGT_ERR_ARG;
    if ((pitch & 3) || ((uintptr_t)d_frames & 3) || (batch_stride & 3) || pitch < (size_t)c->cfg.width) return AGT_ERR_ARG;
    hipEvent_t* pev = (c->prof_ev && c->prof_n < c->prof_cap) ? c->prof_ev + (size_t)c->prof_n * AGT_PROF_EVENTS : nullptr;
    return step_serial(c, d_frames, pitch, batch_stride, B, d_state_out, d_dense_out, pev);
}

// `count` consecutive frames of the stream(s), frame k at d_frames + k * frame_stride: agt_track_frame_dense for each, in order.
// Knowing the next frame, the library lets its pyramid pass ride in the current frame's four-wave PnP launch or second dense launch (step_serial) --
// same records, one launch less in every frame's serial chain.
int agt_track_frames_dense(agt_ctx* c, const uint8_t* d_frames, size_t pitch, size_t batch_stride, size_t frame_stride, int B, int count,
                           double* d_state_out, double* d_dense_out)
{
    if (!c || !d_frames || !d_dense_out || count < 0 || (frame_stride & 3)) return AGT_ERR_ARG;
    if (c->trk_ready != 2 || c->dn_M <= 0) return AGT_ERR_STATE;
    if (B <= 0 || B != c->trk_B) return AGT_ERR_ARG;
    if ((pitch & 3) || ((uintptr_t)d_frames & 3) || (batch_stride & 3) || pitch < (size_t)c->cfg.width) return AGT_ERR_ARG;
    for (int k = 0; k < count; k++) {
        hipEvent_t* pev = (c->prof_ev && c->prof_n < c->prof_cap) ? c->prof_ev + (size_t)c->prof_n * AGT_PROF_EVENTS : nullptr;
        int rc = step_serial(c, d_frames + (size_t)k * frame_stride, pitch, batch_stride, B,
                             d_state_out ? d_state_out + (size_t)k * B * AGT_STATE_STRIDE : nullptr,
                             d_dense_out + (size_t)k * B * AGT_DENSE_STRIDE, pev, k + 1 < count ? d_frames + (size_t)(k + 1) * frame_stride : nullptr);
        if (rc) {
            if (c->dense_pending) { c->dense_pending = 0; (void)agt_launch_dense_final(c->stream, c->dense_final, B); }     // (a frame failed before its LK launch)
            return rc;
        }
    }
    return AGT_OK;
}

int agt_profile_begin(agt_ctx* c, int max_frames)
{
    if (!c || max_frames <= 0 || max_frames > (1 << 20) || c->prof_ev) return AGT_ERR_ARG;
    const size_t n = (size_t)max_frames * AGT_PROF_EVENTS;
    c->prof_ev = new (std::nothrow) hipEvent_t[n];
    if (!c->prof_ev) return AGT_ERR_ALLOC;
    for (size_t i = 0; i < n; i++) {
        hipError_t e = hipEventCreate(&c->prof_ev[i]);
        if (e != hipSuccess) {
            for (size_t j = 0; j < i; j++) (void)hipEventDestroy(c->prof_ev[j]);
            delete[] c->prof_ev; c->prof_ev = nullptr;
            return hip_fail(c, e);
        }
    }
    c->prof_dense = new (std::nothrow) int[max_frames];
    if (!c->prof_dense) { for (size_t j = 0; j < n; j++) (void)hipEventDestroy(c->prof_ev[j]); delete[] c->prof_ev; c->prof_ev = nullptr; return AGT_ERR_ALLOC; }
    for (int i = 0; i < max_frames; i++) c->prof_dense[i] = 0;
    c->prof_cap = max_frames; c->prof_n = 0;
    return AGT_OK;
}

int agt_profile_end(agt_ctx* c, float* ms_out, int* n_frames)
{
    if (!c || !c->prof_ev) return AGT_ERR_ARG;
    hipError_t e = hipStreamSynchronize(c->stream);
    int rc = e == hipSuccess ? AGT_OK : hip_fail(c, e);
    if (rc == AGT_OK && ms_out)
        for (int f = 0; f < c->prof_n && rc == AGT_OK; f++) {
            hipEvent_t* ev = c->prof_ev + (size_t)f * AGT_PROF_EVENTS;
            for (int k = 0; k < 3; k++) {
                e = hipEventElapsedTime(&ms_out[f * AGT_PROF_SPANS + k], ev[k], ev[k + 1]);
                if (e != hipSuccess) { rc = hip_fail(c, e); break; }
            }
            // dense stage: ev[3] = its start, ev[4] after the last Gauss-Newton launch, ev[5] after the final launch
            float acc = 0.f, upd = 0.f;
            if (c->prof_dense[f] > 0 && rc == AGT_OK) {
                e = hipEventElapsedTime(&acc, ev[3], ev[4]);
                if (e == hipSuccess) e = hipEventElapsedTime(&upd, ev[4], ev[5]);
                if (e != hipSuccess) { rc = hip_fail(c, e); break; }
            }
            ms_out[f * AGT_PROF_SPANS + 3] = acc; ms_out[f * AGT_PROF_SPANS + 4] = upd;
        }
    if (n_frames) *n_frames = c->prof_n;
    for (size_t i = 0; i < (size_t)c->prof_cap * AGT_PROF_EVENTS; i++) (void)hipEventDestroy(c->prof_ev[i]);
    delete[] c->prof_ev; c->prof_ev = nullptr; c->prof_cap = c->prof_n = 0;
    delete[] c->prof_dense; c->prof_dense = nullptr;
    return rc;
}

}  // extern "C"
