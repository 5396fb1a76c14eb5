/* agt_cpu_twin.c -- TEST INFRASTRUCTURE (part of the CPU oracle library; never linked into or loaded by the product).
 *
 * SURVEY.md section 8b asks for "a CPU twin of each [C-ABI entry point] with identical signatures (host pointers)".  The product has
 * no CPU path by design, so the twin lives HERE, on top of the oracle's restatement of the OpenCV algorithms
 * (cv_lk.c, cv_pnp.c): for every stateless entry point of include/agt_hip.h -- context, pyrDown, pyramid slots, LK on two slots,
 * solvePnP, projectPoints -- a function agt_cpu_<name> with the SAME parameter list and the same meaning of every argument, in which
 * every "d_" pointer is a HOST pointer and the stream argument is ignored.  tests/test_cpu_twin.py checks the signatures against
 * the header text and runs one call sequence through both libraries (GPU suite: pyramid levels and LK bit-exact, poses <= 1e-9).
 *
 * Reference call sites these replace: cv.solvePnP detect_pose.py:509-526, cv.projectPoints transform_helper.py:106-111 /
 * detect_pose.py:455-461, cv.calcOpticalFlowPyrLK (north-star; hole at detect_pose.py:573-574). */
#include <stdlib.h>
#include <string.h>
#include "cv_oracle.h"
#include "../include/agt_hip.h"

#define TWIN_SLOTS 2

struct agt_cpu_ctx {
    agt_config cfg;
    int eff_max_level;
    int lw[AGT_MAX_LEVELS], lh[AGT_MAX_LEVELS];
    int B[TWIN_SLOTS];
    cvo_pyramid** pyr[TWIN_SLOTS];              /* [B] per slot */
    uint8_t* level[TWIN_SLOTS][AGT_MAX_LEVELS]; /* tight copies of the levels, [B][h][w], for agt_cpu_pyramid_level */
};
typedef struct agt_cpu_ctx agt_cpu_ctx;

int agt_cpu_version(void) { return AGT_VERSION; }

int agt_cpu_create(const agt_config* cfg, void* hip_stream, agt_cpu_ctx** out)
{
    (void)hip_stream;
    if (!cfg || !out) return AGT_ERR_ARG;
    *out = NULL;
    if (cfg->width <= 0 || cfg->height <= 0 || cfg->max_level < 0 || cfg->max_level >= AGT_MAX_LEVELS) return AGT_ERR_ARG;
    if (cfg->max_points <= 0 || cfg->max_points > 256) return AGT_ERR_NPOINTS;
    if (cfg->max_streams <= 0) return AGT_ERR_ARG;
    if (cfg->win != 21 && cfg->win != 15 && cfg->win != 31) return AGT_ERR_UNSUPPORTED;
    agt_cpu_ctx* c = (agt_cpu_ctx*)calloc(1, sizeof(*c));
    if (!c) return AGT_ERR_ALLOC;
    c->cfg = *cfg;
    int w = cfg->width, h = cfg->height;
    for (int l = 0; l <= cfg->max_level; l++) {          /* buildOpticalFlowPyramid's early stop, as agt_create */
        c->lw[l] = w; c->lh[l] = h; c->eff_max_level = l;
        w = (w + 1) / 2; h = (h + 1) / 2;
        if (w <= cfg->win || h <= cfg->win) break;
    }
    *out = c;
    return AGT_OK;
}

static void free_slot(agt_cpu_ctx* c, int s)
{
    if (c->pyr[s]) { for (int b = 0; b < c->B[s]; b++) cvo_pyramid_free(c->pyr[s][b]); free(c->pyr[s]); c->pyr[s] = NULL; }
    for (int l = 0; l < AGT_MAX_LEVELS; l++) { free(c->level[s][l]); c->level[s][l] = NULL; }
    c->B[s] = 0;
}

int agt_cpu_destroy(agt_cpu_ctx* c)
{
    if (!c) return AGT_OK;
    for (int s = 0; s < TWIN_SLOTS; s++) free_slot(c, s);
    free(c);
    return AGT_OK;
}

int agt_cpu_pyr_down_u8(agt_cpu_ctx* c, const uint8_t* d_src, int sw, int sh, size_t spitch, size_t sbatch,
                        uint8_t* d_dst, size_t dpitch, size_t dbatch, int B)
{
    if (!c || !d_src || !d_dst || sw <= 0 || sh <= 0 || B <= 0) return AGT_ERR_ARG;
    if (spitch < (size_t)sw || dpitch < (size_t)((sw + 1) / 2)) return AGT_ERR_ARG;
    for (int b = 0; b < B; b++)
        if (cvo_pyr_down_u8(d_src + (size_t)b * sbatch, sw, sh, (int)spitch, d_dst + (size_t)b * dbatch, (int)dpitch)) return AGT_ERR_ARG;
    return AGT_OK;
}

int agt_cpu_pyramid_build(agt_cpu_ctx* c, int slot, const uint8_t* d_frames, size_t pitch, size_t batch_stride, int B)
{
    if (!c || slot < 0 || slot >= TWIN_SLOTS || !d_frames || B <= 0 || B > c->cfg.max_streams) return AGT_ERR_ARG;
    if (pitch < (size_t)c->cfg.width) return AGT_ERR_ARG;
    free_slot(c, slot);
    c->pyr[slot] = (cvo_pyramid**)calloc((size_t)B, sizeof(cvo_pyramid*));
    if (!c->pyr[slot]) return AGT_ERR_ALLOC;
    c->B[slot] = B;
    for (int b = 0; b < B; b++) {
        c->pyr[slot][b] = cvo_pyramid_build(d_frames + (size_t)b * batch_stride, c->cfg.width, c->cfg.height, (int)pitch,
                                            c->cfg.win, c->cfg.win, c->cfg.max_level);
        if (!c->pyr[slot][b]) return AGT_ERR_ALLOC;
    }
    for (int l = 0; l <= c->eff_max_level; l++) {
        const size_t per = (size_t)c->lw[l] * c->lh[l];
        c->level[slot][l] = (uint8_t*)malloc(per * (size_t)B);
        if (!c->level[slot][l]) return AGT_ERR_ALLOC;
        for (int b = 0; b < B; b++)
            if (cvo_pyramid_level_copy(c->pyr[slot][b], l, c->level[slot][l] + per * (size_t)b, c->lw[l])) return AGT_ERR_STATE;
    }
    return AGT_OK;
}

int agt_cpu_pyramid_level(const agt_cpu_ctx* c, int slot, int level, const uint8_t** d_ptr,
                          int* w, int* h, size_t* pitch, size_t* batch_stride)
{
    if (!c || slot < 0 || slot >= TWIN_SLOTS || level < 0 || level > c->eff_max_level || !c->level[slot][level]) return AGT_ERR_ARG;
    if (d_ptr) *d_ptr = c->level[slot][level];
    if (w) *w = c->lw[level];
    if (h) *h = c->lh[level];
    if (pitch) *pitch = (size_t)c->lw[level];
    if (batch_stride) *batch_stride = (size_t)c->lw[level] * c->lh[level];
    return AGT_OK;
}

int agt_cpu_pyramid_max_level(const agt_cpu_ctx* c) { return c ? c->eff_max_level : AGT_ERR_ARG; }

int agt_cpu_lk_track(agt_cpu_ctx* c, int prev_slot, int next_slot,
                     const float* d_prev_pts, float* d_next_pts, uint8_t* d_status, float* d_err,
                     int n, int B, int crit_type, int crit_max_count, double crit_eps,
                     int flags, double min_eig_threshold)
{
    if (!c || prev_slot < 0 || prev_slot >= TWIN_SLOTS || next_slot < 0 || next_slot >= TWIN_SLOTS) return AGT_ERR_ARG;
    if (!d_prev_pts || !d_next_pts || !d_status || n < 0 || B <= 0) return AGT_ERR_ARG;
    if (n == 0) return AGT_OK;
    if (c->B[prev_slot] < B || c->B[next_slot] < B) return AGT_ERR_STATE;
    for (int b = 0; b < B; b++) {
        /* CVO_ACC_EXACT: the accumulation the HIP kernels are held to (deviation 1) */
        const int rc = cvo_lk_on_pyramids(c->pyr[prev_slot][b], c->pyr[next_slot][b], d_prev_pts + (size_t)b * n * 2, d_next_pts + (size_t)b * n * 2,
                                          d_status + (size_t)b * n, d_err ? d_err + (size_t)b * n : NULL, n, c->cfg.win, c->cfg.win, c->cfg.max_level,
                                          crit_type, crit_max_count, crit_eps, flags, min_eig_threshold, CVO_ACC_EXACT, 1);
        if (rc) return AGT_ERR_ARG;
    }
    return AGT_OK;
}

static double get_elem(const void* p, int dtype, size_t i) { return dtype == AGT_F64 ? ((const double*)p)[i] : (double)((const float*)p)[i]; }

int agt_cpu_solve_pnp(agt_cpu_ctx* c, const void* d_obj, size_t obj_batch_stride, const void* d_img, int dtype,
                      const uint8_t* d_mask, int n, int B,
                      const double* K, const double* dist, int ndist,
                      double* d_pose, int use_guess, int32_t* d_info, double* d_err)
{
    if (!c || !d_obj || !d_img || !K || !d_pose || B <= 0) return AGT_ERR_ARG;
    if (n < 3 || n > 256) return AGT_ERR_NPOINTS;
    if (dtype != AGT_F32 && dtype != AGT_F64) return AGT_ERR_ARG;
    if (ndist != 0 && ndist != 4 && ndist != 5 && ndist != 8 && ndist != 12 && ndist != 14) return AGT_ERR_DIST;
    double* obj = (double*)malloc(sizeof(double) * 5 * (size_t)n);
    if (!obj) return AGT_ERR_ALLOC;
    double* img = obj + 3 * (size_t)n;
    for (int b = 0; b < B; b++) {
        int m = 0;
        for (int i = 0; i < n; i++) {
            if (d_mask && !d_mask[(size_t)b * n + i]) continue;
            for (int k = 0; k < 3; k++) obj[3 * m + k] = get_elem(d_obj, dtype, (size_t)b * obj_batch_stride + 3 * (size_t)i + k);
            for (int k = 0; k < 2; k++) img[2 * m + k] = get_elem(d_img, dtype, ((size_t)b * n + i) * 2 + k);
            m++;
        }
        double* pose = d_pose + 6 * (size_t)b;
        int iters = 0, ok = 0, fl = 0;
        /* cv::solvePnP needs >= 4 points (3 with a guess); the non-planar DLT start needs 6 */
        if (m >= 4 || (use_guess && m >= 3)) {
            double r[3] = { pose[0], pose[1], pose[2] }, t[3] = { pose[3], pose[4], pose[5] };
            if (cvo_solve_pnp_iterative(obj, img, m, K, dist, ndist, r, t, use_guess, &iters) == 0) {
                ok = 1;
                for (int k = 0; k < 3; k++) { pose[k] = r[k]; pose[3 + k] = t[k]; }
                if (d_err) d_err[b] = cvo_mean_reproj_error(obj, img, m, r, t, K, dist, ndist);
            }
        } else fl |= AGT_PNP_TOO_FEW;
        if (d_info) { d_info[4 * b + AGT_INFO_OK] = ok; d_info[4 * b + AGT_INFO_ITERS] = iters; d_info[4 * b + AGT_INFO_NUSED] = m; d_info[4 * b + AGT_INFO_FLAGS] = fl; }
        if (!ok && d_err) d_err[b] = 0.0;
    }
    free(obj);
    return AGT_OK;
}

int agt_cpu_project_points(agt_cpu_ctx* c, const void* d_obj, size_t obj_batch_stride, int dtype, int n, int B,
                           const double* d_pose, const double* K, const double* dist, int ndist,
                           void* d_img_out, double* d_jac)
{
    if (!c || !d_obj || !d_pose || !K || !d_img_out || n <= 0 || B <= 0) return AGT_ERR_ARG;
    if (dtype != AGT_F32 && dtype != AGT_F64) return AGT_ERR_ARG;
    double* obj = (double*)malloc(sizeof(double) * 5 * (size_t)n);
    if (!obj) return AGT_ERR_ALLOC;
    double* out = obj + 3 * (size_t)n;
    double* jbuf = d_jac ? (double*)malloc(sizeof(double) * 12 * (size_t)n) : NULL;
    if (d_jac && !jbuf) { free(obj); return AGT_ERR_ALLOC; }
    for (int b = 0; b < B; b++) {
        for (size_t i = 0; i < 3 * (size_t)n; i++) obj[i] = get_elem(d_obj, dtype, (size_t)b * obj_batch_stride + i);
        const double* pose = d_pose + 6 * (size_t)b;
        if (cvo_project_points(obj, n, pose, pose + 3, K, dist, ndist, out, jbuf, jbuf ? jbuf + 6 * (size_t)n : NULL)) { free(obj); free(jbuf); return AGT_ERR_ARG; }
        for (size_t i = 0; i < 2 * (size_t)n; i++) {
            if (dtype == AGT_F64) ((double*)d_img_out)[(size_t)b * n * 2 + i] = out[i];
            else ((float*)d_img_out)[(size_t)b * n * 2 + i] = (float)out[i];
        }
        if (d_jac)          /* [B][2n][6]: d/dr | d/dt, as agt_project_points */
            for (size_t row = 0; row < 2 * (size_t)n; row++)
                for (int k = 0; k < 3; k++) {
                    d_jac[((size_t)b * 2 * n + row) * 6 + k] = jbuf[row * 3 + k];
                    d_jac[((size_t)b * 2 * n + row) * 6 + 3 + k] = jbuf[6 * (size_t)n + row * 3 + k];
                }
    }
    free(obj); free(jbuf);
    return AGT_OK;
}

/* twins of the synchronous host-array entry points (round 5): the same calls, already on host pointers */
int agt_cpu_solve_pnp_host(agt_cpu_ctx* c, const void* h_obj, const void* h_img, int dtype, int n,
                           const double* K, const double* dist, int ndist,
                           double* h_pose, int use_guess, int32_t* h_info, double* h_err)
{
    if (!c || !h_obj || !h_img || !h_pose) return AGT_ERR_ARG;
    if (!use_guess && n < 4) return AGT_ERR_NPOINTS;
    double pose[6] = { 0, 0, 0, 0, 0, 0 };
    if (use_guess) memcpy(pose, h_pose, sizeof(pose));
    int rc = agt_cpu_solve_pnp(c, h_obj, 0, h_img, dtype, NULL, n, 1, K, dist, ndist, pose, use_guess, h_info, h_err);
    if (rc == AGT_OK) memcpy(h_pose, pose, sizeof(pose));
    return rc;
}

int agt_cpu_project_points_host(agt_cpu_ctx* c, const void* h_obj, int dtype, int n, const double* h_pose,
                                const double* K, const double* dist, int ndist, void* h_img_out, double* h_jac)
{
    if (!c || !h_obj || !h_pose || !h_img_out) return AGT_ERR_ARG;
    if (dtype != AGT_F32 && dtype != AGT_F64) return AGT_ERR_ARG;
    if (n <= 0 || n > 256) return AGT_ERR_NPOINTS;
    return agt_cpu_project_points(c, h_obj, 0, dtype, n, 1, h_pose, K, dist, ndist, h_img_out, h_jac);
}
