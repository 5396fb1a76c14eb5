// agt_pyramid.hip -- stand-alone cv::pyrDown launch (body: agt_pyramid_body.h).
// Replaces the pyramid build inside cv.calcOpticalFlowPyrLK (north-star step).
// Algorithmic bytes per launch: sw*sh read + dw*dh written (per image).
#include <cstdlib>
#include "agt_pyramid2_body.h"
#include "agt_pyramid3_body.h"
#include "agt_pyramid4_body.h"

namespace {

__global__ __launch_bounds__(agt_pyr::NT) void pyr_down_kernel(const AgtPyrArgs A)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    // XCD-aware tile order (agt_kernels.h agt_xcd_order): workgroups are dealt round-robin to the X XCDs (each with its own L2), so
    // workgroup b takes tile (b % X) * per_xcd + b / X -- every XCD walks a CONTIGUOUS row-major run of
    // tiles and the 128-B lines shared by neighbouring tiles (16-B side halos, 3 halo rows) hit in its L2
    // instead of being fetched once per XCD.
    const int t = agt_xcd_order((int)blockIdx.x, (int)gridDim.x, A.xshift);
    const int per_img = A.gx * A.gy;
    if (t >= per_img * A.B) return;
    const int bz = t / per_img, r = t - bz * per_img;
    const int by = r / A.gx;
    agt_pyr::pyr_down_body(A, r - by * A.gx, by, A.src + (long)bz * A.sbatch, A.dst + (long)bz * A.dbatch, lds);
}

// register-rolling form (agt_pyramid3_body.h): A.gx = workgroups per image, A.pad = output rows per strip; same XCD-aware order
#ifndef AGT_PYR3_ATTR
#define AGT_PYR3_ATTR
#endif
__global__ __launch_bounds__(agt_pyr::NT) AGT_PYR3_ATTR void pyr_roll_kernel(const AgtPyrArgs A)
{
    const int t = agt_xcd_order((int)blockIdx.x, (int)gridDim.x, A.xshift);
    if (t >= A.gx * A.B) return;
    const int bz = t / A.gx;
    agt_pyr3::pyr_roll_body(A, t - bz * A.gx, A.src + (long)bz * A.sbatch, A.dst + (long)bz * A.dbatch);
}

// two levels per pass (agt_pyramid2_body.h): same XCD-aware tile order, tiles of 64 x 16 L2 pixels
__global__ __launch_bounds__(agt_pyr::NT) void pyr_down2_kernel(const AgtPyrArgs A0, const AgtPyrArgs A1)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const int t = agt_xcd_order((int)blockIdx.x, (int)gridDim.x, A0.xshift);
    const int per_img = A0.gx * A0.gy;
    if (t >= per_img * A0.B) return;
    const int bz = t / per_img, r = t - bz * per_img;
    const int by = r / A0.gx;
    agt_pyr2::pyr_down2_body(A0, A1, r - by * A0.gx, by, A0.src + (long)bz * A0.sbatch, A0.dst + (long)bz * A0.dbatch,
                             A1.dst + (long)bz * A1.dbatch, lds);
}

// Fused upload (agt_track_host_frame): ONE gray frame read from pinned host memory (A0.src: its device address), level 0 copied to
// HBM (`copy`), levels 1 and 2 written -- the two-level register-rolling pass with COPY (agt_pyramid4_body.h)
__global__ __launch_bounds__(agt_pyr::NT) void pyr_upload2_kernel(const AgtPyrArgs A0, const AgtPyrArgs A1, uint8_t* copy, const int cpitch)
{
    const int t = agt_xcd_order((int)blockIdx.x, (int)gridDim.x, A0.xshift);
    if (t >= A0.gx) return;
    agt_pyr4::pyr_roll2_body<true, false>(A0, A1, t, A0.src, A0.dst, A1.dst, copy, cpitch);
}

// two levels per pass, register-rolling form with alternating strip directions (agt_pyramid4_body.h): A0.gx = workgroups per image,
// A0.pad = level-2 rows per strip
__global__ __launch_bounds__(agt_pyr::NT) void pyr_roll2_kernel(const AgtPyrArgs A0, const AgtPyrArgs A1)
{
#ifdef AGT_PYR_PRIO         // experiment builds: issue priority of the HBM-bound pass beside the LK kernels' waves
    __builtin_amdgcn_s_setprio(AGT_PYR_PRIO);
#endif
    const int t = agt_xcd_order((int)blockIdx.x, (int)gridDim.x, A0.xshift);
    if (t >= A0.gx * A0.B) return;
    const int bz = t / A0.gx;
    agt_pyr4::pyr_roll2_body(A0, A1, t - bz * A0.gx, A0.src + (long)bz * A0.sbatch, A0.dst + (long)bz * A0.dbatch, A1.dst + (long)bz * A1.dbatch);
}

}  // namespace

void agt_pyr2_grid(int w2, int h2, int* gx, int* gy)
{
    *gx = (w2 + agt_pyr2::TW2 - 1) / agt_pyr2::TW2;
    *gy = (h2 + agt_pyr2::TH2 - 1) / agt_pyr2::TH2;
}
int agt_pyr2_lds_bytes(void) { return agt_pyr2::PYR2_LDS_BYTES; }

void agt_pyr2_args(const uint8_t* src, int sw, int sh, long spitch, long sbatch, uint8_t* dst1, long dpitch1, long dbatch1,
                   uint8_t* dst2, long dpitch2, long dbatch2, int B, AgtPyrArgs* pA0, AgtPyrArgs* pA1)
{
    AgtPyrArgs& A0 = *pA0; AgtPyrArgs& A1 = *pA1;
    A0.src = src; A0.sw = sw; A0.sh = sh; A0.spitch = spitch; A0.sbatch = sbatch;
    A0.dst = dst1; A0.dw = (sw + 1) / 2; A0.dh = (sh + 1) / 2; A0.dpitch = dpitch1; A0.dbatch = dbatch1;
    A1.src = dst1; A1.sw = A0.dw; A1.sh = A0.dh; A1.spitch = dpitch1; A1.sbatch = dbatch1;
    A1.dst = dst2; A1.dw = (A0.dw + 1) / 2; A1.dh = (A0.dh + 1) / 2; A1.dpitch = dpitch2; A1.dbatch = dbatch2;
    agt_pyr2_grid(A1.dw, A1.dh, &A0.gx, &A0.gy);              // the tile grid of the pass rides in A0
    A1.gx = A0.gx; A1.gy = A0.gy;
    A0.B = A1.B = B; A0.pad = A1.pad = 0;
    A0.xshift = A1.xshift = agt_chip_current().xshift; A0.rsv_ = A1.rsv_ = 0;
}

// The register-rolling form of the two-level pass where it applies: A0.pad = level-2 rows per strip, A0.gx = workgroups per image,
// A0.gy = 1 (A1 likewise); else the tiled form is left as agt_pyr2_args set it up (A0.pad = 0).  src_align / dst_align: OR of every
// source / destination address of the launch (both destination levels); frames: images per stream in the launch.
void agt_pyr2_plan(AgtPyrArgs* pA0, AgtPyrArgs* pA1, uintptr_t src_align, uintptr_t dst_align, int frames, int oh_cap)
{
    AgtPyrArgs& A0 = *pA0; AgtPyrArgs& A1 = *pA1;
    A0.pad = A1.pad = 0;
    agt_pyr2_grid(A1.dw, A1.dh, &A0.gx, &A0.gy);
    A1.gx = A0.gx; A1.gy = A0.gy;
    const bool ok = ((src_align | (uintptr_t)A0.spitch | (uintptr_t)A0.sbatch | (uintptr_t)A0.sw) & 15) == 0 &&
                    ((dst_align | (uintptr_t)A0.dpitch | (uintptr_t)A0.dbatch) & 7) == 0 && ((dst_align | (uintptr_t)A1.dpitch | (uintptr_t)A1.dbatch) & 3) == 0 &&
                    A0.sw >= 32 && A0.sh >= 32 && (long)A0.sh * A0.spitch < (1L << 31) && (long)A0.dh * A0.dpitch < (1L << 31);
    // Round 4 measured this pass and did not ship it: every strip walked top-down, its 9 halo rows came from memory a second time
    // ((4 oh2 + 9) / (4 oh2) of the image: FETCH_SIZE 76.1 MB for 59.0 MB of frames at oh2 = 8) and a lane spent ~70 VALU per level-0
    // row; 64 x 720p: 24.0-25.5 us per pass against 24.9 for two single-level passes.  Round 5 ships it for launches of >= 16 images:
    // alternating strip directions (neighbouring strips read their shared rows at the same time: FETCH_SIZE 60.4 MB) and the lean
    // horizontal / vertical passes (agt_pyramid_body.h hgroup8b / vgroup8b: ~50 VALU per row).  The pipelined cold-pair step of
    // BASELINE configs[2] (which is VALU- and HBM-bound together): 54.6-56.2 -> 48.1-49.1 us; traffic of the pass 94.5 -> 78.9 MB
    // against the 77.4 MB of SURVEY 8d (profiles/r05_experiments.md).  Smaller launches keep the tiled pass (one frame: 6.5-7.3 us
    // tiled, 5.9-8.5 rolling).
    const long images = (long)A0.B * (frames > 0 ? frames : 1);
    int want = images >= 16 ? 1 : 0;
#ifdef AGT_DEBUG_KNOBS      // AGT_PYR4=0 / 1 forces the choice, AGT_PYR4_OH=n the strip height, AGT_PYR4_REV=0 top-down strips only
    { static const int on = [] { const char* e = getenv("AGT_PYR4"); return e ? atoi(e) : -1; }(); if (on >= 0) want = on; }
    { static const int rv = [] { const char* e = getenv("AGT_PYR4_REV"); return e ? atoi(e) : 1; }(); A0.rsv_ = A1.rsv_ = rv ? 0 : 1; }
#endif
    if (!ok || !want) return;
    // strip height (level-2 rows, even): ~2 waves on each SIMD where the launch has that many units (the pass shares the chip with
    // the LK kernels of other batches; measured on 64 x 720p inside the pipelined step: oh2 = 4: 53.5 us, 6: 50.3, 8: 48.8-49.1,
    // 10: 48.1, 12: 49.7, 16: 52.8), strips of at most 16 rows (73 level-0 rows: 14 % of them the halo), at least 2 (17 rows)
    const int ncol = ((A0.sw >> 4) + agt_pyr4::TILE_GROUPS - 1) / agt_pyr4::TILE_GROUPS;
    const long want_units = 32L * agt_chip_current().cus;                 // 2 waves x 4 units x 4 SIMDs per CU
    long per_image = (want_units + images - 1) / images;
    long strips = (per_image + ncol - 1) / ncol;
    if (strips < 1) strips = 1;
    constexpr int Q = agt_pyr4::L2_PER_TRIP;
    int oh = (int)((A1.dh + strips - 1) / strips + Q - 1) / Q * Q;
    // oh_cap (round 6): 16 for a launch of its own (agt_pyramid_build, agt_pyramid_build_pair: the pass IS the step's long pole there and
    // every strip pays 9 halo rows of loads and arithmetic); AGT_SPLIT_PYR_OH = 6 (agt_api.hip) for the pyramid role of the split pipeline,
    // whose launch of up to 1,024 images runs for hundreds of microseconds BESIDE the per-frame LK launches: 64 streams x 16 frames, us per
    // step over four boxes 16: 36.7-37.6, 10: 36.0-36.2, 8: 35.7-36.3, 6: 35.1-35.9, 4: 35.4-36.3, 2: 40.6 (profiles/r06_experiments.md 17)
    // -- short-lived pyramid waves give their slots back sooner
    if (oh_cap < Q) oh_cap = Q;
    oh = oh < Q ? Q : (oh > oh_cap ? oh_cap : oh);
#ifdef AGT_DEBUG_KNOBS
    { static const int f = [] { const char* e = getenv("AGT_PYR4_OH"); return e ? atoi(e) : 0; }(); if (f > 0) oh = (f + Q - 1) / Q * Q; }
#endif
    A0.pad = A1.pad = oh;
    A0.gx = A1.gx = agt_pyr4::roll2_blocks(A0.sw, A1.dh, oh);
    A0.gy = A1.gy = 1;
}

// Fused upload + two-level pyramid of ONE frame: src = device address of the caller's gray frame in pinned host memory; copy = the
// frame's level 0 in HBM.  hipErrorInvalidValue when the geometry does not fit the rolling form (the caller then copies and builds).
hipError_t agt_launch_pyr_upload2(hipStream_t stream, const uint8_t* src, int sw, int sh, long spitch, uint8_t* copy, long cpitch,
                                  uint8_t* dst1, long dpitch1, uint8_t* dst2, long dpitch2)
{
    AgtPyrArgs A0, A1;
    agt_pyr2_args(src, sw, sh, spitch, 0, dst1, dpitch1, 0, dst2, dpitch2, 0, 1, &A0, &A1);
    const bool ok = (((uintptr_t)src | (uintptr_t)spitch | (uintptr_t)sw | (uintptr_t)copy | (uintptr_t)cpitch) & 15) == 0 &&
                    (((uintptr_t)dst1 | (uintptr_t)dpitch1) & 7) == 0 && (((uintptr_t)dst2 | (uintptr_t)dpitch2) & 3) == 0 &&
                    sw >= 32 && sh >= 32 && (long)sh * spitch < (1L << 31) && (long)sh * cpitch < (1L << 31) && (long)A0.dh * dpitch1 < (1L << 31);
    if (!ok) return hipErrorInvalidValue;
    // two level-2 rows per strip (17 level-0 rows a lane): the most units one frame gives, measured 22.9 us from pinned host memory
    // (4: 24.7, 8: 26.2); the rows a strip reads twice come out of the L2, not over PCIe again
    const int oh = agt_pyr4::L2_PER_TRIP;
    A0.pad = A1.pad = oh;
    A0.gx = A1.gx = agt_pyr4::roll2_blocks(sw, A1.dh, oh);
    A0.gy = A1.gy = 1;
    hipLaunchKernelGGL(pyr_upload2_kernel, dim3(agt_xcd_grid(A0.gx, A0.xshift)), dim3(agt_pyr::NT), 0, stream, A0, A1, copy, (int)cpitch);
    return hipGetLastError();
}

// src (sw x sh) -> dst1 ((sw+1)/2 x (sh+1)/2) -> dst2, both written, one launch
hipError_t agt_launch_pyr_down2(hipStream_t stream, const uint8_t* src, int sw, int sh, long spitch, long sbatch,
                                uint8_t* dst1, long dpitch1, long dbatch1, uint8_t* dst2, long dpitch2, long dbatch2, int B)
{
    AgtPyrArgs A0, A1;
    agt_pyr2_args(src, sw, sh, spitch, sbatch, dst1, dpitch1, dbatch1, dst2, dpitch2, dbatch2, B, &A0, &A1);
    agt_pyr2_plan(&A0, &A1, (uintptr_t)src, (uintptr_t)dst1 | (uintptr_t)dst2, 1);
    const long tiles = (long)A0.gx * A0.gy * B;
    if (A0.pad) hipLaunchKernelGGL(pyr_roll2_kernel, dim3(agt_xcd_grid(tiles, A0.xshift)), dim3(agt_pyr::NT), 0, stream, A0, A1);
    else hipLaunchKernelGGL(pyr_down2_kernel, dim3(agt_xcd_grid(tiles, A0.xshift)), dim3(agt_pyr::NT), agt_pyr2::PYR2_LDS_BYTES, stream, A0, A1);
    return hipGetLastError();
}

void agt_pyr_grid(int dw, int dh, int* gx, int* gy)
{
    *gx = (dw + agt_pyr::TW - 1) / agt_pyr::TW;
    *gy = (dh + agt_pyr::TH - 1) / agt_pyr::TH;
}

// Geometry of one pyrDown pass for A.src / A.dst (pointers, pitches, sizes and B filled in): the register-rolling form where
// it applies (A.pad = output rows per strip, A.gx = workgroups per image, A.gy = 1), else the tiled form (A.pad = 0, tile grid).
// `src_align` / `dst_align`: OR of every source / destination address the launch will see (frames of a group, batch strides).
void agt_pyr_plan(AgtPyrArgs* pA, uintptr_t src_align, uintptr_t dst_align, int frames)
{
    AgtPyrArgs& A = *pA;
    A.pad = 0;
    A.xshift = agt_chip_current().xshift; A.rsv_ = 0;
    agt_pyr_grid(A.dw, A.dh, &A.gx, &A.gy);
    const bool ok = ((src_align | (uintptr_t)A.spitch | (uintptr_t)A.sbatch | (uintptr_t)A.sw) & 15) == 0 &&
                    ((dst_align | (uintptr_t)A.dpitch | (uintptr_t)A.dbatch) & 7) == 0 &&
                    A.sw >= 32 && A.sh >= 8 && (long)A.sh * A.spitch < (1L << 31) && (long)A.dh * A.dpitch < (1L << 31);
#ifdef AGT_DEBUG_KNOBS      // diagnostic library only: AGT_PYR3=0 keeps the tiled kernel, AGT_PYR3_OH=n forces the strip height
    { static const int on = [] { const char* e = getenv("AGT_PYR3"); return e ? atoi(e) : 1; }(); if (!on) return; }
#endif
    if (!ok) return;
    // strip height: enough units (four per wave) to put ~3 waves on each of the chip's SIMDs (1024 on MI355X: 12,288 units) -- a wave keeps 8 KB of reads in
    // flight --, strips no longer than 16 output rows (measured on 64 x 720p, L0 -> L1: 8 rows 17.8 us, 16: 17.0, 24: 21.2,
    // 32: 22.1, 48: 27 -- the tiled kernel: 18.6), no shorter than 4 (3 halo rows per strip are re-read through the L2)
    const long images = (long)A.B * (frames > 0 ? frames : 1);
    const int ncol = ((A.sw >> 4) + 15) >> 4;
    const long want_units = 48L * agt_chip_current().cus;                 // 3 waves x 4 units x 4 SIMDs per CU
    long per_image = (want_units + images - 1) / images;
    long strips = (per_image + ncol - 1) / ncol;
    if (strips < 1) strips = 1;
    int oh = (int)(A.dh / strips) & ~3;
    oh = oh < 4 ? 4 : (oh > 16 ? 16 : oh);
#ifdef AGT_DEBUG_KNOBS
    { static const int f = [] { const char* e = getenv("AGT_PYR3_OH"); return e ? atoi(e) : 0; }(); if (f > 0) oh = f & ~3; }
#endif
    A.pad = oh;
    A.gx = agt_pyr3::roll_blocks(A.sw, A.dh, oh);
    A.gy = 1;
}

hipError_t agt_launch_pyr_down(hipStream_t stream, const uint8_t* src, int sw, int sh, long spitch, long sbatch,
                               uint8_t* dst, long dpitch, long dbatch, int B)
{
    AgtPyrArgs A;
    A.src = src; A.sw = sw; A.sh = sh; A.spitch = spitch; A.sbatch = sbatch;
    A.dst = dst; A.dw = (sw + 1) / 2; A.dh = (sh + 1) / 2; A.dpitch = dpitch; A.dbatch = dbatch;
    A.B = B;
    agt_pyr_plan(&A, (uintptr_t)src, (uintptr_t)dst, 1);
    const long tiles = (long)A.gx * A.gy * B;
    if (A.pad) hipLaunchKernelGGL(pyr_roll_kernel, dim3(agt_xcd_grid(tiles, A.xshift)), dim3(agt_pyr::NT), 0, stream, A);
    else hipLaunchKernelGGL(pyr_down_kernel, dim3(agt_xcd_grid(tiles, A.xshift)), dim3(agt_pyr::NT), agt_pyr::PYR_LDS_BYTES, stream, A);
    return hipGetLastError();
}
