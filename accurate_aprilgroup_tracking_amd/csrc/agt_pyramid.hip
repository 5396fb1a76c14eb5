// agt_pyramid.hip -- stand-alone cv::pyrDown launch (body: agt_pyramid_body.h).
// Replaces the pyramid build inside cv.calcOpticalFlowPyrLK (north-star step).
// Algorithmic bytes per launch: sw*sh read + dw*dh written (per image).
#include "agt_pyramid_body.h"

namespace {

__global__ __launch_bounds__(agt_pyr::NT) void pyr_down_kernel(const AgtPyrArgs A)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    // XCD-aware tile order: workgroups are dealt round-robin to the 8 XCDs (each with its own L2), so
    // workgroup b takes tile (b % 8) * per_xcd + b / 8 -- every XCD walks a CONTIGUOUS row-major run of
    // tiles and the 128-B lines shared by neighbouring tiles (16-B side halos, 3 halo rows) hit in its L2
    // instead of being fetched once per XCD.
    const int per_xcd = (int)gridDim.x >> 3;
    const int t = (int)(blockIdx.x & 7) * per_xcd + (int)(blockIdx.x >> 3);
    const int per_img = A.gx * A.gy;
    if (t >= per_img * A.B) return;
    const int bz = t / per_img, r = t - bz * per_img;
    const int by = r / A.gx;
    agt_pyr::pyr_down_body(A, r - by * A.gx, by, A.src + (long)bz * A.sbatch, A.dst + (long)bz * A.dbatch, lds);
}

}  // namespace

void agt_pyr_grid(int dw, int dh, int* gx, int* gy)
{
    *gx = (dw + agt_pyr::TW - 1) / agt_pyr::TW;
    *gy = (dh + agt_pyr::TH - 1) / agt_pyr::TH;
}

hipError_t agt_launch_pyr_down(hipStream_t stream, const uint8_t* src, int sw, int sh, long spitch, long sbatch,
                               uint8_t* dst, long dpitch, long dbatch, int B)
{
    AgtPyrArgs A;
    A.src = src; A.sw = sw; A.sh = sh; A.spitch = spitch; A.sbatch = sbatch;
    A.dst = dst; A.dw = (sw + 1) / 2; A.dh = (sh + 1) / 2; A.dpitch = dpitch; A.dbatch = dbatch;
    agt_pyr_grid(A.dw, A.dh, &A.gx, &A.gy);
    A.B = B;
    const long tiles = (long)A.gx * A.gy * B;
    hipLaunchKernelGGL(pyr_down_kernel, dim3((unsigned)((tiles + 7) / 8 * 8)), dim3(agt_pyr::NT), agt_pyr::PYR_LDS_BYTES, stream, A);
    return hipGetLastError();
}
