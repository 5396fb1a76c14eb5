// agt_step.hip -- one heterogeneous launch per frame: software pipelining ACROSS frames.
//
// The per-frame chain pyrDown(L0->L1) -> pyrDown(L1->L2) -> LK -> PnP is a serial dependency
// inside one frame, and each link is a latency-bound kernel that fills a few percent of the
// chip.  Different links of DIFFERENT frames are independent, so one launch runs
//     pyramid stage s of frame t-s,  LK of frame t-(L-1),  PnP of frame t-L
// side by side in disjoint block ranges (no inter-block communication, no events, no second
// stream): the launch boundary is the only ordering, and a step costs max(link) instead of
// sum(links).  Results are bit-identical to the serial order -- every link reads exactly the
// buffers the serial order would give it (rings of 4 in the context).
//
// Bodies: agt_pyramid_body.h, agt_lk_body.h, agt_pnp_body.h (shared with the stand-alone kernels).
#undef AGT_LK_STAMPS
#undef AGT_PNP_STAMPS
#include "agt_pyramid_body.h"
#include "agt_lk_body.h"
#include "agt_pnp_body.h"

namespace {

constexpr int STEP_THREADS = 256;

template <int WIN, int NW, int NLEV>
__global__ __launch_bounds__(STEP_THREADS) void step_kernel(const AgtStepParams S)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    int blk = blockIdx.x;
#pragma unroll
    for (int s = 0; s < AGT_MAX_LEVELS - 1; s++) {
        if (blk < S.n_pyr[s]) {
            const AgtPyrArgs& A = S.pyr[s];
            const int per_img = A.gx * A.gy;
            const int bz = blk / per_img, r = blk - bz * per_img;
            const int by = r / A.gx, bx = r - by * A.gx;
            agt_pyr::pyr_down_body(A, bx, by, bz, lds);
            return;
        }
        blk -= S.n_pyr[s];
    }
    if (blk < S.n_lk) {
        // NW = 4: the workgroup is one corner; NW = 1: each wave is its own corner
        constexpr int CPB = STEP_THREADS / (AGT_WAVE * NW);
        const int wave = threadIdx.x / AGT_WAVE;
        const long corner = (long)blk * CPB + (NW == 1 ? wave : 0);
        if (corner >= (long)S.lk.n * S.lk_B) return;
        const int b = (int)(corner / S.lk.n), pt = (int)(corner - (long)b * S.lk.n);
        uint8_t* my = lds + (NW == 1 ? (size_t)wave * ((agt_lk::lk_lds_bytes<WIN, NW>(S.lk.max_level + 1) + 15) & ~(size_t)15) : 0);
        agt_lk::lk_body<WIN, NW, NLEV>(S.lk, pt, b, my);
        return;
    }
    blk -= S.n_lk;
    if (blk < S.n_pnp) {
        if (threadIdx.x >= AGT_WAVE) return;
        agt_pnp::PnpShared& sh = *reinterpret_cast<agt_pnp::PnpShared*>(lds);
        agt_pnp::pnp_body<float, 1>(S.pnp, blk, sh);        // fused path: n <= 64 (agt_step_supported)
    }
}

template <int WIN, int NW>
hipError_t launch_step_t(hipStream_t stream, const AgtStepParams& S)
{
    constexpr int CPB = STEP_THREADS / (AGT_WAVE * NW);
    AgtStepParams P = S;
    size_t lds = 0;
    int blocks = 0;
    for (int s = 0; s < AGT_MAX_LEVELS - 1; s++) if (P.n_pyr[s] > 0) { blocks += P.n_pyr[s]; lds = lds > (size_t)agt_pyr::PYR_LDS_BYTES ? lds : (size_t)agt_pyr::PYR_LDS_BYTES; }
    if (P.n_lk > 0) {
        const long corners = (long)P.lk.n * P.lk_B;
        P.n_lk = (int)((corners + CPB - 1) / CPB);
        blocks += P.n_lk;
        const size_t per = (agt_lk::lk_lds_bytes<WIN, NW>(P.lk.max_level + 1) + 15) & ~(size_t)15;
        const size_t need = per * (NW == 1 ? CPB : 1);
        lds = lds > need ? lds : need;
    }
    if (P.n_pnp > 0) { blocks += P.n_pnp; lds = lds > sizeof(agt_pnp::PnpShared) ? lds : sizeof(agt_pnp::PnpShared); }
    if (blocks == 0) return hipSuccess;
    if (P.lk.max_level < 3) hipLaunchKernelGGL((step_kernel<WIN, NW, 3>), dim3(blocks), dim3(STEP_THREADS), lds, stream, P);
    else hipLaunchKernelGGL((step_kernel<WIN, NW, AGT_MAX_LEVELS>), dim3(blocks), dim3(STEP_THREADS), lds, stream, P);
    return hipGetLastError();
}

}  // namespace

bool agt_step_supported(int win) { return win == 21; }
bool agt_step_fits(int n, int B) { return n <= AGT_WAVE && (long)n * B <= 128; }

hipError_t agt_launch_step(hipStream_t stream, const AgtStepParams& S, int win)
{
    if (win != 21) return hipErrorInvalidValue;
    const bool wide = S.n_lk > 0 ? agt_lk_wide(S.lk.n, S.lk_B) : true;
    return wide ? launch_step_t<21, 4>(stream, S) : launch_step_t<21, 1>(stream, S);
}
