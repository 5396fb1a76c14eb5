// agt_pnp.hip -- stand-alone cv::solvePnP(SOLVEPNP_ITERATIVE) / cv::projectPoints launches
// (solver body and design notes: agt_pnp_body.h).
#include "agt_pnp_body.h"

namespace {

using agt_pnp::load_cam;

template <typename T, int PPL>
__global__ __launch_bounds__(AGT_WAVE) void pnp_kernel(const AgtPnpParams P)
{
    __shared__ agt_pnp::PnpShared sh;
    agt_pnp::pnp_body<T, PPL>(P, blockIdx.x, sh, P.img, P.mask, P.state_out);
}

// n > 64: four waves.  A solve that starts from a guess is shared by all of them (agt_pnp_body.h, COOP); one without (first
// frame of a tracker, after a gate rejection, cv2-shaped calls without useExtrinsicGuess) is wave 0's, four points per lane.
constexpr int COOP_WAVES = agt_pnp::MAX_PPL;
template <typename T>
__global__ __launch_bounds__(AGT_WAVE * COOP_WAVES) void pnp_coop_kernel(const AgtPnpParams P)
{
    __shared__ agt_pnp::PnpShared sh;
    const int b = blockIdx.x;
    const bool guess = P.track ? (agt_uniform(P.track[b].has_guess) && P.enhance_ape) : P.use_guess != 0;
    if (guess) agt_pnp::pnp_body<T, 1, agt_pnp::PnpNoHook, false, COOP_WAVES>(P, b, sh, P.img, P.mask, P.state_out);
    else if (threadIdx.x < AGT_WAVE) agt_pnp::pnp_body<T, agt_pnp::MAX_PPL>(P, b, sh, P.img, P.mask, P.state_out);
}

template <typename T>
__global__ __launch_bounds__(256) void project_kernel(const AgtProjParams P)
{
    const int b = blockIdx.y;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P.n) return;
    AgtCamera cam;
    load_cam<T>(P.cam, cam);
    double param[6];
#pragma unroll
    for (int k = 0; k < 6; k++) param[k] = P.pose[(long)b * 6 + k];
    const T* obj = reinterpret_cast<const T*>(P.obj) + (long)b * P.obj_bstride;
    T* out = reinterpret_cast<T*>(P.img_out) + (long)b * P.n * 2;
    double R[9], G[9], u, v, jr[6], jt[6];
    const double X = (double)obj[i * 3], Y = (double)obj[i * 3 + 1], Z = (double)obj[i * 3 + 2];
    if (P.jac) {
        agt_rodrigues<true>(param, R, G);
        agt_project<true>(cam, R, G, param + 3, X, Y, Z, u, v, jr, jt);
        double* J = P.jac + ((long)b * P.n + i) * 12;
#pragma unroll
        for (int k = 0; k < 3; k++) { J[k] = jr[k]; J[3 + k] = jt[k]; J[6 + k] = jr[3 + k]; J[9 + k] = jt[3 + k]; }
    } else {
        agt_rodrigues<false>(param, R, G);
        agt_project<false>(cam, R, G, param + 3, X, Y, Z, u, v, nullptr, nullptr);
    }
    out[i * 2] = (T)u; out[i * 2 + 1] = (T)v;
}

template <typename T>
hipError_t launch_pnp_t(hipStream_t stream, const AgtPnpParams& p, int B)
{
    dim3 grid(B), block(AGT_WAVE);
    if (p.n <= AGT_WAVE) hipLaunchKernelGGL((pnp_kernel<T, 1>), grid, block, 0, stream, p);
    else hipLaunchKernelGGL((pnp_coop_kernel<T>), grid, dim3(AGT_WAVE * COOP_WAVES), 0, stream, p);
    return hipGetLastError();
}

}  // namespace

hipError_t agt_launch_pnp(hipStream_t stream, const AgtPnpParams& p, int B)
{
    if (p.n > AGT_WAVE * agt_pnp::MAX_PPL) return hipErrorInvalidValue;
    return p.dtype == AGT_F64 ? launch_pnp_t<double>(stream, p, B) : launch_pnp_t<float>(stream, p, B);
}

hipError_t agt_launch_project(hipStream_t stream, const AgtProjParams& p, int B)
{
    dim3 grid((p.n + 255) / 256, B), block(256);
    if (p.dtype == AGT_F64) hipLaunchKernelGGL(project_kernel<double>, grid, block, 0, stream, p);
    else hipLaunchKernelGGL(project_kernel<float>, grid, block, 0, stream, p);
    return hipGetLastError();
}
