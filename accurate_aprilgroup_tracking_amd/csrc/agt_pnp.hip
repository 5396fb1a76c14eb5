// agt_pnp.hip -- stand-alone cv::solvePnP(SOLVEPNP_ITERATIVE) / cv::projectPoints launches
// (solver body and design notes: agt_pnp_body.h).
#include "agt_pnp_body.h"
#include "agt_pyramid2_body.h"

namespace {

using agt_pnp::load_cam;

template <typename T, int PPL>
__global__ __launch_bounds__(AGT_WAVE) void pnp_kernel(const AgtPnpParams P)
{
    __shared__ agt_pnp::PnpShared sh;
#ifdef AGT_PNP_PRIO         // experiment builds (tools/build_variant.sh): issue priority of the solver's wave beside other kernels' waves
    __builtin_amdgcn_s_setprio(AGT_PNP_PRIO);
#endif
    agt_pnp::pnp_body<T, PPL>(P, blockIdx.x, sh, P.img, P.mask, P.state_out);
    if (P.host_seq && blockIdx.x == 0) agt_host_seq_store(P.host_seq, P.host_seq_base, threadIdx.x == 0);
}

// n > 64: four waves.  A solve that starts from a guess is shared by all of them (agt_pnp_body.h, COOP); one without (first
// frame of a tracker, after a gate rejection, cv2-shaped calls without useExtrinsicGuess) is wave 0's, four points per lane.
constexpr int COOP_WAVES = agt_pnp::MAX_PPL;
// Clip submission of the serial step (agt_api.hip step_serial): the two-level pyramid pass of the NEXT frame rides in this launch
// as extra workgroups (blockIdx.x >= n_solve; tile t of stream s at n_solve + s * n_pyr + t) -- the solve is one workgroup per
// stream, the rest of the chip idles beside it, and alone the pass was a 6.5 us launch at the head of the next frame's chain.
template <typename T>
__global__ __launch_bounds__(AGT_WAVE * COOP_WAVES) void pnp_coop_kernel(const AgtPnpParams P, const AgtPyrArgs Y0, const AgtPyrArgs Y1,
                                                                         const int n_solve, const int n_pyr)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t tile_lds[];       // pyramid tiles only (0 bytes without the job)
    __shared__ agt_pnp::PnpShared sh;
    if ((int)blockIdx.x >= n_solve) {
        const int t = (int)blockIdx.x - n_solve;
        const int st = t / n_pyr, tile = t - st * n_pyr;
        const int by = tile / Y0.gx, bx = tile - by * Y0.gx;
        agt_pyr2::pyr_down2_body(Y0, Y1, bx, by, Y0.src + (long)st * Y0.sbatch, Y0.dst + (long)st * Y0.dbatch, Y1.dst + (long)st * Y1.dbatch, tile_lds);
        return;
    }
    const int b = blockIdx.x;
    const bool guess = P.track ? (agt_uniform(P.track[b].has_guess) && P.enhance_ape) : P.use_guess != 0;
    // every wave has read the decision before wave 0 (alone, in the branch without a guess) may rewrite track[b].has_guess
    __syncthreads();
    if (guess) agt_pnp::pnp_body<T, 1, agt_pnp::PnpNoHook, false, COOP_WAVES>(P, b, sh, P.img, P.mask, P.state_out);
    else if (threadIdx.x < AGT_WAVE) agt_pnp::pnp_body<T, agt_pnp::MAX_PPL>(P, b, sh, P.img, P.mask, P.state_out);
    if (P.host_seq && b == 0 && threadIdx.x < AGT_WAVE) agt_host_seq_store(P.host_seq, P.host_seq_base, threadIdx.x == 0);      // (wave 0 writes the record in both bodies)
}

// cv::projectPoints for point i of stream b
template <typename T>
__device__ __forceinline__ void project_one(const AgtProjParams& P, int b, int i)
{
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
__global__ __launch_bounds__(256) void project_kernel(const AgtProjParams P)
{
    const int b = blockIdx.y;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < P.n) project_one<T>(P, b, i);
    if (P.host_seq) {
        // one-block launch of agt_project_points_host: every wave's outputs (host-mapped memory) are performed at system scope before the
        // barrier, then one lane tells the polling host thread
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
        __syncthreads();
        if (threadIdx.x < AGT_WAVE) agt_host_seq_store(P.host_seq, P.host_seq_base, threadIdx.x == 0);
    }
}

template <typename T>
hipError_t launch_pnp_t(hipStream_t stream, const AgtPnpParams& p, int B, const AgtPyrArgs* ride)
{
    dim3 grid(B), block(AGT_WAVE);
    if (p.n <= AGT_WAVE) {
        if (ride) return hipErrorInvalidValue;                    // (one-wave workgroups cannot carry pyramid tiles: agt_pnp_can_ride)
        hipLaunchKernelGGL((pnp_kernel<T, 1>), grid, block, 0, stream, p);
    } else {
        const AgtPyrArgs none = AgtPyrArgs();
        const int n_pyr = ride ? ride[0].gx * ride[0].gy : 0;
        hipLaunchKernelGGL((pnp_coop_kernel<T>), dim3((unsigned)(B + n_pyr * B)), dim3(AGT_WAVE * COOP_WAVES),
                           ride ? (size_t)agt_pyr2::PYR2_LDS_BYTES : 0, stream, p, ride ? ride[0] : none, ride ? ride[1] : none, B, n_pyr > 0 ? n_pyr : 1);
    }
    return hipGetLastError();
}

}  // namespace

bool agt_pnp_can_ride(int n) { return n > AGT_WAVE && n <= AGT_WAVE * agt_pnp::MAX_PPL; }

hipError_t agt_launch_pnp(hipStream_t stream, const AgtPnpParams& p, int B, const AgtPyrArgs* ride)
{
    if (p.n > AGT_WAVE * agt_pnp::MAX_PPL) return hipErrorInvalidValue;
    return p.dtype == AGT_F64 ? launch_pnp_t<double>(stream, p, B, ride) : launch_pnp_t<float>(stream, p, B, ride);
}

hipError_t agt_launch_project(hipStream_t stream, const AgtProjParams& p, int B)
{
    dim3 grid((p.n + 255) / 256, B), block(256);
    if (p.dtype == AGT_F64) hipLaunchKernelGGL(project_kernel<double>, grid, block, 0, stream, p);
    else hipLaunchKernelGGL(project_kernel<float>, grid, block, 0, stream, p);
    return hipGetLastError();
}
