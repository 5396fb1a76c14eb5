// agt_step_dense.hip -- the chained launches of the tracker's dense stage (BASELINE configs[4]): the LK launch that first finishes the
// previous frame's photometric refinement and re-seeds its corner (lk_reseed_kernel), and the same with the frame's cooperative pose
// solve and the next frame's pyramid tiles behind it in ONE launch (lk_pnp_coop_kernel).  Its own translation unit since round 5:
// agt_step.hip is compiled without the machine-level loop-invariant code motion (Makefile), which these kernels want to keep.
#undef AGT_LK_STAMPS          // (the diagnostic build's in-kernel stamps belong to agt_lk.hip / agt_pnp.hip / agt_step.hip)
#undef AGT_PNP_STAMPS
#undef AGT_STEP_STAMPS
#include "agt_step_args.h"

namespace {

// Clip submission of the tracker's dense stage (agt_track_frames_dense): the LK launch of frame t + 1 first finishes the dense stage
// of frame t.  Every workgroup (one corner, four waves) derives the last Gauss-Newton update from the block rows itself
// (agt_dense_body.h dense_finish: same rows, same order, same bits in every workgroup -- the scheme of the accumulate launches'
// prologue), projects ITS corner at the refined pose (the re-seed) and tracks it from there; the workgroup of a stream's corner 0
// also publishes pose / statistics / record.  Replaces dense_final_kernel, a one-workgroup launch of 5.5 us in the frame's serial
// chain, by ~2.5 us at the head of this launch.  The stream's done word is read, never written here (the other corners read it).
template <int NLEV>
__device__ __forceinline__ void lk_reseed_role(const AgtStepParams& S, const AgtStepTables& T, KParams KS, const agt_dense::DenseParams& F, bool has_final,
                                               int bid, int nblk, uint8_t* lds)
{
    static_assert(sizeof(agt_dense::DenseShared) <= agt_lk::lk_chain_lds_bytes<NLEV>(), "the prologue's LDS fits the tracker's");
    const int blk = agt_xcd_order(bid, nblk, 3);      // XCD-aware corner order, the literal 8-way deal (see pyr_role); nblk is a multiple of 8
    if (blk >= S.lk.n * S.lk_B) return;
    const int b = blk / S.lk.n, pt = blk - b * S.lk.n;
    const long pidx = (long)b * S.lk.n + pt;
    // the corner as the previous frame's LK / PnP left it: requested first, used when the stage did not refine the pose
    float px = S.lk.prev_pts[pidx * 2], py = S.lk.prev_pts[pidx * 2 + 1];
    int pst = S.lk.prev_status ? (int)S.lk.prev_status[pidx] : 1;
    if (has_final) {
        const float X = F.obj[pt * 3], Y = F.obj[pt * 3 + 1], Z = F.obj[pt * 3 + 2];
        agt_dense::DenseShared& sh = *reinterpret_cast<agt_dense::DenseShared*>(lds);
        double param[6];
        const bool refined = agt_dense::dense_finish(F, sh, b, S.lk_B, pt == 0, false, param);
        if (F.seed_pts && F.rec && refined) {
            AgtCamera cam;
            agt_pnp::load_cam<float>(F.cam, cam);
            double R[9], G[9];
            agt_rodrigues<false>(param, R, G);
            double u, v;
            agt_project<false>(cam, R, G, param + 3, (double)X, (double)Y, (double)Z, u, v, nullptr, nullptr);
            px = (float)u; py = (float)v; pst = 1;
            if (threadIdx.x == 0) { F.seed_pts[pidx * 2] = px; F.seed_pts[pidx * 2 + 1] = py; F.seed_status[pidx] = 1; }
        }
        __syncthreads();                // the prologue's LDS is the tracker's from here on
    }
    auto frame = [&](int) {
        agt_lk::LkFrameIo<NLEV> io;
        io.grouped = true; io.prev_pts = S.lk.prev_pts; io.err = nullptr; io.have_pos = true; io.px = px; io.py = py; io.pst = pst;
#pragma unroll
        for (int l = 0; l < NLEV; l++) { io.imgI[l] = T.lk.img[0][l]; io.imgJ[l] = T.lk.img[1][l]; }
        io.next_pts = T.lk.next[0]; io.status = T.lk.status[0]; io.done = T.lk.done[0];
        return io;
    };
    agt_lk::lk_frames_w4<NLEV>(&KS->lk, pt, b, lds, 1, frame);
}

template <int NLEV>
__global__ __launch_bounds__(AGT_WAVE * 4) void lk_reseed_kernel(const AgtStepParams S, const AgtStepTables T, const agt_dense::DenseParams F)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    lk_reseed_role<NLEV>(S, T, kernarg_params(), F, true, (int)blockIdx.x, (int)gridDim.x, lds);
}

// ... and the frame's pose solve chained to it in the SAME launch (n > 64 corners: the four-wave cooperative solver, one workgroup
// per stream behind the S.n_lk tracker workgroups), waiting for the frame's arrival count as the pose role of the fused step does:
// the launch boundary between LK and PnP (~2 us at the end of a 25 us launch) and the solver's start-up leave the serial chain.
// has_final == 0: no dense stage is pending (first frame of a clip).
// Behind the solver's workgroups: the two-level pyramid pass of the NEXT frame (clip submission; n_pyr tiles per stream, Y0 / Y1 as in
// pnp_coop_kernel) -- the kernel's registers allow one workgroup per CU, the trackers and the solver hold 241 of the 256, the tiles
// take the rest and the CUs the trackers leave.
template <int NLEV>
__global__ __launch_bounds__(AGT_WAVE * 4) void lk_pnp_coop_kernel(const AgtStepParams S, const AgtStepTables T, const agt_dense::DenseParams F, const int has_final,
                                                                   const AgtPyrArgs Y0, const AgtPyrArgs Y1, const int n_pyr)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    if ((int)blockIdx.x >= S.n_lk + S.n_pnp) {
        const int t = (int)blockIdx.x - S.n_lk - S.n_pnp;
        const int st = t / n_pyr, tile = t - st * n_pyr;
        const int by = tile / Y0.gx, bx = tile - by * Y0.gx;
        agt_pyr2::pyr_down2_body(Y0, Y1, bx, by, Y0.src + (long)st * Y0.sbatch, Y0.dst + (long)st * Y0.dbatch, Y1.dst + (long)st * Y1.dbatch, lds);
        return;
    }
    if ((int)blockIdx.x >= S.n_lk) {
        // the solve as the stand-alone pnp_coop_kernel runs it (tracker state in global memory: the role form of the group launches,
        // pnp_role_coop, keeps it in LDS across frames and spills 236 registers for it), behind the wait of a chained launch
        agt_pnp::PnpShared& sh = *reinterpret_cast<agt_pnp::PnpShared*>(lds);
        const int b = (int)blockIdx.x - S.n_lk;
        if (threadIdx.x == 0) {
            const int* fault = &S.pnp.track[b].chain_fault;
            const unsigned target = (unsigned)T.pnp.target[0];
            unsigned polls = 0;
            int timed_out = 0;
            while ((int)(__hip_atomic_load(T.pnp.wait[0] + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - target) < 0) {
                __builtin_amdgcn_s_sleep(4);
                if (++polls > AGT_CHAIN_POLLS || ((polls & 15) == 1 && __hip_atomic_load(fault, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) { timed_out = 1; break; }
            }
            *(volatile int*)&sh.late = timed_out;
        }
        __syncthreads();
        const int late = agt_uniform(*(volatile int*)&sh.late);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        const bool guess = agt_uniform(S.pnp.track[b].has_guess) && S.pnp.enhance_ape;
        __syncthreads();        // every wave has read the decision before wave 0 (alone, without a guess) may rewrite has_guess
        const int xf = late ? AGT_TRK_CHAIN_TIMEOUT : 0;
        if (guess) agt_pnp::pnp_body<float, 1, agt_pnp::PnpNoHook, false, PNP_COOP>(S.pnp, b, sh, T.pnp.img[0], T.pnp.mask[0], T.pnp.so[0], xf);
        else if (threadIdx.x < AGT_WAVE) agt_pnp::pnp_body<float, agt_pnp::MAX_PPL>(S.pnp, b, sh, T.pnp.img[0], T.pnp.mask[0], T.pnp.so[0], xf);
        return;
    }
    lk_reseed_role<NLEV>(S, T, kernarg_params(), F, has_final != 0, (int)blockIdx.x, S.n_lk, lds);
}

}  // namespace

// the one-frame LK role launch of step_serial with the previous frame's dense stage finished in its prologue (lk_reseed_kernel; F
// null: nothing pending), and, with S.n_pnp > 0, the frame's cooperative pose solve chained to it in the same launch (lk_pnp_coop_kernel)
hipError_t agt_launch_lk_reseed(hipStream_t stream, const AgtStepParams& S, const AgtStepTables& T, int win, const AgtDenseFinal* F, const AgtPyrArgs* ride)
{
    if (ride && S.n_pnp <= 0) return hipErrorInvalidValue;          // (the pyramid tiles ride in the chained form only)
    if (win != 21 || S.n_lk <= 0 || S.lk_nf != 1 || !agt_lk_wide(S.lk.n, S.lk_B) || S.lk.flags != 0 || S.lk.err != nullptr) return hipErrorInvalidValue;
    const bool chain = S.n_pnp > 0;
    if (!chain && !F) return hipErrorInvalidValue;
    if (chain && (S.pnp_nf != 1 || S.pnp.n <= AGT_WAVE || S.pnp.n > AGT_WAVE * PNP_COOP || !T.pnp.wait[0] || !T.lk.done[0])) return hipErrorInvalidValue;
    AgtStepParams P = S;
    if (!chain) { P.n_pnp = 0; P.pnp_nf = 0; }
    for (int s = 0; s < AGT_MAX_LEVELS - 1; s++) { P.n_pyr[s] = 0; P.pyr_nf[s] = 0; }
    const long corners = (long)P.lk.n * P.lk_B;
    P.xshift = agt_chip_current().xshift; P.rsv_ = 0; P.lk.xshift = P.xshift;
    const unsigned grid8 = agt_xcd_grid(corners, 3);              // (lk_reseed_role deals 8 ways on every device)
    P.n_lk = (int)grid8;
    agt_dense::DenseParams D;
    static_assert(sizeof(D) <= sizeof(F->bytes), "AgtDenseFinal holds a DenseParams");
    memset(&D, 0, sizeof(D));
    if (F) {
        memcpy(&D, F->bytes, sizeof(D));
        if (D.N != P.lk.n || (D.seed_pts && D.seed_pts != P.lk.prev_pts)) return hipErrorInvalidValue;      // (a re-seeded corner set IS this launch's start)
    }
    const bool small = P.lk.max_level < 3;
    size_t per = small ? lk_role_lds<21, 4, 3>(P.lk.max_level + 1) : lk_role_lds<21, 4, AGT_MAX_LEVELS>(P.lk.max_level + 1);
    if (!chain) {
        if (small) hipLaunchKernelGGL((lk_reseed_kernel<3>), dim3(grid8), dim3(AGT_WAVE * 4), per, stream, P, T, D);
        else hipLaunchKernelGGL((lk_reseed_kernel<AGT_MAX_LEVELS>), dim3(grid8), dim3(AGT_WAVE * 4), per, stream, P, T, D);
        return hipGetLastError();
    }
    if (per < sizeof(agt_pnp::PnpShared)) per = sizeof(agt_pnp::PnpShared);
    const AgtPyrArgs none = AgtPyrArgs();
    const int n_pyr = ride ? ride[0].gx * ride[0].gy : 0;
    if (ride && per < (size_t)agt_pyr2::PYR2_LDS_BYTES) per = (size_t)agt_pyr2::PYR2_LDS_BYTES;
    const unsigned grid = grid8 + (unsigned)P.n_pnp + (unsigned)(n_pyr * P.lk_B);
    if (small) hipLaunchKernelGGL((lk_pnp_coop_kernel<3>), dim3(grid), dim3(AGT_WAVE * 4), per, stream, P, T, D, F ? 1 : 0, ride ? ride[0] : none, ride ? ride[1] : none, n_pyr > 0 ? n_pyr : 1);
    else hipLaunchKernelGGL((lk_pnp_coop_kernel<AGT_MAX_LEVELS>), dim3(grid), dim3(AGT_WAVE * 4), per, stream, P, T, D, F ? 1 : 0, ride ? ride[0] : none, ride ? ride[1] : none, n_pyr > 0 ? n_pyr : 1);
    return hipGetLastError();
}
