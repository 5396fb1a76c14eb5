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
#ifdef AGT_STEP_LK_STAMPS       // experiment builds only: the LK role's in-kernel stamps under their own symbol (tools/_exp)
#define AGT_LK_STAMPS
#define agt_lk_stamps agt_lk_stamps_step
#define agt_debug_lk_stamps agt_debug_lk_stamps_step
#else
#undef AGT_LK_STAMPS
#endif
#undef AGT_PNP_STAMPS
#include "agt_step_args.h"

// Role timeline of the fused step (diagnostic build only, -DAGT_STEP_STAMPS; tools/stepstamps.py): s_memtime at entry and
// exit of the PnP block, of the first LK block and the latest exit of any LK / pyramid block.
#ifdef AGT_STEP_STAMPS
__device__ unsigned long long agt_step_stamps[16];
#define SSTAMP_SET(i) do { if (threadIdx.x == 0) agt_step_stamps[i] = __builtin_amdgcn_s_memtime(); } while (0)
#define SSTAMP_MAX(i) do { if (threadIdx.x == 0) atomicMax(&agt_step_stamps[i], (unsigned long long)__builtin_amdgcn_s_memtime()); } while (0)
#define SSTAMP_MIN(i) do { if (threadIdx.x == 0) atomicMin(&agt_step_stamps[i], (unsigned long long)__builtin_amdgcn_s_memtime()); } while (0)
extern "C" int agt_debug_step_stamps(unsigned long long* host16, int reset)
{
    int rc = (int)hipMemcpyFromSymbol(host16, HIP_SYMBOL(agt_step_stamps), sizeof(agt_step_stamps));
    if (reset) {
        unsigned long long z[16];
        for (int i = 0; i < 16; i++) z[i] = (i == 6) ? ~0ull : 0ull;
        rc |= (int)hipMemcpyToSymbol(HIP_SYMBOL(agt_step_stamps), z, sizeof(z));
    }
    return rc;
}
// per-frame timeline of the PnP role of stream 0 (tools/rolestamps.py): [frame][0] loop top, [1] corners acquired, [2] previous
// frame's state acquired, [3] frame done
__device__ unsigned long long agt_role_stamps[AGT_MAX_GROUP * 4];
#define RSTAMP(k, i) do { if (blk == 0 && lane == 0) agt_role_stamps[(k) * 4 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
extern "C" int agt_debug_role_stamps(unsigned long long* host) { return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(agt_role_stamps), sizeof(agt_role_stamps)); }
#else
#define SSTAMP_SET(i)
#define SSTAMP_MAX(i)
#define SSTAMP_MIN(i)
#define RSTAMP(k, i)
#endif


// ---- Copies whose ordering must not rest on type-based alias rules (round 6; profiles/r06_aperture_violation.md).
// The per-frame pointer tables (kernel arguments) and the tracker state are copied to / from LDS as raw 8-byte words while every
// other access to them is typed (pointers, doubles, ints).  Written through `uint32_t*` / `double*` views, the copies were, to the
// compiler's type-based alias analysis, unrelated to the typed reads that follow them: the order rested on the fences and asm
// clobbers between them alone.  A character copy aliases every type, so the
// dependency from a copy to each typed read (and from each typed write to the copy back) is one the optimiser and the machine
// scheduler both see, under any scheduling strategy.  A `may_alias` 64-bit word is that character copy at its natural alignment
// (or a 32-bit one).  One word per lane and trip; n8 words, `stride` lanes, lane index `t`.
typedef unsigned long long __attribute__((may_alias)) agt_word_t;
__device__ __forceinline__ void agt_copy_words_to_lds(void* dst_lds, const void* src, int n8, int t, int stride)
{
    for (int i = t; i < n8; i += stride) static_cast<agt_word_t*>(dst_lds)[i] = static_cast<const agt_word_t*>(src)[i];
}
typedef unsigned int __attribute__((may_alias)) agt_halfword_t;
__device__ __forceinline__ void agt_copy_halfwords_to_lds(void* dst_lds, const void* src, int n4, int t, int stride)
{
    for (int i = t; i < n4; i += stride) static_cast<agt_halfword_t*>(dst_lds)[i] = static_cast<const agt_halfword_t*>(src)[i];
}
// A pointer that came out of an LDS table is dereferenced only if it looks like one: user-space device and host-mapped addresses
// on this platform have bits 47.. clear.  Anything else (stale LDS, a clobbered table) would be an access "beyond the largest legal
// address" -- HSA_STATUS_ERROR_MEMORY_APERTURE_VIOLATION, which aborts the queue and the process.  The roles turn it into the
// fail-stop of a chained wait that gave up instead (record flagged, stream frozen, host told): the reference never crashes on a
// bad frame (detect_pose.py:570-574).
__device__ __forceinline__ bool agt_ptr_plausible(unsigned long long p) { return (p >> 47) == 0; }
// ... the LK role's: every image, output and counter pointer a frame of the group took from the LDS copy of the tables (wave-uniform)
template <int NLEV>
__device__ __forceinline__ bool lk_table_bad(const agt_lk::LkFrameIo<NLEV>& io, int max_level)
{
    unsigned long long acc = (unsigned long long)io.next_pts | (unsigned long long)io.status | (unsigned long long)io.done;
    bool null_img = io.next_pts == nullptr || io.status == nullptr;
#pragma unroll
    for (int l = 0; l < NLEV; l++)
        if (l <= max_level) {
            acc |= (unsigned long long)io.imgI[l] | (unsigned long long)io.imgJ[l];
            null_img = null_img || io.imgI[l] == nullptr || io.imgJ[l] == nullptr;
        }
    return agt_uniform((int)(null_img || !agt_ptr_plausible(acc))) != 0;
}

#ifndef AGT_LKG_OCC
#define AGT_LKG_OCC 4            // waves per SIMD the one-wave-per-corner LK group kernel is register-allocated for (4: 128 VGPRs)
#endif

// (defined in the translation unit compiled without MachineLICM: see pnp_group_coop_kernel below)
hipError_t agt_launch_pnp_group_coop(hipStream_t stream, const AgtStepParams& P, const AgtStepTables& T);
hipError_t agt_launch_step_deep(hipStream_t stream, const AgtStepParams& P, const AgtStepTables& T, int nw, int blocks, size_t lds);

namespace {

// ---- LK role: workgroup `blk` of the role, THREADS threads.  NW = 4: the workgroup is one corner; NW = 1: each wave is
// its own corner.  Consecutive frames of a corner are tracked in-kernel (position carried in registers).
// XCD_MAP (the role as its own launch, grid rounded up to a multiple of 8): workgroup g takes block (g mod 8) * (grid / 8) + g / 8 of
// the role -- every XCD walks a contiguous run of corners, so the overlapping tiles of a tag's corners meet in one L2 (see agt_lk.hip).
template <int WIN, int NW, int NLEV, int THREADS, bool XCD_MAP = false>
__device__ __forceinline__ void lk_role(const AgtStepParams& S, const AgtStepTables& T, KParams KS, KTables KT, int blk, uint8_t* lds)
{
    constexpr int CPB = THREADS / (AGT_WAVE * NW);
    if constexpr (XCD_MAP) blk = agt_xcd_order(blk, (int)gridDim.x, KS->xshift);
    const int wave = threadIdx.x / AGT_WAVE;
    const long corner = (long)blk * CPB + (NW == 1 ? wave : 0);
    if (corner >= (long)S.lk.n * S.lk_B) return;
    const int b = (int)(corner / S.lk.n), pt = (int)(corner - (long)b * S.lk.n);
    constexpr size_t LKB = sizeof(AgtLkTables);
    const size_t per = lk_role_lds<WIN, NW, NLEV>(S.lk.max_level + 1);
    uint8_t* my = lds + (NW == 1 ? (size_t)wave * per : 0);
    // Frames 2.. of the group take their image / output pointers from a copy of the tables in LDS: a dependent
    // scalar load from the kernel-argument segment in the middle of the chain costs a memory round trip (device
    // memory with this runtime's defaults, host memory across PCIe under HIP_FORCE_DEV_KERNARG=0: profiles/
    // r05_experiments.md section 15).  The copy is requested here, before anything else, and lands while the first
    // frame's corner position and tiles are still in flight; the first frame itself uses the statically indexed `S`.
    AgtLkTables* tab = reinterpret_cast<AgtLkTables*>(my + per - LKB);
    {
        static_assert(sizeof(AgtLkTables) % 8 == 0, "table copy: 8-byte words");
        const int tid = NW == 1 ? (int)(threadIdx.x & (AGT_WAVE - 1)) : (int)threadIdx.x;
        // (vector loads through a generic view of the kernel-argument segment: one word per lane, all in flight at once; a character
        // copy -- see agt_copy_words_to_lds -- so the typed reads of tab->... below depend on it for every alias analysis)
        if (S.lk_nf > 1) agt_copy_words_to_lds(tab, (const void*)(const __attribute__((address_space(4))) void*)&KT->lk, (int)(LKB / 8), tid, AGT_WAVE * NW);
    }
    if constexpr (WIN == 21 && NW == 4) {
        // one workgroup per corner: the frame-chained body (agt_lk_chain_body.h) for what the tracker asks for.  A launch of a
        // single frame has nothing to hand over, but the body's shorter iteration (LDS-atomic exchange, early tap loads: round 3)
        // now outweighs its on-demand first frame: one frame per launch 47.4 -> 52.4 k frames/s at 48 corners, configs[4]'s
        // LK(240) 28.8 -> 25.5 us (round 2 had measured 23.2 vs 20.0 us against it)
        if (S.lk.flags == 0 && S.lk.err == nullptr) {
            auto frame = [&](int k) {
                agt_lk::LkFrameIo<NLEV> io;
                io.grouped = true; io.prev_pts = S.lk.prev_pts; io.err = nullptr; io.have_pos = false; io.px = io.py = 0.f; io.pst = 1;
                if (k == 0) {
#pragma unroll
                    for (int l = 0; l < NLEV; l++) { io.imgI[l] = T.lk.img[0][l]; io.imgJ[l] = T.lk.img[1][l]; }
                    io.next_pts = T.lk.next[0]; io.status = T.lk.status[0]; io.done = T.lk.done[0];
                } else {
#pragma unroll
                    for (int l = 0; l < NLEV; l++) { io.imgI[l] = tab->img[k][l]; io.imgJ[l] = tab->img[k + 1][l]; }
                    io.next_pts = tab->next[k]; io.status = tab->status[k]; io.done = tab->done[k];
                    io.bad = lk_table_bad(io, S.lk.max_level);
                    if (io.bad && threadIdx.x == 0 && S.pnp.fault) __hip_atomic_store(S.pnp.fault, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                }
                return io;
            };
            agt_lk::lk_frames_w4<NLEV>(&KS->lk, pt, b, my, S.lk_nf, frame);
            return;
        }
    }
    float px = 0.f, py = 0.f; int pst = 1;
    auto frame_io = [&](int k) {
        agt_lk::LkFrameIo<NLEV> io;
        io.grouped = true; io.prev_pts = S.lk.prev_pts; io.err = nullptr; io.have_pos = k > 0; io.px = px; io.py = py; io.pst = pst;
        if (k == 0) {
#pragma unroll
            for (int l = 0; l < NLEV; l++) { io.imgI[l] = T.lk.img[0][l]; io.imgJ[l] = T.lk.img[1][l]; }
            io.next_pts = T.lk.next[0]; io.status = T.lk.status[0]; io.done = T.lk.done[0];
        } else {
            agt_lk::block_sync<NW>();          // the previous frame's LDS tiles are free again; the table copy is visible
#pragma unroll
            for (int l = 0; l < NLEV; l++) { io.imgI[l] = tab->img[k][l]; io.imgJ[l] = tab->img[k + 1][l]; }
            io.next_pts = tab->next[k]; io.status = tab->status[k]; io.done = tab->done[k];
            io.bad = lk_table_bad(io, S.lk.max_level);
            if (io.bad && (threadIdx.x & (AGT_WAVE * NW - 1)) == 0 && S.pnp.fault) __hip_atomic_store(S.pnp.fault, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        return io;
    };
    int k = 0;
    if constexpr (WIN == 21 && NW == 1) {
        // one wave per corner: the row-segment body while the window stays inside the image and the corner is alive (with four
        // waves per corner the general body is kept: inside this kernel the row-segment form measured 16.7 us per frame against
        // 15.9, although it wins by 2.5 us as the stand-alone lk_kernel).  Two loops in sequence, NOT one loop with both bodies:
        // side by side in one loop body the two trackers are register-allocated as one (29 VGPR spills at a 168-register budget,
        // 79 at 128; each alone fits 137).  A corner that needs the general body once (window at the image border, status lost)
        // finishes the group's remaining frames in the second loop -- the general body tracks every case, only slower.
        const long pidx = (long)b * S.lk.n + pt;
        for (; k < S.lk_nf; k++) {
            const float ppx = k ? px : S.lk.prev_pts[pidx * 2], ppy = k ? py : S.lk.prev_pts[pidx * 2 + 1];
            const int alive = k ? pst : (S.lk.prev_status ? S.lk.prev_status[pidx] : 1);
            if (!agt_uniform((int)(alive != 0 && !(S.lk.flags & 0x10000) && agt_lk::rs_interior(ppx, ppy, S.lk.max_level, S.lk.prev[0].w, S.lk.prev[0].h))))
                break;
#ifdef AGT_STEP_LK_STAMPS
            const unsigned long long tq0 = __builtin_amdgcn_s_memtime();
#endif
            const agt_lk::LkFrameIo<NLEV> io = frame_io(k);
            if (io.bad) return;
            agt_lk::lk_body_rs<NW, NLEV>(&KS->lk, pt, b, my, io, ppx, ppy, px, py, pst);
#ifdef AGT_STEP_LK_STAMPS
            if ((threadIdx.x & 63) == 0) {       // experiment: slowest / summed per-frame time over all corners, frames in the row-segment loop
                const unsigned long long dt = __builtin_amdgcn_s_memtime() - tq0;
                atomicMax(&agt_lk_stamps[40 + (k < 15 ? k : 15)], dt);
                atomicAdd(&agt_lk_stamps[56], dt); atomicAdd(&agt_lk_stamps[57], 1ull);
            }
#endif
        }
    }
    for (; k < S.lk_nf; k++) {
        const agt_lk::LkFrameIo<NLEV> io = frame_io(k);
        if (io.bad) return;
        agt_lk::lk_body<WIN, NW, NLEV>(&KS->lk, pt, b, my, io, px, py, pst);
#ifdef AGT_STEP_LK_STAMPS
        if ((threadIdx.x & 63) == 0) atomicAdd(&agt_lk_stamps[58], 1ull);
#endif
    }
}

// ---- pyramid role: workgroup `blk` of the role (all stages concatenated); `base` = launch-wide index of the role's first
// workgroup (decides which XCD a workgroup sits on).  Everything is read from the kernel-argument segment (scalar loads):
// a reference to the by-value argument would make the compiler copy the whole structure to scratch.
// ROLL2: the register-rolling two-level pass may be planned (split pipeline: pyr_group_kernel); the fused step kernel is planned with
// the tiled two-level pass only (agt_api.hip launch_group) and does not carry the rolling body -- it sits on the edge of its registers
template <bool ROLL2>
__device__ __forceinline__ void pyr_role(KParams KS, KTables KT, int blk, int base, uint8_t* lds)
{
#pragma unroll
    for (int s = 0; s < AGT_MAX_LEVELS - 1; s++) {
        const int n = KS->n_pyr[s];
        if (blk < n) {
            AgtPyrArgs A;
            A.src = nullptr; A.dst = nullptr; A.pad = 0; A.rsv_ = 0;
            A.spitch = KS->pyr[s].spitch; A.sbatch = KS->pyr[s].sbatch; A.dpitch = KS->pyr[s].dpitch; A.dbatch = KS->pyr[s].dbatch;
            A.sw = KS->pyr[s].sw; A.sh = KS->pyr[s].sh; A.dw = KS->pyr[s].dw; A.dh = KS->pyr[s].dh;
            A.gx = KS->pyr[s].gx; A.gy = KS->pyr[s].gy; A.B = KS->pyr[s].B; A.pad = KS->pyr[s].pad;
            // XCD-aware tile order (see agt_pyramid.hip): workgroup index % X is the XCD (X = 2^xshift XCDs); the stage's
            // workgroups on XCD j take a contiguous run of tiles, runs laid out in XCD order.
            // (the fused step kernel keeps the LITERAL 8-way deal of rounds 1-4 -- a correct order on every device, the tuned one on a whole
            // MI355X: read from the arguments, the shift cost the chained launches 0.5-1.7 % (c4 82.4 -> 81.5 k frames/s; the kernel sits on
            // the edge of its register allocation); the group launch of the split pipeline follows the device's XCD count)
            const int xs = ROLL2 ? KS->xshift : 3, xm = (1 << xs) - 1;
            const int j = (blk + base) & xm;
            int tile = (blk - ((j - base) & xm)) >> xs;
            for (int q = 0; q < j; q++) tile += (n - ((q - base) & xm) + xm) >> xs;
            const int per_img = A.gx * A.gy;
            const int bz = tile / per_img, r = tile - bz * per_img;      // bz = frame * B + stream
            const int by = r / A.gx, bx = r - by * A.gx;
            const int fr = bz / A.B, st = bz - fr * A.B;
            if (s == 0 && KS->pyr_fused) {
                // levels 1 and 2 in one pass (agt_pyramid2_body.h); the level 1 -> 2 geometry sits in pyr[1]
                AgtPyrArgs A1;
                A1.src = nullptr; A1.dst = nullptr; A1.pad = 0;
                A1.spitch = KS->pyr[1].spitch; A1.sbatch = KS->pyr[1].sbatch; A1.dpitch = KS->pyr[1].dpitch; A1.dbatch = KS->pyr[1].dbatch;
                A1.sw = KS->pyr[1].sw; A1.sh = KS->pyr[1].sh; A1.dw = KS->pyr[1].dw; A1.dh = KS->pyr[1].dh;
                A1.gx = A.gx; A1.gy = A.gy; A1.B = A.B;
                if (ROLL2 && A.pad) {      // register-rolling form, alternating strip directions (agt_pyramid.hip agt_pyr2_plan): bx = workgroup of the image, no LDS
                    A.rsv_ = KS->pyr[0].rsv_;
                    agt_pyr4::pyr_roll2_body(A, A1, bx, KT->pyr_src[0][fr] + (long)st * A.sbatch, KT->pyr_dst[0][fr] + (long)st * A.dbatch,
                                             KT->pyr_dst[1][fr] + (long)st * A1.dbatch);
                    return;
                }
                agt_pyr2::pyr_down2_body(A, A1, bx, by, KT->pyr_src[0][fr] + (long)st * A.sbatch, KT->pyr_dst[0][fr] + (long)st * A.dbatch,
                                         KT->pyr_dst[1][fr] + (long)st * A1.dbatch, lds);
                return;
            }
            if (A.pad) {        // register-rolling form (agt_pyramid3_body.h): bx = workgroup of the image, no LDS
                agt_pyr3::pyr_roll_body(A, bx, KT->pyr_src[s][fr] + (long)st * A.sbatch, KT->pyr_dst[s][fr] + (long)st * A.dbatch);
                return;
            }
            agt_pyr::pyr_down_body(A, bx, by, KT->pyr_src[s][fr] + (long)st * A.sbatch, KT->pyr_dst[s][fr] + (long)st * A.dbatch, lds);
            return;
        }
        blk -= n;
        base += n;
    }
}

// ---- PnP role: one wave per stream, consecutive frames: frame k+1 starts from the tracker state frame k left in
// global memory (written and read by this one wave; the barrier orders the two).  Frames 2.. read their pointers
// from an LDS copy of the tables requested up front (see the LK role).
// Chained launch: a frame whose corners come from the LK role of THIS launch carries the address of its arrival counter;
// the wave waits until the counter reaches the frame's corner count (every lk_publish of the frame has been acknowledged by
// memory), then drops whatever its own L1 / L2 hold of the corner arrays (acquire at device scope).  The LK workgroups
// have lower indices than this one, so they were dispatched before it and never wait for anything themselves; should the
// count still not arrive within AGT_CHAIN_POLLS polls (>= 30 ms of executing time) the wave gives up: the frame is NOT solved,
// its record is flagged AGT_TRK_CHAIN_TIMEOUT and invalid, the stream's tracker state is frozen behind a sticky fault word
// (AgtTrackState::chain_fault; every later record of the stream is flagged too, until agt_tracker_reset), the context's
// host-mapped fault word is set (agt_synchronize / agt_tracker_join then return AGT_ERR_CHAIN), and neither the remaining
// frames of the launch nor later launches wait again for that stream -- the launch always drains.
// NWV = 2 (fused step): two waves of the workgroup alternate over the frames.  The wave of frame k+1 waits for that frame's
// corners, requests and counts them, and only then waits (LDS word sh.seq) for frame k's state update to be complete: the
// ~1.3 us of HBM latency and bookkeeping at the head of a solve run under the tail of the previous one.  The waves share one
// scratch area: a wave touches it only between its wait on sh.seq and its own increment of it.  wave: this wave's index.
template <int PPL, int NWV = 1>
__device__ __forceinline__ void pnp_role(const AgtStepParams& S, const AgtStepTables& T, KTables KT, int blk, agt_pnp::PnpShared& sh, int wave = 0)
{
    static_assert(sizeof(AgtPnpTables) == 40 * AGT_MAX_GROUP && sizeof(sh.tab) == sizeof(AgtPnpTables), "table layout");
    static_assert(sizeof(AgtTrackState) % 8 == 0 && sizeof(AgtTrackState) / 8 <= AGT_WAVE, "state copy: one double per lane");
    const int lane = (int)(threadIdx.x & (AGT_WAVE - 1));
    // The stream's tracker state lives in LDS (sh.ts) for the frames of the launch: frame k + 1 starts from what frame k left
    // there.  Through global memory the hand-over cost ~1.2 us per frame on the critical path -- the writer waits for its
    // stores to be acknowledged before it may signal, the reader's loads go out to L2.  Loaded here, written back by the wave
    // that solves the launch's last frame.
    if (NWV == 1) {
        // (character copies: agt_copy_words_to_lds.  The table is copied whenever a later frame will read it -- S.pnp_nf > 1 -- and
        // read only then: frame 0 takes its pointers from the kernel arguments, frames k >= 1 exist only when S.pnp_nf > 1)
        if (S.pnp_nf > 1) agt_copy_words_to_lds(sh.tab, (const void*)(const __attribute__((address_space(4))) void*)&KT->pnp, (int)(sizeof(AgtPnpTables) / 8), lane, AGT_WAVE);
        agt_copy_words_to_lds(&sh.ts, S.pnp.track + blk, (int)(sizeof(AgtTrackState) / 8), lane, AGT_WAVE);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    }
    // (NWV == 2: the caller has copied the tables and the state and zeroed sh.seq with the whole workgroup, behind a barrier)
    // (measured and dropped, round 3: s_setprio(3) for this wave -- in split mode it shares its SIMD with up to three VALU-bound
    // LK waves; the step time did not move at any stream count, the LK group launch is the longer chain there)
    int late = 0;
    for (int k = wave; k < S.pnp_nf; k += NWV) {
        const void* img = T.pnp.img[0]; const uint8_t* mask = T.pnp.mask[0]; double* so = T.pnp.so[0];
        const unsigned* wait = T.pnp.wait[0]; unsigned target = (unsigned)T.pnp.target[0];
        RSTAMP(k, 0);
        if (k) {
            if (NWV == 1) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_wave_barrier(); }      // (tables, state: LDS)
            unsigned long long t_img = sh.tab[k], t_mask = sh.tab[AGT_MAX_GROUP + k], t_so = sh.tab[2 * AGT_MAX_GROUP + k];
            unsigned long long t_wait = sh.tab[3 * AGT_MAX_GROUP + k];
            target = (unsigned)sh.tab[4 * AGT_MAX_GROUP + k];
            // a table entry that cannot be an address is never dereferenced: the frame takes the give-up path of a chained wait
            // (nothing solved, record -- if its destination is believable -- flagged, stream frozen, host told)
            if (!agt_uniform((int)(t_img != 0 && agt_ptr_plausible(t_img | t_mask | t_so | t_wait)))) {
                late = 1;
                t_img = (unsigned long long)T.pnp.img[0]; t_mask = 0; t_wait = 0;
                if (!agt_ptr_plausible(t_so)) t_so = 0;
            }
            img = (const void*)t_img; mask = (const uint8_t*)t_mask; so = (double*)t_so; wait = (const unsigned*)t_wait;
        }
        if (wait) {
            if (!late) {
                int timed_out = 0;
                if (lane == 0) {
                    // The give-up budget counts POLLS, not wall time (a preempted queue -- ranks sharing a card, a
                    // debugger -- must not trip it): 2^16 polls of >= 0.5 us each are >= 30 ms of this wave EXECUTING.  While
                    // polling the wave also watches for a give-up of the workgroup's other wave (sh.late) and for the stream's
                    // sticky fault word (set by an earlier launch): a faulted stream never waits again.
                    const int* fault = &S.pnp.track[blk].chain_fault;
                    unsigned polls = 0;
                    while ((int)(__hip_atomic_load(wait + blk, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - target) < 0) {
                        __builtin_amdgcn_s_sleep(4);
                        if ((NWV > 1 && *(volatile int*)&sh.late) || ++polls > AGT_CHAIN_POLLS ||
                            ((polls & 15) == 1 && __hip_atomic_load(fault, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) { timed_out = 1; break; }
                    }
                    if (timed_out && NWV > 1) *(volatile int*)&sh.late = 1;
                }
                late = agt_uniform(__shfl(timed_out, 0));
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        RSTAMP(k, 1);
        auto before_state = [&]() {
            RSTAMP(k, 1);
            if (NWV > 1) {
                // frame k - 1 belongs to the other wave: its tracker state (global memory) and the scratch area are ours once
                // it has counted itself in
                while (agt_uniform(*(volatile int*)&sh.seq) < k) __builtin_amdgcn_s_sleep(1);
                asm volatile("" ::: "memory");                  // (state and scratch are LDS: read in order behind sh.seq)
            }
            RSTAMP(k, 2);
        };
        agt_pnp::pnp_body<float, PPL, decltype(before_state), true>(S.pnp, blk, sh, img, mask, so, late ? AGT_TRK_CHAIN_TIMEOUT : 0, before_state);
        // the state (LDS) is written: LDS operations of a wave complete in order, the other wave reads sh.seq first.  The
        // frame's record and the write-back below are global stores nobody in this launch waits for.
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        RSTAMP(k, 3);
        if (k == S.pnp_nf - 1 && lane < (int)(sizeof(AgtTrackState) / 8))
            reinterpret_cast<agt_word_t*>(S.pnp.track + blk)[lane] = reinterpret_cast<const agt_word_t*>(&sh.ts)[lane];      // (character copy: see agt_copy_words_to_lds)
        if (NWV > 1) {
            if (lane == 0) *(volatile int*)&sh.seq = k + 1;
        }
        // agt_track_host_frame: the host thread polls for this frame's record (behind the hand-over to the other wave: the
        // system-scope release waits for the record's stores to reach the host)
        // ONLY the launch's last frame is reported: frames k and k + 1 are solved by different waves, nothing orders
        // their stores, and a word that steps back from `want` to `want - 1` would leave the host polling until its time-out.  The
        // host waits for the newest frame only, launches are stream-ordered, and a launch's last frame is written by one wave.
        if (S.pnp.host_seq && blk == 0 && k == S.pnp_nf - 1) agt_host_seq_store(S.pnp.host_seq, S.pnp.host_seq_base + (unsigned long long)k, lane == 0);
    }
}

// ---- PnP role for n > 64 (split launches only): the workgroup's four waves share each frame's solve (agt_pnp_body.h, COOP).
// Frames are taken in order by the whole workgroup; the tracker state lives in LDS as above, wave 0 updates it, a barrier at
// the head of each frame hands it to the others.  A frame without a guess (first frame, after a gate rejection) is solved by
// wave 0 alone with four points per lane -- the DLT initialisation is one-wave code.  A chained wait (not used by the split
// launches of today, kept so the tables mean the same everywhere) is polled by one lane and its outcome shared through LDS.
__device__ __forceinline__ void pnp_role_coop(const AgtStepParams& S, const AgtStepTables& T, KTables KT, int blk, agt_pnp::PnpShared& sh)
{
    // Round 5: the solver bodies of the stand-alone pnp_coop_kernel / lk_pnp_coop_kernel -- tracker state in GLOBAL
    // memory (P.track[blk]), read by every wave at the head of a frame and written by wave 0 at its end -- instead of the
    // LDS-resident state of rounds 3-4 (154 VGPR spills, 1,600 B of scratch: one function with the one-wave solver, 512 registers
    // were not enough).  A frame costs the stream 376 B of state traffic and two barriers more; 129 spills / 376 B of scratch are
    // left (the cooperative loop ALONE needs 365 registers: it is the PPL = 4 one-wave body beside it, inside a loop, that does
    // not fit -- two separate loops for the two solvers spill the same).  Frame k's pointers come straight from the kernel arguments.
    const int tid = (int)threadIdx.x, wave = tid >> 6;
    int late = 0;
#pragma nounroll
    for (int k = 0; k < S.pnp_nf; k++) {
        // frame k - 1's state update (wave 0, global stores) is complete and visible to the other waves of the workgroup: the
        // barrier orders it, the acquire below drops this CU's cached copy of the lines
        __syncthreads();
        const void* img = (const void*)KT->pnp.img[k]; const uint8_t* mask = KT->pnp.mask[k]; double* so = KT->pnp.so[k];
        const unsigned* wait = KT->pnp.wait[k]; const unsigned target = (unsigned)KT->pnp.target[k];
        if (wait) {
            if (!late) {
                if (tid == 0) {
                    const int* fault = &S.pnp.track[blk].chain_fault;
                    unsigned polls = 0;
                    int timed_out = 0;
                    while ((int)(__hip_atomic_load(wait + blk, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - target) < 0) {
                        __builtin_amdgcn_s_sleep(4);
                        if (++polls > AGT_CHAIN_POLLS || ((polls & 15) == 1 && __hip_atomic_load(fault, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) { timed_out = 1; break; }
                    }
                    *(volatile int*)&sh.late = timed_out;
                }
                __syncthreads();
                late = agt_uniform(*(volatile int*)&sh.late);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        const bool guess = agt_uniform(__hip_atomic_load(&S.pnp.track[blk].has_guess, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) != 0 && S.pnp.enhance_ape;
        __syncthreads();        // every wave has read the decision before wave 0 (alone, without a guess) may rewrite has_guess
        const int xf = late ? AGT_TRK_CHAIN_TIMEOUT : 0;
        // The solver sees the parameters and its thread index through values the compiler cannot follow across frames (an empty
        // volatile asm on the kernel-argument pointer, the stream index and -- inside the body -- the thread index): everything it derives
        // from them is computed per frame.  Hoisted out of the frame loop those values (reciprocals of the camera constants, lane-selected
        // polynomial coefficients, addresses) were alive across BOTH solver bodies: 129 VGPRs spilled, 376 B of scratch.
        KParams KF = kernarg_params();
        int bk = blk;
        asm volatile("" : "+s"(KF), "+s"(bk));
        const AgtPnpParams& PF = *(const AgtPnpParams*)&KF->pnp;
        if (guess) agt_pnp::pnp_body<float, 1, agt_pnp::PnpNoHook, false, PNP_COOP, true>(PF, bk, sh, img, mask, so, xf);
        else if (wave == 0) agt_pnp::pnp_body<float, agt_pnp::MAX_PPL, agt_pnp::PnpNoHook, false, 1, true>(PF, bk, sh, img, mask, so, xf);
        if (wave == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");          // the state (and the record) of frame k are written before the next frame's barrier lets anybody read them
            if (S.pnp.host_seq && blk == 0 && k == S.pnp_nf - 1) agt_host_seq_store(S.pnp.host_seq, S.pnp.host_seq_base + (unsigned long long)k, tid == 0);
        }
    }
}

// One heterogeneous launch: block ranges [LK | PnP | pyr stage 0 | stage 1 | ..].
// OCC: waves per SIMD the register allocation must leave room for (1 = unconstrained).
template <int WIN, int NW, int NLEV, bool PNP, int OCC>
__global__ __launch_bounds__(STEP_THREADS) __attribute__((amdgpu_waves_per_eu(OCC))) void step_kernel(const AgtStepParams S, const AgtStepTables T)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    KParams KS = kernarg_params();
    KTables KT = kernarg_tables();
    // Workgroups are dispatched in index order: the long serial chains (LK, then PnP) take the lowest indices so they
    // start at t = 0 and the short, bandwidth-bound pyramid tiles fill in around them.  LK comes first: in a chained
    // launch the PnP workgroups wait for corners of the LK role, which must therefore never queue up behind them.
    int blk = blockIdx.x;
    SSTAMP_MIN(6);                                   // earliest entry of any block
    if (blk < S.n_lk) {
        if (blk == 0) SSTAMP_SET(2);
        lk_role<WIN, NW, NLEV, STEP_THREADS>(S, T, KS, KT, blk, lds);
        if (blk == 0) SSTAMP_SET(3);
        SSTAMP_MAX(4);
        return;
    }
    blk -= S.n_lk;
    if (PNP && blk < S.n_pnp) {
        // fused path (n <= 64): two waves alternate over the frames (see pnp_role); the whole workgroup sets their scratch up
        agt_pnp::PnpShared& sh = *reinterpret_cast<agt_pnp::PnpShared*>(lds);
        // (32-bit words here: the same copy in 64-bit words leaves step_kernel<21,4,6> a dead 48-byte stack object -- a private segment
        // for every launch; tests/test_kernel_resources.py holds every kernel of the library to none)
        agt_copy_halfwords_to_lds(sh.tab, (const void*)(const __attribute__((address_space(4))) void*)&KT->pnp, (int)(sizeof(AgtPnpTables) / 4), (int)threadIdx.x, STEP_THREADS);
        if (threadIdx.x == 0) { sh.seq = 0; sh.late = 0; }
        agt_copy_words_to_lds(&sh.ts, S.pnp.track + blk, (int)(sizeof(AgtTrackState) / 8), (int)threadIdx.x, STEP_THREADS);
        __syncthreads();
        const int wave = (int)(threadIdx.x / AGT_WAVE);
        if (wave >= 2) return;
        if (blk == 0 && wave == 0) SSTAMP_SET(0);
        pnp_role<1, 2>(S, T, KT, blk, sh, wave);
        if (blk == 0 && wave == 0) SSTAMP_SET(1);
        return;
    }
    if (PNP) blk -= S.n_pnp;
    pyr_role<false>(KS, KT, blk, (PNP ? S.n_pnp : 0) + S.n_lk, lds);
    SSTAMP_MAX(5);
}

// ---- split mode (more corners in flight than the fused launch takes): one kernel per role, so that each has its own
// register allocation, block shape and LDS size.  Same role code, same tables.
// LK: one corner per workgroup of 64 * NW threads (the shape of the stand-alone lk_kernel; as a role of the 256-thread
// step_kernel the one-wave-per-corner variant needed 240 B of scratch and ran at half the speed)
template <int WIN, int NW, int NLEV, int OCC>
__global__ __launch_bounds__(AGT_WAVE * NW) __attribute__((amdgpu_waves_per_eu(OCC))) void lk_group_kernel(const AgtStepParams S, const AgtStepTables T)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    lk_role<WIN, NW, NLEV, AGT_WAVE * NW, true>(S, T, kernarg_params(), kernarg_tables(), blockIdx.x, lds);
}

#ifndef AGT_STEP_NOLICM_TU
#ifdef AGT_PYRG_NUM_VGPR     // experiment builds: register ceiling of the split pipeline's pyramid launch
#define AGT_PYRG_VGPR_ATTR __attribute__((amdgpu_num_vgpr(AGT_PYRG_NUM_VGPR)))
#else
#define AGT_PYRG_VGPR_ATTR
#endif
__global__ __launch_bounds__(agt_pyr::NT) AGT_PYRG_VGPR_ATTR void pyr_group_kernel(const AgtStepParams S, const AgtStepTables T)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    pyr_role<true>(kernarg_params(), kernarg_tables(), blockIdx.x, 0, lds);
}
#endif

// PPL: points per lane (n <= 64 * PPL)
template <int PPL>
__global__ __launch_bounds__(AGT_WAVE) void pnp_group_kernel(const AgtStepParams S, const AgtStepTables T)
{
    __shared__ agt_pnp::PnpShared sh;
    pnp_role<PPL>(S, T, kernarg_tables(), blockIdx.x, sh);
}

// ---- The two kernels of this file that are compiled WITHOUT the machine-level loop-invariant code motion (agt_step_nolicm.hip
// includes this file with AGT_STEP_NOLICM_TU defined; Makefile: -mllvm -disable-machine-licm): both run a FRAME loop around whole
// solver / tracker bodies, and the FP64 constants (register PAIRS: not rematerialisable) and addresses MachineLICM hoists in front
// of that loop stay alive across everything inside it -- pnp_group_coop_kernel 512 VGPRs + 376 B of scratch -> 382 / none (with the
// per-frame opaque inputs of pnp_role_coop), step_kernel<21,4,6> 501 VGPRs / 16 B -> 408 / none.  Everything else keeps the pass:
// the 20-frame c2 blocks lose 0.5 % without it (67.8 k against 68.2 k frames/s, three same-box runs each; profiles/r05_experiments.md).
#ifdef AGT_STEP_NOLICM_TU
__global__ __launch_bounds__(AGT_WAVE * PNP_COOP) void pnp_group_coop_kernel(const AgtStepParams S, const AgtStepTables T)
{
    __shared__ agt_pnp::PnpShared sh;
    pnp_role_coop(S, T, kernarg_tables(), blockIdx.x, sh);
}
#endif

#ifndef AGT_STEP_NOLICM_TU
// roles: AGT_STEP_ALL = one fused launch; AGT_STEP_PYR / AGT_STEP_LK = that role alone, from the kernel compiled without
// the FP64 PnP role (the one-wave-per-corner LK keeps its four waves per SIMD; each launch sizes its own LDS);
// AGT_STEP_PNP = the PnP role alone
template <int WIN, int NW>
hipError_t launch_step_t(hipStream_t stream, const AgtStepParams& S, const AgtStepTables& T, int roles)
{
    constexpr int CPB = STEP_THREADS / (AGT_WAVE * NW);
    AgtStepParams P = S;
    P.xshift = agt_chip_current().xshift; P.rsv_ = 0; P.lk.xshift = P.xshift;
    for (int s = 0; s < AGT_MAX_LEVELS - 1; s++) P.pyr[s].xshift = P.xshift;
    if (roles == AGT_STEP_PNP) {
        if (P.n_pnp <= 0) return hipSuccess;
        if (P.pnp.n <= AGT_WAVE) { hipLaunchKernelGGL((pnp_group_kernel<1>), dim3(P.n_pnp), dim3(AGT_WAVE), 0, stream, P, T); return hipGetLastError(); }
        // 64 < n <= 256 (the four-wave solver): the group's frames inside ONE launch (pnp_group_coop_kernel).  With the frame loop around
        // both solver bodies the function needs more than its 512 registers: 154 VGPR spills / 1,600 B of scratch with the LDS-resident
        // state of rounds 3-4, 129 / 376 B with the state in global memory (round 5, shipped).  The spill-free alternative -- the
        // stand-alone pnp_coop_kernel once per frame (419 registers, no scratch) -- was measured and loses: 240 corners, 16 frames per
        // group, us per step group / per frame: 2 streams 28.5 / 36.9, 4: 37.6 / 44.4, 8: 45.4 / 52.5 (the host enqueues one more launch
        // per frame; profiles/r05_experiments.md).  Zero scratch needs the guess-less initialisation (DLT / homography, one-wave code
        // with four points per lane) in cooperative form, so that no PPL = 4 body sits beside the cooperative one: open.
        // The knobs build can take the per-frame path (AGT_PNP_COOP_GROUP=0).
        bool group = true;
#ifdef AGT_DEBUG_KNOBS
        { static const int f = [] { const char* e = getenv("AGT_PNP_COOP_GROUP"); return e ? atoi(e) : 1; }(); group = f != 0; }
#endif
        if (group) return agt_launch_pnp_group_coop(stream, P, T);
        for (int k = 0; k < P.pnp_nf; k++) {
            if (T.pnp.wait[k]) return hipErrorInvalidValue;          // (split launches are ordered by events: no in-kernel wait to honour)
            AgtPnpParams q = P.pnp;
            q.img = T.pnp.img[k]; q.mask = T.pnp.mask[k]; q.state_out = T.pnp.so[k];
            q.host_seq = (k == P.pnp_nf - 1) ? P.pnp.host_seq : nullptr;
            q.host_seq_base = P.pnp.host_seq_base + (unsigned long long)k;
            const hipError_t e = agt_launch_pnp(stream, q, P.n_pnp);
            if (e != hipSuccess) return e;
        }
        return hipSuccess;
    }
    if (!(roles & AGT_STEP_PNP)) { P.n_pnp = 0; P.pnp_nf = 0; }
    if (!(roles & AGT_STEP_LK)) { P.n_lk = 0; P.lk_nf = 0; }
    if (!(roles & AGT_STEP_PYR)) { for (int s = 0; s < AGT_MAX_LEVELS - 1; s++) { P.n_pyr[s] = 0; P.pyr_nf[s] = 0; } }
#ifdef AGT_DEBUG_KNOBS      // diagnostic library only: AGT_LK_RS=0 keeps every corner on the general LK body
    { static const int rs = [] { const char* e = getenv("AGT_LK_RS"); return e ? atoi(e) : 1; }(); if (!rs) P.lk.flags |= 0x10000; }
#endif
#ifdef AGT_DEBUG_KNOBS      // diagnostic library only (make dbg): drop roles from the launch to time the others
    { static const int skip = [] { const char* e = getenv("AGT_STEP_SKIP"); return e ? atoi(e) : 0; }();
      if (skip & 1) P.n_pnp = 0; if (skip & 2) P.n_lk = 0; if (skip & 4) { for (int s = 0; s < AGT_MAX_LEVELS - 1; s++) P.n_pyr[s] = 0; } }
#endif
    size_t lds = 0;
    int blocks = 0;
    // (the register-rolling forms of the pyramid passes, pyr[s].pad != 0, use no LDS: a launch of them alone must not be held to
    // the tiled kernels' occupancy)
    size_t pyr_lds = 0;
    for (int s = 0; s < AGT_MAX_LEVELS - 1; s++)
        if (P.n_pyr[s] > 0) {
            blocks += P.n_pyr[s];
            const size_t need = P.pyr[s].pad ? 0 : ((P.pyr_fused && s == 0) ? (size_t)agt_pyr2::PYR2_LDS_BYTES : (size_t)agt_pyr::PYR_LDS_BYTES);
            pyr_lds = pyr_lds > need ? pyr_lds : need;
        }
    lds = pyr_lds;
    if (P.n_lk > 0) {
        const long corners = (long)P.lk.n * P.lk_B;
        P.n_lk = (int)((corners + CPB - 1) / CPB);
        blocks += P.n_lk;
        const size_t per = P.lk.max_level < 3 ? lk_role_lds<WIN, NW, 3>(P.lk.max_level + 1) : lk_role_lds<WIN, NW, AGT_MAX_LEVELS>(P.lk.max_level + 1);
        const size_t need = per * (NW == 1 ? CPB : 1);
        lds = lds > need ? lds : need;
    }
    if (P.n_pnp > 0) { blocks += P.n_pnp; lds = lds > sizeof(agt_pnp::PnpShared) ? lds : sizeof(agt_pnp::PnpShared); }
    if (blocks == 0) return hipSuccess;
    const bool small = P.lk.max_level < 3;
    if (roles == AGT_STEP_PYR) {
        hipLaunchKernelGGL(pyr_group_kernel, dim3(blocks), dim3(agt_pyr::NT), pyr_lds, stream, P, T);
        return hipGetLastError();
    }
    if (roles == AGT_STEP_LK) {
#ifndef AGT_DEBUG_KNOBS
        // The one-wave-per-corner LK role as a group launch (in-kernel frame loop) is a measured dead end -- round 4, spill-free at
        // last (two frame loops in sequence instead of one with both tracker bodies): 64 streams 54.2 us per step against 42.1 with
        // per-frame launches in two half-batch chains, 24 streams 36.9 against 30.6; the corners that need 30 iterations need them
        // in EVERY frame, so a corner's own chain over the group is as long as the chain of per-frame maxima.  Sub-groups of 2 / 4 / 8
        // frames per launch (to save the ~4 us launch boundary and ~2 us kernel start per frame) lose as well: 64 streams 55.3 / 52.4 /
        // 51.5 us, 24 streams 39.9 / 37.5 / 35.9 against 30.7 -- inside the frame loop a frame costs the one-wave kernel ~10 us more
        // than as a launch of its own.  Only the knobs build (AGT_SPLIT_LK_GROUP=2) still carries the instantiation.
        if constexpr (NW == 1) return hipErrorInvalidValue;
        else {
#endif
        constexpr int OCCL = (WIN == 21 && NW == 1) ? AGT_LKG_OCC : 1;
        const long corners = (long)P.lk.n * P.lk_B;
        size_t per = small ? lk_role_lds<WIN, NW, 3>(P.lk.max_level + 1) : lk_role_lds<WIN, NW, AGT_MAX_LEVELS>(P.lk.max_level + 1);
        if (NW == 1 && (size_t)P.lk.lds_min > per) per = (size_t)P.lk.lds_min;      // agt_lk_occupancy_cu also caps the one-wave group launch (knobs build only: ADVICE r5)
        const unsigned grid8 = agt_xcd_grid(corners, P.xshift);           // (XCD-aware corner order: lk_role; blocks past the last corner exit)
        if (small) hipLaunchKernelGGL((lk_group_kernel<WIN, NW, 3, OCCL>), dim3(grid8), dim3(AGT_WAVE * NW), per, stream, P, T);
        else hipLaunchKernelGGL((lk_group_kernel<WIN, NW, AGT_MAX_LEVELS, OCCL>), dim3(grid8), dim3(AGT_WAVE * NW), per, stream, P, T);
        return hipGetLastError();
#ifndef AGT_DEBUG_KNOBS
        }
#endif
    }
    if (!(roles & (AGT_STEP_LK | AGT_STEP_PNP))) return hipErrorInvalidValue;       // (LK | PnP without the pyramid role: diagnostics)
    // The FP64 PnP role gets the whole register file (256 VGPR + AGPR spill space): one workgroup per CU, which is why the fused
    // launch is only used while <= 256 corners are in flight (agt_step_fits).  Round 2 also shipped an OCC = 2 build for 257-2048
    // corners (registers capped at 256: ~500 VGPR spills, 1 KB of scratch in the PnP role); round 3's split mode with the LK role
    // as one group launch beats it at every stream count (profiles/r03_stream_sweep.txt), so it is gone.
#ifndef AGT_DEBUG_KNOBS
    // (the fused launch needs <= 256 corners in flight -- agt_step_fits -- and <= 1024 corners are always tracked by four waves each
    // -- agt_lk_wide --, so step_kernel<21, 1, *> was unreachable object code: 476 VGPRs, scratch.  Only the knobs build, whose
    // AGT_LK_WIDE_MAX can force the one-wave body on small batches, still instantiates it: VERDICT r4 #4)
    if constexpr (NW == 1) return hipErrorInvalidValue;
    else {
#endif
    if (small) hipLaunchKernelGGL((step_kernel<WIN, NW, 3, true, 1>), dim3(blocks), dim3(STEP_THREADS), lds, stream, P, T);
    else return agt_launch_step_deep(stream, P, T, NW, blocks, lds);
    return hipGetLastError();
#ifndef AGT_DEBUG_KNOBS
    }
#endif
}
#endif      // !AGT_STEP_NOLICM_TU

}  // namespace

#ifdef AGT_STEP_NOLICM_TU
hipError_t agt_launch_pnp_group_coop(hipStream_t stream, const AgtStepParams& P, const AgtStepTables& T)
{
    hipLaunchKernelGGL(pnp_group_coop_kernel, dim3(P.n_pnp), dim3(AGT_WAVE * PNP_COOP), 0, stream, P, T);
    return hipGetLastError();
}

// the fused step of pyramids with more than three levels (cv2's default maxLevel = 3 is four)
hipError_t agt_launch_step_deep(hipStream_t stream, const AgtStepParams& P, const AgtStepTables& T, int nw, int blocks, size_t lds)
{
#ifdef AGT_DEBUG_KNOBS      // (the one-wave-per-corner fused kernel: knobs build only, see launch_step_t)
    if (nw == 1) { hipLaunchKernelGGL((step_kernel<21, 1, AGT_MAX_LEVELS, true, 1>), dim3(blocks), dim3(STEP_THREADS), lds, stream, P, T); return hipGetLastError(); }
#endif
    if (nw != 4) return hipErrorInvalidValue;
    hipLaunchKernelGGL((step_kernel<21, 4, AGT_MAX_LEVELS, true, 1>), dim3(blocks), dim3(STEP_THREADS), lds, stream, P, T);
    return hipGetLastError();
}
#else

bool agt_step_supported(int win) { return win == 21; }
// Fused launch (all roles in one kernel, PnP chained to LK through arrival counters) while <= 256 corners are in flight:
// one 256-thread workgroup per CU (the PnP role's registers) holds every LK workgroup of <= 5 streams of 48 corners at once.
// Beyond that the same pipeline runs in split mode, one launch per role (AGT_STEP_PYR / _LK / _PNP): round-3 sweep at 16 frames
// per group, us per step fused (round 2's OCC = 2 build) vs split with the LK role as one group launch: 6 streams 23.2 / 19.6,
// 8: 24.6 / 20.4, 12: 33.9 / 25.4, 16: 38.0 / 33.5, 21: 46.7 / 36.9, 32: 42.9 / 40.6, 42: 56.0 / 46.2 (profiles/r03_stream_sweep.txt).
bool agt_step_fits(int n, int B)
{
    // one LK workgroup per CU, all co-resident (the PnP role's registers leave room for one workgroup per CU): the chip's CU count,
    // 256 on a whole MI355X
    long cap = agt_chip_current().cus;
#ifdef AGT_DEBUG_KNOBS
    { static const long f = [] { const char* e = getenv("AGT_STEP_MAX_CORNERS"); return e ? atol(e) : 0L; }(); if (f > 0) cap = f; }
#endif
    return n <= AGT_WAVE && (long)n * B <= cap;
}

hipError_t agt_launch_step(hipStream_t stream, const AgtStepParams& S, const AgtStepTables& T, int win, int roles)
{
    if (win != 21) return hipErrorInvalidValue;
    const bool wide = S.n_lk > 0 ? agt_lk_wide(S.lk.n, S.lk_B) : true;
    return wide ? launch_step_t<21, 4>(stream, S, T, roles) : launch_step_t<21, 1>(stream, S, T, roles);
}
#endif      // !AGT_STEP_NOLICM_TU
