// agt_step_args.h -- what the translation units of the step launches share (agt_step.hip: the fused step and the group launches of the
// split pipeline; agt_step_dense.hip: the dense stage's chained launches): the kernel-argument views and the LK role's LDS size.
#pragma once
#include <cstdlib>
#include <cstring>
#include <cstddef>
#include "agt_pyramid2_body.h"
#include "agt_pyramid3_body.h"
#include "agt_pyramid4_body.h"
#include "agt_lk_rs_body.h"
#include "agt_lk_chain_body.h"
#include "agt_pnp_body.h"
#include "agt_dense_body.h"

namespace {

constexpr int STEP_THREADS = 256;
// chained launch: polls of an arrival counter before the waiting wave gives up (each poll is a device-scope load + s_sleep, >= 0.5 us)
constexpr unsigned AGT_CHAIN_POLLS = 1u << 16;

typedef const __attribute__((address_space(4))) AgtStepParams* KParams;
typedef const __attribute__((address_space(4))) AgtStepTables* KTables;

// The per-frame tables (second kernel argument) are indexed with run-time frame numbers; they are read straight
// from the kernel-argument segment -- indexing a by-value argument dynamically forces a copy into scratch.
__device__ __forceinline__ KParams kernarg_params() { return (KParams)__builtin_amdgcn_kernarg_segment_ptr(); }
__device__ __forceinline__ KTables kernarg_tables()
{
    static_assert(alignof(AgtStepTables) == 8 && alignof(AgtStepParams) == 8, "kernel-argument layout");
    return (KTables)((const __attribute__((address_space(4))) char*)__builtin_amdgcn_kernarg_segment_ptr() + ((sizeof(AgtStepParams) + 7) & ~(size_t)7));
}

// LDS of one corner of the LK role: the tracker's tiles (general body, or the frame-chained body where that one applies)
// and, behind them, the copy of the per-frame tables
template <int WIN, int NW, int NLEV>
__host__ __device__ constexpr size_t lk_role_lds(int levels)
{
    size_t body = agt_lk::lk_lds_bytes<WIN, NW>(levels);
    if (WIN == 21 && NW == 4 && agt_lk::lk_chain_lds_bytes<NLEV>() > body) body = agt_lk::lk_chain_lds_bytes<NLEV>();
    return (body + sizeof(AgtLkTables) + 15) & ~(size_t)15;
}

// waves of the cooperative pose solve (64 < n <= 256 corners)
constexpr int PNP_COOP = agt_pnp::MAX_PPL;

}  // namespace
