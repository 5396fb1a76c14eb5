// agt_step_nolicm.hip -- the two kernels of agt_step.hip that are compiled without the machine-level loop-invariant code motion
// (Makefile: -mllvm -disable-machine-licm on this file only): pnp_group_coop_kernel and step_kernel<21,4,6>, with their launchers.
// Reason, numbers and the rest of the file: agt_step.hip, at pnp_group_coop_kernel.
#define AGT_STEP_NOLICM_TU
#undef AGT_STEP_STAMPS          // (the diagnostic build's role stamps and their accessors live in agt_step.hip's own object)
#undef AGT_STEP_LK_STAMPS
#include "agt_step.hip"
