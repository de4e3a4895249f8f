// 16-byte global stores under a chosen cache policy, shared by the LinearObs expansion kernels (d2d_obs.hip) and the
// write-ceiling probe (d2d_probe.hip).  gfx950 only.
#pragma once
#include <hip/hip_runtime.h>

namespace d2d {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// One 16-byte store under a chosen cache policy (gfx942+ scope / streaming bits of the global_store encoding): 0 plain, 1 nt
// (what __builtin_nontemporal_store emits), 2 sc1 (agent scope), 3 sc0 sc1 (system scope: written through), 4 sc0 sc1 nt, 5 sc1 nt.
// The obs stream is written once and never read by the GPU again; which policy drains it fastest is measured, not assumed
// (d2d_probe_write_staged, D2D_TUNE_OBS_NONTEMPORAL).
template <int POLICY>
__device__ __forceinline__ void store16(f32x4* p, f32x4 v) {
    if (POLICY == 0) *p = v;
    else if (POLICY == 1) __builtin_nontemporal_store(v, p);
    // (no "memory" clobber: nothing in these kernels reads what they store, and a clobber would pin every LDS read of the next
    // row behind the store of this one - the obs kernel with the clobber lost 13 % where the fill, which has no reads, gained 5 %)
    else if (POLICY == 2) asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v));
    else if (POLICY == 3) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(v));
    else if (POLICY == 4) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1 nt" ::"v"(p), "v"(v));
    else asm volatile("global_store_dwordx4 %0, %1, off sc1 nt" ::"v"(p), "v"(v));
}

}  // namespace d2d
