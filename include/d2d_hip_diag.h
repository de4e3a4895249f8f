/*
 * d2d_hip_diag.h - measurement equipment around libd2d_hip.so.  NOT part of the drop-in boundary (that is d2d_hip.h, which
 * INTEGRATION.md binds): nothing here replaces an interface of the reference.  Two things live here:
 *
 *  (1) tuning keys / values that only a DIAGNOSTIC build of libd2d_hip.so accepts
 *      (D2D_BUILD_DIAG=1 python -m gym_d2d_amd.build; a release build answers D2D_ERR_UNSUPPORTED): the A/B shapes the
 *      kernels were tuned against, the ablation switch of the generic step kernel and the phase stamps of tools/phase_times.py;
 *  (2) the write-ceiling probe, a library of its own (libd2d_probe.so, csrc/d2d_probe.hip): what bench.py's
 *      `box_write_ceiling` and the placement studies under profiles/ were measured with.
 */
#ifndef D2D_HIP_DIAG_H
#define D2D_HIP_DIAG_H

#include "d2d_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- (1) diagnostic builds of libd2d_hip.so: further d2d_set_tuning keys ------------------------------------------------ */
#define D2D_TUNE_OBS_VARIANT 4   /* 0 (default): flat 32 KB slabs, T staged in LDS; 1: T read from global; 3: the row-aligned
                                    kernel of rounds 1-3                                                                    */
#define D2D_TUNE_STEP_ABLATE 9   /* the one key that DOES change results: bit mask of kernel parts to skip (1 interferer walk,
                                    2 mask build, 4 mask clear, 8 result stores, 16 table store, 32 rb/pwr stores, 64 pass-0/1
                                    barriers, 128 per-env loads hit L2, ...) so that the rest can be timed                   */
#define D2D_TUNE_OBS_STAGGER 16  /* wave w of an obs workgroup sleeps w * value * 64 clocks before its stores; 0 = off       */
/* ... and further VALUES of release keys: D2D_TUNE_OBS_NONTEMPORAL 2 sc1, 3 sc0 sc1, 4 sc0 sc1 nt, 5 sc1 nt (the scope bits
 * of the gfx942+ store encoding); D2D_TUNE_STEP_WALK 1 = membership masks, flattened walk.                                  */

/* diagnostic builds only: copy the shader-clock stamps of the last step (D2D_TUNE_STEP_ABLATE bit 8192) to the host          */
int d2d_debug_stamps(d2d_handle* h, void* host, size_t bytes);

/* ---- (2) libd2d_probe.so: streaming-store probe ---------------------------------------------------------------------------
 * Writes a scratch buffer of `bytes` (>= 64 MiB; rounded down to whole groups of eight 512-row regions) `iters` times with
 * pure fill kernels - nothing to compute - in a family of store geometries that contains the obs kernel's own, and with
 * hipMemsetAsync, and reports the BEST sustained rate: the box's write ceiling as far as these kernels can demonstrate one.
 * per_variant (n entries, may be NULL) receives the first n rates (variant v: block {768,1024,512,256}[v & 3], rows per
 * workgroup {2,4,8,32}[(v >> 2) & 3], nontemporal unless v & 16; index D2D_PROBE_VARIANTS = hipMemsetAsync).  Runs on the null
 * stream of `device` and synchronises the device around itself.  Returns 0, or 1 with text in d2d_probe_last_error().        */
#define D2D_PROBE_VARIANTS 32
int d2d_probe_write_variants(int32_t device, size_t bytes, int32_t iters, double* best_gb_per_s, double* per_variant, int32_t n);
/* One member of the family with the obs kernel's TIMING structure added: variant = the geometry index above + 32 (every
 * workgroup first stages one row of a table through LDS behind a barrier and stores what it reads back) + 64 (wave w sleeps
 * w * stagger * 64 clocks before its first store) + 128 * k (k = 1 .. 4: the store's cache policy sc1 / sc0 sc1 / sc0 sc1 nt /
 * sc1 nt instead of nt or plain).  dst_dev = NULL writes a scratch buffer of `bytes`; a device pointer writes THAT memory.  */
int d2d_probe_write_staged(int32_t device, void* dst_dev, size_t bytes, int32_t variant, int32_t stagger, int32_t iters,
                           double* gb_per_s);
const char* d2d_probe_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* D2D_HIP_DIAG_H */
