// libd2d_probe.so - the streaming-store probe behind bench.py's `box_write_ceiling` and the round-3/4 placement studies.
// Measurement equipment, NOT part of the drop-in boundary: a library of its own (declared in include/d2d_hip_diag.h) so that
// libd2d_hip.so exports the product ABI and nothing else.  gfx950 only; no handle: the probe runs on the null stream of the
// given device and synchronises the device around itself.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>

#include "d2d_store.h"

namespace d2d {

// Streaming-store probe (libd2d_probe.so, include/d2d_hip_diag.h): pure fill kernels - no table, no LDS, no source selection - in a family
// of store geometries that contains the obs kernel's own (one thread per float4 column of a 6N-float row: 768 threads at
// N = 512, two rows per workgroup, XCD-grouped dispatch order, nontemporal 16-byte stores).  The best of the family (and
// of the runtime's own hipMemsetAsync) is the box's write ceiling as far as this library can demonstrate one.
// STAGE reproduces the obs kernel's TIMING structure around the same stores: bit 0 - every workgroup first stages one row
// (T float4, 12 KiB at 768 threads) from `src` into LDS behind a barrier and stores what it reads back from LDS; bit 1 - wave w of
// the workgroup sleeps w * stagger x 64 clocks before its first store.  (Does the obs kernel out-write its own geometry run as
// a plain fill - 7.19 vs 6.39 TB/s in round 3 - because its load + barrier phase spreads the waves' stores in time?)
template <int POLICY, int STAGE>
__global__ __launch_bounds__(1024) void fill_kernel(f32x4* dst, const f32x4* src, unsigned chunks, unsigned rows_per_wg, unsigned rows_per_env, int xcd,
                                                    float value, int stagger) {
    extern __shared__ __align__(16) float fill_lds[];
    unsigned env, chunk;
    if (xcd) {
        const unsigned bid = blockIdx.x, lane8 = bid & 7u, rest = bid >> 3;
        chunk = rest % chunks; env = (rest / chunks) * 8u + lane8;
    } else {
        env = blockIdx.x / chunks; chunk = blockIdx.x % chunks;
    }
    const unsigned T = blockDim.x;                             // float4 per row
    f32x4 v = {value, value, value, value};
    if (STAGE & 1) {
        f32x4* l4 = reinterpret_cast<f32x4*>(fill_lds);
        l4[threadIdx.x] = src[(size_t)env * T + threadIdx.x];
        __syncthreads();
        v = l4[(threadIdx.x + 1u) % T];
    }
    if (STAGE & 2) {
        const int n = (int)(threadIdx.x >> 6) * stagger;
        for (int k = 0; k < n; ++k) __builtin_amdgcn_s_sleep(1);
    }
    f32x4* o4 = dst + ((size_t)env * rows_per_env + (size_t)chunk * rows_per_wg) * T + threadIdx.x;
#pragma unroll 2
    for (unsigned i = 0; i < rows_per_wg; ++i) store16<POLICY>(o4 + (size_t)i * T, v);
}

// Variant v of the family: block in {768, 1024, 512, 256} x rows per workgroup in {2, 4, 8, 32} x {nt, plain}, XCD-grouped
// order; v == 0 is the obs kernel's geometry.  Bits 5-6 select the staged forms above (32: LDS stage + barrier, 64: per-wave
// sleep stagger of `stagger` x 64 clocks, 96: both); v / 128 = 1 .. 4 replaces the store's cache policy by sc1, sc0 sc1,
// sc0 sc1 nt, sc1 nt (store16).  "Envs" are regions of 512 rows; n_float4 is rounded DOWN to whole
// groups of 8 regions; returns the float4 actually written through *written.
int fill_variants() { return 4 * 4 * 2; }

hipError_t launch_fill(float* dst, size_t n_float4, float value, hipStream_t stream, int variant, size_t* written, const float* src, int stagger) {
    static const unsigned blocks[4] = {768, 1024, 512, 256}, rows[4] = {2, 4, 8, 32};
    const unsigned T = blocks[variant & 3], rows_per_wg = rows[(variant >> 2) & 3], rows_per_env = 512, chunks = rows_per_env / rows_per_wg;
    const bool nt = ((variant >> 4) & 1) == 0;
    const int stage = (variant >> 5) & 3;
    const int policy = (variant >> 7) > 0 ? (variant >> 7) + 1 : (nt ? 1 : 0);
    const size_t env_f4 = (size_t)rows_per_env * T;
    const size_t envs = (n_float4 / env_f4) & ~(size_t)7;
    if (written) *written = envs * env_f4;
    if (envs == 0) return hipSuccess;
    if ((stage & 1) && !src) return hipErrorInvalidValue;
    const dim3 grid((unsigned)(envs * chunks)), block(T);
    const size_t lds = (stage & 1) ? (size_t)T * 16 : 0;
    f32x4* d4 = reinterpret_cast<f32x4*>(dst);
    const f32x4* s4 = reinterpret_cast<const f32x4*>(src);
#define D2D_FILL(P, ST) hipLaunchKernelGGL((fill_kernel<P, ST>), grid, block, lds, stream, d4, s4, chunks, rows_per_wg, rows_per_env, 1, value, stagger)
#define D2D_FILL_P(P)                                                                                               \
    switch (stage) { case 0: D2D_FILL(P, 0); break; case 1: D2D_FILL(P, 1); break; case 2: D2D_FILL(P, 2); break; default: D2D_FILL(P, 3); break; }
    switch (policy) {
        case 0: D2D_FILL_P(0); break; case 1: D2D_FILL_P(1); break; case 2: D2D_FILL_P(2); break;
        case 3: D2D_FILL_P(3); break; case 4: D2D_FILL_P(4); break; default: D2D_FILL_P(5); break;
    }
#undef D2D_FILL_P
#undef D2D_FILL
    return hipGetLastError();
}

}  // namespace d2d

namespace {

thread_local std::string g_probe_error;

int probe_fail(const std::string& msg) noexcept {
    try { g_probe_error = msg; } catch (...) { }
    return 1;
}

struct DeviceScope {
    int prev = -1;
    hipError_t err;
    explicit DeviceScope(int want) { err = hipGetDevice(&prev); if (err == hipSuccess && prev != want) err = hipSetDevice(want); else prev = -1; }
    ~DeviceScope() { if (prev >= 0) (void)hipSetDevice(prev); }
};

// one variant of the fill family over `dst`: GB/s sustained over `iters` launches behind one warm-up launch
int time_fill(float* dst, size_t bytes, const float* src, int variant, int stagger, int iters, hipEvent_t e0, hipEvent_t e1, double* rate) {
    size_t written = bytes / 16;
    hipError_t err = hipSuccess;
    for (int k = -1; k < iters && err == hipSuccess; ++k) {                     // k == -1: warm-up / page touch
        if (k == 0) err = hipEventRecord(e0, nullptr);
        if (err != hipSuccess) break;
        if (variant < 0) err = hipMemsetAsync(dst, k & 0xFF, written * 16, nullptr);     // the runtime's own fill
        else err = d2d::launch_fill(dst, bytes / 16, (float)k, nullptr, variant, &written, src, stagger);
        if (variant < 0 && k == -1 && err == hipSuccess) {                      // same byte count as the family's variant 0 writes
            size_t w0 = 0;
            err = d2d::launch_fill(dst, bytes / 16, 0.0f, nullptr, 0, &w0, nullptr, 0);
            written = w0;
        }
    }
    if (err == hipSuccess) err = hipEventRecord(e1, nullptr);
    if (err == hipSuccess) err = hipEventSynchronize(e1);
    float ms = 0.f;
    if (err == hipSuccess) err = hipEventElapsedTime(&ms, e0, e1);
    if (err != hipSuccess) return probe_fail(std::string("write probe: ") + hipGetErrorString(err));
    *rate = (double)(written * 16) * iters / (ms * 1e-3) / 1e9;
    return 0;
}

const size_t kGroup = (size_t)8 * 512 * 1024 * 16;         // whole groups of 8 regions of 512 rows (64 MiB at the widest row)

}  // namespace

extern "C" {

const char* d2d_probe_last_error(void) { return g_probe_error.c_str(); }

int d2d_probe_write_variants(int32_t device, size_t bytes, int32_t iters, double* best_gb_per_s, double* per_variant, int32_t n) try {
    if (!best_gb_per_s || iters < 1 || n < 0 || (n > 0 && !per_variant)) return probe_fail("bad argument");
    if (bytes < kGroup) return probe_fail("the probe needs at least 64 MiB");
    DeviceScope scope(device);
    if (scope.err != hipSuccess) return probe_fail(std::string("hipSetDevice: ") + hipGetErrorString(scope.err));
    float* tmp = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    hipError_t err = hipDeviceSynchronize();
    if (err == hipSuccess) err = hipMalloc(&tmp, bytes);
    if (err == hipSuccess) err = hipEventCreate(&e0);
    if (err == hipSuccess) err = hipEventCreate(&e1);
    const int variants = d2d::fill_variants();
    double best = 0.0;
    int rc = err == hipSuccess ? 0 : probe_fail(std::string("d2d_probe_write_variants: ") + hipGetErrorString(err));
    for (int v = 0; v <= variants && rc == 0; ++v) {          // v == variants: the runtime's own fill (hipMemsetAsync)
        double rate = 0.0;
        rc = time_fill(tmp, bytes, nullptr, v < variants ? v : -1, 0, iters, e0, e1, &rate);
        if (v < n) per_variant[v] = rate;
        if (rate > best) best = rate;
    }
    if (e0) (void)hipEventDestroy(e0);                        // every exit releases what was allocated
    if (e1) (void)hipEventDestroy(e1);
    if (tmp) (void)hipFree(tmp);
    if (rc) return rc;
    *best_gb_per_s = best;
    return 0;
} catch (...) { return probe_fail("C++ exception"); }

int d2d_probe_write_staged(int32_t device, void* dst_dev, size_t bytes, int32_t variant, int32_t stagger, int32_t iters, double* gb_per_s) try {
    if (!gb_per_s || iters < 1) return probe_fail("bad argument");
    if (variant < 0 || variant >= 5 * 4 * d2d::fill_variants()) return probe_fail("variant must be in [0, 640)");
    if (stagger < 0 || stagger > 64) return probe_fail("stagger must be in [0, 64]");
    if (bytes < kGroup) return probe_fail("the probe needs at least 64 MiB");
    DeviceScope scope(device);
    if (scope.err != hipSuccess) return probe_fail(std::string("hipSetDevice: ") + hipGetErrorString(scope.err));
    // the staged forms read one 1024-float4 row per region of 512 rows: a table 1 / 512 of the destination, as in the obs kernel
    float *tmp = static_cast<float*>(dst_dev), *src = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    const size_t src_bytes = bytes / 512 + (size_t)1024 * 16;
    hipError_t err = hipDeviceSynchronize();
    if (err == hipSuccess) err = hipMalloc(&src, src_bytes);
    if (err == hipSuccess && !dst_dev) err = hipMalloc(&tmp, bytes);
    if (err == hipSuccess) err = hipMemsetAsync(src, 0, src_bytes, nullptr);
    if (err == hipSuccess) err = hipEventCreate(&e0);
    if (err == hipSuccess) err = hipEventCreate(&e1);
    int rc = err == hipSuccess ? time_fill(tmp, bytes, src, variant, stagger, iters, e0, e1, gb_per_s)
                               : probe_fail(std::string("d2d_probe_write_staged: ") + hipGetErrorString(err));
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    if (src) (void)hipFree(src);
    if (!dst_dev && tmp) (void)hipFree(tmp);
    return rc;
} catch (...) { return probe_fail("C++ exception"); }

}  // extern "C"
