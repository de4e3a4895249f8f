// What does s_memtime count?  One wave spins for a fixed number of ticks; HIP events give the wall time.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/clock_rate.hip -o /tmp/clock_rate && /tmp/clock_rate
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ void spin(unsigned long long ticks, unsigned long long* out) {
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    unsigned long long t = t0;
    while (t - t0 < ticks) t = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) out[blockIdx.x] = t - t0;
}

// a VALU-saturating loop: n dependent-free FMAs per lane on every SIMD, to see the clock under load
__global__ void fma_burn(int iters, float* out, unsigned long long* ticks) {
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    float a = threadIdx.x, b = 1.0001f, c = 0.5f, d = 0.25f;
    for (int i = 0; i < iters; ++i) { a = fmaf(a, b, c); d = fmaf(d, b, c); c = fmaf(c, b, a); b = fmaf(b, 1.0f, 1e-9f); }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a + b + c + d;
    if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
}

int main() {
    unsigned long long* d; float* f;
    hipMalloc(&d, 8 * 65536); hipMalloc(&f, 4 * 4096 * 512);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (unsigned long long ticks : {1000000ull, 10000000ull}) {
        spin<<<1, 64>>>(ticks, d);
        hipDeviceSynchronize();
        hipEventRecord(e0); spin<<<1, 64>>>(ticks, d); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("spin %llu ticks: %.3f us -> %.1f ticks/us\n", ticks, ms * 1e3, ticks / (ms * 1e3));
    }
    for (int iters : {20000, 200000}) {
        fma_burn<<<4096, 512>>>(iters, f, d);
        hipDeviceSynchronize();
        hipEventRecord(e0); fma_burn<<<4096, 512>>>(iters, f, d); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        unsigned long long h[16]; hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
        // 4096 x 8 waves x iters x 4 FMA over 1024 SIMDs, 4 cycles each if the SIMD is 16 lanes wide
        const double simd_cycles = 4096.0 * 8 * iters * 4 * 4 / 1024.0;
        printf("fma_burn iters %d: %.1f us; wave 0 of wg 0 saw %llu ticks; FMA issue cycles per SIMD %.0f -> %.2f GHz if VALU bound at 4 cycles/instr\n",
               iters, ms * 1e3, h[0], simd_cycles, simd_cycles / (ms * 1e3) / 1e3);
    }
    return 0;
}
