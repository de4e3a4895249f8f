// What does one wave64 instruction of each kind cost the SIMD that issues it?  (DESIGN.md 4.1 prices the rollout kernel in
// "VALU equivalents": 4 cycles for a full-rate instruction, 16 for a transcendental, and asks what the f32 -> f64 conversion
// and the f64 add of its exact interference sum cost.)  1, 4 and 8 waves per SIMD, 8 independent chains per lane,
// 8 x 2048 instructions per wave between two s_memtime reads (the counter runs at about the core clock); printed relative to
// v_mul_f32, with the launch's wall time per instruction and SIMD beside it.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/issue_rates.hip -o /tmp/issue_rates && /tmp/issue_rates
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

#define REP 2048

#define KERNEL(NAME, DECL, BODY, SINK)                                                                        \
    __global__ __launch_bounds__(256) void NAME(unsigned long long* out, float seed) {                         \
        DECL;                                                                                                 \
        const unsigned long long t0 = __builtin_readcyclecounter();                                           \
        _Pragma("unroll 1") for (int k = 0; k < REP; ++k) { BODY; }                                           \
        const unsigned long long t1 = __builtin_readcyclecounter();                                           \
        SINK;                                                                                                 \
        if ((threadIdx.x & 63) == 0) out[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0;          \
    }

#define F8 float a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3, a4 = seed + 4, a5 = seed + 5, a6 = seed + 6, a7 = seed + 7
#define SINKF if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 123.456f) out[1] = 1
#define ASM8(OP) asm volatile(OP " %0, %0, %0\n" OP " %1, %1, %1\n" OP " %2, %2, %2\n" OP " %3, %3, %3\n" OP " %4, %4, %4\n" OP " %5, %5, %5\n" OP " %6, %6, %6\n" OP " %7, %7, %7" \
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7))
#define ASM8U(OP) asm volatile(OP " %0, %0\n" OP " %1, %1\n" OP " %2, %2\n" OP " %3, %3\n" OP " %4, %4\n" OP " %5, %5\n" OP " %6, %6\n" OP " %7, %7" \
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7))

KERNEL(k_mul_f32, F8, ASM8("v_mul_f32"), SINKF)
KERNEL(k_fma_f32, F8, asm volatile("v_fma_f32 %0, %0, %0, %0\nv_fma_f32 %1, %1, %1, %1\nv_fma_f32 %2, %2, %2, %2\nv_fma_f32 %3, %3, %3, %3\nv_fma_f32 %4, %4, %4, %4\nv_fma_f32 %5, %5, %5, %5\nv_fma_f32 %6, %6, %6, %6\nv_fma_f32 %7, %7, %7, %7"
                                    : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)), SINKF)
KERNEL(k_rcp_f32, F8, ASM8U("v_rcp_f32"), SINKF)
KERNEL(k_log_f32, F8, ASM8U("v_log_f32"), SINKF)
KERNEL(k_exp_f32, F8, ASM8U("v_exp_f32"), SINKF)
KERNEL(k_max3_i32, F8, asm volatile("v_max3_i32 %0, %0, %1, %2\nv_max3_i32 %1, %1, %2, %3\nv_max3_i32 %2, %2, %3, %4\nv_max3_i32 %3, %3, %4, %5\nv_max3_i32 %4, %4, %5, %6\nv_max3_i32 %5, %5, %6, %7\nv_max3_i32 %6, %6, %7, %0\nv_max3_i32 %7, %7, %0, %1"
                                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)), SINKF)
KERNEL(k_mul_hi_u32, F8, ASM8("v_mul_hi_u32"), SINKF)

#define D8 double d0 = seed, d1 = seed + 1, d2 = seed + 2, d3 = seed + 3, d4 = seed + 4, d5 = seed + 5, d6 = seed + 6, d7 = seed + 7
#define SINKD if (d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7 == 123.456) out[1] = 1
KERNEL(k_add_f64, D8, asm volatile("v_add_f64 %0, %0, %0\nv_add_f64 %1, %1, %1\nv_add_f64 %2, %2, %2\nv_add_f64 %3, %3, %3\nv_add_f64 %4, %4, %4\nv_add_f64 %5, %5, %5\nv_add_f64 %6, %6, %6\nv_add_f64 %7, %7, %7"
                                    : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7)), SINKD)
KERNEL(k_fma_f64, D8, asm volatile("v_fma_f64 %0, %0, %0, %0\nv_fma_f64 %1, %1, %1, %1\nv_fma_f64 %2, %2, %2, %2\nv_fma_f64 %3, %3, %3, %3\nv_fma_f64 %4, %4, %4, %4\nv_fma_f64 %5, %5, %5, %5\nv_fma_f64 %6, %6, %6, %6\nv_fma_f64 %7, %7, %7, %7"
                                    : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7)), SINKD)
// f32 -> f64: the sources are floats that stay put, the destinations eight doubles
KERNEL(k_cvt_f64_f32, F8; D8, asm volatile("v_cvt_f64_f32 %0, %8\nv_cvt_f64_f32 %1, %9\nv_cvt_f64_f32 %2, %10\nv_cvt_f64_f32 %3, %11\nv_cvt_f64_f32 %4, %12\nv_cvt_f64_f32 %5, %13\nv_cvt_f64_f32 %6, %14\nv_cvt_f64_f32 %7, %15"
                                            : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7)
                                            : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(a4), "v"(a5), "v"(a6), "v"(a7)), SINKD)
KERNEL(k_cvt_f32_f64, F8; D8, asm volatile("v_cvt_f32_f64 %0, %8\nv_cvt_f32_f64 %1, %9\nv_cvt_f32_f64 %2, %10\nv_cvt_f32_f64 %3, %11\nv_cvt_f32_f64 %4, %12\nv_cvt_f32_f64 %5, %13\nv_cvt_f32_f64 %6, %14\nv_cvt_f32_f64 %7, %15"
                                            : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                                            : "v"(d0), "v"(d1), "v"(d2), "v"(d3), "v"(d4), "v"(d5), "v"(d6), "v"(d7)), SINKF)
// packed f32 add on register pairs
KERNEL(k_pk_add_f32, D8, asm volatile("v_pk_add_f32 %0, %0, %0\nv_pk_add_f32 %1, %1, %1\nv_pk_add_f32 %2, %2, %2\nv_pk_add_f32 %3, %3, %3\nv_pk_add_f32 %4, %4, %4\nv_pk_add_f32 %5, %5, %5\nv_pk_add_f32 %6, %6, %6\nv_pk_add_f32 %7, %7, %7"
                                       : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7)), SINKD)
// LDS: a 16-byte read per lane from lane-contiguous addresses (the data is never used: issue cost only)
__global__ __launch_bounds__(256) void k_ds_read_b128(unsigned long long* out, float seed) {
    __shared__ float4 buf[64 * 8];
    buf[threadIdx.x] = make_float4(seed, seed, seed, seed);
    __syncthreads();
    const unsigned addr = (threadIdx.x & 63) * 16u;
    float acc = 0.0f;
    const unsigned long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
    for (int k = 0; k < REP; ++k) {
        float4 v0, v1, v2, v3, v4, v5, v6, v7;
        asm volatile("ds_read_b128 %0, %8\nds_read_b128 %1, %8 offset:1024\nds_read_b128 %2, %8 offset:2048\nds_read_b128 %3, %8 offset:3072\n"
                     "ds_read_b128 %4, %8 offset:4096\nds_read_b128 %5, %8 offset:5120\nds_read_b128 %6, %8 offset:6144\nds_read_b128 %7, %8 offset:7168\ns_waitcnt lgkmcnt(0)"
                     : "=v"(v0), "=v"(v1), "=v"(v2), "=v"(v3), "=v"(v4), "=v"(v5), "=v"(v6), "=v"(v7) : "v"(addr) : "memory");
        acc += v0.x + v7.w;
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (acc == 123.456f) out[1] = 1;
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0;
}

typedef void (*kern_t)(unsigned long long*, float);

static double g_ns = 0.0;                  // wall time of the launch (HIP events), ns

static double run(kern_t k, unsigned long long* dev, int waves, int block) {
    std::vector<unsigned long long> host(waves);
    std::vector<double> med, wall;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int rep = 0; rep < 5; ++rep) {
        (void)hipEventRecord(e0, 0);
        hipLaunchKernelGGL(k, dim3(waves * 64 / block), dim3(block), 0, 0, dev, 1.0f);
        (void)hipEventRecord(e1, 0);
        (void)hipEventSynchronize(e1);
        float ms = 0.0f; (void)hipEventElapsedTime(&ms, e0, e1);
        wall.push_back(ms * 1e6);
        (void)hipMemcpy(host.data(), dev, waves * sizeof(unsigned long long), hipMemcpyDeviceToHost);
        std::sort(host.begin(), host.end());
        med.push_back((double)host[waves / 2] / (8.0 * REP));
    }
    std::sort(med.begin(), med.end()); std::sort(wall.begin(), wall.end());
    g_ns = wall[2];
    return med[2];
}

int main() {
    unsigned long long* dev;
    (void)hipMalloc(&dev, 65536 * sizeof(unsigned long long));
    struct { const char* name; kern_t k; } tests[] = {
        {"v_mul_f32", k_mul_f32}, {"v_fma_f32", k_fma_f32}, {"v_rcp_f32", k_rcp_f32}, {"v_log_f32", k_log_f32}, {"v_exp_f32", k_exp_f32},
        {"v_max3_i32", k_max3_i32}, {"v_mul_hi_u32", k_mul_hi_u32}, {"v_pk_add_f32", k_pk_add_f32}, {"v_add_f64", k_add_f64},
        {"v_fma_f64", k_fma_f64}, {"v_cvt_f64_f32", k_cvt_f64_f32}, {"v_cvt_f32_f64", k_cvt_f32_f64}, {"ds_read_b128 (+ wait per 8)", k_ds_read_b128}};
    // one wave per SIMD (1024 one-wave workgroups over 256 CUs x 4 SIMDs), then four and eight per SIMD (workgroups of four waves:
    // a CU takes at most 16 workgroups, so one-wave workgroups stop at four per SIMD)
    for (int waves : {1024, 4096, 8192}) {
        const int block = waves == 1024 ? 64 : 256;
        printf("%d waves of 64 lanes (%d per SIMD): s_memtime ticks per instruction and wave; relative to v_mul_f32; launch wall time / (instructions per SIMD)\n", waves, waves / 1024);
        double base = 0.0;
        for (auto& t : tests) {
            const double c = run(t.k, dev, waves, block);
            if (base == 0.0) base = c;
            printf("  %-28s %8.3f ticks   x %.2f   %7.3f ns per instruction and SIMD (launch %.1f us)\n", t.name, c, c / base,
                   g_ns / (8.0 * REP * (waves / 1024)), g_ns * 1e-3);
        }
    }
    (void)hipFree(dev);
    return 0;
}
