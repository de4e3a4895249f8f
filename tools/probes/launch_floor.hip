// What does it cost to launch W waves that do nothing?  (the step kernel's 6 us floor at 4096 x 512.)
//   hipcc --offload-arch=gfx950 -O3 tools/probes/launch_floor.hip -o /tmp/launch_floor && /tmp/launch_floor
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

template <int VG>
__global__ __launch_bounds__(1024) void empty_kernel(int* out, int never) {
    extern __shared__ int lds[];
    if (VG > 0) {
        // keep VG registers live so the wave allocates them
        float r[VG > 0 ? VG : 1];
#pragma unroll
        for (int k = 0; k < VG; ++k) r[k] = (float)(threadIdx.x + k) * 1.0001f;
        float s = 0;
#pragma unroll
        for (int k = 0; k < VG; ++k) s += r[k] * r[(k * 7 + 3) % VG];
        if (s == 12345.678f) out[0] = (int)s;
    }
    if (threadIdx.x == never) { lds[threadIdx.x] = 1; out[blockIdx.x] = lds[0]; }
}

template <int VG>
float run(int grid, int block, size_t lds, int* out) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&empty_kernel<VG>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    std::vector<float> t;
    for (int rep = 0; rep < 30; ++rep) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL((empty_kernel<VG>), dim3(grid), dim3(block), lds, 0, out, 5000);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); t.push_back(ms * 1e3f);
    }
    std::sort(t.begin(), t.end());
    return t[t.size() / 2];
}

int main() {
    int* out; hipMalloc(&out, 1 << 20);
    printf("grid x block, LDS bytes, live VGPRs -> median us per launch (HIP events)\n");
    for (int block : {64, 256, 512, 1024}) {
        const int grid = 4096 * 512 / block;
        for (size_t lds : {(size_t)0, (size_t)16 * 1024, (size_t)38 * 1024}) {
            printf("%6d x %4d  lds %6zu  vgpr~8  : %7.2f us   vgpr~48 : %7.2f us\n", grid, block, lds, run<0>(grid, block, lds, out),
                   run<40>(grid, block, lds, out));
        }
    }
    printf("1024 x 512 (persistent grid), lds 38912: %7.2f us\n", run<40>(1024, 512, 38 * 1024, out));
    // round 5: the rollout kernel's shapes (15.4 KB of LDS per env; one or two links per thread) and a persistent grid of them
    printf("4096 x 512, lds 15800, vgpr~48: %7.2f us\n", run<40>(4096, 512, 15800, out));
    printf("4096 x 256, lds 15800, vgpr~48: %7.2f us\n", run<40>(4096, 256, 15800, out));
    printf("4096 x 128, lds 15800, vgpr~48: %7.2f us\n", run<40>(4096, 128, 15800, out));
    printf("2048 x 256 (persistent), lds 15800: %7.2f us\n", run<40>(2048, 256, 15800, out));
    printf("   1 x 64: %7.2f us\n", run<0>(1, 64, 0, out));
    return 0;
}
