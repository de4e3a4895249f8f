#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
#include <random>
__global__ void eval(const float* x, float* ex, float* lg, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    ex[i] = __builtin_amdgcn_exp2f(x[i]);
    lg[i] = __builtin_amdgcn_logf(fabsf(x[i]) + 0.5f);
}
int main() {
    const int n = 1 << 22;
    std::mt19937_64 rng(1);
    std::vector<float> x(n), ex(n), lg(n);
    std::uniform_real_distribution<double> u(-3.0, 3.0);
    for (int i = 0; i < n; ++i) x[i] = (float)u(rng);
    float *dx, *de, *dl; hipMalloc(&dx, n * 4); hipMalloc(&de, n * 4); hipMalloc(&dl, n * 4);
    hipMemcpy(dx, x.data(), n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(eval, dim3(n / 256), dim3(256), 0, 0, dx, de, dl, n);
    hipMemcpy(ex.data(), de, n * 4, hipMemcpyDeviceToHost); hipMemcpy(lg.data(), dl, n * 4, hipMemcpyDeviceToHost);
    double we = 0, wl = 0, wlr = 0; double binmax[6] = {0};
    for (int i = 0; i < n; ++i) {
        const double r = std::exp2((double)x[i]);
        const double e = std::fabs(ex[i] - r) / r; we = std::max(we, e);
        binmax[(int)std::floor(x[i] + 3.0) % 6] = std::max(binmax[(int)std::floor(x[i] + 3.0) % 6], e);
        const double a = (double)(std::fabs(x[i]) + 0.5f), rl = std::log2(a);
        wl = std::max(wl, std::fabs(lg[i] - rl)); if (std::fabs(rl) > 1e-3) wlr = std::max(wlr, std::fabs(lg[i] - rl) / std::fabs(rl));
    }
    printf("v_exp_f32 max rel err on [-3,3]: %.3e  per unit interval from -3: %.2e %.2e %.2e %.2e %.2e %.2e\n", we, binmax[0], binmax[1], binmax[2], binmax[3], binmax[4], binmax[5]);
    printf("v_log_f32 on [0.5,3.5]: max abs %.3e, max rel %.3e\n", wl, wlr);
    return 0;
}
