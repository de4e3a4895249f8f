// Accuracy of the step kernel's (d^2)^(-e/2) (pow_neg_half, csrc/d2d_step.hip) and 10^(p/10) (pow10_tenth) on the hardware's
// v_log_f32 / v_exp_f32, against double precision, over the ranges the path uses.  Same code as the kernel's (kept in sync by hand).
//   hipcc --offload-arch=gfx950 -O3 tools/probes/pow_accuracy.hip -o /tmp/pow_acc && /tmp/pow_acc
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstring>
#include <cstdio>
#include <vector>
#include <random>

__device__ __forceinline__ float pow_neg_half(float d2, float e) {
    const float m = __builtin_amdgcn_frexp_mantf(d2);
    const float fe = (float)__builtin_amdgcn_frexp_expf(d2);
    const float l = __builtin_amdgcn_logf(m);
    const float h = -0.5f * e;
    const float p = h * fe;
    const float perr = fmaf(h, fe, -p);
    const float ip = rintf(p);
    const float fr = (p - ip) + fmaf(h, l, perr);
    return __builtin_amdgcn_ldexpf(__builtin_amdgcn_exp2f(fr), (int)ip);
}

// the same with floating-point contraction OFF inside the function: HIP's default -ffp-contract=fast may fuse `h * fe - ip` into
// one FMA, which no longer rounds p - and then the exact residual `perr` of the ROUNDED product is added on top of an unrounded one
__device__ __forceinline__ float pow_neg_half_nocontract(float d2, float e) {
#pragma clang fp contract(off)
    const float m = __builtin_amdgcn_frexp_mantf(d2);
    const float fe = (float)__builtin_amdgcn_frexp_expf(d2);
    const float l = __builtin_amdgcn_logf(m);
    const float h = -0.5f * e;
    const float p = h * fe;
    const float perr = fmaf(h, fe, -p);
    const float ip = rintf(p);
    const float fr = (p - ip) + fmaf(h, l, perr);
    return __builtin_amdgcn_ldexpf(__builtin_amdgcn_exp2f(fr), (int)ip);
}

// round 3's final form: -e/2 as a head (12 leading bits) + tail pair built from the DOUBLE exponent on the host
__device__ __forceinline__ float pow_neg_half_headtail(float d2, float hx, float hy) {
#pragma clang fp contract(off)
    const float m = __builtin_amdgcn_frexp_mantf(d2);
    const float fe = (float)__builtin_amdgcn_frexp_expf(d2);
    const float l = __builtin_amdgcn_logf(m);
    const float p = hx * fe;
    const float ip = rintf(p);
    const float fr = (p - ip) + fmaf(hx, l, hy * (l + fe));
    return __builtin_amdgcn_ldexpf(__builtin_amdgcn_exp2f(fr), (int)ip);
}

// candidate: one Newton step on l = log2(m) through exp2 (m = 2^l  ->  l += (m - 2^l) / (2^l ln 2)), so that the log's error
// is the exp's; then the same split
__device__ __forceinline__ float pow_neg_half_refined(float d2, float e) {
    const float m = __builtin_amdgcn_frexp_mantf(d2);
    const float fe = (float)__builtin_amdgcn_frexp_expf(d2);
    float l = __builtin_amdgcn_logf(m);
    const float t = __builtin_amdgcn_exp2f(l);
    const float dl = (m - t) * __builtin_amdgcn_rcpf(t) * 1.44269504088896340736f;      // residual of the log, ~1e-7
    const float h = -0.5f * e;
    const float p = h * fe;
    const float perr = fmaf(h, fe, -p);
    const float ip = rintf(p);
    const float fr = (p - ip) + (fmaf(h, l, perr) + h * dl);
    return __builtin_amdgcn_ldexpf(__builtin_amdgcn_exp2f(fr), (int)ip);
}

__global__ void eval2(const float* d2, const float* hx, const float* hy, float* out, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = pow_neg_half_headtail(d2[i], hx[i], hy[i]);
}

__global__ void eval(const float* d2, const float* e, float* a, float* b, float* c, float* l, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    a[i] = pow_neg_half(d2[i], e[i]);
    b[i] = pow_neg_half_refined(d2[i], e[i]);
    c[i] = pow_neg_half_nocontract(d2[i], e[i]);
    l[i] = __builtin_amdgcn_logf(__builtin_amdgcn_frexp_mantf(d2[i]));
}

int main() {
    const int n = 1 << 22;
    std::mt19937_64 rng(1);
    std::vector<float> d2(n), e(n), a(n), b(n), c(n), l(n);
    std::uniform_real_distribution<double> ld(0.0, 6.0), ex(2.05, 4.6);
    for (int i = 0; i < n; ++i) { const double d = std::pow(10.0, ld(rng) / 2.0); d2[i] = (float)(d * d); e[i] = (float)ex(rng); }
    float *dd, *de, *da, *db, *dc, *dl;
    hipMalloc(&dd, n * 4); hipMalloc(&de, n * 4); hipMalloc(&da, n * 4); hipMalloc(&db, n * 4); hipMalloc(&dc, n * 4); hipMalloc(&dl, n * 4);
    hipMemcpy(dd, d2.data(), n * 4, hipMemcpyHostToDevice); hipMemcpy(de, e.data(), n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(eval, dim3(n / 256), dim3(256), 0, 0, dd, de, da, db, dc, dl, n);
    hipMemcpy(a.data(), da, n * 4, hipMemcpyDeviceToHost); hipMemcpy(b.data(), db, n * 4, hipMemcpyDeviceToHost); hipMemcpy(c.data(), dc, n * 4, hipMemcpyDeviceToHost); hipMemcpy(l.data(), dl, n * 4, hipMemcpyDeviceToHost);
    double wa = 0, wb = 0, wc = 0, wl = 0, sa = 0, sb = 0, sc = 0;
    for (int i = 0; i < n; ++i) {
        const double ref = std::pow((double)d2[i], -0.5 * (double)e[i]);
        const double ra = std::fabs(a[i] - ref) / ref, rb = std::fabs(b[i] - ref) / ref, rc = std::fabs(c[i] - ref) / ref;
        int ex2; const double m = std::frexp((double)d2[i], &ex2);
        const double el = std::fabs(l[i] - std::log2(m));
        wa = std::max(wa, ra); wb = std::max(wb, rb); wc = std::max(wc, rc); sc += rc; wl = std::max(wl, el); sa += ra; sb += rb;
    }
    // against the DOUBLE exponent (what the oracle and the reference use): the float exponent's own rounding shows up here
    {
        std::vector<double> ed(n); std::vector<float> hx(n), hy(n), o(n), ef(n);
        std::mt19937_64 r2(7);
        for (int i = 0; i < n; ++i) {
            ed[i] = ex(r2); ef[i] = (float)ed[i];
            const double hd = -0.5 * ed[i]; float head = (float)hd; unsigned hb; memcpy(&hb, &head, 4); hb &= 0xFFFFF000u; memcpy(&head, &hb, 4);
            hx[i] = head; hy[i] = (float)(hd - (double)head);
        }
        float *dhx, *dhy, *dout; hipMalloc(&dhx, n * 4); hipMalloc(&dhy, n * 4); hipMalloc(&dout, n * 4);
        hipMemcpy(dhx, hx.data(), n * 4, hipMemcpyHostToDevice); hipMemcpy(dhy, hy.data(), n * 4, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(eval2, dim3(n / 256), dim3(256), 0, 0, dd, dhx, dhy, dout, n);
        hipMemcpy(o.data(), dout, n * 4, hipMemcpyDeviceToHost);
        hipMemcpy(de, ef.data(), n * 4, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(eval, dim3(n / 256), dim3(256), 0, 0, dd, de, da, db, dc, dl, n);
        hipMemcpy(c.data(), dc, n * 4, hipMemcpyDeviceToHost);
        double wn = 0, sn = 0, wo = 0, so = 0;
        for (int i = 0; i < n; ++i) {
            const double ref = std::pow((double)d2[i], -0.5 * ed[i]);
            const double rn = std::fabs(o[i] - ref) / ref, ro = std::fabs(c[i] - ref) / ref;
            wn = std::max(wn, rn); sn += rn; wo = std::max(wo, ro); so += ro;
        }
        printf("{\"against_the_double_exponent\": {\"float_exponent_contract_off_max_rel\": %.3e, \"mean\": %.3e, \"head_tail_exponent_max_rel\": %.3e, \"head_tail_mean\": %.3e}}\n", wo, so / n, wn, sn / n);
    }
    printf("{\"samples\": %d, \"pow_neg_half_max_rel\": %.3e, \"mean_rel\": %.3e, \"refined_max_rel\": %.3e, \"refined_mean_rel\": %.3e, \"contract_off_max_rel\": %.3e, \"contract_off_mean_rel\": %.3e, \"v_log_f32_max_abs_err_on_[0.5,1)\": %.3e}\n",
           n, wa, sa / n, wb, sb / n, wc, sc / n, wl);
    return 0;
}
