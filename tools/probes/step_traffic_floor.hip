// The compact-obs step kernel's DATA MOVEMENT with nothing to compute: the same launch shape (one 512-thread workgroup per
// env, thread = link), the same reads (4-byte action from a fresh buffer every launch, 16-byte position row, the next env's
// action row touched one residency round ahead) and the same writes (four 4-byte result planes, one 24-byte table row, the
// reward row by one wave) - 64 algorithmic bytes per link - but no LDS, no barriers, no walk, no math.  What this takes is
// the floor for a kernel of this shape on this box; the step kernel's distance from it is what its computation costs.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/step_traffic_floor.hip -o /tmp/step_floor && /tmp/step_floor
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int EXPORT, int PREFETCH>
__global__ __launch_bounds__(512) void traffic_kernel(const int* __restrict__ act, const f32x4* __restrict__ lpos, float* sinr, float* snr, float* rate,
                                                      float* cap, float* table, float* reward, int* rb, int* pw, int B, int N, int pf_envs) {
    const int b = blockIdx.x, i = threadIdx.x;
    const unsigned row = (unsigned)b * (unsigned)N + (unsigned)i;
    const int a = act[row];
    const f32x4 p = lpos[row];
    int pf = 0;
    if (PREFETCH) { const int bq = b + pf_envs; pf = act[(unsigned)(bq < B ? bq : B - 1) * (unsigned)N + i]; }
    const float v = (float)a + p.x;
    sinr[row] = v; snr[row] = p.y; rate[row] = p.z; cap[row] = p.w;
    if (EXPORT) { rb[row] = a; pw[row] = a + 1; }
    f32x2* t = reinterpret_cast<f32x2*>(table + (size_t)row * 6);
    t[0] = f32x2{p.x, p.y}; t[1] = f32x2{p.z, p.w}; t[2] = f32x2{v, v};
    if (i < 128) { const f32x4 r4 = {v, v, v, v}; reinterpret_cast<f32x4*>(reward + (size_t)b * N)[i] = r4; }
    asm volatile("" ::"v"(pf));
}

// The same bytes in fewer streams: the four result planes as one [B, N, 4] array (one 16-byte store per lane), the decoded
// (rb, pwr) as one [B, N, 2] array - does the number of concurrent write streams matter, or only the bytes?
template <int EXPORT>
__global__ __launch_bounds__(512) void packed_kernel(const int* __restrict__ act, const f32x4* __restrict__ lpos, f32x4* results, float* table,
                                                     float* reward, int2* rbpw, int B, int N, int pf_envs) {
    const int b = blockIdx.x, i = threadIdx.x;
    const unsigned row = (unsigned)b * (unsigned)N + (unsigned)i;
    const int a = act[row];
    const f32x4 p = lpos[row];
    const int bq = b + pf_envs;
    const int pf = act[(unsigned)(bq < B ? bq : B - 1) * (unsigned)N + i];
    const float v = (float)a + p.x;
    results[row] = f32x4{v, p.y, p.z, p.w};
    if (EXPORT) rbpw[row] = make_int2(a, a + 1);
    f32x2* t = reinterpret_cast<f32x2*>(table + (size_t)row * 6);
    t[0] = f32x2{p.x, p.y}; t[1] = f32x2{p.z, p.w}; t[2] = f32x2{v, v};
    if (i < 128) { const f32x4 r4 = {v, v, v, v}; reinterpret_cast<f32x4*>(reward + (size_t)b * N)[i] = r4; }
    asm volatile("" ::"v"(pf));
}

template <int EXPORT>
float run_packed(int B, int N, int sets, const int* act, const f32x4* lpos, f32x4* results, float* table, float* reward, int2* rbpw) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    std::vector<float> t;
    int k = 0;
    for (int rep = 0; rep < 12; ++rep) {
        hipEventRecord(e0, 0);
        for (int l = 0; l < 32; ++l, ++k)
            hipLaunchKernelGGL((packed_kernel<EXPORT>), dim3(B), dim3(N), 0, 0, act + (size_t)(k % sets) * B * N, lpos, results, table, reward, rbpw, B, N, 1024);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); t.push_back(ms * 1e3f / 32);
    }
    std::sort(t.begin(), t.end());
    return t[t.size() / 2];
}

template <int EXPORT, int PREFETCH>
float run(int B, int N, int sets, const int* act, const f32x4* lpos, float** out, int* rb, int* pw) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    std::vector<float> t;
    int k = 0;
    for (int rep = 0; rep < 12; ++rep) {
        hipEventRecord(e0, 0);
        for (int l = 0; l < 32; ++l, ++k)
            hipLaunchKernelGGL((traffic_kernel<EXPORT, PREFETCH>), dim3(B), dim3(N), 0, 0, act + (size_t)(k % sets) * B * N, lpos, out[0], out[1], out[2],
                               out[3], out[4], out[5], rb, pw, B, N, 1024);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); t.push_back(ms * 1e3f / 32);
    }
    std::sort(t.begin(), t.end());
    return t[t.size() / 2];
}

int main() {
    const int B = 4096, N = 512, sets = 64;
    int* act; f32x4* lpos; float* out[6]; int *rb, *pw;
    hipMalloc(&act, (size_t)sets * B * N * 4); hipMemset(act, 1, (size_t)sets * B * N * 4);
    hipMalloc(&lpos, (size_t)B * N * 16); hipMemset(lpos, 0, (size_t)B * N * 16);
    for (int k = 0; k < 6; ++k) hipMalloc(&out[k], (size_t)B * N * (k == 4 ? 24 : 4));
    hipMalloc(&rb, (size_t)B * N * 4); hipMalloc(&pw, (size_t)B * N * 4);
    f32x4* results; int2* rbpw;
    hipMalloc(&results, (size_t)B * N * 16); hipMalloc(&rbpw, (size_t)B * N * 8);
    for (int round = 0; round < 3; ++round) {
        printf("{\"round\": %d, \"packed_results_us\": %.2f, \"packed_results_and_rb_pwr_us\": %.2f}\n", round,
               run_packed<0>(B, N, sets, act, lpos, results, out[4], out[5], rbpw), run_packed<1>(B, N, sets, act, lpos, results, out[4], out[5], rbpw));
        const float a = run<0, 1>(B, N, sets, act, lpos, out, rb, pw), b = run<1, 1>(B, N, sets, act, lpos, out, rb, pw);
        const float c = run<0, 0>(B, N, sets, act, lpos, out, rb, pw);
        printf("{\"round\": %d, \"traffic_only_us\": %.2f, \"with_rb_pwr_planes_us\": %.2f, \"without_prefetch_us\": %.2f, \"algorithmic_GBps\": %.0f}\n", round, a, b, c,
               (double)B * N * 64 / a / 1e3);
    }
    return 0;
}
