// Batched device-side Simulator.reset for gfx950: uniform-in-disc placement of CUEs / DUE transmitters and the
// "nearby, but inside the cell" rejection sampler for DUE receivers.
//
// Reference: Simulator.reset simulator.py:61-75; get_random_position position.py:18-28;
// get_random_position_nearby position.py:31-45.  The reference draws from Python's global `random`; here the
// stream is counter-based (Philox4x32-10, Salmon et al. SC'11) so that every (env, device, try) has its own
// reproducible draw independent of launch geometry or sharding:
//     counter = (global env index, device index, try, episode)      key = 64-bit seed
//     word 0 -> theta = 2*pi*u, u = (word >> 8) * 2^-24 in [0,1);  word 1 -> r = radius*sqrt(u), u = ((word >> 8) + 0.5) * 2^-24 in (0,1)
// One thread per (env, device).  A DUE receiver thread re-derives its transmitter's position from the
// transmitter's own counter instead of waiting for another thread, so there is no intra-kernel dependency.
#include "d2d_internal.h"

namespace d2d {

__device__ __forceinline__ void philox4x32_10(unsigned c0, unsigned c1, unsigned c2, unsigned c3, unsigned k0,
                                              unsigned k1, unsigned out[4]) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const unsigned hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        const unsigned hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        const unsigned n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

__device__ __forceinline__ float2 disc_offset(unsigned w_theta, unsigned w_r, float radius) {
    const float u1 = (float)(w_theta >> 8) * 5.9604644775390625e-08f;   // [0, 1): 2^-24, exact
    // radius draw on the OPEN interval (0, 1): u2 == 0 would put a CUE exactly on the base station or a DUE receiver
    // exactly on its transmitter (distance 0 -> the step's log10(0) 'math domain error'); with [0, 1) that happens
    // once per 2^24 draws, i.e. about every 8th reset of a 4096 x 512-device batch
    const float u2 = ((float)(w_r >> 8) + 0.5f) * 5.9604644775390625e-08f;
    const float theta = 6.283185307179586f * u1;                        // position.py:24,41
    const float r = radius * sqrtf(u2);                                 // position.py:25,42
    float sn, cs;
    sincosf(theta, &sn, &cs);
    return make_float2(r * cs, r * sn);                                 // position.py:26-27
}

struct ResetArgs {
    int B, D, C;                 // envs, devices per env, num_cues
    float cell_radius, d2d_radius;
    unsigned seed_lo, seed_hi, episode;
    unsigned long long env_offset;
    const unsigned char* fixed_mask;   // [D] or null
    const float* fixed_xy;             // [D,2]
    float* pos_x;
    float* pos_y;
    int max_tries;
};

__global__ __launch_bounds__(256) void reset_kernel(const ResetArgs a) {
    const size_t gid = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (gid >= (size_t)a.B * a.D) return;
    const int b = (int)(gid / a.D), d = (int)(gid % a.D);
    const unsigned env = (unsigned)(a.env_offset + b);
    float2 p = make_float2(0.f, 0.f);                                    // 'mbs' at the origin, simulator.py:63-64
    const bool fixed = a.fixed_mask && a.fixed_mask[d];
    if (d == 0) {
        // base station
    } else if (fixed) {
        p = make_float2(a.fixed_xy[2 * d], a.fixed_xy[2 * d + 1]);      // simulator.py:65-66
    } else {
        unsigned w[4];
        const int k = d - 1 - a.C;                                       // >= 0 for DUE devices
        if (k < 0 || (k & 1) == 0) {
            philox4x32_10(env, (unsigned)d, 0u, a.episode, a.seed_lo, a.seed_hi, w);
            p = disc_offset(w[0], w[1], a.cell_radius);                  // simulator.py:67-68
        } else {
            // DUE receiver: anchor = its transmitter (device d-1), simulator.py:69-72
            float2 anchor;
            if (a.fixed_mask && a.fixed_mask[d - 1]) {
                anchor = make_float2(a.fixed_xy[2 * (d - 1)], a.fixed_xy[2 * (d - 1) + 1]);
            } else {
                philox4x32_10(env, (unsigned)(d - 1), 0u, a.episode, a.seed_lo, a.seed_hi, w);
                anchor = disc_offset(w[0], w[1], a.cell_radius);
            }
            p = anchor;                                                  // only if every try is rejected
            const float r2 = a.cell_radius * a.cell_radius;
            for (int t = 0; t < a.max_tries; ++t) {                      // position.py:39-44
                philox4x32_10(env, (unsigned)d, (unsigned)t, a.episode, a.seed_lo, a.seed_hi, w);
                const float2 o = disc_offset(w[0], w[1], a.d2d_radius);
                const float x = anchor.x + o.x, y = anchor.y + o.y;
                if (!(x * x + y * y > r2)) { p = make_float2(x, y); break; }
            }
        }
    }
    a.pos_x[gid] = p.x;
    a.pos_y[gid] = p.y;
}

hipError_t launch_reset(int B, int D, int C, float cell_radius, float d2d_radius, unsigned long long seed,
                        unsigned long long episode, unsigned long long env_offset, const unsigned char* fixed_mask,
                        const float* fixed_xy, float* pos_x, float* pos_y, hipStream_t stream) {
    ResetArgs a;
    a.B = B; a.D = D; a.C = C;
    a.cell_radius = cell_radius; a.d2d_radius = d2d_radius;
    a.seed_lo = (unsigned)(seed & 0xFFFFFFFFull); a.seed_hi = (unsigned)(seed >> 32);
    a.episode = (unsigned)episode;
    a.env_offset = env_offset;
    a.fixed_mask = fixed_mask; a.fixed_xy = fixed_xy;
    a.pos_x = pos_x; a.pos_y = pos_y;
    a.max_tries = 64;
    const size_t total = (size_t)B * D;
    hipLaunchKernelGGL(reset_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, a);
    return hipGetLastError();
}

}  // namespace d2d
