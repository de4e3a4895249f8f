// Batched device-side Simulator.reset for gfx950: uniform-in-disc placement of CUEs / DUE transmitters and the
// "nearby, but inside the cell" rejection sampler for DUE receivers.
//
// Reference: Simulator.reset simulator.py:61-75; get_random_position position.py:18-28;
// get_random_position_nearby position.py:31-45.  The reference draws from Python's global `random`; here the
// stream is counter-based (Philox4x32-10, Salmon et al. SC'11) so that every (env, device, try) has its own
// reproducible draw independent of launch geometry or sharding:
//     counter = (global env index, device index, try, episode)      key = 64-bit seed
//     word 0 -> theta = 2*pi*u, u = (word >> 8) * 2^-24 in [0,1);  word 1 -> r = radius*sqrt(u), u = ((word >> 8) + 0.5) * 2^-24 in (0,1)
// One thread per (env, placement unit): base station, CUE, or DUE pair (transmitter, then its receiver's rejection loop).
#include "d2d_internal.h"

namespace d2d {

__device__ __forceinline__ void philox4x32_10(unsigned c0, unsigned c1, unsigned c2, unsigned c3, unsigned k0,
                                              unsigned k1, unsigned out[4]) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const unsigned long long p0 = (unsigned long long)0xD2511F53u * c0;     // one v_mad_u64_u32 per 32x32 -> 64 product
        const unsigned long long p1 = (unsigned long long)0xCD9E8D57u * c2;
        const unsigned hi0 = (unsigned)(p0 >> 32), lo0 = (unsigned)p0, hi1 = (unsigned)(p1 >> 32), lo1 = (unsigned)p1;
        const unsigned n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// sin and cos of 2*pi*u for u = k * 2^-24 in [0, 1).  The quadrant reduction is EXACT in these units (4u splits into an
// integer quadrant and a fraction with no rounding), so the only errors are one rounding of x = f * pi/2 and the two
// single-precision minimax polynomials on [-pi/4, pi/4] (Cephes sinf / cosf coefficients, < 1 ulp) - tighter than
// sincosf(fl(2*pi*u)), whose argument rounding alone is worth 3.7e-7 near 2*pi, and a fraction of its instructions
// (no large-argument path).
__device__ __forceinline__ void sincos_turns(float u, float& sn, float& cs) {
    const float t = 4.0f * u;                                           // [0, 4), exact
    float q = floorf(t);
    float f = t - q;                                                    // [0, 1), exact
    if (f > 0.5f) { f -= 1.0f; q += 1.0f; }                             // [-0.5, 0.5], exact
    const float x = f * 1.5707963267948966f, z = x * x;
    const float s = fmaf(x * z, fmaf(fmaf(-1.9515295891e-4f, z, 8.3321608736e-3f), z, -1.6666654611e-1f), x);
    const float c = fmaf(z * z, fmaf(fmaf(2.443315711809948e-5f, z, -1.388731625493765e-3f), z, 4.166664568298827e-2f),
                         fmaf(-0.5f, z, 1.0f));
    const int k = (int)q;
    const float a = (k & 1) ? c : s, b = (k & 1) ? s : c;               // quarter turns: (s,c) -> (c,-s) -> (-s,-c) -> (-c,s)
    sn = (k & 2) ? -a : a;
    cs = ((k + 1) & 2) ? -b : b;
}

__device__ __forceinline__ float2 disc_offset(unsigned w_theta, unsigned w_r, float radius) {
    const float u1 = (float)(w_theta >> 8) * 5.9604644775390625e-08f;   // [0, 1): 2^-24, exact
    // radius draw on the OPEN interval (0, 1): u2 == 0 would put a CUE exactly on the base station or a DUE receiver
    // exactly on its transmitter (distance 0 -> the step's log10(0) 'math domain error'); with [0, 1) that happens
    // once per 2^24 draws, i.e. about every 8th reset of a 4096 x 512-device batch
    const float u2 = ((float)(w_r >> 8) + 0.5f) * 5.9604644775390625e-08f;
    const float r = radius * sqrtf(u2);                                 // position.py:25,42
    float sn, cs;
    sincos_turns(u1, sn, cs);                                           // theta = 2*pi*u1, position.py:24,41
    return make_float2(r * cs, r * sn);                                 // position.py:26-27
}

struct ResetArgs {
    int B, D, C;                 // envs, devices per env, num_cues
    unsigned units;              // placement units per env: base station + C CUEs + P DUE pairs
    unsigned units_magic;        // floor(2^32 / units)
    unsigned total;              // B * units
    float cell_radius, d2d_radius;
    unsigned seed_lo, seed_hi, episode;
    unsigned long long env_offset;
    const unsigned char* fixed_mask;   // [D] or null
    const float* fixed_xy;             // [D,2]
    float* pos_x;
    float* pos_y;
    float4* lpos;                // [B, N] per-link (tx_x, tx_y, rx_x, rx_y) rows, or null: written here when the link list
    int N;                       // is the standard one (link i = unit i + 1), which saves the separate gather kernel
    int max_tries;
};

__device__ __forceinline__ float2 draw(const ResetArgs& a, unsigned env, unsigned d, unsigned t, float radius) {
    unsigned w[4];
    philox4x32_10(env, d, t, a.episode, a.seed_lo, a.seed_hi, w);
    return disc_offset(w[0], w[1], radius);
}

// One thread per placement UNIT of an env: the base station, one CUE, or one DUE PAIR.  A pair thread draws the
// transmitter and then runs the receiver's rejection loop against it, so the anchor is drawn once (a thread per device
// would re-derive it) and a wave holds one kind of work instead of alternating transmitter / receiver lanes.
__global__ __launch_bounds__(256) void reset_kernel(const ResetArgs a) {
    const unsigned gid = blockIdx.x * 256u + threadIdx.x;
    if (gid >= a.total) return;
    unsigned b = __umulhi(gid, a.units_magic);                           // gid / units: estimate is exact or one short
    unsigned u = gid - b * a.units;
    if (u >= a.units) { u -= a.units; ++b; }
    const unsigned env = (unsigned)(a.env_offset + b);
    const size_t base = (size_t)b * (size_t)a.D;
    if (u <= (unsigned)a.C) {                                            // base station (origin, simulator.py:63-64) or a CUE
        float2 p = make_float2(0.f, 0.f);
        if (u > 0) {
            if (a.fixed_mask && a.fixed_mask[u]) p = make_float2(a.fixed_xy[2 * u], a.fixed_xy[2 * u + 1]);   // simulator.py:65-66
            else p = draw(a, env, u, 0u, a.cell_radius);                 // simulator.py:67-68
        }
        a.pos_x[base + u] = p.x;
        a.pos_y[base + u] = p.y;
        if (a.lpos && u > 0) a.lpos[(size_t)b * (size_t)a.N + (u - 1u)] = make_float4(p.x, p.y, 0.f, 0.f);   // uplink to the origin
        return;
    }
    const unsigned d = (unsigned)a.C + 1u + 2u * (u - (unsigned)a.C - 1u);   // transmitter; its receiver is d + 1
    float2 tx, rx;
    if (a.fixed_mask && a.fixed_mask[d]) tx = make_float2(a.fixed_xy[2 * d], a.fixed_xy[2 * d + 1]);
    else tx = draw(a, env, d, 0u, a.cell_radius);
    if (a.fixed_mask && a.fixed_mask[d + 1]) {
        rx = make_float2(a.fixed_xy[2 * d + 2], a.fixed_xy[2 * d + 3]);
    } else {
        rx = tx;                                                         // only if every try is rejected
        const float r2 = a.cell_radius * a.cell_radius;
        for (int t = 0; t < a.max_tries; ++t) {                          // simulator.py:69-72, position.py:39-44
            const float2 o = draw(a, env, d + 1u, (unsigned)t, a.d2d_radius);
            const float x = tx.x + o.x, y = tx.y + o.y;
            if (!(x * x + y * y > r2)) { rx = make_float2(x, y); break; }
        }
    }
    a.pos_x[base + d] = tx.x;  a.pos_x[base + d + 1] = rx.x;
    a.pos_y[base + d] = tx.y;  a.pos_y[base + d + 1] = rx.y;
    if (a.lpos) a.lpos[(size_t)b * (size_t)a.N + (u - 1u)] = make_float4(tx.x, tx.y, rx.x, rx.y);
}

hipError_t launch_reset(int B, int D, int C, float cell_radius, float d2d_radius, unsigned long long seed,
                        unsigned long long episode, unsigned long long env_offset, const unsigned char* fixed_mask,
                        const float* fixed_xy, float* pos_x, float* pos_y, float4* lpos, int N, hipStream_t stream) {
    ResetArgs a;
    a.lpos = lpos; a.N = N;
    a.B = B; a.D = D; a.C = C;
    a.units = 1u + (unsigned)C + (unsigned)((D - 1 - C) / 2);
    a.units_magic = a.units == 1u ? 0xFFFFFFFFu : (unsigned)(0x100000000ull / a.units);   // one short at most: the kernel corrects
    const unsigned long long total = (unsigned long long)B * a.units;
    if (total >= 0xFFFFFF00ull) return hipErrorInvalidValue;
    if (total == 0) return hipSuccess;
    a.total = (unsigned)total;
    a.cell_radius = cell_radius; a.d2d_radius = d2d_radius;
    a.seed_lo = (unsigned)(seed & 0xFFFFFFFFull); a.seed_hi = (unsigned)(seed >> 32);
    a.episode = (unsigned)episode;
    a.env_offset = env_offset;
    a.fixed_mask = fixed_mask; a.fixed_xy = fixed_xy;
    a.pos_x = pos_x; a.pos_y = pos_y;
    a.max_tries = 64;
    hipLaunchKernelGGL(reset_kernel, dim3((a.total + 255u) / 256u), dim3(256), 0, stream, a);
    return hipGetLastError();
}

}  // namespace d2d
