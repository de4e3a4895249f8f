// LinearObsFunction expansion for gfx950 (MI355X): obs[b, i, :] = concat(T[b,i], T[b,0..i-1], T[b,i+1..N-1]).
//
// Reference: LinearObsFunction.get_state, envs/obs_fn.py:43-53 (hot loop #2: O(N^2) list.extend + N np.array
// calls, 86 % of the reference's step time at N = 512).
//
// This is the HBM-dominant kernel of the path: 24*N bytes written per agent-step against 24 read.  It is a
// pure streaming store, so the design is about the store side only:
//   * T[b] (6N floats) is staged once per workgroup in LDS;
//   * the output block of an env is contiguous ([N][6N] floats), so a workgroup walks a contiguous slab of it
//     with one 16-byte store per lane per iteration - every wave-instruction writes 1 KiB of consecutive
//     addresses (full 128-B lines, no partial-line read-modify-write); since round 4 the slab is FLAT (two
//     consecutive pieces of 1024 float4, whatever the row length: obs_expand_flat_kernel), before that two whole rows;
//   * row i is T shifted by 6 floats for columns [6, 6(i+1)) and unshifted after that; both boundaries are
//     even, so every aligned float2 of the output maps to one aligned float2 of T: two ds_read_b64 feed each
//     global_store_dwordx4.  (Not conflict-free, as this comment claimed until round 5: the two halves of a float4 that
//     straddles the shift boundary, and the lanes on either side of a row end, hit the same banks - SQ_LDS_BANK_CONFLICT /
//     SQ_LDS_IDX_ACTIVE = 0.37, profiles/r5_pmc_stress_linear.json.  It costs nothing measurable: SQ_WAIT_INST_LDS is 0.04 %
//     of the wave cycles of a kernel that runs at 0.977 of the box's best store-only kernel.)
//   * stores are nontemporal (written once, never re-read by this kernel);
//   * blockIdx is remapped so the chunks of one env share an XCD (its T stays in that XCD's L2).
#include "d2d_internal.h"
#include "d2d_store.h"

namespace d2d {

// source float index inside T_flat for output column f (even) of row i
__device__ __forceinline__ unsigned src_col(unsigned f, unsigned i) {
    const unsigned head = 6u * i;
    return f < 6u ? head + f : (f < head + 6u ? f - 6u : f);
}

template <int VEC, int NT>           // NT: the store policy of store16 (0 plain, 1 nt, 2 .. 5 scope variants; VEC == 2: 0 / 1 only)
__global__ __launch_bounds__(1024) void obs_expand_kernel(const ObsArgs a) {
    extern __shared__ __align__(16) float t_flat[];          // [6N]
    const unsigned N = a.N, tid = threadIdx.x, T = blockDim.x;
    unsigned env, chunk;
    if (a.xcd_remap) {
        // blocks b, b+8, b+16, ... share an XCD (round-robin dispatch): give them the chunks of ONE env (G = 1), or
        // of G envs interleaved chunk by chunk (G = xcd_remap > 1: more independent write fronts per XCD)
        const unsigned bid = blockIdx.x, lane8 = bid & 7u, rest = bid >> 3, G = (unsigned)a.xcd_remap;
        const unsigned per_group = a.chunks * G, group = rest / per_group, within = rest % per_group;
        chunk = within / G;
        env = (group * G + within % G) * 8u + lane8;
    } else {
        env = blockIdx.x / a.chunks;
        chunk = blockIdx.x % a.chunks;
    }
    const unsigned row_floats = 6u * N;

    // stage T[env] (coalesced; 8-byte granules are always aligned because 6N is even)
    {
        const f32x2* src = reinterpret_cast<const f32x2*>(a.table + (size_t)env * row_floats);
        f32x2* dst = reinterpret_cast<f32x2*>(t_flat);
        for (unsigned k = tid; k < row_floats / 2; k += T) dst[k] = src[k];
    }
    __syncthreads();

    const unsigned r0 = chunk * a.rows_per_wg;
    const unsigned r1 = min(r0 + (unsigned)a.rows_per_wg, N);
    const unsigned q_per_row = a.q_per_row;
    const unsigned total = (r1 - r0) * q_per_row;
    float* out = a.obs + ((size_t)env * N + r0) * row_floats;   // contiguous slab of rows [r0, r1)
    if (a.stagger > 0) {                                        // D2D_TUNE_OBS_STAGGER (A/B): wave w waits w * stagger x 64 clocks
        const int n = (int)(tid >> 6) * a.stagger;
        for (int k = 0; k < n; ++k) __builtin_amdgcn_s_sleep(1);
    }

    if (VEC == 4 && q_per_row == T) {
        // The default geometry: one thread per float4 column (block == 6N / 4), so q = tid for every row of the slab and
        // the source selection is two compares against per-thread constants (column f comes from 6i + f if f < 6, from
        // f - 6 once i >= f / 6, else from f) - no division, no three-way selection per store.
        const unsigned f0 = tid * 4u, f1 = f0 + 2u;
        const unsigned t0 = f0 / 6u, t1 = f1 / 6u;
        const bool own0 = f0 < 6u, own1 = f1 < 6u;
        f32x4* o4 = reinterpret_cast<f32x4*>(out) + tid;
        const f32x2* t2 = reinterpret_cast<const f32x2*>(t_flat);
        const auto row = [&](unsigned i) {
            const unsigned head = 6u * i;
            const unsigned s0 = own0 ? head + f0 : (i >= t0 ? f0 - 6u : f0);
            const unsigned s1 = own1 ? head + f1 : (i >= t1 ? f1 - 6u : f1);
            const f32x2 lo = t2[s0 >> 1], hi = t2[s1 >> 1];
            const f32x4 v = {lo.x, lo.y, hi.x, hi.y};
            return v;
        };
        // rows in pairs, all four LDS reads ahead of the two stores (written out: the scope-policy stores are asm statements,
        // which the unroller leaves alone)
        unsigned i = r0;
        for (; i + 1u < r1; i += 2u) {
            const f32x4 va = row(i), vb = row(i + 1u);
            store16<NT>(o4 + (size_t)(i - r0) * T, va);
            store16<NT>(o4 + (size_t)(i + 1u - r0) * T, vb);
        }
        if (i < r1) store16<NT>(o4 + (size_t)(i - r0) * T, row(i));
        return;
    }
#pragma unroll 4
    for (unsigned idx = tid; idx < total; idx += T) {
        const unsigned lr = (unsigned)(((unsigned long long)idx * a.q_magic) >> 40);   // idx / q_per_row
        const unsigned q = idx - lr * q_per_row;
        const unsigned i = r0 + lr;
        const unsigned f = q * VEC;
        if (VEC == 4) {
            const f32x2 lo = reinterpret_cast<const f32x2*>(t_flat)[src_col(f, i) >> 1];      // even index: one ds_read_b64
            const f32x2 hi = reinterpret_cast<const f32x2*>(t_flat)[src_col(f + 2u, i) >> 1];
            const f32x4 v = {lo.x, lo.y, hi.x, hi.y};
            store16<NT>(reinterpret_cast<f32x4*>(out + (size_t)idx * 4), v);
        } else {
            const f32x2 v = reinterpret_cast<const f32x2*>(t_flat)[src_col(f, i) >> 1];      // even index: one ds_read_b64
            f32x2* p = reinterpret_cast<f32x2*>(out + (size_t)idx * 2);
            if (NT) __builtin_nontemporal_store(v, p); else *p = v;
        }
    }
}

// Flat slabs - the default for 16-byte rows since round 4: the env's [N][6N] block is one flat array of N * 6N / 4 float4 and a workgroup
// of T threads writes `flat_passes` x T consecutive float4 of it, whatever the row length - the shape of the fastest fill of the
// probe family (1024 threads x 2 passes = 32 KB per workgroup, 16 KB contiguous per pass), where the row-aligned default writes
// 2 x 12 KB.  A lane's (row, column) differs per pass: two multiply-shift divisions per lane and launch.
template <int NT>
__global__ __launch_bounds__(1024) void obs_expand_flat_kernel(const ObsArgs a) {
    extern __shared__ __align__(16) float t_flat[];          // [6N]
    const unsigned N = a.N, tid = threadIdx.x, T = blockDim.x;
    unsigned env, chunk;
    if (a.xcd_remap) {
        const unsigned bid = blockIdx.x, lane8 = bid & 7u, rest = bid >> 3;
        chunk = rest % a.chunks; env = (rest / a.chunks) * 8u + lane8;
    } else {
        env = blockIdx.x / a.chunks; chunk = blockIdx.x % a.chunks;
    }
    const unsigned row_floats = 6u * N, q_per_row = a.q_per_row, total = N * q_per_row;
    {
        const f32x2* src = reinterpret_cast<const f32x2*>(a.table + (size_t)env * row_floats);
        f32x2* dst = reinterpret_cast<f32x2*>(t_flat);
        for (unsigned k = tid; k < row_floats / 2; k += T) dst[k] = src[k];
    }
    __syncthreads();
    if (a.stagger > 0) {                                                  // D2D_TUNE_OBS_STAGGER (A/B)
        const int n = (int)(tid >> 6) * a.stagger;
        for (int k = 0; k < n; ++k) __builtin_amdgcn_s_sleep(1);
    }
    const f32x2* t2 = reinterpret_cast<const f32x2*>(t_flat);
    f32x4* out = reinterpret_cast<f32x4*>(a.obs + (size_t)env * N * row_floats);
    const unsigned base = chunk * (unsigned)a.rows_per_wg * T;            // rows_per_wg = passes per workgroup here
    f32x4 v[4];
    unsigned idx[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        idx[p] = base + (unsigned)p * T + tid;
        if (p < a.rows_per_wg && idx[p] < total) {
            const unsigned i = (unsigned)(((unsigned long long)idx[p] * a.q_magic) >> 40), q = idx[p] - i * q_per_row;
            const unsigned f = q * 4u;
            const f32x2 lo = t2[src_col(f, i) >> 1], hi = t2[src_col(f + 2u, i) >> 1];
            v[p] = f32x4{lo.x, lo.y, hi.x, hi.y};
        }
    }
#pragma unroll
    for (int p = 0; p < 4; ++p)
        if (p < a.rows_per_wg && idx[p] < total) store16<NT>(out + idx[p], v[p]);
}

// The same expansion written as float64 (d2d_set_obs_dtype: the reference's observation dtype, obs_fn.py:47,51 builds float64
// arrays): every aligned float2 of T becomes one 16-byte double2 store, so the widening happens on the way out and the [B,N,6N]
// block is written once - 48 N bytes per agent-step instead of 24 N written, 24 N re-read and 48 N written by a separate cast.
template <bool NT>
__global__ __launch_bounds__(1024) void obs_expand_f64_kernel(const ObsArgs a) {
    extern __shared__ __align__(16) float t_flat[];          // [6N]
    typedef double f64x2 __attribute__((ext_vector_type(2)));
    const unsigned N = a.N, tid = threadIdx.x, T = blockDim.x;
    unsigned env, chunk;
    if (a.xcd_remap) {
        const unsigned bid = blockIdx.x, lane8 = bid & 7u, rest = bid >> 3, G = (unsigned)a.xcd_remap;
        const unsigned per_group = a.chunks * G, group = rest / per_group, within = rest % per_group;
        chunk = within / G;
        env = (group * G + within % G) * 8u + lane8;
    } else {
        env = blockIdx.x / a.chunks;
        chunk = blockIdx.x % a.chunks;
    }
    const unsigned row_floats = 6u * N, q2 = row_floats / 2u;
    {
        const f32x2* src = reinterpret_cast<const f32x2*>(a.table + (size_t)env * row_floats);
        f32x2* dst = reinterpret_cast<f32x2*>(t_flat);
        for (unsigned k = tid; k < q2; k += T) dst[k] = src[k];
    }
    __syncthreads();
    const unsigned r0 = chunk * a.rows_per_wg;
    const unsigned r1 = min(r0 + (unsigned)a.rows_per_wg, N);
    const f32x2* t2 = reinterpret_cast<const f32x2*>(t_flat);
    double* out = reinterpret_cast<double*>(a.obs) + ((size_t)env * N + r0) * row_floats;
    for (unsigned i = r0; i < r1; ++i) {
        const unsigned head = 6u * i;
        f64x2* o = reinterpret_cast<f64x2*>(out + (size_t)(i - r0) * row_floats);
#pragma unroll 2
        for (unsigned c = tid; c < q2; c += T) {
            const unsigned f = 2u * c;
            const f32x2 v = t2[src_col(f, i) >> 1];
            const f64x2 d = {(double)v.x, (double)v.y};
            if (NT) __builtin_nontemporal_store(d, o + c); else o[c] = d;
        }
        (void)head;
    }
}

// Flat slabs in float64 (round 5): the shape of obs_expand_flat_kernel for the reference's observation dtype.  A 16-byte piece is
// one double2 = ONE aligned float2 of T, so the env's [N][6N] block is a flat array of N * 3N pieces; a 1024-thread workgroup
// writes `passes` x 1024 consecutive pieces (16 KB per pass) whatever the row length, every lane's (row, column) found once per
// piece by multiply-shift, all LDS reads issued ahead of the stores.  (The row-aligned kernel above walked a row with a strided
// loop and a three-way source select per element: 6.6 TB/s where this shape's float32 twin does 7.1 - 7.2, profiles/r4_obs_float64.jsonl.)
template <int NT>
__global__ __launch_bounds__(1024) void obs_expand_flat_f64_kernel(const ObsArgs a) {
    extern __shared__ __align__(16) float t_flat[];          // [6N]
    typedef double f64x2 __attribute__((ext_vector_type(2)));
    const unsigned N = a.N, tid = threadIdx.x, T = blockDim.x;
    unsigned env, chunk;
    if (a.xcd_remap) {
        const unsigned bid = blockIdx.x, lane8 = bid & 7u, rest = bid >> 3;
        chunk = rest % a.chunks; env = (rest / a.chunks) * 8u + lane8;
    } else {
        env = blockIdx.x / a.chunks; chunk = blockIdx.x % a.chunks;
    }
    const unsigned row_floats = 6u * N, q_per_row = a.q_per_row, total = N * q_per_row;      // q_per_row = 3N double2 per row
    {
        const f32x2* src = reinterpret_cast<const f32x2*>(a.table + (size_t)env * row_floats);
        f32x2* dst = reinterpret_cast<f32x2*>(t_flat);
        for (unsigned k = tid; k < row_floats / 2; k += T) dst[k] = src[k];
    }
    __syncthreads();
    const f32x2* t2 = reinterpret_cast<const f32x2*>(t_flat);
    f64x2* out = reinterpret_cast<f64x2*>(reinterpret_cast<double*>(a.obs) + (size_t)env * N * row_floats);
    const unsigned base = chunk * (unsigned)a.rows_per_wg * T;            // rows_per_wg = passes per workgroup here
    f32x2 v[4];
    unsigned idx[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        idx[p] = base + (unsigned)p * T + tid;
        if (p < a.rows_per_wg && idx[p] < total) {
            const unsigned i = (unsigned)(((unsigned long long)idx[p] * a.q_magic) >> 40), c = idx[p] - i * q_per_row;
            v[p] = t2[src_col(2u * c, i) >> 1];
        }
    }
#pragma unroll
    for (int p = 0; p < 4; ++p)
        if (p < a.rows_per_wg && idx[p] < total) {
            const f64x2 d = {(double)v[p].x, (double)v[p].y};
            if (NT) __builtin_nontemporal_store(d, out + idx[p]); else out[idx[p]] = d;
        }
}

// Variant without LDS staging or barrier (float4 rows only): every thread fetches its two source float2 straight
// from T in global memory (L1 / the XCD's L2 after the first touch) and stores.  Selected by D2D_TUNE_OBS_VARIANT=1;
// kept for A/B measurement against the staged kernel (tools/tune_obs.py).
__global__ __launch_bounds__(1024) void obs_expand_direct_kernel(const ObsArgs a) {
    const unsigned N = a.N, tid = threadIdx.x, T = blockDim.x;
    unsigned env, chunk;
    if (a.xcd_remap) {
        const unsigned bid = blockIdx.x, lane8 = bid & 7u, rest = bid >> 3;
        chunk = rest % a.chunks;
        env = (rest / a.chunks) * 8u + lane8;
    } else {
        env = blockIdx.x / a.chunks;
        chunk = blockIdx.x % a.chunks;
    }
    const unsigned row_floats = 6u * N;
    const float* t_flat = a.table + (size_t)env * row_floats;
    const unsigned r0 = chunk * a.rows_per_wg;
    const unsigned r1 = min(r0 + (unsigned)a.rows_per_wg, N);
    const unsigned q_per_row = a.q_per_row;
    const unsigned total = (r1 - r0) * q_per_row;
    float* out = a.obs + ((size_t)env * N + r0) * row_floats;
#pragma unroll 2
    for (unsigned idx = tid; idx < total; idx += T) {
        const unsigned lr = (unsigned)(((unsigned long long)idx * a.q_magic) >> 40);
        const unsigned q = idx - lr * q_per_row;
        const unsigned i = r0 + lr;
        const unsigned f = q * 4u;
        const f32x2 lo = reinterpret_cast<const f32x2*>(t_flat)[src_col(f, i) >> 1];      // global_load_dwordx2, L1/L2
        const f32x2 hi = reinterpret_cast<const f32x2*>(t_flat)[src_col(f + 2u, i) >> 1];
        const f32x4 v = {lo.x, lo.y, hi.x, hi.y};
        __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(out + (size_t)idx * 4));
    }
}

hipError_t launch_obs_expand(const ObsArgs& a, hipStream_t stream) {
    const size_t lds = (size_t)a.N * 6 * sizeof(float);
    dim3 grid((unsigned)a.B * (unsigned)a.chunks), block(a.block > 0 ? a.block : 256);
    if (a.out_f64) {
        if (a.variant == 2 && a.rows_per_wg <= 4) {            // flat slabs (the default)
            if (a.nontemporal) hipLaunchKernelGGL((obs_expand_flat_f64_kernel<1>), grid, block, lds, stream, a);
            else hipLaunchKernelGGL((obs_expand_flat_f64_kernel<0>), grid, block, lds, stream, a);
        } else if (a.nontemporal) hipLaunchKernelGGL(obs_expand_f64_kernel<true>, grid, block, lds, stream, a);
        else hipLaunchKernelGGL(obs_expand_f64_kernel<false>, grid, block, lds, stream, a);
        return hipGetLastError();
    }
    if (a.variant == 1 && a.vec == 4) {
        hipLaunchKernelGGL(obs_expand_direct_kernel, grid, block, 0, stream, a);
        return hipGetLastError();
    }
    if (a.variant == 2 && a.vec == 4 && a.rows_per_wg <= 4) {
        switch (a.nontemporal) {
            case 0: hipLaunchKernelGGL((obs_expand_flat_kernel<0>), grid, block, lds, stream, a); break;
            case 2: hipLaunchKernelGGL((obs_expand_flat_kernel<2>), grid, block, lds, stream, a); break;
            case 3: hipLaunchKernelGGL((obs_expand_flat_kernel<3>), grid, block, lds, stream, a); break;
            case 4: hipLaunchKernelGGL((obs_expand_flat_kernel<4>), grid, block, lds, stream, a); break;
            case 5: hipLaunchKernelGGL((obs_expand_flat_kernel<5>), grid, block, lds, stream, a); break;
            default: hipLaunchKernelGGL((obs_expand_flat_kernel<1>), grid, block, lds, stream, a); break;
        }
        return hipGetLastError();
    }
    if (a.vec == 4) {
        switch (a.nontemporal) {
            case 0: hipLaunchKernelGGL((obs_expand_kernel<4, 0>), grid, block, lds, stream, a); break;
            case 2: hipLaunchKernelGGL((obs_expand_kernel<4, 2>), grid, block, lds, stream, a); break;
            case 3: hipLaunchKernelGGL((obs_expand_kernel<4, 3>), grid, block, lds, stream, a); break;
            case 4: hipLaunchKernelGGL((obs_expand_kernel<4, 4>), grid, block, lds, stream, a); break;
            case 5: hipLaunchKernelGGL((obs_expand_kernel<4, 5>), grid, block, lds, stream, a); break;
            default: hipLaunchKernelGGL((obs_expand_kernel<4, 1>), grid, block, lds, stream, a); break;
        }
    } else {
        if (a.nontemporal) hipLaunchKernelGGL((obs_expand_kernel<2, 1>), grid, block, lds, stream, a);
        else hipLaunchKernelGGL((obs_expand_kernel<2, 0>), grid, block, lds, stream, a);
    }
    return hipGetLastError();
}

}  // namespace d2d
