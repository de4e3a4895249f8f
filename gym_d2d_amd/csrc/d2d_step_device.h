// Device-side helpers shared by the step kernels (d2d_step.hip: the generic / fused kernels; d2d_rollout.hip: the rollout
// kernel): lane exchange, the power-law arithmetic, link records and their decode, the LDS carve-up.  gfx950 only.
#pragma once
#include "d2d_internal.h"
#include <type_traits>

namespace d2d {

#define LIKELY(x) __builtin_expect(!!(x), 1)
#define UNLIKELY(x) __builtin_expect(!!(x), 0)
#define COLD_LOOP _Pragma("clang loop vectorize(disable) interleave(disable) unroll(disable)")

typedef unsigned long long u64;
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

#define FLAG_ZERO_DISTANCE 1
#define FLAG_RB_OOR 2
#define FLAG_NON_FINITE 4

#define LINK_UPLINK 1
#define LINK_DOWNLINK 2
#define LINK_SIDELINK 3

// DPP lane exchange (VALU, ~8 cycles) instead of ds_bpermute (LDS crossbar, > 100 cycles) for the in-row steps.
template <int CTRL>
__device__ __forceinline__ float dpp_f32(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}

// Sum over the 64 lanes, returned wave-uniform: quad butterfly (xor 1, xor 2), half-row mirror, row mirror - every lane of a
// 16-lane row then holds its row's sum (each lane adds its partner's partial: a + b == b + a) - then the gfx9 row
// broadcasts (lane 15 of a row into the next row, lane 31 into rows 2-3) leave the total in lane 63, which is read with
// v_readlane.  All DPP: no ds_bpermute round trip through the LDS crossbar (two of them before, > 100 cycles each).
__device__ __forceinline__ float wave_sum(float v) {
    v += dpp_f32<0xB1>(v);          // quad_perm:[1,0,3,2]
    v += dpp_f32<0x4E>(v);          // quad_perm:[2,3,0,1]
    v += dpp_f32<0x141>(v);         // row_half_mirror
    v += dpp_f32<0x140>(v);         // row_mirror
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x142, 0xA, 0xF, false));   // row_bcast:15 -> rows 1, 3
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x143, 0xC, 0xF, false));   // row_bcast:31 -> rows 2, 3
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

// The sums of the wave's lower and upper 32 lanes, each by the tree wave_sum builds over 32 lanes (its first five levels): what
// a kernel with two ADJACENT links per lane needs to reproduce wave_sum over 64 links - the lane's own pair is level one.
__device__ __forceinline__ void wave_sum_halves(float v, float& lo, float& hi) {
    v += dpp_f32<0xB1>(v);
    v += dpp_f32<0x4E>(v);
    v += dpp_f32<0x141>(v);
    v += dpp_f32<0x140>(v);
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x142, 0xA, 0xF, false));   // row_bcast:15 -> rows 1, 3
    lo = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 31));
    hi = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

// (d^2)^h for h = -e/2 given as a head + tail pair (h.x: the 12 leading bits of h, so that h.x * exponent is exact; h.y = h - h.x,
// both built by the host in double precision, d2d_capi.hip::refresh_tables), to ~2.5e-7 relative.  log2 of the mantissa and the
// integer exponent are handled separately so the error does not scale with |log2(d^2)| (a plain exp2(h*log2(x)) loses ~2e-6), and
// the EXPONENT ITSELF carries more than float precision: a float32 e is off by up to 6e-8 relative, which (d^2)^(-e/2) amplifies by
// ln(d^2) * e/2 - 1.2e-6 at 300 m with COST-Hata's e = 3.5, the largest single term of the power-law modes' error until round 3.
__device__ __forceinline__ float pow_neg_half(float d2, float2 h) {
    // Contraction OFF in here: hipcc's default -ffp-contract=fast fuses products into the additions below, and the head / tail
    // arithmetic then adds exact residuals to unrounded terms.  Measured on the hardware (tools/probes/pow_accuracy.hip, the r2 form
    // of this function): max relative error 1.56e-6 as the compiler fused it, 2.5e-7 as written.
#pragma clang fp contract(off)
    const float m = __builtin_amdgcn_frexp_mantf(d2);       // [0.5, 1)
    const float fe = (float)__builtin_amdgcn_frexp_expf(d2);
    // v_log_f32 = log2, in [-1, 0); log2(0) = -inf is held at -1e30 so that the tail product below stays finite and d2 == 0
    // still ends in exp2(+huge) = +inf (0 * -inf or +x * -inf in the tail made it NaN, unlike the 1 / d^2 mode)
    const float l = fmaxf(__builtin_amdgcn_logf(m), -1.0e30f);
    const float p = h.x * fe;                               // exact: 12 bits x at most 8
    const float ip = rintf(p);
    const float fr = (p - ip) + fmaf(h.x, l, h.y * (l + fe));
    return __builtin_amdgcn_ldexpf(__builtin_amdgcn_exp2f(fr), (int)ip);
}

// 10^(p/10) for an integer power level p (dBm -> mW), ~1.5e-7 relative: 2^(p*c), c = log2(10)/10 split into a 12-bit
// head (p*c_hi exact for |p| < 4096) and a tail, so only the fractional part of the exponent reaches v_exp_f32.
__device__ __forceinline__ float pow10_tenth(int p) {
    // Contraction spelled out: under hipcc's default -ffp-contract=fast the compiler fused `(xh - ip) + fp * c_lo` into one fma
    // in one kernel and left a rounded product + add in another (v_pk_mul_f32 of the two constants) - a last-bit difference
    // between kernels that must agree bit for bit.  The fma form is the one every kernel had before round 5.
#pragma clang fp contract(off)
    const float fp = (float)p;
    const float xh = fp * 0.3321533203125f;                 // 2721 / 8192: exact product
    const float ip = floorf(xh);
    const float fr = fmaf(fp, 3.948917623623e-05f, xh - ip);   // c - c_hi
    return __builtin_amdgcn_ldexpf(__builtin_amdgcn_exp2f(fr), (int)ip);
}

// PL_POWK: (d^2)^(-n/2) for NP pairs at once, where every transmitter's exponent n lies within 1/2 of one integer k, the same for the
// whole launch (path_loss.py:48-66 with ple near k; COST-Hata's slopes 3.6 - 4.4, path_loss.py:90-123: k = 4):
//   (d^2)^(-k/2)   r = v_rcp(d^2) multiplied up: r, r r, (r r) r, ((r r) r) r; one v_rsq on top when k is odd - behind branches on k,
//                  which is wave-uniform, taken once for the whole group of pairs;
//   (d^2)^phi      exp2(phi log2(d^2)), phi = -(n - k) / 2 in [-1/4, 1/4]: v_log's relative ulp on log2(d^2) <= 20 is an absolute
//                  3e-7 in the exponent at most (the general split, pow_neg_half, exists because at |h| ~ 2 the same ulp is 2.4e-6).
// About 4e-7 relative in all.  Every kernel calls this with the same association of the products (NP = 1 for a single pair), so
// they agree bit for bit.  A zero distance ends non-finite (inf from the reciprocal, or NaN from inf * 0), as with 1 / d^2.
// KC != 0: k is known at compile time (the rollout kernel branches ONCE on the common case k == 4 and runs a copy of its pair section
// without the tests on k: a taken branch costs a wave its instruction buffer); the operations, hence the bits, are the same.
template <int NP, int KC = 0>
__device__ __forceinline__ void pow_k_gains(const float (&d2)[NP], const float (&phi)[NP], int k_runtime, float (&g)[NP]) {
    const int k = KC ? KC : k_runtime;
    float r[NP], e[NP];
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        r[p] = __builtin_amdgcn_rcpf(d2[p]);
        e[p] = __builtin_amdgcn_exp2f(phi[p] * __builtin_amdgcn_logf(d2[p]));
    }
    if (k >= 2) {
#pragma unroll
        for (int p = 0; p < NP; ++p) g[p] = r[p];
    } else {
#pragma unroll
        for (int p = 0; p < NP; ++p) g[p] = 1.0f;
    }
    // r, r r, (r r) r, ((r r) r) r: a wave-uniform trip count (a loop, not four tests: fewer scalar masks alive in the generic kernels;
    // unrolled away when k is a compile-time constant)
    if (KC) {
#pragma unroll
        for (int q = 4; q <= KC; q += 2) {
#pragma unroll
            for (int p = 0; p < NP; ++p) g[p] *= r[p];
        }
    } else {
        COLD_LOOP
        for (int q = 4; q <= k; q += 2) {
#pragma unroll
            for (int p = 0; p < NP; ++p) g[p] *= r[p];
        }
    }
    if (k & 1) {
#pragma unroll
        for (int p = 0; p < NP; ++p) g[p] *= __builtin_amdgcn_rsqf(d2[p]);
    }
#pragma unroll
    for (int p = 0; p < NP; ++p) g[p] *= e[p];
}

template <int MODE>
__device__ __forceinline__ float pair_gain(float d2, float2 h, int k = 0) {
    if (MODE == PL_INV_SQUARE) return __builtin_amdgcn_rcpf(d2);
    if (MODE == PL_POWK) {
        const float d[1] = {d2}, f[1] = {h.x};
        float g[1];
        pow_k_gains<1>(d, f, k, g);
        return g[0];
    }
    return pow_neg_half(d2, h);
}

// Philox4x32-10 (same generator as csrc/d2d_reset.hip), used for the per-call Gaussian of ShadowingPathLoss.
__device__ __forceinline__ void philox_step(unsigned c0, unsigned c1, unsigned c2, unsigned c3, unsigned k0, unsigned k1,
                                            unsigned& o0, unsigned& o1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const unsigned hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        const unsigned hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        const unsigned n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    o0 = c0; o1 = c1;
}

// cos(2 pi u) for u = k * 2^-24 in [0, 1): the quadrant reduction is EXACT in these units (4u splits into an integer quadrant and a
// fraction with no rounding), then one rounding of x = f * pi / 2 and a minimax polynomial on [-pi/4, pi/4] (the reset sampler's
// sincos_turns, csrc/d2d_reset.hip).  cosf(fl(2 pi u)) carries the argument's rounding - 3.7e-7 near 2 pi, times a Gaussian of up
// to 5.9, times chi dB: most of the 1e-5 bar where |sinr_db| < 1.
__device__ __forceinline__ float cos_turns(float u) {
    const float t = 4.0f * u;                                           // [0, 4), exact
    float q = floorf(t);
    float f = t - q;                                                    // [0, 1), exact
    if (f > 0.5f) { f -= 1.0f; q += 1.0f; }                             // [-0.5, 0.5], exact
    const float x = f * 1.5707963267948966f, z = x * x;
    const float s = fmaf(x * z, fmaf(fmaf(-1.9515295891e-4f, z, 8.3321608736e-3f), z, -1.6666654611e-1f), x);
    const float c = fmaf(z * z, fmaf(fmaf(2.443315711809948e-5f, z, -1.388731625493765e-3f), z, 4.166664568298827e-2f), fmaf(-0.5f, z, 1.0f));
    const int k = (int)q;
    const float b = (k & 1) ? s : c;                                    // quarter turns: cos -> -sin -> -cos -> sin
    return ((k + 1) & 2) ? -b : b;
}

// Linear-domain factor 10^(-X/10), X ~ N(0, chi^2) dB, for the call (tx link j -> rx link i, kind) of this step.
// ShadowingPathLoss.__call__ draws gauss(0, chi) on EVERY call with d > d0 (path_loss.py:76-79): the signal term of
// the SINR (kind 0, j == i), every interferer term (kind 0, j != i) and the SNR's own re-evaluation of the signal
// path loss (kind 1, simulator.py:114) are independent draws.
// Box-Muller to ~1e-7 absolute in z (round 6; the oracle does the same transform of the same words in double): u1 = (k + 0.5) 2^-24 is
// a float32 value only for k < 2^23 - above, k + 0.5 needs 25 bits and rounds, and near u1 = 1 the Gaussian's radius sqrt(-2 ln u1)
// amplifies that by 1 / (z u1) (z = 0.01: 3e-6; the largest k: the radius itself) - so the upper half goes through
// log1p(-(1 - u1)), whose argument (2^24 - k - 0.5) 2^-24 IS exact; the angle by cos_turns.
__device__ __forceinline__ float shadow_factor(const StepArgs& a, unsigned env, int j, int i, unsigned kind) {
    unsigned w0, w1;
    philox_step(env, a.shadow_step, (unsigned)j | ((unsigned)i << 16), kind, a.shadow_seed_lo, a.shadow_seed_hi, w0, w1);
    const unsigned k1 = w0 >> 8;
    const float lo_half = logf(((float)(k1 & 0x7FFFFFu) + 0.5f) * 5.9604644775390625e-08f);                 // k1 < 2^23: exact argument
    const float hi_half = log1pf(-(((float)(0x1000000u - k1) - 0.5f) * 5.9604644775390625e-08f));          // k1 >= 2^23: 1 - u1 exact
    const float nl = -(k1 < 0x800000u ? lo_half : hi_half);                                                 // -ln(u1), u1 in (0, 1)
    const float u2 = (float)(w1 >> 8) * 5.9604644775390625e-08f;               // [0, 1)
    const float z = sqrtf(2.0f * nl) * cos_turns(u2);                          // Box-Muller
    return exp2f(-0.33219280948873623f * a.shadow_chi * z);                     // 10^(-chi z / 10)
}

// LDS layout of ONE env (byte offsets, all computed on the host by step_lds_layout and passed in StepArgs::lds):
//   0    red[16] f32   wave partial sums            64   flags[4] i32   0: env flags  1: reward violated  2: ticket
//   80   link[N] float4  tx_x, tx_y, effective tx power (mW, incl. tx side of the PL constant), rb bits
//   then ONLY what the configuration reads back from LDS:
//        aux[N] i32 (tx_dev | type << 24)                       always (all-pairs fallback, table route)
//        rx[N] float2                                           strided links only (LPT == 0)
//        sinr[N], sh[N] f32                                     Shannon / CueSinrShannon rewards
//        expo[N] float2 (head, tail of -exponent / 2)          power-law / shadowing path loss
//        lo[N] float2 (low parts of tx_x, tx_y)                 exact positions (OPT_XPOS)
//        tflat[6N] f32                                          fused obs expansion
//   off_mask: mask[W][R] u32 per-RB membership, word-major (lanes with different RBs hit different banks),
//             side[W] u32 sidelink membership, summ[R] u32 (bit w set <=> mask[w][rb] != 0)
// At N = 512, R = 256, inverse-square path loss, SystemCapacity: 10.3 KB + 17.4 KB masks = 27.7 KB per env.
#define LDS_HEAD_BYTES 80u

struct Smem {
    float* red; int* flags; float4* link; float2* rx; float* sinr; float* sh; float2* expo; float2* lo; int* aux; float* tflat;
    unsigned* mask; unsigned* side; unsigned* summ;
    uint4* slots; unsigned* cnt;
};

__device__ __forceinline__ Smem carve(unsigned char* base, const StepLds& l, unsigned R, unsigned W) {
    Smem s;
    s.red = reinterpret_cast<float*>(base);
    s.flags = reinterpret_cast<int*>(base + 64);
    s.link = reinterpret_cast<float4*>(base + LDS_HEAD_BYTES);
    s.aux = reinterpret_cast<int*>(base + l.aux);
    s.rx = reinterpret_cast<float2*>(base + l.rx);
    s.sinr = reinterpret_cast<float*>(base + l.sinr);
    s.sh = reinterpret_cast<float*>(base + l.sh);
    s.expo = reinterpret_cast<float2*>(base + l.expo);
    s.lo = reinterpret_cast<float2*>(base + l.lo);
    s.tflat = reinterpret_cast<float*>(base + l.tflat);
    s.mask = reinterpret_cast<unsigned*>(base + l.mask);
    s.side = s.mask + R * W;
    s.summ = s.side + W;
    s.slots = reinterpret_cast<uint4*>(base + l.lists);
    s.cnt = reinterpret_cast<unsigned*>(base + l.lists + R * 16u);
    return s;
}

// base + 32-bit BYTE offset: selects to a global access with an SGPR base and ONE VGPR offset, so the 4-byte arrays of an env
// share a single offset register and no 64-bit address is formed per array (10 v_lshl_add_u64 per link before).  The host
// keeps B * N * 24 below 2^32 (run_step).
// The offset must be (re)defined in the basic block of the access - instruction selection works per block, and a 64-bit
// offset pair carried in from another block is added with a v_lshl_add_u64 - so callers pass it through here once per block.
__device__ __forceinline__ unsigned fresh(unsigned byte_off) {
    asm volatile("" : "+v"(byte_off));
    return byte_off;
}

template <class T>
__device__ __forceinline__ T* at(T* base, unsigned byte_off) {
    return reinterpret_cast<T*>(reinterpret_cast<unsigned char*>(const_cast<typename std::remove_const<T>::type*>(base)) + byte_off);
}

// Everything the kernel needs about one link, as it comes out of memory.  The loads are INDEPENDENT of one another
// (no link -> device -> position double hop, no power-table lookup, no per-type constant fetched behind the record):
// they are issued back to back in the prologue and first used after pass 0's barrier.
struct LinkRaw {
    int4 ra;             // rec_a
    float4 rb_;          // rec_b: tx_lin, rx_pl, rx_lin, noise_mw
    float4 rc;           // rec_c: sens_db, bw_mhz, exponent, (P | column << 16)
    float2 hh;           // rec_h: head / tail of -exponent / 2 (power-law and shadowing modes only)
    float4 pos;          // tx_x, tx_y, rx_x, rx_y
    float4 plo;          // OPT_XPOS: the low parts of the same four coordinates (StepArgs::lpos_lo), else zero
    int act0, act1;      // raw action, or explicit (rb, pwr)
};

// load through the constant address space: with a wave-uniform address the compiler selects s_load_dwordx4 (the data is
// written by the host between launches only)
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ i32x4 scalar_load16(const void* p) {
    typedef const __attribute__((address_space(4))) i32x4* cptr;
    return *reinterpret_cast<cptr>(reinterpret_cast<unsigned long long>(p));
}
typedef int i32x16 __attribute__((ext_vector_type(16)));
__device__ __forceinline__ i32x16 scalar_load64(const void* p) {
    typedef const __attribute__((address_space(4))) i32x16* cptr;
    return *reinterpret_cast<cptr>(reinterpret_cast<unsigned long long>(p));
}
__device__ __forceinline__ i32x2 scalar_load8(const void* p) {
    typedef const __attribute__((address_space(4))) i32x2* cptr;
    return *reinterpret_cast<cptr>(reinterpret_cast<unsigned long long>(p));
}

__device__ __forceinline__ LinkRaw load_link(const StepArgs& a, unsigned row, unsigned act_row, int i, int action_mode, int col_mode,
                                             bool no_fixed = false, bool srec = false, bool need_h = false, bool xpos = false) {
    LinkRaw in;
    in.act0 = 0; in.act1 = 0;
    in.plo = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    in.hh = make_float2(-1.0f, 0.0f);
    // actions first, branch-free per lane (uniform branches only), so that no join forces a wait on the loads in flight
    if (action_mode == 0) {
        if (no_fixed) {
            in.act0 = *at(a.actions, fresh((act_row + (unsigned)i) * 4u));   // every link has its own column: column = link index
        } else if (a.act_stride > 0) {
            // Fixed links are the first n_fixed links (the traffic-model case, CUE links first) or there are none: the
            // action column follows from the link index alone, so this load waits for nothing.  Arbitrary fixed sets
            // read their column from a host-built per-link array first (a second hop, but a uniform branch: no join
            // that would make the compiler wait for every load in flight).  A fixed link reads column 0 and ignores it.
            int col = i - a.n_fixed;
            if (col_mode != 0) col = a.act_cols[i];
            in.act0 = *at(a.actions, fresh((act_row + (unsigned)(col > 0 ? col : 0)) * 4u));
        }
    } else {
        const unsigned oe = fresh((row + (unsigned)i) * 4u);
        in.act0 = *at(a.rb_in, oe);
        in.act1 = *at(a.pwr_in, oe);
    }
    if (srec) {
        // the records of this wave's 64 links are identical (StepArgs::rec_uniform, checked by the host; device ids aside,
        // which only the table route reads): ONE scalar load per row and wave into SGPRs instead of 64 lanes x 16 bytes
        const int iu = __builtin_amdgcn_readfirstlane(i);
        const i32x4 va = scalar_load16(a.rec_a + iu), vb = scalar_load16(a.rec_b + iu), vc = scalar_load16(a.rec_c + iu);
        in.ra = make_int4(va.x, va.y, va.z, va.w);
        in.rb_ = make_float4(__int_as_float(vb.x), __int_as_float(vb.y), __int_as_float(vb.z), __int_as_float(vb.w));
        in.rc = make_float4(__int_as_float(vc.x), __int_as_float(vc.y), __int_as_float(vc.z), __int_as_float(vc.w));
        if (need_h) { const i32x2 vh = scalar_load8(a.rec_h + iu); in.hh = make_float2(__int_as_float(vh.x), __int_as_float(vh.y)); }
    } else {
        in.ra = a.rec_a[i];
        in.rb_ = a.rec_b[i];
        in.rc = a.rec_c[i];
        if (need_h) in.hh = a.rec_h[i];
    }
    in.pos = *at(a.lpos, fresh((row + (unsigned)i) * 16u));
    if (xpos) in.plo = *at(a.lpos_lo, fresh((row + (unsigned)i) * 16u));
    return in;
}

// One coordinate difference tx - rx.  Plain float32 positions: one subtraction.  OPT_XPOS (float64 positions uploaded as hi + lo
// pairs, d2d_set_positions_f64): (tx_hi - rx_hi) + (tx_lo - rx_lo) - the first difference is exact whenever the two are within a
// factor of two of each other (Sterbenz), i.e. exactly where absolute float32 coordinates lose the difference (a receiver 0.1 m from
// a transmitter 500 m out: 3e-5 m of rounding on each), and the low parts restore what the rounding to float32 took.  Every kernel
// forms it in this order, so they agree bit for bit.  (position.py:11-12: the reference subtracts Python floats.)
__device__ __forceinline__ float coord_diff(float tx_hi, float rx_hi, float tx_lo, float rx_lo) { return (tx_hi - rx_hi) + (tx_lo - rx_lo); }

// (rb, tx power dBm) of a link: fixed by the traffic model (traffic_model.py:15-32), decoded from the raw action
// (d2d_env.py:94-96, Python floor semantics; NB due_min_tx_power_dBm is not added back), or given explicitly.
__device__ __forceinline__ void decode_link(const StepArgs& a, const LinkRaw& in, unsigned act_row, int& rb, int& p, int action_mode,
                                            bool no_fixed = false) {
    if (!no_fixed && (in.ra.x & D2D_REC_FIXED_BIT)) {
        rb = in.ra.z; p = in.ra.w;            // no decode: any power is legal, as in the reference's Action(rb, pwr)
    } else if (action_mode == 0) {
        const int act = in.act0;
        const int P = (int)(__float_as_uint(in.rc.w) & 0xFFFFu);
        int q, r;
        if (__builtin_expect((unsigned)act <= (unsigned)in.ra.w, 1)) {
            // q = floor(act * M / 2^32), M = ceil(2^32 / P): exact while act * (M P - 2^32) < 2^32, i.e. for act < 2^32 / P;
            // the host stores the bound (capped at 2^24 - 1 so that q * P is a 24-bit multiply) next to M.  One
            // v_mul_hi_u32; a negative action is a huge unsigned and takes the other arm, as does a link without a magic
            // (bound 0: only act == 0 passes, and M = 0 decodes it correctly)
            q = (int)__umulhi((unsigned)act, (unsigned)in.ra.z);
            r = act - (int)__umul24((unsigned)q, (unsigned)P);
        } else {
            q = act / P; r = act - q * P;
            if (r < 0) { r += P; q -= 1; }
        }
        rb = q; p = r;
    } else {
        rb = in.act0; p = in.act1;
    }
}

// x / y with v_rcp_f32 (1 ulp) instead of the IEEE division sequence (~10 VALU): 2e-7 relative.  Used where the
// quotient's error is second order (the x / (u - 1) factor of log2(1 + x)).
__device__ __forceinline__ float fast_div(float x, float y) { return x * __builtin_amdgcn_rcpf(y); }

// x / y to within an ulp: v_rcp_f32 + one Newton step on the quotient (4 VALU instead of ~10; no denormal / overflow
// special cases - the operands here are powers in mW, far from both).  The SINR / SNR quotients use this one: at
// |value| < 1 dB the 1e-5 bar is 1e-5 dB absolute = 2.3e-6 relative, and the power-law modes need that headroom.
__device__ __forceinline__ float precise_div(float x, float y) {
    const float r = __builtin_amdgcn_rcpf(y);
    const float q = x * r;
    return fmaf(fmaf(-y, q, x), r, q);
}

// source float index inside T_flat for output column f (even) of row i (obs_fn.py:43-53: own link first, then the
// others in agent order)
__device__ __forceinline__ unsigned obs_src_col(unsigned f, unsigned i) {
    const unsigned head = 6u * i;
    return f < 6u ? head + f : (f < head + 6u ? f - 6u : f);
}

// Ascending sort of eight keys in registers (Batcher's odd-even merge sort, 19 compare-exchanges = 38 VALU, no branches): the
// receiver's view of its RB's member list.  Empty slots hold 0xFFFF and sink to the end.
__device__ __forceinline__ void sort8(unsigned (&v)[8]) {
#define D2D_CE(x, y) { const unsigned lo_ = min(v[x], v[y]); v[y] = max(v[x], v[y]); v[x] = lo_; }
    D2D_CE(0, 1) D2D_CE(2, 3) D2D_CE(4, 5) D2D_CE(6, 7)
    D2D_CE(0, 2) D2D_CE(1, 3) D2D_CE(4, 6) D2D_CE(5, 7)
    D2D_CE(1, 2) D2D_CE(5, 6)
    D2D_CE(0, 4) D2D_CE(1, 5) D2D_CE(2, 6) D2D_CE(3, 7)
    D2D_CE(2, 4) D2D_CE(3, 5)
    D2D_CE(1, 2) D2D_CE(3, 4) D2D_CE(5, 6)
#undef D2D_CE
}

#define LIST_EMPTY 0xFFFFu
#define LIST_SLOTS 8

// masks + sidelink words + summaries of one env, 16 bytes per store (the region is 16-byte aligned and padded)
template <bool FULL>
__device__ __forceinline__ void clear_masks(const Smem& s, int R, int W, int lt, int TPE) {
    uint4* m16 = reinterpret_cast<uint4*>(s.mask);
    const int n16 = (R * W + W + R + 3) >> 2;
    if (FULL) {
        // whole rounds with a wave-uniform trip count (scalar loop, no exec masking), then one predicated tail
        int k0 = 0;
#pragma unroll 1
        for (; k0 + TPE <= n16; k0 += TPE) m16[lt + k0] = make_uint4(0u, 0u, 0u, 0u);
        if (lt < n16 - k0) m16[lt + k0] = make_uint4(0u, 0u, 0u, 0u);
    } else {
        for (int k = lt; k < n16; k += TPE) m16[k] = make_uint4(0u, 0u, 0u, 0u);
    }
}

// LDS by raw byte address.  The kernels' only LDS object is the dynamic block `extern __shared__ smem_raw[]`, which sits at LDS
// address 0 - but as a SYMBOL, whose "+ 0" survives as a real v_add / s_add in front of every access built from a run-time offset
// (eight of them in the rollout kernel's pair loop).  The rollout kernel addresses its arrays by the byte offsets of StepLds
// directly; launch_rollout checks that the kernel has no static LDS in front of the dynamic block.
#define D2D_LDS(T) __attribute__((address_space(3))) T
template <class T> __device__ __forceinline__ T lds_get(unsigned addr) { return *(const D2D_LDS(T)*)(addr); }
template <class T> __device__ __forceinline__ void lds_put(unsigned addr, T v) { *(D2D_LDS(T)*)(addr) = v; }
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// branch weights: block placement moves the rare arms (invalid actions, all-pairs sweep, flag reporting) behind the hot path
// rare arms (a reward rule's search when it fires, the sweep fallbacks): LLVM's loop vectoriser otherwise unrolls and widens
// them into hundreds of instructions whose live values spill the hot path's scalars

// kernel options (template parameter OPT of step_kernel / rollout_kernel)
#define OPT_LISTS 1      /* generic kernels: per-RB member lists instead of the masks (StepArgs::walk == 2) */
#define OPT_SREC 2       /* rollout kernel: link records by scalar loads (StepArgs::rec_uniform) */
#define OPT_NT 4         /* rollout kernel: nontemporal result stores (StepArgs::nt_results) */
#define OPT_PAD 8        /* rollout kernel: N is no multiple of 64 (threads beyond the last link shadow it) */
#define OPT_XPOS 16      /* exact positions: every coordinate is a (hi, lo) float pair (StepArgs::lpos_lo, d2d_set_positions_f64) */

}  // namespace d2d
