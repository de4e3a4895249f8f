// Fused per-step kernel for gfx950 (MI355X): action decode -> received signal -> same-RB interference
// reduction -> SINR / SNR / Shannon rate / capacity -> reward -> compact observation table.
//
// Reference path being replaced (file:line under /root/reference/src/gym_d2d):
//   D2DEnv._decode_action            envs/d2d_env.py:93-101
//   Simulator._calculate_sinrs       simulator.py:89-108      (hot loop #1)
//   Simulator._calculate_snrs        simulator.py:110-116
//   Simulator._calculate_rates       simulator.py:118-127
//   Simulator._calculate_network_capacity  simulator.py:144-154
//   Actions.get_actions_by_rb        actions.py:27-31
//   {SystemCapacity,Shannon,CueSinrShannon}RewardFunction   envs/reward_fn.py:22-78
//   LinearObsFunction._agent_obs     envs/obs_fn.py:55-61
//
// Design (one workgroup = one environment; environments are independent):
//   * every link's transmitter tuple (tx_x, tx_y, effective tx power in mW, rb) is staged once in LDS as a
//     float4, so the interference loop is one ds_read_b128 per candidate interferer;
//   * same-RB interferers are found through per-RB membership bitmasks in LDS (R x ceil(N/64) u64 words,
//     built with ds_or_b64 - order independent), walked in ascending link order with ctz; a masked
//     all-pairs sweep is the fallback (rb outside [0,R), or mask table too large) and produces bit-identical
//     sums because both walk interferers in ascending index order through the same fmaf;
//   * all arithmetic is in the LINEAR domain (mW): sinr = S / (I + N) with one 10*log10 at the end; the
//     literal dB-domain transcription cannot hold 1e-5 relative in fp32 (hard parts);
//   * reductions (capacity sum) are xor-butterfly wave reductions + a fixed-order cross-wave sum:
//     run-to-run deterministic.
#include "d2d_internal.h"

namespace d2d {

typedef unsigned long long u64;

#define FLAG_ZERO_DISTANCE 1
#define FLAG_RB_OOR 2
#define FLAG_NON_FINITE 4

#define LINK_UPLINK 1
#define LINK_DOWNLINK 2
#define LINK_SIDELINK 3

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// (d^2)^(-e/2) for arbitrary exponent e, to ~3e-7 relative.  log2 of the mantissa and the integer exponent
// are handled separately so the error does not scale with |log2(d^2)| (a plain exp2(h*log2(x)) loses ~2e-6).
__device__ __forceinline__ float pow_neg_half(float d2, float e) {
    const float m = __builtin_amdgcn_frexp_mantf(d2);       // [0.5, 1)
    const float fe = (float)__builtin_amdgcn_frexp_expf(d2);
    const float l = __builtin_amdgcn_logf(m);               // v_log_f32 = log2, in [-1, 0)
    const float h = -0.5f * e;
    const float p = h * fe;
    const float perr = fmaf(h, fe, -p);                     // exact residual of the product
    const float ip = rintf(p);
    const float fr = (p - ip) + fmaf(h, l, perr);
    return __builtin_amdgcn_ldexpf(__builtin_amdgcn_exp2f(fr), (int)ip);
}

template <int MODE>
__device__ __forceinline__ float pair_gain(float d2, float e) {
    if (MODE == PL_INV_SQUARE) return __builtin_amdgcn_rcpf(d2);
    return pow_neg_half(d2, e);
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

// Linear-domain factor 10^(-X/10), X ~ N(0, chi^2) dB, for the call (tx link j -> rx link i, kind) of this step.
// ShadowingPathLoss.__call__ draws gauss(0, chi) on EVERY call with d > d0 (path_loss.py:76-79): the signal term of
// the SINR (kind 0, j == i), every interferer term (kind 0, j != i) and the SNR's own re-evaluation of the signal
// path loss (kind 1, simulator.py:114) are independent draws.
__device__ __forceinline__ float shadow_factor(const StepArgs& a, unsigned env, int j, int i, unsigned kind) {
    unsigned w0, w1;
    philox_step(env, a.shadow_step, (unsigned)j | ((unsigned)i << 16), kind, a.shadow_seed_lo, a.shadow_seed_hi, w0, w1);
    const float u1 = ((float)(w0 >> 8) + 0.5f) * 5.9604644775390625e-08f;      // (0, 1)
    const float u2 = (float)(w1 >> 8) * 5.9604644775390625e-08f;               // [0, 1)
    const float z = sqrtf(-2.0f * logf(u1)) * cosf(6.283185307179586f * u2);   // Box-Muller
    return exp2f(-0.33219280948873623f * a.shadow_chi * z);                     // 10^(-chi z / 10)
}

// LDS carve-up.  40 bytes per link + masks.
struct Smem {
    float4* link;   // [N] tx_x, tx_y, effective tx power (mW, incl. tx side of the PL constant), rb bits
    float2* rx;     // [N] rx_x, rx_y
    float* sinr;    // [N] sinr_db
    float* sh;      // [N] log2(1 + sinr_lin)
    float* expo;    // [N] path-loss exponent of the tx
    int* aux;       // [N] tx_dev | link_type << 24
    float* red;     // [32]
    int* flags;     // [4]  0: env flags  1: reward violated
    u64* mask;      // [W][R] per-RB membership (word-major: lanes with different RBs hit different banks), then [W] sidelink membership
    unsigned* summ; // [R] bit w set <=> mask[rb][w] != 0: lets a receiver skip the empty words of its RB
};

__device__ __forceinline__ Smem carve(unsigned char* base, int N) {
    Smem s;
    s.link = reinterpret_cast<float4*>(base);
    s.rx = reinterpret_cast<float2*>(s.link + N);
    s.sinr = reinterpret_cast<float*>(s.rx + N);
    s.sh = s.sinr + N;
    s.expo = s.sh + N;
    s.aux = reinterpret_cast<int*>(s.expo + N);
    s.red = reinterpret_cast<float*>(s.aux + N);
    s.flags = reinterpret_cast<int*>(s.red + 32);
    s.mask = reinterpret_cast<u64*>(s.flags + 4);
    return s;
}

size_t step_lds_bytes(int N, int R, int mask_words) {
    size_t bytes = (size_t)N * 40 + 32 * 4 + 4 * 4;
    bytes = (bytes + 7) & ~(size_t)7;
    if (mask_words > 0) bytes += ((size_t)R * mask_words + mask_words + (size_t)(R + 1) / 2) * 8;   // masks + summaries
    return bytes;
}

// Everything pass 1 needs about one link, in registers.
struct LinkIn {
    int type, txd, rb, p;
    float txx, txy, rxx, rxy, tx_lin, p10;
};

__device__ __forceinline__ LinkIn load_link(const StepArgs& a, const float* px, const float* py, size_t row, int i) {
    LinkIn in;
    // hop 1: link table, per-link constant, action
    in.type = a.link_type[i];
    in.txd = a.link_tx[i];
    const int rxd = a.link_rx[i];
    in.tx_lin = a.lk_tx_lin[i];
    if (a.action_mode == 0) {
        // d2d_env.py:94-96 with Python floor semantics; NB due_min_tx_power_dBm is not added back
        const int act = a.actions[row + i];
        const int P = in.type == LINK_SIDELINK ? a.p_due : (in.type == LINK_UPLINK ? a.p_cue : a.p_mbs);
        int q, r;
        if (act >= 0) {
            // exact for 0 <= act < 2^31 and P < 2^9: q = floor(act * ceil(2^40 / P) / 2^40) (host-computed magic)
            const unsigned long long M = in.type == LINK_SIDELINK ? a.m_due : (in.type == LINK_UPLINK ? a.m_cue : a.m_mbs);
            q = M ? (int)(((unsigned long long)(unsigned)act * M) >> 40) : act / P;
            r = act - q * P;
        } else {
            q = act / P; r = act - q * P;
            if (r < 0) { r += P; q -= 1; }
        }
        in.rb = q; in.p = r;
    } else {
        in.rb = a.rb_in[row + i]; in.p = a.pwr_in[row + i];
    }
    // hop 2: positions of the two devices, 10^(p/10) for the integer power level
    in.txx = px[in.txd]; in.txy = py[in.txd];
    in.rxx = px[rxd]; in.rxy = py[rxd];
    in.p10 = (unsigned)in.p < 128u ? a.pow10_tab[in.p] : exp10f(0.1f * (float)in.p);
    return in;
}

// SINGLE = every thread owns at most one link (N <= blockDim, the normal case up to 1024 links): the per-link loops
// collapse to a single predicated body, which removes their exec-mask bookkeeping from the scalar pipe.
#define FOR_MY_LINKS(i) for (int i = tid, go_ = 1; go_ && i < N; i += T, go_ = !SINGLE)

template <int MODE, bool SINGLE>
__global__ __launch_bounds__(1024) void step_kernel(const StepArgs a) {
    extern __shared__ __align__(16) unsigned char smem_raw[];
    const int N = a.N, R = a.R, D = a.D, W = a.mask_words;
    const int b = blockIdx.x, tid = threadIdx.x, T = blockDim.x;
    const size_t row = (size_t)b * N;
    Smem s = carve(smem_raw, N);
    // carve() aligns the mask region to 8 bytes the same way step_lds_bytes does
    s.mask = reinterpret_cast<u64*>((reinterpret_cast<uintptr_t>(s.flags + 4) + 7) & ~(uintptr_t)7);
    s.summ = reinterpret_cast<unsigned*>(s.mask + (size_t)R * W + W);

    // ---- prologue: issue this thread's first link's loads BEFORE any LDS work or barrier, so their latency (the
    // link table / action hop, then the dependent position / 10^(p/10) hop) overlaps pass 0 and the barrier.
    // Per-link constants are host-flattened arrays (lk_*[N], d2d_capi.hip refresh_tables), so there is no
    // link -> device -> column double hop in the kernel.
    const float* px = a.pos_x + (size_t)b * D;
    const float* py = a.pos_y + (size_t)b * D;
    LinkIn first;
    if (tid < N) first = load_link(a, px, py, row, tid);

    // ---- pass 0: clear masks and flags
    const bool want_masks = W > 0;
    if (want_masks)
        for (int k = tid; k < R * W + W + (R + 1) / 2; k += T) s.mask[k] = 0ull;      // masks + summary words
    if (tid < 4) s.flags[tid] = 0;
    if (tid < 32) s.red[tid] = 0.0f;
    __syncthreads();

    // ---- pass 1: decode + stage the transmitter side of every link
    float4 me0 = make_float4(0.f, 0.f, 0.f, 0.f);
    float2 rx0 = make_float2(0.f, 0.f);
    FOR_MY_LINKS(i) {
        const LinkIn in = i == tid ? first : load_link(a, px, py, row, i);
        const int type = in.type, txd = in.txd;
        const int rb = in.rb, p = in.p;
        const float4 tuple = make_float4(in.txx, in.txy, in.p10 * in.tx_lin, __int_as_float(rb));
        s.link[i] = tuple;
        s.rx[i] = make_float2(in.rxx, in.rxy);
        s.aux[i] = txd | (type << 24);
        if (i == tid) { me0 = tuple; rx0 = make_float2(in.rxx, in.rxy); }   // own link stays in registers for pass 2
        if (MODE == PL_POWER || MODE == PL_SHADOW) s.expo[i] = a.lk_exp[i];
        if (a.rb_out) { a.rb_out[row + i] = rb; a.pwr_out[row + i] = p; }
        if (want_masks) {
            const u64 bit = 1ull << (i & 63);
            if ((unsigned)rb < (unsigned)R) {
                atomicOr(&s.mask[(size_t)(i >> 6) * R + rb], bit);
                atomicOr(&s.summ[rb], 1u << (i >> 6));
            }
            else atomicOr(&s.flags[0], FLAG_RB_OOR);
            if (type == LINK_SIDELINK) atomicOr(&s.mask[(size_t)R * W + (i >> 6)], bit);
        }
    }
    __syncthreads();
    const bool use_masks = want_masks && !(s.flags[0] & FLAG_RB_OOR);
    const float* gtab = MODE == PL_TABLE ? a.gain_table + (size_t)b * a.table_env_stride : nullptr;
    const unsigned genv = (unsigned)(a.env_offset + (unsigned long long)b);   // global env index (RNG counter)

    // ---- pass 2: interference reduction + SINR/SNR/rate/capacity + obs table
    float cap_part = 0.0f;
    int my_flags = 0;
    bool violated = false;
    FOR_MY_LINKS(i) {
        const float4 me = i == tid ? me0 : s.link[i];
        const float2 rx = i == tid ? rx0 : s.rx[i];
        const int rb = __float_as_int(me.w);
        const int txd = i == tid ? first.txd : (s.aux[i] & 0xFFFFFF);
        const int rxd = a.link_rx[i];
        // per-link receiver / transmitter constants: coalesced, issued ahead of the mask walk that hides them
        const float rx_pl = a.lk_rx_pl[i], rx_lin = a.lk_rx_lin[i], noise = a.lk_noise_mw[i];
        const float sens = a.lk_sens_db[i], bw_mhz = a.lk_bw_mhz[i];
        float acc = 0.0f;
        bool zero = false;

        if (use_masks) {
            const u64* m = s.mask + rb;                                   // word w of this RB: m[w * R]
            unsigned live = s.summ[rb];                                   // non-empty words of this RB, ascending
            while (live) {
                const int w = __builtin_ctz(live);
                live &= live - 1;
                u64 word = m[(size_t)w * R];
                if (w == (i >> 6)) word &= ~(1ull << (i & 63));          // .difference({action}), simulator.py:95
                while (word) {
                    const int j = (w << 6) + __builtin_ctzll(word);
                    word &= word - 1;
                    const float4 o = s.link[j];
                    const float dx = o.x - rx.x, dy = o.y - rx.y;
                    const float d2 = fmaf(dx, dx, dy * dy);
                    float g;
                    if (MODE == PL_TABLE) g = gtab[(size_t)(s.aux[j] & 0xFFFFFF) * D + rxd];
                    else { g = pair_gain<MODE>(d2, MODE != PL_INV_SQUARE ? s.expo[j] : 2.0f); zero |= d2 == 0.0f; }
                    if (MODE == PL_SHADOW && d2 > a.shadow_d0sq) g *= shadow_factor(a, genv, j, i, 0u);
                    acc = fmaf(o.z, g, acc);                             // simulator.py:97-101, linear mW
                }
            }
        } else {
#pragma unroll 4
            for (int j = 0; j < N; ++j) {
                const float4 o = s.link[j];                              // same address in every lane: LDS broadcast
                const bool same = (__float_as_int(o.w) == rb) & (j != i);
                const float dx = o.x - rx.x, dy = o.y - rx.y;
                const float d2 = fmaf(dx, dx, dy * dy);
                float g;
                if (MODE == PL_TABLE) g = same ? gtab[(size_t)(s.aux[j] & 0xFFFFFF) * D + rxd] : 0.0f;
                else { g = pair_gain<MODE>(d2, MODE != PL_INV_SQUARE ? s.expo[j] : 2.0f); zero |= same & (d2 == 0.0f); }
                if (MODE == PL_SHADOW && same && d2 > a.shadow_d0sq) g *= shadow_factor(a, genv, j, i, 0u);
                acc = same ? fmaf(o.z, g, acc) : acc;
            }
        }

        // own link: simulator.py:93
        const float dx = me.x - rx.x, dy = me.y - rx.y;
        const float d2 = fmaf(dx, dx, dy * dy);
        float g;
        if (MODE == PL_TABLE) g = gtab[(size_t)txd * D + rxd];
        else { g = pair_gain<MODE>(d2, MODE != PL_INV_SQUARE ? s.expo[i] : 2.0f); zero |= d2 == 0.0f; }
        float sig = me.z * g * rx_pl * rx_lin;                           // mW at the receiver, with rx gains
        float sig_snr = sig;
        if (MODE == PL_SHADOW && d2 > a.shadow_d0sq) {
            sig_snr = sig * shadow_factor(a, genv, i, i, 1u);            // simulator.py:114: a second, independent draw
            sig *= shadow_factor(a, genv, i, i, 0u);                     // simulator.py:93
        }
        const float ix = acc * rx_pl;                                    // interferers: no rx gains (simulator.py:100)
        const float sinr_lin = sig / (ix + noise);
        // dB = 10 log10 x = 3.0103 log2 x, log2 on the transcendental unit (v_log_f32, 1 ulp): abs error < 6e-6 dB
        // at 80 dB and < 1e-6 dB near 0 dB, inside the 1e-5 * max(|ref|, 1) bar with an order of magnitude to spare
        const float sinr_db = 3.01029995663981195f * __builtin_amdgcn_logf(sinr_lin);         // simulator.py:106-107
        const float snr_db = 3.01029995663981195f * __builtin_amdgcn_logf(sig_snr / noise);   // simulator.py:115
        // log2(1 + x) without losing small x: log2(u) * x / (u - 1), u = fl(1 + x)
        const float u1p = 1.0f + sinr_lin, um1 = u1p - 1.0f;
        const float sh = um1 == 0.0f ? sinr_lin * 1.44269504088896340736f
                                     : __builtin_amdgcn_logf(u1p) * (sinr_lin / um1);
        const bool ok = sinr_db > sens;                                  // simulator.py:123,149
        const float rate = ok ? sh : 0.0f;
        const float cap = ok ? bw_mhz * sh : 0.0f;                       // simulator.py:150-151

        a.sinr_db[row + i] = sinr_db;
        a.snr_db[row + i] = snr_db;
        a.rate[row + i] = rate;
        a.cap[row + i] = cap;
        if (a.write_table) {                                             // obs_fn.py:57-60
            float2* t = reinterpret_cast<float2*>(a.table + (row + i) * 6);
            t[0] = make_float2(me.x, me.y);
            t[1] = rx;
            t[2] = make_float2(sinr_db, snr_db);
        }
        // staged only for the reward pass that reads them (reward_fn.py): 1 -> cap; 2 -> own sinr, sh; 3 -> sinr, sh
        if (a.reward_fn >= 2) { s.sinr[i] = sinr_db; s.sh[i] = sh; }    // staged for reward passes 2 / 3 only
        if (a.reward_fn == 1) {
            // SystemCapacityRewardFunction's -1 rule (reward_fn.py:29-41), from this link's side: I am a non-D2D
            // link whose capacity is <= min_capacity and some D2D link shares my RB.  Masks / tuples of ALL links
            // were published by the barrier before this pass, so no further synchronisation is needed here.
            const int type_i = i == tid ? first.type : (s.aux[i] >> 24);
            if (type_i != LINK_SIDELINK && cap <= a.reward_param) {
                bool hit = false;
                if (use_masks) {
                    unsigned live = s.summ[rb];
                    while (live) {
                        const int w = __builtin_ctz(live);
                        live &= live - 1;
                        hit |= (s.mask[(size_t)w * R + rb] & s.mask[(size_t)R * W + w]) != 0ull;
                    }
                } else {
                    for (int k = 0; k < N; ++k)
                        hit |= (k != i) & ((s.aux[k] >> 24) == LINK_SIDELINK) & (__float_as_int(s.link[k].w) == rb);
                }
                violated |= hit;
            }
        }
        cap_part += cap;
        if (zero) my_flags |= FLAG_ZERO_DISTANCE;
        if (!(fabsf(sinr_db) <= 3.0e38f)) my_flags |= FLAG_NON_FINITE;
    }
    if (my_flags) atomicOr(&s.flags[0], my_flags);

    // ---- pass 3: reward
    if (a.reward_fn == 1) {
        // SystemCapacityRewardFunction, reward_fn.py:27-44: mean capacity, or -1 for everyone if any link reported
        // a violation above.  One barrier: wave partial sums + the violation flag.
        const float wsum = wave_sum(cap_part);
        if ((tid & 63) == 0) s.red[tid >> 6] = wsum;
        if (violated) atomicOr(&s.flags[1], 1);
        __syncthreads();
        float total = 0.0f;
        const int nw = (T + 63) >> 6;
        const float4* red4 = reinterpret_cast<const float4*>(s.red);       // 16 slots, zero-padded: fixed-order sum
        for (int w = 0; w < (nw + 3) >> 2; ++w) { const float4 v = red4[w]; total += (v.x + v.y) + (v.z + v.w); }
        const float r = s.flags[1] ? -1.0f : total / (float)N;
        FOR_MY_LINKS(i) a.reward[row + i] = r;
    } else if (a.reward_fn == 2) {
        // ShannonRewardFunction, reward_fn.py:52-57
        FOR_MY_LINKS(i) a.reward[row + i] = s.sinr[i] >= a.reward_param ? s.sh[i] : -1.0f;
        __syncthreads();
    } else if (a.reward_fn == 3) {
        // CueSinrShannonRewardFunction, reward_fn.py:65-78
        __syncthreads();
        FOR_MY_LINKS(i) {
            const int rbi = __float_as_int(s.link[i].w);
            bool bad = false;
            if (use_masks) {
                unsigned live = s.summ[rbi];
                while (live) {
                    const int w = __builtin_ctz(live);
                    live &= live - 1;
                    u64 word = s.mask[(size_t)w * R + rbi] & ~s.mask[(size_t)R * W + w];   // non-sidelink members
                    if (w == (i >> 6)) word &= ~(1ull << (i & 63));
                    while (word) {
                        const int j = (w << 6) + __builtin_ctzll(word);
                        word &= word - 1;
                        bad |= s.sinr[j] < a.reward_param;
                    }
                }
            } else {
                for (int j = 0; j < N; ++j)
                    bad |= (j != i) & ((s.aux[j] >> 24) != LINK_SIDELINK) & (__float_as_int(s.link[j].w) == rbi) &
                           (s.sinr[j] < a.reward_param);
            }
            a.reward[row + i] = bad ? -1.0f : s.sh[i];
        }
        __syncthreads();
    } else {
        __syncthreads();
    }

    if (tid == 0) a.env_flags[b] = s.flags[0];
}

// OR-reduction of the per-env flag words, run only when the host asks (d2d_status_flags): keeps the memset +
// atomic out of the per-step stream.
__global__ __launch_bounds__(256) void flags_or_kernel(const int* env_flags, int B, unsigned* status) {
    unsigned f = 0;
    for (int k = blockIdx.x * 256 + threadIdx.x; k < B; k += gridDim.x * 256) f |= (unsigned)env_flags[k];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) f |= __shfl_xor(f, o);
    if ((threadIdx.x & 63) == 0 && f) atomicOr(status, f);
}

hipError_t launch_flags_or(const int* env_flags, int B, unsigned* status, hipStream_t stream) {
    hipError_t err = hipMemsetAsync(status, 0, 4, stream);
    if (err != hipSuccess) return err;
    int blocks = (B + 255) / 256;
    if (blocks > 64) blocks = 64;
    hipLaunchKernelGGL(flags_or_kernel, dim3(blocks), dim3(256), 0, stream, env_flags, B, status);
    return hipGetLastError();
}

hipError_t launch_step(const StepArgs& a, PlMode mode, hipStream_t stream) {
    int threads = a.threads > 0 ? a.threads : ((a.N + 63) / 64) * 64;
    if (threads > 1024) threads = 1024;
    if (threads < 64) threads = 64;
    const size_t lds = step_lds_bytes(a.N, a.R, a.mask_words);
    dim3 grid(a.B), block(threads);
    hipError_t err = hipSuccess;
    const bool single = a.N <= threads;
#define D2D_LAUNCH_1(M, S)                                                                               \
    do {                                                                                                 \
        if (lds > 48 * 1024)                                                                             \
            err = hipFuncSetAttribute(reinterpret_cast<const void*>(&step_kernel<M, S>),                 \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);             \
        if (err == hipSuccess) {                                                                         \
            hipLaunchKernelGGL((step_kernel<M, S>), grid, block, lds, stream, a);                        \
            err = hipGetLastError();                                                                     \
        }                                                                                                \
    } while (0)
#define D2D_LAUNCH(M)                                                                                    \
    do {                                                                                                 \
        if (single) D2D_LAUNCH_1(M, true); else D2D_LAUNCH_1(M, false);                                  \
    } while (0)
    switch (mode) {
        case PL_INV_SQUARE: D2D_LAUNCH(PL_INV_SQUARE); break;
        case PL_POWER: D2D_LAUNCH(PL_POWER); break;
        case PL_TABLE: D2D_LAUNCH(PL_TABLE); break;
        case PL_SHADOW: D2D_LAUNCH(PL_SHADOW); break;
    }
#undef D2D_LAUNCH
#undef D2D_LAUNCH_1
    return err;
}

}  // namespace d2d
