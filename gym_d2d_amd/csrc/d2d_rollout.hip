// Rollout kernel for gfx950 (MI355X): the step of a learner's rollout - raw agent actions for every link (or all but a prefix with
// fixed actions: traffic-model CUEs), any of the three rewards, one env per workgroup of N / LPT threads, sparse RB occupancy
// (N <= 4 R) - with a STRAIGHT-LINE hot path.
//
// Reference path (file:line under /root/reference/src/gym_d2d): the same as csrc/d2d_step.hip -
//   D2DEnv._decode_action envs/d2d_env.py:93-101, Actions.get_actions_by_rb actions.py:27-31,
//   Simulator._calculate_sinrs / _snrs / _rates / _network_capacity simulator.py:89-154,
//   SystemCapacityRewardFunction envs/reward_fn.py:27-44, ShannonRewardFunction :47-57, CueSinrShannonRewardFunction :60-78,
//   LinearObsFunction's base table envs/obs_fn.py:55-61.
//
// Why a kernel of its own (round 5): at 36 - 64 bytes per link the step is paced by the instructions its waves issue, of every
// kind (a wave gets one in every 7.5 cycles or so: profiles/r5_issue_rates.txt), not by HBM.  The generic
// kernel's member-list walk spent 70 of its 246 VALU per wave on taking the own entry out of the RB's list, sorting the rest
// and clamping the indices, and ~40 of its 102 SALU on exec-mask bookkeeping around per-lane rarities.  This kernel
//   * keeps every per-lane rarity behind wave-uniform ballot branches that the common wave never enters: an action beyond the
//     multiply-high decode's bound (refresh_tables keeps it below R * P, so inside it the quotient is exact AND a valid RB) is
//     decoded by division; a link whose RB lies outside [0, R) enters no list and sweeps all pairs; the ninth and later links
//     of an RB go to a per-env OVERFLOW POOL of (rb, link) pairs that only the members of such an RB ever scan - no
//     membership masks, no workgroup-wide fallback, 17 KB of LDS per env instead of 37 KB;
//   * stores the list entries as LDS BYTE OFFSETS of the members' tuples (u16, link * 16; empty = N * 16, a stand-in tuple
//     of power -0.0), so an entry is an address: no clamp, no shift;
//   * does NOT sort and does NOT take the own entry out: the interference sum starts at minus the own term and adds all
//     eight slots in arrival order.  Every term is a float (24 bits) added in double (53 bits): while the largest and the
//     smallest non-zero term of a lane are less than 2^25 apart, every partial sum of the at most nine values is exact,
//     hence independent of the order and equal - bit for bit - to the ascending-order sum of the mask walk and the
//     all-pairs sweep.  The two extremes are tracked with one v_max3_i32 / v_min3_u32 pair per two terms (the stand-in's
//     term, -0.0, is neutral for both); a lane outside the window (5e-5 of lanes at BASELINE config 3 geometry) re-does its
//     sum in sorted order;
//   * LPT = 2: a thread carries two ADJACENT links (2t and 2t + 1), so every per-WAVE instruction - the scalar record load, barriers,
//     ballots, the wave reduction, the ticket - is paid once per 128 links instead of once per 64, half as many waves are launched,
//     the thread's two actions are one 8-byte load and its two results one 8-byte element of every plane.  The capacity sum
//     keeps the bits of the one-link kernels: their wave sum is a balanced tree over adjacent links, whose first level here is the
//     lane's own pair (wave_sum_halves);
//   * with two links per thread the TABLE ROWS (24 bytes per link) leave through LDS, so that every store instruction writes
//     1024 contiguous bytes (see the results section: 26.7 -> 24.5 us in the table mode);
//   * a link count that is no multiple of 64 (OPT_PAD) is padded to the next one with threads that SHADOW the last link (see the
//     top of the kernel), and its reward is reduced as the generic kernels reduce it for such shapes.
// Same arithmetic as step_kernel everywhere else (tests/test_gpu_step_variants.py holds the two bit-identical).
//
// Worst case: an env whose actions pile more than eight links on one RB costs its members a scan of the pool (<= N entries);
// a lane whose terms then fall outside the exactness window (25 bits at 9 - 15 members ... 18 at 2047) takes them in ascending
// order by selection (up to 32 members) or by one all-pairs sweep - bounded by N pair evaluations per link, what the
// reference's own loop does (simulator.py:95-101).  (A first version swept at an 18-bit window whatever the member count: a
// few hundred lanes per launch at BASELINE config 3, each outliving the launch - 47 us median against a 16 us minimum.)  A
// workload that lives there (N > 4 R on average) is not given this kernel (run_step), and D2D_TUNE_STEP_WALK = 0 keeps the mask
// walk for any other.
#include <cstring>

#include "d2d_step_device.h"

namespace d2d {

#define RO_SLOTS 8
// ABLATION builds (D2D_BUILD_DEFINES="-DD2D_EXP_ABL=<bits>" python -m gym_d2d_amd.build; never shipped: results are WRONG, costs right):
// 1 no input loads, 2 no result-plane stores, 4 no pair evaluation, 8 no reward epilogue, 16 no SINR arithmetic, 32 no action prefetch,
// 64 no second barrier.  Round 6 (profiles/r6_rollout_kernel_ablation.jsonl): skeleton 9.2 us + pairs 4.9 + loads 3.8 + stores 2.2.
#ifndef D2D_EXP_ABL
#define D2D_EXP_ABL 0
#endif
// diagnostic builds: lane 0 of every wave stamps the shader clock at the phase boundaries (tools/phase_times.py)
#if defined(D2D_STEP_ABLATE) && D2D_STEP_ABLATE
#define RO_STAMP(k) do { if (a.dbg && (threadIdx.x & 63) == 0)                                                            \
        a.dbg[((size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * 16 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define RO_STAMP(k) do { } while (0)
#endif
#define RO_ST(ptr, val) do { if (NT) __builtin_nontemporal_store((val), (ptr)); else *(ptr) = (val); } while (0)

// LDS of one env (byte offsets; StepLds): 0 ... 59 unused, 60 dump (u16) | 64 flags[4]: env flags, reward bits, ticket, pool count
// | 80 link[N + 1] tuples | expo[N + 1] (power law) | lo[N + 1] (exact positions) | slots[R + 1] (8 x u16) | cnt[R + 1] | pool[N] (rb, link) | 16 wave sums
// (the reward's group sums) | low[N + 1] (CueSinrShannon).  With two links per thread the region from 80 on becomes the env's table image.
void rollout_lds_layout(int N, int R, int mode, int reward_fn, int xpos, StepLds* out) {
    std::memset(out, 0, sizeof(*out));
    unsigned off = LDS_HEAD_BYTES + ((unsigned)N + 1u) * 16u;
    out->expo = off;
    if (mode == PL_POWER) off += ((unsigned)N + 1u) * 8u;
    else if (mode == PL_POWK) off += (((unsigned)N + 1u) * 4u + 7u) & ~7u;       // PL_POWK: the links' RBs (the tuple's fourth word holds phi)
    out->lo = off; if (xpos) off += ((unsigned)N + 1u) * 8u;      // exact positions: low parts of (tx_x, tx_y), + one for the stand-in
    off = (off + 15u) & ~15u;
    out->lists = off; off += ((unsigned)R + 1u) * 16u + (((unsigned)R + 1u + 3u) & ~3u) * 4u;
    out->pool = off; off += (unsigned)N * 8u;
    off = (off + 15u) & ~15u;
    out->aux = off; off += 64u;                                   // padded link counts: the waves' capacity sums (16 floats)
    out->rx = off;                                                // CueSinrShannon: per link "a non-D2D link below the threshold" (+ one for the stand-in)
    if (reward_fn == 3) off += ((unsigned)N + 1u) * 4u;
    out->env_bytes = (off + 15u) & ~15u;
}

typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float2 lds_f2(unsigned addr) { const f32x2 v = lds_get<f32x2>(addr); return make_float2(v.x, v.y); }
__device__ __forceinline__ unsigned lds_atomic_inc(unsigned addr) {
    return __hip_atomic_fetch_add((D2D_LDS(unsigned)*)(addr), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);      // ds_add_rtn_u32
}
__device__ __forceinline__ void lds_atomic_or(unsigned addr, int bits) {
    __hip_atomic_fetch_or((D2D_LDS(int)*)(addr), bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

template <int MODE, int OPT, int LPT>
__global__ __launch_bounds__(1024) void rollout_kernel(const StepArgs a) {
    constexpr bool SREC = (OPT & OPT_SREC) != 0, NT = (OPT & OPT_NT) != 0, PAD = (OPT & OPT_PAD) != 0;
    // XPOS: float64 positions uploaded as (hi, lo) float pairs (d2d_set_positions_f64): one more 16-byte row per link, one more
    // ds_read_b64 per pair, every difference by coord_diff.  One link per thread (the two-link kernel sits at its 64-VGPR limit).
    constexpr bool XPOS = (OPT & OPT_XPOS) != 0;
    static_assert(!PAD || (LPT == 1 && !SREC), "a link count that is no multiple of 64: one link per thread, per-lane records");
    static_assert(!XPOS || LPT == 1, "exact positions: one link per thread");
    constexpr bool POWLAW = MODE == PL_POWER;
    // PL_POWK (one integer k within 1/2 of every transmitter's exponent, pow_k_gains): the tuple's fourth word is the transmitter's
    // phi instead of its RB - no second LDS read per pair - and the RBs, which only the rare sweeps look at, sit in an array of their own
    constexpr bool POWK = MODE == PL_POWK;
    constexpr bool NF_ONLY = MODE != PL_POWER;                   // a zero distance shows as a non-finite gain: no smallest-distance tracking
    static_assert(MODE == PL_INV_SQUARE || MODE == PL_POWER || MODE == PL_POWK, "the rollout kernel serves the power laws");
    static_assert(LPT == 1 || LPT == 2, "one or two links per thread");
    const int N = a.N, R = a.R, TPE = a.tpe;                     // N == LPT * TPE == LPT * blockDim.x; thread t: links LPT * t + u
    const int tid = threadIdx.x, b = (int)blockIdx.x;
    // OPT_PAD: N is no multiple of 64 and TPE the next one.  A thread beyond the last link SHADOWS link N - 1 - same loads, same
    // arithmetic, the same values stored to the same addresses - except that it enters no list and adds no capacity: two guards
    // instead of a predicate on every access.
    const auto link_of = [&](int u) { const int i = LPT * tid + u; return PAD ? min(i, N - 1) : i; };
    const bool shadow = PAD && tid >= N;
    const unsigned row = (unsigned)b * (unsigned)N;              // element offsets fit 32 bits (run_step refuses B * N * 24 >= 2^32)
    const int n_fixed = SREC ? 0 : a.n_fixed;                    // scalar records: the host offers them without fixed links only
    const unsigned act_row = SREC ? row : (unsigned)b * (unsigned)a.act_stride;
    const bool cfg_export_actions = a.rb_out != nullptr;
    const bool capacity_reward = a.reward_fn == 1;               // SystemCapacity (env-wide mean) | 2: Shannon (per link) | 3: CueSinrShannon (per
    const bool shannon_reward = a.reward_fn == 2;                // link, looks at the other members of the RB: one link per thread only)
    const bool cue_sinr_reward = LPT == 1 && a.reward_fn == 3;
    const unsigned EMPTY = (unsigned)N * 16u;                    // byte offset of the stand-in tuple link[N]
    const unsigned L_LINK = LDS_HEAD_BYTES, L_EXPO = a.lds.expo, L_RB = a.lds.expo, L_LO = a.lds.lo, L_SLOTS = a.lds.lists, L_CNT = a.lds.lists + ((unsigned)R + 1u) * 16u;
    const unsigned L_POOL = a.lds.pool, L_FLAGS = 64u, L_DUMP = 60u, L_RED = a.lds.aux, L_LOW = a.lds.rx;

    RO_STAMP(0);
    // ---- prologue: the links' loads, issued before any LDS work or barrier (their latency overlaps pass 0)
    LinkRaw in[LPT];
    if (SREC && LPT == 2) {                                      // the thread's two actions: adjacent columns, one 8-byte load
        if (D2D_EXP_ABL & 1) { in[0].act0 = (tid * 7919 + b * 13) % 5000; in[LPT - 1].act0 = (tid * 104729 + b * 17) % 5000; } else {
        const i32x2 aa = *reinterpret_cast<const i32x2*>(at(a.actions, fresh((row + 2u * (unsigned)tid) * 4u)));
        in[0].act0 = aa.x; in[LPT - 1].act0 = aa.y; }
    }
#pragma unroll
    for (int u = 0; u < LPT; ++u) {
        const int i = link_of(u);
        if (SREC) {
            if (LPT == 1) in[u].act0 = *at(a.actions, fresh((row + (unsigned)i) * 4u));
            in[u].act1 = 0;
            if (D2D_EXP_ABL & 1) in[u].pos = make_float4((float)((i * 37 + b) % 997), (float)((i * 91 + b) % 983), (float)((i * 53 + b * 3) % 991) + 0.5f, (float)((i * 71 + b) % 977) + 0.5f);
            else in[u].pos = *at(a.lpos, fresh((row + (unsigned)i) * 16u));
            in[u].plo = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            if (XPOS) in[u].plo = *at(a.lpos_lo, fresh((row + (unsigned)i) * 16u));
        } else {
            // (per-lane records: a PREFIX of the links may carry fixed actions - traffic-model CUEs, traffic_model.py:15-32 - and the
            // action array then has a column per remaining link only)
            in[u] = load_link(a, row, act_row, i, 0, 0, false, false, POWLAW || POWK, XPOS);
        }
    }

    // ---- pass 0: slots[R + 1] <- EMPTY.., cnt[R + 1] <- 0 (one contiguous region of 16-byte units), flags, the stand-in tuple
    {
        const int rows = R + 1, nl = rows + ((rows + 3) >> 2);
        const unsigned e2 = EMPTY | (EMPTY << 16);
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int k = tid + u * TPE;
            if (k < nl) { const unsigned f = k < rows ? e2 : 0u; lds_put<u32x4>(L_SLOTS + (unsigned)k * 16u, u32x4{f, f, f, f}); }
        }
        if (UNLIKELY(nl > 2 * TPE)) {                            // more RBs than 1.6 threads: further rounds (workgroup-uniform)
            COLD_LOOP
            for (int k = tid + 2 * TPE; k < nl; k += TPE) { const unsigned f = k < rows ? e2 : 0u; lds_put<u32x4>(L_SLOTS + (unsigned)k * 16u, u32x4{f, f, f, f}); }
        }
        if (tid == TPE - 1) {
            lds_put<f32x4>(L_LINK + EMPTY, f32x4{1.0e18f, 1.0e18f, -0.0f, POWK ? 0.0f : __int_as_float(-1)});
            if (POWLAW) lds_put<f32x2>(L_EXPO + (EMPTY >> 1), f32x2{-1.0f, 0.0f});
            if (POWK) lds_put<int>(L_RB + (EMPTY >> 2), -1);
            if (XPOS) lds_put<f32x2>(L_LO + (EMPTY >> 1), f32x2{0.0f, 0.0f});
        }
        if (tid < 5) lds_put<u32x4>((unsigned)tid * 16u, u32x4{0u, 0u, 0u, 0u});          // sum, dump, flags[4]: 80 bytes
        if (tid >= 8 && tid < 12) lds_put<u32x4>(L_RED + (unsigned)(tid - 8) * 16u, u32x4{0u, 0u, 0u, 0u});   // the 16 group sums of the reward
    }
    // nothing that consumes a loaded value may be scheduled above this barrier (the wave would sit on the HBM round trip
    // before pass 0 instead of behind it)
    __builtin_amdgcn_sched_barrier(0);
    RO_STAMP(1);
    __syncthreads();
    RO_STAMP(2);
    __builtin_amdgcn_sched_barrier(0);
    // The records of a wave's 64 * LPT links are identical (StepArgs::rec_uniform; the host offers two links per thread only
    // where that holds for aligned groups of 128): the whole record in ONE 64-byte scalar load per wave.  rec_grp holds a 64-byte
    // row per group of 64 links, so the wave's first link index is the byte offset of its row.  Issued HERE, behind the barrier:
    // a scalar load shares its counter with the LDS writes of pass 0, so in front of the barrier the wave would sit out this
    // second round trip (kernel arguments -> table pointer -> record) before it may even arrive; here it hides under the links'
    // loads, which are still in flight.
    if (SREC) {
        const i32x16 g = scalar_load64(reinterpret_cast<const unsigned char*>(a.rec_grp) + (unsigned)(__builtin_amdgcn_readfirstlane(tid) * LPT));
#pragma unroll
        for (int u = 0; u < LPT; ++u) {
            in[u].ra = make_int4(g[2], 0, g[0], g[1]);
            in[u].rb_ = make_float4(__int_as_float(g[4]), __int_as_float(g[5]), __int_as_float(g[6]), __int_as_float(g[7]));
            in[u].rc = make_float4(__int_as_float(g[8]), __int_as_float(g[9]), __int_as_float(g[10]), __int_as_float(g[11]));
            in[u].hh = make_float2(__int_as_float(g[12]), __int_as_float(g[13]));
        }
    }

    // ---- pass 1: decode, stage the transmitter tuple, enter the RB's list
    int rb[LPT], pwr[LPT];
    float pz[LPT];                                               // effective tx power (mW), the tuple's third component
    bool oor[LPT];
    unsigned slot[LPT], row_off[LPT];
#pragma unroll
    for (int u = 0; u < LPT; ++u) {
        const int i = link_of(u);
        const unsigned my_off = (unsigned)i << 4;
        const unsigned P = __float_as_uint(in[u].rc.w) & 0xFFFFu;
        // the host keeps the bound below R * P (refresh_tables): an action inside it decodes to rb < R by one multiply-high
        // (a link with a fixed action takes the rare arm too: its record holds (rb, pwr) where the others hold the magic and the bound)
        const bool fixed = !SREC && (in[u].ra.x & D2D_REC_FIXED_BIT) != 0;
        const bool bad = fixed | ((unsigned)in[u].act0 > (unsigned)in[u].ra.w);
        rb[u] = (int)__umulhi((unsigned)in[u].act0, (unsigned)in[u].ra.z);
        int pw = in[u].act0 - (int)__umul24((unsigned)rb[u], P);
        oor[u] = false;
        unsigned rbc = (unsigned)rb[u];                          // the list this link enters
        if (UNLIKELY(__builtin_amdgcn_ballot_w64(bad) != 0ull)) {
            if (bad) {                                           // negative, beyond R * P, or beyond the multiply-high range: divide
                decode_link(a, in[u], act_row, rb[u], pw, 0, SREC);
                oor[u] = (unsigned)rb[u] >= (unsigned)R;         // accepted like the reference does (d2d_env.py:94-96)
                rbc = oor[u] ? (unsigned)R : (unsigned)rb[u];    // row R: where links that enter no list are parked
                if (oor[u]) lds_atomic_or(L_FLAGS, FLAG_RB_OOR);
            }
        }
        pz[u] = pow10_tenth(pw) * in[u].rb_.x;
        // the tuple as two 8-byte halves (one ds_write2_b64): (x, y) goes out of the position row's own registers instead of being
        // copied into a fresh four-register tuple first
        lds_put<f32x2>(L_LINK + my_off, f32x2{in[u].pos.x, in[u].pos.y});
        lds_put<f32x2>(L_LINK + my_off + 8u, f32x2{pz[u], POWK ? in[u].hh.x : __int_as_float(rb[u])});
        if (POWLAW) lds_put<f32x2>(L_EXPO + (my_off >> 1), f32x2{in[u].hh.x, in[u].hh.y});
        if (POWK) lds_put<int>(L_RB + (my_off >> 2), rb[u]);
        if (XPOS) lds_put<f32x2>(L_LO + (my_off >> 1), f32x2{in[u].plo.x, in[u].plo.y});
        pwr[u] = pw;
        row_off[u] = L_SLOTS + rbc * 16u;
        slot[u] = shadow ? 0u : lds_atomic_inc(L_CNT + rbc * 4u);
    }
    // (the slot numbers of the thread's links come back together: one LDS round trip, not one per link)
#pragma unroll
    for (int u = 0; u < LPT; ++u) {
        const int i = link_of(u);
        const bool ovf = slot[u] >= (unsigned)RO_SLOTS;
        if (!shadow) lds_put<unsigned short>(ovf ? L_DUMP : row_off[u] + slot[u] * 2u, (unsigned short)((unsigned)i << 4));
        if (UNLIKELY(__builtin_amdgcn_ballot_w64(ovf) != 0ull)) {
            if (ovf) {                                           // the ninth and later links of an RB: the env's overflow pool
                const unsigned ps = lds_atomic_inc(L_FLAGS + 12u);
                lds_put<u32x2>(L_POOL + ps * 8u, u32x2{(unsigned)rb[u], (unsigned)i});
            }
        }
    }
    if (cfg_export_actions) {                                    // the decoded planes: the thread's LPT adjacent elements in one store
        const unsigned oe = fresh((row + (unsigned)link_of(0)) * 4u);
        if (LPT == 2) {
            RO_ST(reinterpret_cast<i32x2*>(at(a.rb_out, oe)), (i32x2{rb[0], rb[LPT - 1]}));
            RO_ST(reinterpret_cast<i32x2*>(at(a.pwr_out, oe)), (i32x2{pwr[0], pwr[LPT - 1]}));
        } else { RO_ST(at(a.rb_out, oe), rb[0]); RO_ST(at(a.pwr_out, oe), pwr[0]); }
    }
    RO_STAMP(3);
    if (!(D2D_EXP_ABL & 64)) __syncthreads();
    RO_STAMP(4);

    // software prefetch of the action rows of the env the workgroup `prefetch_envs` later will own (see step_kernel).  (The position
    // rows too - one dword per thread touches every line - measured: 20.9 -> 21.5 us with one link per thread, and a 65th VGPR with two)
    int pf = 0;
    {
        const int bq = b + a.prefetch_envs;
        const int bp = bq < a.B ? bq : a.B - 1;
        const unsigned op = fresh(((unsigned)bp * (SREC ? (unsigned)N : (unsigned)a.act_stride) + (unsigned)max(link_of(0) - n_fixed, 0)) * 4u);
        if (D2D_EXP_ABL & 32) pf = 0;
        else if (LPT == 2) { const i32x2 aa = *reinterpret_cast<const i32x2*>(at(a.actions, op)); pf = aa.x ^ aa.y; }
        else pf = *at(a.actions, op);
    }

    float caps[LPT], rates[LPT], sinrs[LPT], snrs[LPT];
    float sh_kept = 0.0f;                                        // CueSinrShannon's value (one link per thread)
#pragma unroll
    for (int u = 0; u < LPT; ++u) {
        const int i = link_of(u);
        const unsigned my_off = (unsigned)i << 4;
        const int type = (in[u].ra.x >> D2D_REC_TYPE_SHIFT) & D2D_REC_TYPE_MASK;
        const float2 rx = make_float2(in[u].pos.z, in[u].pos.w);
        const float2 rxlo = make_float2(in[u].plo.z, in[u].plo.w);   // zero unless XPOS
        const float rx_pl = in[u].rb_.y, rx_lin = in[u].rb_.z, noise = in[u].rb_.w;
        const float sens = in[u].rc.x, bw_mhz = in[u].rc.y;
        int dmin = 0x7F000000;                                   // power law: bits of the smallest squared distance met

        // one (transmitter tuple at LDS offset e) -> (this receiver) term: simulator.py:97-101, linear mW
        const auto term = [&](unsigned e, int& dmin_) {
            const f32x4 o = lds_get<f32x4>(L_LINK + e);
            float dx = o.x - rx.x, dy = o.y - rx.y;
            if (XPOS) { const float2 l = lds_f2(L_LO + (e >> 1)); dx = coord_diff(o.x, rx.x, l.x, rxlo.x); dy = coord_diff(o.y, rx.y, l.y, rxlo.y); }
            const float d2 = fmaf(dx, dx, dy * dy);
            const float g = pair_gain<MODE>(d2, POWLAW ? lds_f2(L_EXPO + (e >> 1)) : make_float2(POWK ? o.w : -1.0f, 0.0f), a.pow_k);
            if (POWLAW) dmin_ = min(dmin_, __float_as_int(d2));
            return o.z * g;
        };
        // the RB of link j as the sweeps see it
        const auto rb_of = [&](unsigned j, const f32x4& o) { return POWK ? lds_get<int>(L_RB + (j << 2)) : __float_as_int(o.w); };
        // the masked all-pairs sweep in ascending link order: the reference's own loop (simulator.py:95-101), the order every
        // other search variant reproduces; what a lane falls back to when nothing cheaper is exact
        const auto sweep = [&](int& dmin_) {
            double s = 0.0;
            COLD_LOOP
            for (int j = 0; j < N; ++j) {
                const f32x4 o = lds_get<f32x4>(L_LINK + ((unsigned)j << 4));
                const bool same = (rb_of((unsigned)j, o) == rb[u]) & (j != i);
                float dx = o.x - rx.x, dy = o.y - rx.y;
                if (XPOS) { const float2 l = lds_f2(L_LO + ((unsigned)j << 3)); dx = coord_diff(o.x, rx.x, l.x, rxlo.x); dy = coord_diff(o.y, rx.y, l.y, rxlo.y); }
                const float d2 = fmaf(dx, dx, dy * dy);
                const float g = pair_gain<MODE>(d2, POWLAW ? lds_f2(L_EXPO + ((unsigned)j << 3)) : make_float2(POWK ? o.w : -1.0f, 0.0f), a.pow_k);
                if (POWLAW) dmin_ = same ? min(dmin_, __float_as_int(d2)) : dmin_;
                s += same ? (double)(o.z * g) : 0.0;
            }
            return s;
        };

        // own link: simulator.py:93 (first: its term opens the interference sum below)
        float d2_own, g_own;
        {
            const float dx = XPOS ? coord_diff(in[u].pos.x, rx.x, in[u].plo.x, rxlo.x) : in[u].pos.x - rx.x;
            const float dy = XPOS ? coord_diff(in[u].pos.y, rx.y, in[u].plo.y, rxlo.y) : in[u].pos.y - rx.y;
            d2_own = fmaf(dx, dx, dy * dy);
            if (POWK) {
                const float d[1] = {d2_own}, f[1] = {in[u].hh.x};
                float g[1];
                if (LIKELY(a.pow_k == 4)) pow_k_gains<1, 4>(d, f, 4, g); else pow_k_gains<1>(d, f, a.pow_k, g);
                g_own = g[0];
            } else g_own = pair_gain<MODE>(d2_own, in[u].hh, a.pow_k);
        }

        // ---- pass 2: the RB's eight slots in one read, every slot a tuple address
        const unsigned rbc = oor[u] ? (unsigned)R : (unsigned)rb[u];
        // (one ds_read_b128 + eight VALU to unpack it.  Eight ds_read_u16 off one address register instead - no VALU at all - were
        // measured 9 % SLOWER, 19.65 -> 21.5 us: an LDS instruction costs this kernel about five VALU instructions,
        // profiles/r5_ab_rollout_kernel.jsonl)
        const u32x4 mlist = lds_get<u32x4>(L_SLOTS + rbc * 16u);
        const unsigned off[RO_SLOTS] = {mlist.x & 0xFFFFu, mlist.x >> 16, mlist.y & 0xFFFFu, mlist.y >> 16,
                                        mlist.z & 0xFFFFu, mlist.z >> 16, mlist.w & 0xFFFFu, mlist.w >> 16};
        // .difference({action}) (simulator.py:95) by arithmetic: the own entry is among the slots, so the sum opens at minus
        // its term (pz * g_own - the very product the slot's evaluation repeats, same operands, same rounding)
        const f32x2 rxv = {rx.x, rx.y}, rxlov = {rxlo.x, rxlo.y};
        double acc = -(double)(pz[u] * g_own);
        // The empty slots' stand-in has power -0.0: its term, -0.0, leaves the sum alone and its bits, 0x80000000, are neutral
        // for BOTH extremes - below every real term as a signed integer (the maximum), above every one as an unsigned (the
        // minimum) - so no instruction is spent on telling empty slots from members.  (A real term of +0.0 - zero power, or an
        // underflow - makes the minimum 0 and sends the lane to the sorted sum: conservative.)
        int tmax = 0;
        unsigned tmin = 0xFFFFFFFFu;
        unsigned members = 0u;                                   // links on my RB, read only when its row is full
#define RO_PAIR(k, o)                                                                                                   \
        {                                                                                                               \
            f32x2 dd = f32x2{(o).x, (o).y} - rxv;                     /* one v_pk_add_f32: (x, y) sit in adjacent registers */ \
            if (XPOS) dd = dd + (lds_get<f32x2>(L_LO + (off[k] >> 1)) - rxlov);   /* coord_diff, both coordinates at once */ \
            const float d2 = fmaf(dd.x, dd.x, dd.y * dd.y);                                                             \
            const float g = pair_gain<MODE>(d2, POWLAW ? lds_f2(L_EXPO + (off[k] >> 1)) : make_float2(-1.0f, 0.0f));    \
            if (POWLAW) dmin = min(dmin, __float_as_int(d2));                                                           \
            const float t = (o).z * g;                               /* simulator.py:97-101, linear mW */               \
            acc += (double)t;                                                                                           \
            tmax = max(tmax, __float_as_int(t)); tmin = min(tmin, __float_as_uint(t));                                  \
        }
        if (POWK) {
            // three pairs at once, twice: squared distances, then pow_k_gains - the reciprocals, logarithms and exponentials of a group in
            // flight together, the k-dependent products behind ONE uniform branch per group instead of one per pair (six at once: 68 VGPRs)
#define RO_POWK_GROUP(KC, NP, FIRST, ...)                                                                               \
            {                                                                                                           \
                const f32x4* const os[NP] = {__VA_ARGS__};                                                              \
                float d2v[NP], phv[NP], gv[NP];                                                                         \
                _Pragma("unroll") for (int q = 0; q < NP; ++q) {                                                        \
                    f32x2 dd = f32x2{os[q]->x, os[q]->y} - rxv;                                                         \
                    if (XPOS) dd = dd + (lds_get<f32x2>(L_LO + (off[FIRST + q] >> 1)) - rxlov);                         \
                    d2v[q] = fmaf(dd.x, dd.x, dd.y * dd.y);                                                             \
                    phv[q] = os[q]->w;                                                                                  \
                }                                                                                                       \
                pow_k_gains<NP, KC>(d2v, phv, a.pow_k, gv);                                                             \
                _Pragma("unroll") for (int q = 0; q < NP; ++q) {                                                        \
                    const float t = os[q]->z * gv[q];                                                                   \
                    acc += (double)t;                                                                                   \
                    tmax = max(tmax, __float_as_int(t)); tmin = min(tmin, __float_as_uint(t));                          \
                }                                                                                                       \
            }
#define RO_POWK_ALL(KC)                                                                                                 \
            {                                                                                                           \
                {                                                                                                       \
                    const f32x4 o0 = lds_get<f32x4>(L_LINK + off[0]), o1 = lds_get<f32x4>(L_LINK + off[1]), o2 = lds_get<f32x4>(L_LINK + off[2]); \
                    RO_POWK_GROUP(KC, 3, 0, &o0, &o1, &o2)                                                              \
                }                                                                                                       \
                __builtin_amdgcn_sched_barrier(0);                                                                      \
                {                                                                                                       \
                    const f32x4 o3 = lds_get<f32x4>(L_LINK + off[3]), o4 = lds_get<f32x4>(L_LINK + off[4]), o5 = lds_get<f32x4>(L_LINK + off[5]); \
                    RO_POWK_GROUP(KC, 3, 3, &o3, &o4, &o5)                                                              \
                }                                                                                                       \
                if (__builtin_amdgcn_ballot_w64(off[6] != EMPTY) != 0ull) {                                             \
                    const f32x4 o6 = lds_get<f32x4>(L_LINK + off[6]), o7 = lds_get<f32x4>(L_LINK + off[7]);             \
                    if (off[7] != EMPTY) members = lds_get<unsigned>(L_CNT + rbc * 4u);                                 \
                    RO_POWK_GROUP(KC, 2, 6, &o6, &o7)                                                                   \
                }                                                                                                       \
            }
            // ONE branch on k for the link: the common case (COST-Hata, ple 3.5 - 4.5) runs a copy without the tests on k
            if (LIKELY(a.pow_k == 4)) RO_POWK_ALL(4) else RO_POWK_ALL(0)
        } else if (POWLAW) {
            // (the power law's pairs carry an exponent pair each and a longer evaluation: three and three in flight - six take the
            // two-link kernel to 73 VGPRs = 6 waves per SIMD, and measured slower with one link too: COST-Hata obs-less 33.9 -> 32.0 us)
            {
                const f32x4 o0 = lds_get<f32x4>(L_LINK + off[0]), o1 = lds_get<f32x4>(L_LINK + off[1]), o2 = lds_get<f32x4>(L_LINK + off[2]);
                RO_PAIR(0, o0) RO_PAIR(1, o1) RO_PAIR(2, o2)
                asm volatile("" ::"v"(o0.w), "v"(o1.w), "v"(o2.w));
            }
            __builtin_amdgcn_sched_barrier(0);
            {
                const f32x4 o3 = lds_get<f32x4>(L_LINK + off[3]), o4 = lds_get<f32x4>(L_LINK + off[4]), o5 = lds_get<f32x4>(L_LINK + off[5]);
                RO_PAIR(3, o3) RO_PAIR(4, o4) RO_PAIR(5, o5)
                asm volatile("" ::"v"(o3.w), "v"(o4.w), "v"(o5.w));
            }
        } else if ((D2D_EXP_ABL & 4) && off[0] != 0x12345u) {
            acc = 1e-12;
        } else {
            const f32x4 o0 = lds_get<f32x4>(L_LINK + off[0]), o1 = lds_get<f32x4>(L_LINK + off[1]), o2 = lds_get<f32x4>(L_LINK + off[2]),
                        o3 = lds_get<f32x4>(L_LINK + off[3]), o4 = lds_get<f32x4>(L_LINK + off[4]), o5 = lds_get<f32x4>(L_LINK + off[5]);
            RO_PAIR(0, o0) RO_PAIR(1, o1) RO_PAIR(2, o2) RO_PAIR(3, o3) RO_PAIR(4, o4) RO_PAIR(5, o5)
            asm volatile("" ::"v"(o0.w), "v"(o1.w), "v"(o2.w), "v"(o3.w), "v"(o4.w), "v"(o5.w));   // .w kept live: ds_read_b128, not b96
        }
        // slots fill in arrival order: a seventh / eighth member exists for some lane of 2 in 3 / 1 in 4 waves
        if (!POWK && !(D2D_EXP_ABL & 4) && __builtin_amdgcn_ballot_w64(off[6] != EMPTY) != 0ull) {
            const f32x4 o6 = lds_get<f32x4>(L_LINK + off[6]), o7 = lds_get<f32x4>(L_LINK + off[7]);
            if (off[7] != EMPTY) members = lds_get<unsigned>(L_CNT + rbc * 4u);
            RO_PAIR(6, o6) RO_PAIR(7, o7)
            asm volatile("" ::"v"(o6.w), "v"(o7.w));
        }
#undef RO_PAIR
#undef RO_POWK_GROUP
#undef RO_POWK_ALL
        // all partial sums exact <=> the sum is the ascending-order sum: largest and smallest non-zero term within 2^25
        // (nine values of 24 bits inside the 53 of a double); compared on the raw bits (conservative by less than one binade)
        // ... and a non-finite OWN term (transmitter and receiver in one place under 1 / d^2: inf) cannot be taken back out of the sum
        // (-inf + inf = NaN where the generic kernels, which leave the own link out, keep the interferers' finite sum): any inf / NaN
        // among the terms sends the lane to the sorted sum, which excludes the own entry
        const bool inexact = ((unsigned)tmax - tmin >= (25u << 23)) | (tmax >= 0x7F800000);
        const bool big = members > (unsigned)RO_SLOTS;           // more members than the row holds: the rest is in the pool
        if (UNLIKELY(__builtin_amdgcn_ballot_w64(inexact | big | oor[u]) != 0ull)) {
            if (oor[u]) {
                // an RB outside [0, R) has no list: whoever shares that value is found by the sweep
                dmin = 0x7F000000;
                acc = sweep(dmin);
            } else if (big) {
                // eight members in the row (own entry possibly among them), the others in the pool, tagged with their RB.  First
                // the order-free attempt: all terms in whatever order they sit, exact inside a window of 53 - 24 - bits(members) bits
                dmin = 0x7F000000;
                double s = 0.0;
                unsigned hi = 0u, lo = 0xFFFFFFFFu;
#pragma unroll
                for (int k = 0; k < RO_SLOTS; ++k) {
                    if (off[k] != my_off) { const float t = term(off[k], dmin); s += (double)t; hi = max(hi, __float_as_uint(t)); lo = min(lo, __float_as_uint(t) - 1u); }
                }
                const unsigned pc = min(lds_get<unsigned>(L_FLAGS + 12u), (unsigned)N);
                COLD_LOOP
                for (unsigned p = 0; p < pc; ++p) {
                    const u32x2 e = lds_get<u32x2>(L_POOL + p * 8u);
                    if (e.x == (unsigned)rb[u] && e.y != (unsigned)i) {
                        const float t = term(e.y << 4, dmin); s += (double)t; hi = max(hi, __float_as_uint(t)); lo = min(lo, __float_as_uint(t) - 1u);
                    }
                }
                const unsigned window = 29u - (32u - (unsigned)__builtin_clz(members));       // 9 .. 15 members: 25 bits; 2047: 18
                if (hi - lo >= (window << 23)) {
                    // ... else in ascending link order.  A lane that sweeps all N pairs outlives its whole launch (20 us and more
                    // against 3 per workgroup), so the sweep is kept for RBs where it is no worse than anything else could be;
                    // up to 32 members: selection - the smallest member offset not yet taken, members x (8 + pool) compares
                    dmin = 0x7F000000;
                    if (members <= 32u) {
                        s = 0.0;
                        unsigned from = 0u;
                        COLD_LOOP
                        while (true) {
                            unsigned best = 0xFFFFFFFFu;
#pragma unroll
                            for (int k = 0; k < RO_SLOTS; ++k) if (off[k] >= from && off[k] != my_off) best = min(best, off[k]);
                            COLD_LOOP
                            for (unsigned p = 0; p < pc; ++p) {
                                const u32x2 e = lds_get<u32x2>(L_POOL + p * 8u);
                                const unsigned c = e.y << 4;
                                if (e.x == (unsigned)rb[u] && c >= from && c != my_off) best = min(best, c);
                            }
                            if (best == 0xFFFFFFFFu) break;
                            s += (double)term(best, dmin);
                            from = best + 16u;
                        }
                    } else s = sweep(dmin);
                }
                acc = s;
            } else if (inexact) {
                unsigned v[RO_SLOTS];
#pragma unroll
                for (int k = 0; k < RO_SLOTS; ++k) v[k] = off[k] == my_off ? 0xFFFFu : off[k];   // own entry out; sorts last
                sort8(v);
                double s = 0.0;
                int dummy = 0;
                COLD_LOOP
                for (int k = 0; k < RO_SLOTS - 1; ++k) {
                    if (v[k] >= EMPTY) break;
                    s += (double)term(v[k], dummy);
                }
                acc = s;
            }
        }

        // ---- SINR / SNR / rate / capacity: the arithmetic of step_kernel, operation for operation
        if (POWLAW) dmin = min(dmin, __float_as_int(d2_own));
        const float sig = pz[u] * g_own * rx_pl * rx_lin;            // mW at the receiver, with rx gains
        const float accf = (float)acc;
        // interferers: no rx gains (simulator.py:100); the fma every kernel's `accf * rx_pl + noise` contracted to, spelled out
#if D2D_EXP_ABL & 16
        const float sinr_lin = sig + accf, sinr_db = sinr_lin * rx_pl, snr_db = sig * noise, sh = sinr_lin + 1.0f;
#else
        const float sinr_lin = precise_div(sig, fmaf(accf, rx_pl, noise));
        const float sinr_db = 3.01029995663981195f * __builtin_amdgcn_logf(sinr_lin);                 // simulator.py:106-107
        const float snr_db = 3.01029995663981195f * __builtin_amdgcn_logf(precise_div(sig, noise));   // simulator.py:115
        const float u1p = 1.0f + sinr_lin, um1 = u1p - 1.0f;
        const float sh_big = __builtin_amdgcn_logf(u1p) * fast_div(sinr_lin, um1 == 0.0f ? 1.0f : um1);
        const float sh = um1 == 0.0f ? sinr_lin * 1.44269504088896340736f : sh_big;
#endif
        const bool ok = sinr_db > sens;                              // simulator.py:123,149
        const float rate = ok ? sh : 0.0f;
        const float cap = ok ? bw_mhz * sh : 0.0f;                   // simulator.py:150-151
        caps[u] = cap; rates[u] = rate; sinrs[u] = sinr_db; snrs[u] = snr_db;

        // per-lane rarities behind ONE wave-uniform branch: SystemCapacity's -1 rule (reward_fn.py:29-41: I am a non-D2D link whose
        // capacity is <= min_capacity and some D2D link shares my RB), a non-finite SINR / zero distance
        const bool rule = capacity_reward && type != LINK_SIDELINK && cap <= a.reward_param;
        // ShannonRewardFunction (reward_fn.py:52-57) is per link: log2(1 + SINR), or -1 below the threshold.  (A plain 4-byte store
        // per link, in here: the value would otherwise stay live across the other link's evaluation - a 65th VGPR.)
        if (shannon_reward) *at(a.reward, fresh((row + (unsigned)i) * 4u)) = sinr_db >= a.reward_param ? sh : -1.0f;
        if (LPT == 1) sh_kept = sh;
        const bool nonfinite = NF_ONLY ? !(fabsf(sinr_db) <= 3.0e38f) : (dmin == 0 || !(fabsf(sinr_db) <= 3.0e38f));
        if (UNLIKELY(__builtin_amdgcn_ballot_w64(rule | nonfinite) != 0ull)) {
            if (rule) {
                bool hit = false;
                const auto sidelink = [&](unsigned j) { return ((a.side_words[j >> 5] >> (j & 31u)) & 1u) != 0u; };
                if (oor[u]) {
                    COLD_LOOP
                    for (int k = 0; k < N; ++k)
                        hit |= (k != i) & sidelink((unsigned)k) & (rb_of((unsigned)k, lds_get<f32x4>(L_LINK + ((unsigned)k << 4))) == rb[u]);
                } else {
#pragma unroll
                    for (int k = 0; k < RO_SLOTS; ++k)
                        if (off[k] != EMPTY && off[k] != my_off) hit |= sidelink(off[k] >> 4);
                    if (big) {
                        const unsigned pc = min(lds_get<unsigned>(L_FLAGS + 12u), (unsigned)N);
                        COLD_LOOP
                        for (unsigned p = 0; p < pc; ++p) {
                            const u32x2 e = lds_get<u32x2>(L_POOL + p * 8u);
                            if (e.x == (unsigned)rb[u] && e.y != (unsigned)i) hit |= sidelink(e.y);
                        }
                    }
                }
                if (hit) lds_atomic_or(L_FLAGS + 4u, 1);
            }
            int my_flags = 0;
            if (NF_ONLY) {
                // 1 / d^2 gains: a zero distance (own link: signal inf; an interferer: accumulator inf) always ends in a non-finite SINR
                if (nonfinite) { my_flags |= FLAG_NON_FINITE; if (d2_own == 0.0f || !(accf <= 3.0e38f)) my_flags |= FLAG_ZERO_DISTANCE; }
            } else {
                if (dmin == 0) my_flags |= FLAG_ZERO_DISTANCE;
                if (!(fabsf(sinr_db) <= 3.0e38f)) my_flags |= FLAG_NON_FINITE;
            }
            if (my_flags) lds_atomic_or(L_FLAGS, my_flags);
        }
        RO_STAMP(5 + u);
    }

    if (cue_sinr_reward) {
        // ---- CueSinrShannonRewardFunction, reward_fn.py:65-78: -1 where another member of my RB is a non-D2D link whose SINR is below
        // the threshold, else log2(1 + SINR).  Every link publishes that predicate about ITSELF; after one barrier a link reads it
        // for the members of its RB's row (the stand-in's entry is 0), the pool where the row overflowed, every link where the RB is
        // out of range - the search of pass 2 once more, with a 4-byte read per member instead of a pair evaluation.
        const int i = link_of(0);
        const unsigned my_off = (unsigned)i << 4;
        const int type = (in[0].ra.x >> D2D_REC_TYPE_SHIFT) & D2D_REC_TYPE_MASK;
        lds_put<int>(L_LOW + ((unsigned)i << 2), (type != LINK_SIDELINK) & (sinrs[0] < a.reward_param) ? 1 : 0);
        if (tid == 0) lds_put<int>(L_LOW + ((unsigned)N << 2), 0);
        __syncthreads();
        int low = 0;
        if (oor[0]) {
            COLD_LOOP
            for (int k = 0; k < N; ++k)
                low |= (k != i) & ((POWK ? lds_get<int>(L_RB + ((unsigned)k << 2)) : __float_as_int(lds_get<f32x4>(L_LINK + ((unsigned)k << 4)).w)) == rb[0]) ? lds_get<int>(L_LOW + ((unsigned)k << 2)) : 0;
        } else {
            const u32x4 ml = lds_get<u32x4>(L_SLOTS + (unsigned)rb[0] * 16u);
            const unsigned o[RO_SLOTS] = {ml.x & 0xFFFFu, ml.x >> 16, ml.y & 0xFFFFu, ml.y >> 16, ml.z & 0xFFFFu, ml.z >> 16, ml.w & 0xFFFFu, ml.w >> 16};
#pragma unroll
            for (int k = 0; k < RO_SLOTS; ++k) { const int f = lds_get<int>(L_LOW + (o[k] >> 2)); low |= o[k] != my_off ? f : 0; }
            if (__builtin_amdgcn_ballot_w64(o[7] != EMPTY) != 0ull) {
                if (o[7] != EMPTY && lds_get<unsigned>(L_CNT + (unsigned)rb[0] * 4u) > (unsigned)RO_SLOTS) {
                    const unsigned pc = min(lds_get<unsigned>(L_FLAGS + 12u), (unsigned)N);
                    COLD_LOOP
                    for (unsigned p = 0; p < pc; ++p) {
                        const u32x2 e = lds_get<u32x2>(L_POOL + p * 8u);
                        if (e.x == (unsigned)rb[0] && e.y != (unsigned)i) low |= lds_get<int>(L_LOW + (e.y << 2));
                    }
                }
            }
        }
        *at(a.reward, fresh((row + (unsigned)i) * 4u)) = low ? -1.0f : sh_kept;
    }

    // ---- reward, first half - ISSUED AHEAD OF THE RESULT STORES (round 6): the wave's capacity sum and its ticket are a chain of
    // dependent long-latency steps (five DPP levels, an LDS write, the ticket's round trip); behind the stores it was the last thing
    // a wave did and every workgroup's exit waited for it - 1.3 us of the launch by ablation
    // (profiles/r6_rollout_kernel_ablation.jsonl); in front of them it runs under the stores' issue.
    // Barrier-free ticket reduction (see step_kernel): DPP wave sum into the wave's own slot, the wave that draws the last ticket
    // finishes the env
    const int lane = tid & 63;
    int ticket = 0;
    if (!PAD) {
    // one float sum per GROUP OF 64 LINKS, with the roundings of the one-link-per-thread kernels' wave sum - a balanced tree over
    // adjacent links: pairs, fours, ... 32 + 32.  LPT = 2: a lane's two links ARE the first level, the DPP steps the next four, and
    // the two halves of the wave end as the two groups' sums (wave_sum_halves).  Every wave parks its group sums in its own slots of
    // red[16] (no atomic: nobody else writes them) and then takes a ticket; the wave that draws the last one adds the slots in INDEX
    // order - the generic kernels' `(v.x + v.y) + (v.z + v.w)` per four - so the env's total has the same bits whatever LPT, whatever
    // the order the waves arrive in, and whatever kernel reduced it.  (Until round 6 this was a 32.32 fixed-point atomic sum: exact,
    // hence order-free too, but truncating below 2^-32 Mbps per wave - a last-digit difference from the float sums of envs that share
    // a workgroup once capacities are 1e-6 Mbps, found by the fuzzer at a path-loss exponent of 5.6 - and two conversions, a 64-bit
    // atomic and an overflow guard per link dearer.)
    if (LPT == 2) {
        float lo, hi;
        wave_sum_halves(caps[0] + caps[LPT - 1], lo, hi);
        if (lane == 0) lds_put<f32x2>(L_RED + (unsigned)(tid >> 6) * 8u, f32x2{lo, hi});
    } else {
        const float wsum = wave_sum(shadow ? 0.0f : caps[0]);
        if (lane == 0) lds_put<float>(L_RED + (unsigned)(tid >> 6) * 4u, wsum);
    }
    asm volatile("" ::"v"(pf));                                      // the prefetched words are consumed here (no instruction)
    // This wave's LDS writes above precede its ticket in the LDS queue (in order per wave); the compiler barrier keeps them
    // above it in the instruction stream.  The last wave's reads below stay behind its own ticket (acquire).
    asm volatile("" ::: "memory");
    if (lane == 0) ticket = __hip_atomic_fetch_add((D2D_LDS(int)*)(L_FLAGS + 8u), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }

    // ---- results: the thread's LPT links are adjacent elements of every plane and adjacent rows of the table
    {
        const unsigned o4 = fresh((row + (unsigned)link_of(0)) * 4u);
        if ((D2D_EXP_ABL & 2) && sinrs[0] != 1.2345f) {
        } else
        if (LPT == 2) {
            RO_ST(reinterpret_cast<f32x2*>(at(a.sinr_db, o4)), (f32x2{sinrs[0], sinrs[LPT - 1]}));
            RO_ST(reinterpret_cast<f32x2*>(at(a.snr_db, o4)), (f32x2{snrs[0], snrs[LPT - 1]}));
            RO_ST(reinterpret_cast<f32x2*>(at(a.rate, o4)), (f32x2{rates[0], rates[LPT - 1]}));
            RO_ST(reinterpret_cast<f32x2*>(at(a.cap, o4)), (f32x2{caps[0], caps[LPT - 1]}));
        } else {
            RO_ST(at(a.sinr_db, o4), sinrs[0]);
            RO_ST(at(a.snr_db, o4), snrs[0]);
            RO_ST(at(a.rate, o4), rates[0]);
            RO_ST(at(a.cap, o4), caps[0]);
        }
        if (a.write_table) {                                         // obs_fn.py:57-60: (tx, rx, sinr, snr) per link, 24 bytes
            const unsigned o4t = fresh((row + (unsigned)link_of(0)) * 4u);
            unsigned char* t = reinterpret_cast<unsigned char*>(at(a.table, (o4t << 2) + (o4t << 1)));
            if (LPT == 2) {                                          // 48 contiguous bytes
                const float4 p0 = in[0].pos, p1 = in[LPT - 1].pos;
                const f32x4 v0 = {p0.x, p0.y, p0.z, p0.w}, v1 = {sinrs[0], snrs[0], p1.x, p1.y}, v2 = {p1.z, p1.w, sinrs[LPT - 1], snrs[LPT - 1]};
                if (a.lds.env_bytes - LDS_HEAD_BYTES >= (unsigned)N * 24u) {
                    // Through LDS, so that every store instruction writes 1024 CONTIGUOUS bytes: a lane's own 48 bytes, stored
                    // directly, leave every 128-byte line to be completed by three instructions - fine for L2 when it merges them
                    // (28.9 us), ruinous with the nt hint (37 us: partial lines go out as they are), and either way behind full
                    // lines (24.3 us; 26.7 with one link per thread.  profiles/r5_table_rows_through_lds.jsonl).  The staging area is
                    // the tuples and lists themselves, which nobody reads after this barrier: the wave's 128 rows at their place in
                    // the env's table image, written and read back by the same wave (no second barrier).
                    __syncthreads();
                    const unsigned mine = LDS_HEAD_BYTES + (unsigned)tid * 48u;
                    const unsigned wave0 = LDS_HEAD_BYTES + (unsigned)(tid & ~63) * 48u + (unsigned)(tid & 63) * 16u;
                    lds_put<f32x4>(mine, v0); lds_put<f32x4>(mine + 16u, v1); lds_put<f32x4>(mine + 32u, v2);
                    const f32x4 w0 = lds_get<f32x4>(wave0), w1 = lds_get<f32x4>(wave0 + 1024u), w2 = lds_get<f32x4>(wave0 + 2048u);
                    const unsigned o4w = fresh((row + (unsigned)(LPT * (tid & ~63))) * 4u);
                    unsigned char* tw = reinterpret_cast<unsigned char*>(at(a.table, (o4w << 2) + (o4w << 1))) + (unsigned)(tid & 63) * 16u;
                    RO_ST(reinterpret_cast<f32x4*>(tw), w0);
                    RO_ST(reinterpret_cast<f32x4*>(tw + 1024), w1);
                    RO_ST(reinterpret_cast<f32x4*>(tw + 2048), w2);
                } else {                                             // (an env whose LDS is smaller than its table: R < N / 4.  Never nt)
                    *reinterpret_cast<f32x4*>(t) = v0; *reinterpret_cast<f32x4*>(t + 16) = v1; *reinterpret_cast<f32x4*>(t + 32) = v2;
                }
            } else {
                const float4 p0 = in[0].pos;
                RO_ST(reinterpret_cast<f32x2*>(t), (f32x2{p0.x, p0.y}));
                RO_ST(reinterpret_cast<f32x2*>(t + 8), (f32x2{p0.z, p0.w}));
                RO_ST(reinterpret_cast<f32x2*>(t + 16), (f32x2{sinrs[0], snrs[0]}));
            }
        }
    }

    RO_STAMP(7);
    if (PAD) {
        // ---- reward for a link count that is no multiple of 64: what the generic kernels do for such shapes, operation for
        // operation (their wave partition is this kernel's): the waves' float sums through LDS, one barrier, a fixed-order total
        const float wsum = wave_sum(shadow ? 0.0f : caps[0]);
        if ((tid & 63) == 0) lds_put<float>(L_RED + (unsigned)(tid >> 6) * 4u, wsum);
        asm volatile("" ::"v"(pf));
        __syncthreads();
        float total = 0.0f;
        for (int w = 0; w < (TPE + 255) >> 8; ++w) { const f32x4 v = lds_get<f32x4>(L_RED + (unsigned)w * 16u); total += (v.x + v.y) + (v.z + v.w); }
        if (capacity_reward) {
            const float r = lds_get<int>(L_FLAGS + 4u) ? -1.0f : total * a.inv_n;
            if (a.reward_env) { if (tid == 0) a.reward_env[b] = r; }
            else *at(a.reward, fresh((row + (unsigned)link_of(0)) * 4u)) = r;
        }
        if (tid == 0) a.env_flags[b] = lds_get<int>(L_FLAGS);
        RO_STAMP(8);
        return;
    }
    if ((D2D_EXP_ABL & 8) && caps[0] != 1.2345f) return;
    // ---- reward, second half: the wave that drew the last ticket finishes the env
    ticket = __builtin_amdgcn_readfirstlane(ticket);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    if (ticket == (TPE >> 6) - 1) {
        // SystemCapacityRewardFunction, reward_fn.py:27-44: mean capacity, or -1 for everyone on a violation
        float total = 0.0f;
        for (int w = 0; w < (N + 255) >> 8; ++w) {                   // (atomic loads: LDS reads that cannot be hoisted above the ticket)
            const float v0 = __hip_atomic_load((D2D_LDS(float)*)(L_RED + (unsigned)w * 16u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            const float v1 = __hip_atomic_load((D2D_LDS(float)*)(L_RED + (unsigned)w * 16u + 4u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            const float v2 = __hip_atomic_load((D2D_LDS(float)*)(L_RED + (unsigned)w * 16u + 8u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            const float v3 = __hip_atomic_load((D2D_LDS(float)*)(L_RED + (unsigned)w * 16u + 12u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            total += (v0 + v1) + (v2 + v3);
        }
        const int viol = __hip_atomic_fetch_or((D2D_LDS(int)*)(L_FLAGS + 4u), 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        const float r = (viol & 1) ? -1.0f : total * a.inv_n;        // (an inf / NaN capacity rides through the float sum by itself)
        if (!capacity_reward) {
            // per-link rewards are out already; the ticket only decides who publishes the env's flags
        } else if (a.reward_env) {                                     // D2D_REWARD_PER_ENV: the scalar once, not N copies
            if (lane == 0) a.reward_env[b] = r;
        } else {
            const f32x4 r4 = {r, r, r, r};
            for (int k = lane * 4; k < N; k += 256) RO_ST(reinterpret_cast<f32x4*>(at(a.reward, fresh((row + (unsigned)k) * 4u))), r4);
        }
        if (lane == 0) a.env_flags[b] = __hip_atomic_fetch_or((D2D_LDS(int)*)(L_FLAGS), 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    RO_STAMP(8);
}

hipError_t launch_rollout(const StepArgs& a, PlMode mode, int opt, int block_threads, hipStream_t stream) {
    dim3 grid((unsigned)a.B), block(block_threads);
    const size_t lds = a.lds.env_bytes;
    hipError_t err = hipSuccess;
#define D2D_RO_1(M, O, L)                                                                                \
    do {                                                                                                 \
        static int checked = 0;                                                                          \
        if (!checked) {                                 /* raw LDS addressing: the dynamic block must start at LDS address 0 */ \
            hipFuncAttributes fa;                                                                        \
            err = hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(&rollout_kernel<M, O, L>));    \
            if (err == hipSuccess && fa.sharedSizeBytes != 0) err = hipErrorInvalidConfiguration;        \
            if (err == hipSuccess) checked = 1;                                                          \
        }                                                                                                \
        if (err == hipSuccess && lds > 48 * 1024)                                                        \
            err = hipFuncSetAttribute(reinterpret_cast<const void*>(&rollout_kernel<M, O, L>),           \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);             \
        if (err == hipSuccess) {                                                                         \
            hipLaunchKernelGGL((rollout_kernel<M, O, L>), grid, block, lds, stream, a);                  \
            err = hipGetLastError();                                                                     \
        }                                                                                                \
    } while (0)
#define D2D_RO_L(M, L)                                                                                   \
    switch (opt & (OPT_SREC | OPT_NT)) {                                                                 \
        case 0: D2D_RO_1(M, 0, L); break;                                                                \
        case OPT_SREC: D2D_RO_1(M, OPT_SREC, L); break;                                                  \
        case OPT_NT: D2D_RO_1(M, OPT_NT, L); break;                                                      \
        default: D2D_RO_1(M, OPT_SREC | OPT_NT, L); break;                                               \
    }
#define D2D_RO_X(M)                                     /* exact positions: one link per thread */         \
    switch (opt & (OPT_SREC | OPT_NT | OPT_PAD)) {                                                       \
        case 0: D2D_RO_1(M, OPT_XPOS, 1); break;                                                         \
        case OPT_SREC: D2D_RO_1(M, OPT_XPOS | OPT_SREC, 1); break;                                       \
        case OPT_NT: D2D_RO_1(M, OPT_XPOS | OPT_NT, 1); break;                                           \
        case OPT_SREC | OPT_NT: D2D_RO_1(M, OPT_XPOS | OPT_SREC | OPT_NT, 1); break;                     \
        case OPT_PAD: D2D_RO_1(M, OPT_XPOS | OPT_PAD, 1); break;                                         \
        default: D2D_RO_1(M, OPT_XPOS | OPT_PAD | OPT_NT, 1); break;                                     \
    }
#define D2D_RO(M) do {                                                                                   \
        if (opt & OPT_XPOS) { D2D_RO_X(M) }                                                              \
        else if (opt & OPT_PAD) { if (opt & OPT_NT) D2D_RO_1(M, OPT_PAD | OPT_NT, 1); else D2D_RO_1(M, OPT_PAD, 1); }  \
        else if (a.lpt == 2) { D2D_RO_L(M, 2) } else { D2D_RO_L(M, 1) }                                  \
    } while (0)
    if (mode == PL_INV_SQUARE) D2D_RO(PL_INV_SQUARE); else if (mode == PL_POWK) D2D_RO(PL_POWK); else D2D_RO(PL_POWER);
#undef D2D_RO
#undef D2D_RO_X
#undef D2D_RO_L
#undef D2D_RO_1
    return err;
}

}  // namespace d2d
