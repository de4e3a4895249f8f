// Rollout kernel for gfx950 (MI355X): the step of a learner's rollout - raw agent actions for every link, SystemCapacity
// reward, one env per 512-thread (N-thread) workgroup, sparse RB occupancy (N <= 4 R) - with a STRAIGHT-LINE fast path.
//
// Reference path (file:line under /root/reference/src/gym_d2d): the same as csrc/d2d_step.hip -
//   D2DEnv._decode_action envs/d2d_env.py:93-101, Actions.get_actions_by_rb actions.py:27-31,
//   Simulator._calculate_sinrs / _snrs / _rates / _network_capacity simulator.py:89-154,
//   SystemCapacityRewardFunction envs/reward_fn.py:27-44, LinearObsFunction's base table envs/obs_fn.py:55-61.
//
// Why a kernel of its own (round 5): at 36 - 64 bytes per link the step is paced by its instruction streams, not by HBM
// (profiles/r4_elasticity_step_kernel.json: +96 SALU per wave = +25 %, +64 VALU = +10 %).  The generic kernel's member-list
// walk spent 70 of its 246 VALU per wave on taking the own entry out of the RB's list, sorting the rest and clamping the
// indices, and ~40 of its 102 SALU on exec-mask bookkeeping around per-lane rarities.  This kernel
//   * keeps every per-lane rarity (action beyond the multiply-high decode's range or beyond R * P, a ninth link on an RB)
//     out of the instruction stream: such a lane raises ONE workgroup flag in pass 1 and the whole workgroup then takes the
//     general path (exact decode, membership masks, nested walk - the generic kernel's code) behind one scalar branch;
//   * stores the list entries as LDS BYTE OFFSETS of the members' tuples (u16, link * 16; empty = N * 16, the zero-power
//     stand-in tuple), so an entry is an address: no clamp, no shift;
//   * does NOT sort and does NOT take the own entry out: the interference sum starts at minus the own term and adds all
//     eight slots in arrival order.  Every term is a float (24 bits) added in double (53 bits): while the largest and the
//     smallest non-zero term of a lane are less than 2^25 apart, every partial sum of the at most nine values is exact,
//     hence independent of the order and equal - bit for bit - to the ascending-order sum of the mask walk and the
//     all-pairs sweep.  The two extremes are tracked with one v_max3 / v_min3 pair per two terms; a lane outside the
//     window (1e-6 of lanes without, 5e-5 with the own term at BASELINE config 3 geometry) re-does its sum in sorted order.
// Same arithmetic as step_kernel everywhere else (tests/test_gpu_step_variants.py holds the two bit-identical).
#include "d2d_step_device.h"

namespace d2d {

#define RO_SLOTS 8
#define RO_ST(ptr, val) do { if (NT) __builtin_nontemporal_store((val), (ptr)); else *(ptr) = (val); } while (0)

__device__ __forceinline__ float2 lds_f2(unsigned addr) { const f32x2 v = lds_get<f32x2>(addr); return make_float2(v.x, v.y); }

template <int MODE, int OPT>
__global__ __launch_bounds__(1024) void rollout_kernel(const StepArgs a) {
    constexpr bool SREC = (OPT & OPT_SREC) != 0, NT = (OPT & OPT_NT) != 0;
    constexpr bool POWLAW = MODE == PL_POWER;
    static_assert(MODE == PL_INV_SQUARE || MODE == PL_POWER, "the rollout kernel serves the power laws");
    extern __shared__ __align__(16) unsigned char smem_raw[];
    const int N = a.N, R = a.R, W = a.mask_words, TPE = a.tpe;
    const int tid = threadIdx.x, i = tid, b = (int)blockIdx.x;
    const unsigned row = (unsigned)b * (unsigned)N;              // element offsets fit 32 bits (run_step refuses B * N * 24 >= 2^32)
    const Smem s = carve(smem_raw, a.lds, R, W);
    const bool cfg_export_actions = a.rb_out != nullptr;
    const unsigned EMPTY = (unsigned)N * 16u;                    // byte offset of the stand-in tuple link[N]

    // ---- prologue: this link's loads, issued before any LDS work or barrier (their latency overlaps pass 0)
    LinkRaw in;
    if (SREC) {
        // the records of this wave's 64 links are identical (StepArgs::rec_uniform): the whole record in one 64-byte scalar load
        in.act0 = *at(a.actions, fresh((row + (unsigned)i) * 4u));
        in.act1 = 0;
        const i32x16 g = scalar_load64(reinterpret_cast<const unsigned char*>(a.rec_grp) + (unsigned)__builtin_amdgcn_readfirstlane(tid));
        in.ra = make_int4(g[2], 0, g[0], g[1]);
        in.rb_ = make_float4(__int_as_float(g[4]), __int_as_float(g[5]), __int_as_float(g[6]), __int_as_float(g[7]));
        in.rc = make_float4(__int_as_float(g[8]), __int_as_float(g[9]), __int_as_float(g[10]), __int_as_float(g[11]));
        in.hh = make_float2(__int_as_float(g[12]), __int_as_float(g[13]));
        in.pos = *at(a.lpos, fresh((row + (unsigned)i) * 16u));
    } else {
        in = load_link(a, row, row, i, 0, 0, true, false, POWLAW);
    }

    // LDS byte addresses (StepLds): tuples at 80, the lists' slots[R] (16 bytes each) then cnt[R], flags at 64, the sums at 0
    const unsigned L_LINK = LDS_HEAD_BYTES, L_SLOTS = a.lds.lists, L_CNT = a.lds.lists + (unsigned)R * 16u, L_EXPO = a.lds.expo;
    const unsigned L_FLAGS = 64u, L_DUMP = 60u;                  // red[15]: where a ninth link's entry goes

    // ---- pass 0: slots[R] <- EMPTY.., cnt[R] <- 0 (one contiguous region of 16-byte units), flags, the stand-in tuple
    {
        const int nl = R + ((R + 3) >> 2);
        const unsigned e2 = EMPTY | (EMPTY << 16);
        if (tid < nl) { const unsigned f = tid < R ? e2 : 0u; lds_put<u32x4>(L_SLOTS + (unsigned)tid * 16u, u32x4{f, f, f, f}); }
        if (UNLIKELY(nl > TPE)) {                                // more RBs than 0.8 N: further rounds (workgroup-uniform)
            COLD_LOOP
            for (int k = tid + TPE; k < nl; k += TPE) { const unsigned f = k < R ? e2 : 0u; lds_put<u32x4>(L_SLOTS + (unsigned)k * 16u, u32x4{f, f, f, f}); }
        }
        if (tid == TPE - 1) {
            lds_put<f32x4>(L_LINK + EMPTY, f32x4{1.0e18f, 1.0e18f, 0.0f, __int_as_float(-1)});
            if (POWLAW) lds_put<f32x2>(L_EXPO + (EMPTY >> 1), f32x2{-1.0f, 0.0f});
        }
        if (tid < 5) lds_put<u32x4>((unsigned)tid * 16u, u32x4{0u, 0u, 0u, 0u});          // red[16] + flags[4]
    }
    // nothing that consumes a loaded value may be scheduled above this barrier (the wave would sit on the HBM round trip
    // before pass 0 instead of behind it)
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();
    __builtin_amdgcn_sched_barrier(0);

    // ---- pass 1: decode (multiply-high arm only), stage the transmitter tuple, enter the RB's list
    const int type = (in.ra.x >> D2D_REC_TYPE_SHIFT) & D2D_REC_TYPE_MASK;
    const float2 rx = make_float2(in.pos.z, in.pos.w);
    const unsigned P = __float_as_uint(in.rc.w) & 0xFFFFu;
    // the host keeps the bound below R * P (refresh_tables): an action inside it decodes to rb < R, one beyond it (or
    // negative: a huge unsigned) sends the workgroup down the general path
    const bool bad = (unsigned)in.act0 > (unsigned)in.ra.w;
    int rb = (int)__umulhi((unsigned)in.act0, (unsigned)in.ra.z);
    int pw = in.act0 - (int)__umul24((unsigned)rb, P);
    float4 me = make_float4(in.pos.x, in.pos.y, pow10_tenth(pw) * in.rb_.x, __int_as_float(rb));
    const unsigned my_off = (unsigned)i << 4;
    lds_put<f32x4>(L_LINK + my_off, f32x4{me.x, me.y, me.z, me.w});
    if (POWLAW) lds_put<f32x2>(L_EXPO + (my_off >> 1), f32x2{in.hh.x, in.hh.y});
    if (cfg_export_actions) { const unsigned oe = fresh((row + (unsigned)i) * 4u); RO_ST(at(a.rb_out, oe), rb); RO_ST(at(a.pwr_out, oe), pw); }
    {
        const unsigned rbc = min((unsigned)rb, (unsigned)(R - 1));                  // a bad lane's garbage stays inside cnt[]
        const unsigned slot = __hip_atomic_fetch_add((D2D_LDS(unsigned)*)(L_CNT + rbc * 4u), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);   // ds_add_rtn_u32
        const bool ovf = slot >= (unsigned)RO_SLOTS;
        const unsigned dst = ovf ? L_DUMP : L_SLOTS + rbc * 16u + slot * 2u;
        lds_put<unsigned short>(dst, (unsigned short)my_off);
        const bool odd = bad | ovf;
        if (UNLIKELY(__builtin_amdgcn_ballot_w64(odd) != 0ull)) { if (odd) atomicOr(&s.flags[3], 1); }
    }
    __syncthreads();
    const bool general = __builtin_amdgcn_readfirstlane(lds_get<int>(L_FLAGS + 12u)) != 0;

    // software prefetch of the action row of the env the workgroup `prefetch_envs` later will own (see step_kernel)
    int pf;
    {
        const int bq = b + a.prefetch_envs;
        const int bp = bq < a.B ? bq : a.B - 1;
        pf = *at(a.actions, fresh(((unsigned)bp * (unsigned)N + (unsigned)i) * 4u));
    }

    const float rx_pl = in.rb_.y, rx_lin = in.rb_.z, noise = in.rb_.w;
    const float sens = in.rc.x, bw_mhz = in.rc.y;
    int dmin = 0x7F000000;                                           // power law: bits of the smallest squared distance met
    double acc;
    bool use_masks = false;                                          // general path: this lane walked the masks
    uint4 mlist = make_uint4(0u, 0u, 0u, 0u);

    // own link: simulator.py:93 (first: its term opens the interference sum below)
    float d2_own, g_own;
    {
        const float dx = me.x - rx.x, dy = me.y - rx.y;
        d2_own = fmaf(dx, dx, dy * dy);
        g_own = pair_gain<MODE>(d2_own, in.hh);
    }

    if (LIKELY(!general)) {
        // ---- pass 2, fast: the RB's eight slots in one read, every slot a tuple address
        { const u32x4 m4 = lds_get<u32x4>(L_SLOTS + (unsigned)rb * 16u); mlist = make_uint4(m4.x, m4.y, m4.z, m4.w); }
        const unsigned off[RO_SLOTS] = {mlist.x & 0xFFFFu, mlist.x >> 16, mlist.y & 0xFFFFu, mlist.y >> 16,
                                        mlist.z & 0xFFFFu, mlist.z >> 16, mlist.w & 0xFFFFu, mlist.w >> 16};
        // .difference({action}) (simulator.py:95) by arithmetic: the own entry is among the slots, so the sum opens at minus
        // its term (me.z * g_own - the very product the slot's evaluation repeats, same operands, same rounding)
        acc = -(double)(me.z * g_own);
        unsigned tmax = 0u, tmin = 0xFFFFFFFFu;
#define RO_PAIR(k, o)                                                                                                   \
        {                                                                                                               \
            const float dx = (o).x - rx.x, dy = (o).y - rx.y;                                                           \
            const float d2 = fmaf(dx, dx, dy * dy);                                                                     \
            const float g = pair_gain<MODE>(d2, POWLAW ? lds_f2(L_EXPO + (off[k] >> 1)) : make_float2(-1.0f, 0.0f)); \
            if (POWLAW) dmin = min(dmin, __float_as_int(d2));                                                           \
            const float t = (o).z * g;                               /* simulator.py:97-101, linear mW */               \
            acc += (double)t;                                                                                           \
            tmax = max(tmax, __float_as_uint(t)); tmin = min(tmin, __float_as_uint(t) - 1u);   /* zero terms: ignored */ \
        }
        {
            const f32x4 o0 = lds_get<f32x4>(L_LINK + off[0]), o1 = lds_get<f32x4>(L_LINK + off[1]),
                         o2 = lds_get<f32x4>(L_LINK + off[2]), o3 = lds_get<f32x4>(L_LINK + off[3]),
                         o4 = lds_get<f32x4>(L_LINK + off[4]), o5 = lds_get<f32x4>(L_LINK + off[5]);
            RO_PAIR(0, o0) RO_PAIR(1, o1) RO_PAIR(2, o2) RO_PAIR(3, o3) RO_PAIR(4, o4) RO_PAIR(5, o5)
            asm volatile("" ::"v"(o0.w), "v"(o1.w), "v"(o2.w), "v"(o3.w), "v"(o4.w), "v"(o5.w));   // .w kept live: ds_read_b128, not b96
        }
        // slots fill in arrival order: a seventh / eighth member exists for some lane of 2 in 3 / 1 in 4 waves
        if (__builtin_amdgcn_ballot_w64(off[6] != EMPTY) != 0ull) {
            const f32x4 o6 = lds_get<f32x4>(L_LINK + off[6]), o7 = lds_get<f32x4>(L_LINK + off[7]);
            RO_PAIR(6, o6) RO_PAIR(7, o7)
            asm volatile("" ::"v"(o6.w), "v"(o7.w));
        }
#undef RO_PAIR
        // all partial sums exact <=> the sum is the ascending-order sum: largest and smallest non-zero term within 2^25
        // (nine values of 24 bits inside the 53 of a double); compared on the raw bits (conservative by less than one binade)
        const bool inexact = tmax - tmin >= (25u << 23);
        if (UNLIKELY(__builtin_amdgcn_ballot_w64(inexact) != 0ull)) {
            if (inexact) {
                unsigned v[RO_SLOTS];
#pragma unroll
                for (int k = 0; k < RO_SLOTS; ++k) v[k] = off[k] == my_off ? 0xFFFFu : off[k];   // own entry out; sorts last
                sort8(v);
                double acc2 = 0.0;
                COLD_LOOP
                for (int k = 0; k < RO_SLOTS - 1; ++k) {
                    if (v[k] >= EMPTY) break;
                    const f32x4 o = lds_get<f32x4>(L_LINK + v[k]);
                    const float dx = o.x - rx.x, dy = o.y - rx.y;
                    const float d2 = fmaf(dx, dx, dy * dy);
                    const float g = pair_gain<MODE>(d2, POWLAW ? lds_f2(L_EXPO + (v[k] >> 1)) : make_float2(-1.0f, 0.0f));
                    acc2 += (double)(o.z * g);
                }
                acc = acc2;
            }
        }
    } else {
        // ---- pass 2, general (a workgroup in which some lane met a rarity): exact decode, tuples rewritten, membership masks
        // built behind two more barriers, nested mask walk; a lane whose own rb lies outside [0, R) sweeps all pairs.  The
        // generic kernel's code (d2d_step.hip), cost proportional to the same-RB pairs for any action distribution.
        {
            int rb2, p2;
            decode_link(a, in, row, rb2, p2, 0, true);
            rb = rb2; pw = p2;
            me = make_float4(in.pos.x, in.pos.y, pow10_tenth(pw) * in.rb_.x, __int_as_float(rb));
            s.link[i] = me;
            if (cfg_export_actions) { const unsigned oe = fresh((row + (unsigned)i) * 4u); RO_ST(at(a.rb_out, oe), rb); RO_ST(at(a.pwr_out, oe), pw); }
        }
        clear_masks<true>(s, R, W, tid, TPE);
        __syncthreads();
        const bool in_range = (unsigned)rb < (unsigned)R;
        if (in_range) {
            atomicOr(&s.mask[__umul24((unsigned)(i >> 5), (unsigned)R) + (unsigned)rb], 1u << (i & 31));
            atomicOr(&s.summ[rb], 1u << (i >> 5));
        } else atomicOr(&s.flags[0], FLAG_RB_OOR);
        __syncthreads();
        use_masks = in_range;
        acc = 0.0;
        if (in_range) {
            unsigned live = s.summ[rb];
            const int iw = i >> 5;
            const unsigned self = 1u << (i & 31);
            if (live) {
                int w = __builtin_ctz(live);
                live &= live - 1u;
                const unsigned R4 = (unsigned)R * 4u;
                const unsigned char* mrow_b = reinterpret_cast<const unsigned char*>(s.mask + rb);
                unsigned bits = *reinterpret_cast<const unsigned*>(mrow_b + __umul24((unsigned)w, R4));
                while (true) {
                    const bool more = live != 0u;
                    const int wn = more ? __builtin_ctz(live) : w;
                    const unsigned bits_n = *reinterpret_cast<const unsigned*>(mrow_b + __umul24((unsigned)wn, R4));
                    if (w == iw) bits &= ~self;                      // .difference({action}), simulator.py:95
                    while (bits) {
                        const int j = (w << 5) + __builtin_ctz(bits);
                        bits &= bits - 1u;
                        const float4 o = s.link[j];
                        const float dx = o.x - rx.x, dy = o.y - rx.y;
                        const float d2 = fmaf(dx, dx, dy * dy);
                        const float g = pair_gain<MODE>(d2, POWLAW ? s.expo[j] : make_float2(-1.0f, 0.0f));
                        if (POWLAW) dmin = min(dmin, __float_as_int(d2));
                        acc += (double)(o.z * g);
                    }
                    if (!more) break;
                    live &= live - 1u;
                    w = wn;
                    bits = bits_n;
                }
            }
        } else {
            COLD_LOOP
            for (int j = 0; j < N; ++j) {
                const float4 o = s.link[j];
                const bool same = (__float_as_int(o.w) == rb) & (j != i);
                const float dx = o.x - rx.x, dy = o.y - rx.y;
                const float d2 = fmaf(dx, dx, dy * dy);
                const float g = pair_gain<MODE>(d2, POWLAW ? s.expo[j] : make_float2(-1.0f, 0.0f));
                if (POWLAW) dmin = same ? min(dmin, __float_as_int(d2)) : dmin;
                acc += same ? (double)(o.z * g) : 0.0;
            }
        }
    }

    // ---- SINR / SNR / rate / capacity: the arithmetic of step_kernel, operation for operation
    if (POWLAW) dmin = min(dmin, __float_as_int(d2_own));
    const float sig = me.z * g_own * rx_pl * rx_lin;                 // mW at the receiver, with rx gains
    const float accf = (float)acc;
    // interferers: no rx gains (simulator.py:100); the fma every kernel's `accf * rx_pl + noise` contracted to, spelled out
    const float sinr_lin = precise_div(sig, fmaf(accf, rx_pl, noise));
    const float sinr_db = 3.01029995663981195f * __builtin_amdgcn_logf(sinr_lin);                 // simulator.py:106-107
    const float snr_db = 3.01029995663981195f * __builtin_amdgcn_logf(precise_div(sig, noise));   // simulator.py:115
    const float u1p = 1.0f + sinr_lin, um1 = u1p - 1.0f;
    const float sh_big = __builtin_amdgcn_logf(u1p) * fast_div(sinr_lin, um1 == 0.0f ? 1.0f : um1);
    const float sh = um1 == 0.0f ? sinr_lin * 1.44269504088896340736f : sh_big;
    const bool ok = sinr_db > sens;                                  // simulator.py:123,149
    const float rate = ok ? sh : 0.0f;
    const float cap = ok ? bw_mhz * sh : 0.0f;                       // simulator.py:150-151
    {
        const unsigned o4 = fresh((row + (unsigned)i) * 4u);
        RO_ST(at(a.sinr_db, o4), sinr_db);
        RO_ST(at(a.snr_db, o4), snr_db);
        RO_ST(at(a.rate, o4), rate);
        RO_ST(at(a.cap, o4), cap);
    }
    if (a.write_table) {                                             // obs_fn.py:57-60
        const unsigned o4t = fresh((row + (unsigned)i) * 4u);
        float2* t = reinterpret_cast<float2*>(at(a.table, (o4t << 2) + (o4t << 1)));
        if (NT) {
            f32x2* tv = reinterpret_cast<f32x2*>(t);
            const f32x2 v0 = {me.x, me.y}, v1 = {rx.x, rx.y}, v2 = {sinr_db, snr_db};
            __builtin_nontemporal_store(v0, tv); __builtin_nontemporal_store(v1, tv + 1); __builtin_nontemporal_store(v2, tv + 2);
        } else {
            t[0] = make_float2(me.x, me.y);
            t[1] = rx;
            t[2] = make_float2(sinr_db, snr_db);
        }
    }

    // ---- reward: barrier-free ticket reduction (see step_kernel): DPP wave sum, 32.32 fixed-point capacity total in LDS (the
    // 64-bit integer sum does not depend on arrival order), the wave that draws the last ticket finishes the env
    const int lane = tid & 63;
    const float wsum = wave_sum(cap);
    // per-lane rarities behind ONE wave-uniform branch: SystemCapacity's -1 rule (reward_fn.py:29-41: I am a non-D2D link whose
    // capacity is <= min_capacity and some D2D link shares my RB), a non-finite SINR / zero distance, a capacity too large
    // for the fixed-point sum (64 lanes x 6e7 stays below the 4e9 the accumulator takes per wave: no test of the sum outside)
    const bool rule = type != LINK_SIDELINK && cap <= a.reward_param;
    const bool nonfinite = MODE == PL_INV_SQUARE ? !(fabsf(sinr_db) <= 3.0e38f) : (dmin == 0 || !(fabsf(sinr_db) <= 3.0e38f));
    const bool huge = !(cap <= 6.0e7f);
    bool add_sum = true;                                             // wave-uniform
    if (UNLIKELY(__builtin_amdgcn_ballot_w64(rule | nonfinite | huge) != 0ull)) {
        if (rule) {
            bool hit = false;
            if (!general) {
                const unsigned words[4] = {mlist.x, mlist.y, mlist.z, mlist.w};
#pragma unroll
                for (int k = 0; k < RO_SLOTS; ++k) {
                    const unsigned e = (words[k >> 1] >> ((k & 1) * 16)) & 0xFFFFu, j = e >> 4;
                    if (e != EMPTY && j != (unsigned)i) hit |= ((a.side_words[j >> 5] >> (j & 31u)) & 1u) != 0u;
                }
            } else if (use_masks) {
                COLD_LOOP
                for (int w = 0; w < W; ++w) hit |= (s.mask[(unsigned)w * (unsigned)R + (unsigned)rb] & a.side_words[w]) != 0u;
            } else {
                COLD_LOOP
                for (int k = 0; k < N; ++k)
                    hit |= (k != i) & (((a.rec_a[k].x >> D2D_REC_TYPE_SHIFT) & D2D_REC_TYPE_MASK) == LINK_SIDELINK) &
                           (__float_as_int(s.link[k].w) == rb);
            }
            if (hit) atomicOr(&s.flags[1], 1);
        }
        int my_flags = 0;
        if (MODE == PL_INV_SQUARE) {
            // 1 / d^2 gains: a zero distance (own link: signal inf; an interferer: accumulator inf) always ends in a non-finite SINR
            if (nonfinite) { my_flags |= FLAG_NON_FINITE; if (d2_own == 0.0f || !(accf <= 3.0e38f)) my_flags |= FLAG_ZERO_DISTANCE; }
        } else {
            if (dmin == 0) my_flags |= FLAG_ZERO_DISTANCE;
            if (!(fabsf(sinr_db) <= 3.0e38f)) my_flags |= FLAG_NON_FINITE;
        }
        if (my_flags) atomicOr(&s.flags[0], my_flags);
        // a non-finite (or absurdly large) wave sum cannot go through the fixed-point accumulator: the env's reward is then what
        // a float sum gives - inf, or NaN once a NaN is among the parts
        if (!(wsum <= 4.0e9f)) { add_sum = false; atomicOr(&s.flags[1], wsum != wsum ? 4 : 2); }
    }
    asm volatile("" ::"v"(pf));                                      // the prefetched word is consumed here (no instruction)
    // This wave's LDS atomics above precede its ticket in the LDS queue (in order per wave); the compiler barrier keeps them
    // above it in the instruction stream.  The last wave's reads below stay behind its own ticket (acquire).
    asm volatile("" ::: "memory");
    int ticket = 0;
    if (lane == 0) {
        if (add_sum) __hip_atomic_fetch_add((D2D_LDS(unsigned long long)*)(0u), to_fixed_32_32(wsum), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        asm volatile("" ::: "memory");
        ticket = __hip_atomic_fetch_add((D2D_LDS(int)*)(L_FLAGS + 8u), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    ticket = __builtin_amdgcn_readfirstlane(ticket);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    if (ticket == (TPE >> 6) - 1) {
        // SystemCapacityRewardFunction, reward_fn.py:27-44: mean capacity, or -1 for everyone on a violation
        const unsigned long long tot = atomicAdd(reinterpret_cast<unsigned long long*>(s.red), 0ull);   // LDS read that cannot be hoisted
        const float total = (float)tot * 2.3283064365386963e-10f;
        const int viol = atomicOr(&s.flags[1], 0);
        const float r = (viol & 1) ? -1.0f : ((viol & 4) ? __int_as_float(0x7FC00000) : ((viol & 2) ? __int_as_float(0x7F800000) : total * a.inv_n));
        if (a.reward_env) {                                            // D2D_REWARD_PER_ENV: the scalar once, not N copies
            if (lane == 0) a.reward_env[b] = r;
        } else {
            const f32x4 r4 = {r, r, r, r};
            for (int k = lane * 4; k < N; k += 256) RO_ST(reinterpret_cast<f32x4*>(at(a.reward, fresh((row + (unsigned)k) * 4u))), r4);
        }
        if (lane == 0) a.env_flags[b] = atomicOr(&s.flags[0], 0);
    }
}

hipError_t launch_rollout(const StepArgs& a, PlMode mode, int opt, int block_threads, size_t lds, hipStream_t stream) {
    dim3 grid((unsigned)a.B), block(block_threads);
    hipError_t err = hipSuccess;
#define D2D_RO_1(M, O)                                                                                   \
    do {                                                                                                 \
        if (lds > 48 * 1024)                                                                             \
            err = hipFuncSetAttribute(reinterpret_cast<const void*>(&rollout_kernel<M, O>),              \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);             \
        if (err == hipSuccess) {                                                                         \
            hipLaunchKernelGGL((rollout_kernel<M, O>), grid, block, lds, stream, a);                     \
            err = hipGetLastError();                                                                     \
        }                                                                                                \
    } while (0)
#define D2D_RO(M)                                                                                        \
    switch (opt & (OPT_SREC | OPT_NT)) {                                                                 \
        case 0: D2D_RO_1(M, 0); break;                                                                   \
        case OPT_SREC: D2D_RO_1(M, OPT_SREC); break;                                                     \
        case OPT_NT: D2D_RO_1(M, OPT_NT); break;                                                         \
        default: D2D_RO_1(M, OPT_SREC | OPT_NT); break;                                                  \
    }
    if (mode == PL_INV_SQUARE) { D2D_RO(PL_INV_SQUARE) } else { D2D_RO(PL_POWER) }
#undef D2D_RO
#undef D2D_RO_1
    return err;
}

}  // namespace d2d
