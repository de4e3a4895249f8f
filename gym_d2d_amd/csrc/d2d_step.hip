// Fused per-step kernel for gfx950 (MI355X): action decode -> received signal -> same-RB interference
// reduction -> SINR / SNR / Shannon rate / capacity -> reward -> compact observation table (-> LinearObs expansion
// in the same launch when N is small).
//
// Reference path being replaced (file:line under /root/reference/src/gym_d2d):
//   D2DEnv._decode_action            envs/d2d_env.py:93-101
//   Simulator._calculate_sinrs       simulator.py:89-108      (hot loop #1)
//   Simulator._calculate_snrs        simulator.py:110-116
//   Simulator._calculate_rates       simulator.py:118-127
//   Simulator._calculate_network_capacity  simulator.py:144-154
//   Actions.get_actions_by_rb        actions.py:27-31
//   {SystemCapacity,Shannon,CueSinrShannon}RewardFunction   envs/reward_fn.py:22-78
//   LinearObsFunction._agent_obs / get_state   envs/obs_fn.py:43-61
//   UplinkTrafficModel / DownlinkTrafficModel  traffic_model.py:15-32 (links with fixed (rb, pwr) in their record)
//
// Design (environments are independent): a workgroup owns `epw` environments, `tpe` threads each
// (thread = link); at N = 512 that is one env per 512-thread workgroup, at N = 50 several 64-thread envs share one.
//   * every per-link input is ONE coalesced, independent load: the link's (tx_x, tx_y, rx_x, rx_y) row (built once per
//     reset, launch_link_positions), its action, and three 16-byte rows of the host-built link record (L2-resident; ONE
//     scalar load per row and wave in the rollout kernel when the host found the records uniform within every group of 64
//     links); 10^(p/10) is computed (v_exp_f32 with an exact hi/lo exponent split), so nothing in the prologue is a
//     dependent second hop;
//   * every link's transmitter tuple (tx_x, tx_y, effective tx power in mW, rb | index) is staged in LDS as a float4,
//     so the interference loop is one ds_read_b128 per candidate interferer;
//   * same-RB interferers (Actions.get_actions_by_rb) are found through per-RB membership bitmasks in LDS
//     (R x ceil(N/32) u32 words built with ds_or_b32 - order independent - plus one summary word per RB naming its
//     non-empty words) and walked in ascending link order with ctz: non-empty words outside, members inside, the next
//     word requested before the current one's members are consumed.  Beyond 1024 links per env (the summary word's reach)
//     per-RB member lists take over (slot counter + eight u16 slots per RB, sorted in registers by the receiver).  A masked
//     all-pairs sweep is the fallback (rb outside [0,R), nothing fits in LDS, a list overflows where no masks exist).  All
//     of them visit interferers in ascending link index through the same fmaf, hence produce identical bits.  (Round 5: where
//     RBs are sparsely occupied - N <= 4 R - the rollout configuration has a kernel of its own built on the lists,
//     csrc/d2d_rollout.hip; this file's HOT level 1 is the rollout configuration with the MASKS, for dense occupancy.)
//     Measured and rejected, evidence under profiles/ (NEGATIVE_RESULTS.md): a stable counting sort by RB, a flattened walk;
//   * what paces the kernel is the memory pipeline: bytes and vector-memory requests (dropping the optional decoded
//     (rb, pwr) planes: -2.3 us; record rows by scalar loads: -2 us; one more prefetch load per lane: +1 us), not LDS
//     latency chains, store-instruction counts or the walk's imbalance (profiles/r3_ab_*.jsonl);
//   * all arithmetic is in the LINEAR domain (mW): sinr = S / (I + N) with one 10*log10 at the end; the literal
//     dB-domain transcription cannot hold 1e-5 relative in fp32 (hard parts);
//   * the capacity sum is a DPP wave reduction into the wave's own LDS slot + a fixed-order float sum of the slots (by the wave that
//     draws the last ticket in the barrier-free epilogue of the one-env-per-workgroup kernels, behind one barrier elsewhere): the
//     same bits on every path, run-to-run deterministic;
//   * small envs (N <= 128) expand their LinearObs block inside this launch: the workgroup's envs are one contiguous region,
//     walked in passes of blockDim consecutive float4 with a per-lane (env, row, column) position that is advanced, not
//     recomputed, and the LDS reads of the next pass issued ahead of the store of this one (round 4: 15.7 -> 13.4 us at
//     1024 x 50 on one box, profiles/r4_ab_builds_default_r2head_r3head_r4.jsonl);
//   * what a learner does not read is not written: D2D_OBS_NONE (no table), D2D_REWARD_PER_ENV (SystemCapacity's scalar once per
//     env), d2d_set_export_actions(0) (no decoded rb / pwr planes) take the rollout kernel from 64 + 8 to 36 bytes per link.
#include "d2d_step_device.h"

namespace d2d {

void step_lds_layout(int N, int R, int mask_words, int fuse_obs, int lpt, int reward_fn, int mode, int lists, int xpos, StepLds* out) {
    // lists: one more tuple (and exponent) behind the last link - the far-away, zero-power stand-in an empty list slot reads
    const unsigned NL = (unsigned)N + (lists ? 1u : 0u);
    unsigned off = LDS_HEAD_BYTES + NL * 16u;
    out->aux = off; off += (unsigned)N * 4u;
    out->rx = off; if (lpt == 0) off += (unsigned)N * 8u;
    out->sinr = off; out->sh = off + (unsigned)N * 4u; if (reward_fn >= 2) off += (unsigned)N * 8u;
    off = (off + 7u) & ~7u;
    out->expo = off; if (mode == PL_POWER || mode == PL_SHADOW || mode == PL_POWK) off += NL * 8u;   // (head, tail) of -exponent / 2 per link (PL_POWK: (phi, 0))
    off = (off + 7u) & ~7u;
    out->lo = off; if (xpos) off += NL * 8u;                                         // low parts of (tx_x, tx_y) per link (exact positions)
    out->tflat = off; if (fuse_obs) off += (unsigned)N * 24u;
    off = (off + 15u) & ~15u;                            // the mask region is cleared with 16-byte stores
    out->mask = off;
    if (mask_words > 0) off += ((unsigned)R * mask_words + mask_words + (unsigned)R) * 4u;
    off = (off + 15u) & ~15u;
    // member lists: slots[R] (eight u16 link indices per RB, 0xFFFF = empty) then cnt[R] u32 (padded to 16 bytes) -
    // one contiguous region, (re)initialised with one 16-byte store per lane
    out->lists = off;
    if (lists) off += (unsigned)R * 16u + (((unsigned)R + 3u) & ~3u) * 4u;
    out->env_bytes = (off + 15u) & ~15u;
}

size_t step_lds_bytes_per_env(int N, int R, int mask_words, int fuse_obs, int lpt, int reward_fn, int mode, int lists, int xpos) {
    StepLds l;
    step_lds_layout(N, R, mask_words, fuse_obs, lpt, reward_fn, mode, lists, xpos, &l);
    return l.env_bytes;
}

// LPT  = links per thread held in registers: 1 (thread = link, N <= tpe), 2 (links lt and lt + tpe: half the waves per
//        env, two independent dependency chains per wave), 0 = strided (N > 2 * 1024: records re-read per link).
// FULL = LPT > 0, one env per workgroup and N == LPT * blockDim: every lane owns exactly LPT links of an existing env, so
//        no lane predication (exec-mask save / restore on the scalar pipe) is generated anywhere outside the walk.
#define FOR_MY_LINKS(u, i)                                                                  \
    _Pragma("unroll UNROLL_LINKS") for (int u = 0; u < (LPT > 0 ? LPT : 0x7fffffff); ++u)  \
        if (const int i = lt + u * TPE; !(FULL || (active && i < N))) { if (LPT == 0) break; } else
#define KEPT(u) (LPT > 0 ? (u) : 0)                /* register slot of link u (strided kernels keep only their first) */
#define IN_REGS(u) (LPT > 0 || (u) == 0)

// result stores of the rollout kernel: nontemporal when nothing re-reads them from L2 right behind this launch (OPT_NT)
#define ST(ptr, val) do { if (NT) __builtin_nontemporal_store((val), (ptr)); else *(ptr) = (val); } while (0)
#ifndef D2D_STEP_ABLATE
#define D2D_STEP_ABLATE 0       /* diagnostic build: honour StepArgs::ablate (parts of the kernel skipped; round 2 - 4 studies) */
#endif
#define ABL(bit) (D2D_STEP_ABLATE && (a.ablate & (bit)))
// diagnostic builds: lane 0 of every wave stamps the shader clock at the phase boundaries (tools/phase_times.py)
#if D2D_STEP_ABLATE
#define STAMP(k) do { if (a.dbg && (threadIdx.x & 63) == 0)                                                              \
        a.dbg[((size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * 16 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define STAMP(k) do { } while (0)
#endif

// HOT: the configuration a rollout runs in, with its uniform choices folded at compile time.  Level 1 - raw agent actions for
// every link (column = link index, no fixed links), SystemCapacity reward, compact table written, decoded (rb, pwr) exported,
// nested walk, action prefetch on.  Level 2 - the same with a fixed prefix allowed (traffic-model CUEs) and the 16-byte fused
// LinearObs expansion, for several small envs per workgroup.  Twelve scalar compare-and-branch pairs and the code behind
// their other arms leave the instruction stream (the kernel's SQ_WAIT_INST_ANY - waves waiting to be issued - is a quarter of its wave cycles).
// LISTS: same-RB interferers come from per-RB member lists instead of the membership masks (StepArgs::walk == 2).  Pass 1 draws
// a slot from the RB's counter (ds_add_rtn_u32) and writes its link index there (ds_write_b16); a receiver reads its RB's
// eight slots with ONE ds_read_b128, takes its own entry out, sorts the rest in registers (ascending link index: the same
// accumulation order as the mask walk and the all-pairs sweep, hence the same bits) and then knows every interferer at once -
// three dependent LDS round trips (list, tuples) instead of the mask walk's summary -> word -> tuple chain per word, and 5 KB
// to initialise per env instead of 17 KB of masks.  An env in which some RB drew a ninth link (6 % of envs under uniformly
// random actions at 512 links on 256 RBs) raises a workgroup flag; its workgroup then clears and builds the masks behind two
// extra barriers and walks them, so the cost stays proportional to the same-RB pairs for any action distribution.
template <int MODE, int LPT, bool FULL, int HOT = 0, int OPT = 0>
__global__ __launch_bounds__(1024) void step_kernel(const StepArgs a) {
    constexpr int KEEP = LPT > 0 ? LPT : 1;
    constexpr int UNROLL_LINKS = LPT > 0 ? LPT : 1;                // strided kernels: the per-link bodies are not unrolled
    constexpr bool LISTS = (OPT & OPT_LISTS) != 0, SREC = (OPT & OPT_SREC) != 0, NT = (OPT & OPT_NT) != 0;
    constexpr bool XPOS = (OPT & OPT_XPOS) != 0;                   // float64 positions as (hi, lo) pairs: coord_diff (d2d_step_device.h)
    static_assert(!XPOS || HOT == 0, "the specialised kernels serve float32 positions (device-side resets)");
    constexpr bool POWLAW = MODE == PL_POWER || MODE == PL_SHADOW || MODE == PL_POWK;   // per-link exponents (rec_h, LDS expo[])
    // a zero distance shows as a non-finite gain (1 / d^2, and PL_POWK's reciprocal): no smallest-distance tracking per pair
    constexpr bool NF_ONLY = MODE == PL_INV_SQUARE || MODE == PL_POWK;
    const int cfg_action_mode = HOT ? 0 : a.action_mode;
    const int cfg_col_mode = HOT ? 0 : a.col_mode;
    const int cfg_reward_fn = HOT ? 1 : a.reward_fn;
    const int cfg_write_table = HOT == 2 ? 1 : a.write_table;     // HOT 1 serves D2D_OBS_TABLE and D2D_OBS_NONE (one uniform branch around three stores)
    // (the flattened walk is an A/B shape of the power-law kernels: the table / shadowing kernels, whose pair evaluation is
    // hundreds of instructions, carry the nested one only)
    const int cfg_walk = (HOT || LISTS || MODE == PL_SHADOW || MODE == PL_TABLE) ? 0 : a.walk;
    const bool cfg_export_actions = a.rb_out != nullptr;          // the info dict's rb / tx_pwr_dbm (d2d_set_export_actions)
    extern __shared__ __align__(16) unsigned char smem_raw[];
    const int N = a.N, R = a.R, D = a.D, W = a.mask_words, TPE = a.tpe;
    const int tid = threadIdx.x;
    // env slot in this workgroup (wave-uniform: TPE % 64 == 0); tid / TPE by multiply-shift, exact for tid < 1024
    const int e = FULL ? 0 : (int)(((unsigned)tid * a.tpe_magic) >> 20);
    const int lt = FULL ? tid : tid - e * TPE;
    const int b = FULL ? (int)blockIdx.x : (int)blockIdx.x * a.epw + e;
    const bool active = FULL || (e < a.epw && b < a.B);
    // element offsets fit 32 bits (the host refuses B * N * 6 >= 2^31): one VGPR offset + SGPR base per access
    const unsigned row = (unsigned)b * (unsigned)N;
    const unsigned act_row = (unsigned)b * (unsigned)a.act_stride;
    Smem s = carve(smem_raw + (FULL ? 0u : (unsigned)(e < a.epw ? e : 0) * a.lds.env_bytes), a.lds, R, W);
    int* wg_flags = reinterpret_cast<int*>(smem_raw + 64);       // flags[] of env slot 0: [3] = some RB of some env of this workgroup overflowed its list
    STAMP(0);
    if (ABL(1024)) {                      // diagnostic: workgroup launch only
        if (tid == 4095) a.env_flags[b] = 1;
        return;
    }

    // ---- prologue: issue this thread's link's loads BEFORE any LDS work or barrier, so their latency overlaps pass 0.
    // Inactive lanes (lt >= N, spare env slots) load a clamped duplicate instead of branching around the loads: a
    // join after conditional loads makes the compiler wait for all of them right here.
    const unsigned b_ld = ABL(128) ? 0u : (unsigned)(active ? b : a.B - 1);
    LinkRaw first[KEEP];
#pragma unroll
    for (int u = 0; u < KEEP; ++u) {
        const int i = lt + u * TPE;
        first[u] = load_link(a, b_ld * (unsigned)N, b_ld * (unsigned)a.act_stride, FULL || i < N ? i : N - 1, cfg_action_mode, cfg_col_mode, HOT == 1, SREC, POWLAW, XPOS);
    }
    // ---- pass 0: clear masks and flags
    const bool have_masks = HOT || W > 0;                          // mask region allocated
    const bool want_masks = !LISTS && (HOT || (W > 0 && !ABL(4)));  // masks built in pass 1 (LISTS: only by an overflowing env, later)
    if (active) {
        if (LISTS) {
            // slots[R] <- 0xFFFF.., cnt[R] <- 0: one contiguous region, one 16-byte store per lane and round
            const int nl = R + ((R + 3) >> 2);
#pragma unroll 1
            for (int k = lt; k < nl; k += TPE) { const unsigned f = k < R ? 0xFFFFFFFFu : 0u; s.slots[k] = make_uint4(f, f, f, f); }
            if (lt == TPE - 1) {
                s.link[N] = make_float4(1.0e18f, 1.0e18f, 0.0f, __int_as_float(-1));
                if (POWLAW) s.expo[N] = make_float2(-1.0f, 0.0f);
                if (XPOS) s.lo[N] = make_float2(0.0f, 0.0f);
            }
        } else if (want_masks) clear_masks<FULL>(s, R, W, lt, TPE);
        if (lt < 5) reinterpret_cast<uint4*>(s.red)[lt] = make_uint4(0u, 0u, 0u, 0u);   // red[16] + flags[4]: 80 bytes
    }
    // nothing that consumes a loaded value may be scheduled above this barrier: the wave would sit on the HBM round trip
    // before pass 0 instead of behind it
    __builtin_amdgcn_sched_barrier(0);
    STAMP(1);
    // Envs of one wave each (HOT level 2 at N <= 64: BASELINE config 2): what passes 0 - 2 share through LDS belongs to ONE wave,
    // whose LDS operations complete in issue order - no workgroup barrier is needed until the reward pass publishes the tables
    // for the expansion (two of the kernel's three barriers; 0.35 us of a 13.4 us step, r4_phase_times_default_fused_kernel.json)
    const bool solo = HOT == 2 && TPE == 64;
#define PASS_BARRIER() do { if (solo) { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); } else __syncthreads(); } while (0)
    if (!ABL(64)) PASS_BARRIER();
    STAMP(2);
    __builtin_amdgcn_sched_barrier(0);
    if (ABL(256)) {                       // diagnostic: launch + prologue loads + pass 0 only
        if (first[0].act0 == 0x7fffffff && first[KEEP - 1].pos.x == 1.2345f && first[0].rb_.x == first[0].rc.x && first[0].ra.y == 77) a.env_flags[b] = 1;
        return;
    }

    // ---- pass 1: decode + stage the transmitter side of every link
    float4 me0[KEEP];
    unsigned myslot[KEEP];                                               // LISTS: where this link sits in its RB's list
#pragma unroll
    for (int u = 0; u < KEEP; ++u) { me0[u] = make_float4(0.f, 0.f, 0.f, 0.f); myslot[u] = 0u; }
    FOR_MY_LINKS(u, i) {
        const LinkRaw in = IN_REGS(u) ? first[KEPT(u)] : load_link(a, row, act_row, i, cfg_action_mode, cfg_col_mode, HOT == 1, false, POWLAW, XPOS);
        int rb, p;
        decode_link(a, in, act_row, rb, p, cfg_action_mode, HOT == 1);
        const int type = (in.ra.x >> D2D_REC_TYPE_SHIFT) & D2D_REC_TYPE_MASK;
        const float4 tuple = make_float4(in.pos.x, in.pos.y, pow10_tenth(p) * in.rb_.x, __int_as_float(rb));
        s.link[i] = tuple;
        if (LPT == 0) s.rx[i] = make_float2(in.pos.z, in.pos.w);
        // tx_dev | link_type << 24 (HOT: only the cold all-pairs route wants the type, and reads it from the record); the low bits
        // are the transmitter's ROW in the gain table: its device, or - a table by (tx link, rx link) - the link itself
        if (!HOT) s.aux[i] = MODE == PL_TABLE && a.table_by_link ? (i | (in.ra.x & 0x0F000000)) : (in.ra.x & 0x0FFFFFFF);
        if (IN_REGS(u)) me0[KEPT(u)] = tuple;                            // own links stay in registers for pass 2
        if (POWLAW) s.expo[i] = in.hh;
        if (XPOS) s.lo[i] = make_float2(in.plo.x, in.plo.y);
        if (cfg_export_actions && !ABL(32)) { const unsigned oe = fresh((row + (unsigned)i) * 4u); ST(at(a.rb_out, oe), rb); ST(at(a.pwr_out, oe), p); }
        if (LISTS) {
            if (LIKELY((unsigned)rb < (unsigned)R)) {
                const unsigned slot = atomicAdd(&s.cnt[rb], 1u);                         // ds_add_rtn_u32
                if (LIKELY(slot < (unsigned)LIST_SLOTS)) reinterpret_cast<unsigned short*>(s.slots)[(unsigned)rb * LIST_SLOTS + slot] = (unsigned short)i;
                else atomicOr(&wg_flags[3], 1);
                if (IN_REGS(u)) myslot[KEPT(u)] = slot;
            }
            else atomicOr(&s.flags[0], FLAG_RB_OOR);
            if (cfg_reward_fn == 3 && have_masks && (i & 31) == 0) s.side[i >> 5] = a.side_words[i >> 5];
        }
        if (want_masks && !ABL(2)) {
            const unsigned bit = 1u << (i & 31);
            if (LIKELY((unsigned)rb < (unsigned)R)) {
                atomicOr(&s.mask[__umul24((unsigned)(i >> 5), (unsigned)R) + (unsigned)rb], bit);
                atomicOr(&s.summ[rb], 1u << (i >> 5));
            }
            else atomicOr(&s.flags[0], FLAG_RB_OOR);
            // sidelink membership words are a property of the link list: built on the host (StepArgs::side_words).  Only
            // CueSinrShannon reads them in its hot loop and wants them in LDS; SystemCapacity's -1 rule reads the global copy
            if (cfg_reward_fn == 3 && (i & 31) == 0) s.side[i >> 5] = a.side_words[i >> 5];
        } else if (!LISTS && UNLIKELY((unsigned)rb >= (unsigned)R)) atomicOr(&s.flags[0], FLAG_RB_OOR);   // all-pairs sweep: flagged all the same
    }
    STAMP(3);
    if (!ABL(64)) PASS_BARRIER();
#undef PASS_BARRIER
    STAMP(4);
    // LISTS: did any RB of any env of this workgroup draw a ninth link?  Then every env of the workgroup builds its masks now
    // (two more barriers, workgroup-uniform) and walks them instead.
    bool lists_on = false, masks_late = false;
    if (LISTS) {
        const bool ovf = __builtin_amdgcn_readfirstlane(wg_flags[3]) != 0;
        lists_on = !ovf && active;
        if (UNLIKELY(ovf) && have_masks) {
            if (active) clear_masks<FULL>(s, R, W, lt, TPE);
            __syncthreads();
            FOR_MY_LINKS(u, i) {
                const int rb = IN_REGS(u) ? __float_as_int(me0[KEPT(u)].w) : __float_as_int(s.link[i].w);
                if ((unsigned)rb < (unsigned)R) {
                    atomicOr(&s.mask[__umul24((unsigned)(i >> 5), (unsigned)R) + (unsigned)rb], 1u << (i & 31));
                    atomicOr(&s.summ[rb], 1u << (i >> 5));
                }
            }
            __syncthreads();
            masks_late = true;
        }
    }
    const bool masks_on = (LISTS ? masks_late : want_masks) && active;
    const bool skip_walk = ABL(7);
    if (ABL(512)) {                       // diagnostic: everything up to the end of pass 1
        if (me0[0].z == 1.2345f) a.env_flags[b] = 1;
        return;
    }

    // Software prefetch, issued once this env's own loads have landed (so no wait for them can catch it): touch the
    // action row of the env that the workgroup `prefetch_envs` later will own - a multiple of 8, so it lands in the L2 of
    // the XCD that will read it (round-robin dispatch).  Fresh actions are the one input that is never cache resident, and
    // with four 512-thread workgroups per CU an HBM miss at the head of every workgroup is exposed four rounds deep
    // (+7 us at 4096 x 512: profiles/r2_action_prefetch.txt).  The value is consumed by an empty asm at the very end.
    int pf = 0;
    if (HOT || (a.prefetch_envs > 0 && cfg_action_mode == 0 && a.act_stride > 0)) {
        const int bq = b + a.prefetch_envs;
        const int bp = bq < a.B ? bq : a.B - 1;                         // clamped, not branched around (a join would wait)
#pragma unroll
        for (int u = 0; u < KEEP; ++u) {
            const int i = lt + u * TPE;
            const int col = HOT == 1 || i < a.act_stride ? i : a.act_stride - 1;
            pf ^= *at(a.actions, fresh(((unsigned)bp * (unsigned)a.act_stride + (unsigned)col) * 4u));
        }
    }

    const float* gtab = MODE == PL_TABLE ? a.gain_table + (size_t)b * a.table_env_stride : nullptr;
    const unsigned genv = (unsigned)(a.env_offset + (unsigned long long)b);   // global env index (RNG counter)

    // ---- pass 2: interference reduction + SINR/SNR/rate/capacity + obs table
    float cap_part = 0.0f;
    int my_flags = 0;
    bool violated = false;
    FOR_MY_LINKS(u, i) {
        const LinkRaw in = IN_REGS(u) ? first[KEPT(u)] : load_link(a, row, act_row, i, cfg_action_mode, cfg_col_mode, HOT == 1, false, POWLAW, XPOS);   // strided links: records re-read (L2)
        const float4 me = IN_REGS(u) ? me0[KEPT(u)] : s.link[i];
        const float2 rx = make_float2(in.pos.z, in.pos.w);
        const float2 rxlo = make_float2(in.plo.z, in.plo.w);               // zero unless XPOS
        // tx (tuple o of link j) - rx, per coordinate: position.py:11-12
#define D2D_DXY(o, j)                                                                                                   \
        float dx, dy;                                                                                                   \
        if (XPOS) { const float2 l_ = s.lo[j]; dx = coord_diff((o).x, rx.x, l_.x, rxlo.x); dy = coord_diff((o).y, rx.y, l_.y, rxlo.y); } \
        else { dx = (o).x - rx.x; dy = (o).y - rx.y; }
        const int rb = __float_as_int(me.w);
        // a link whose own rb lies outside [0, R) has no mask row: it (and only it) takes the all-pairs sweep - links
        // with an in-range rb never share it with such a link, so their mask walk is complete
        const bool use_masks = masks_on && (unsigned)rb < (unsigned)R;
        const int type = (in.ra.x >> D2D_REC_TYPE_SHIFT) & D2D_REC_TYPE_MASK;
        const int txd = in.ra.x & D2D_REC_TXDEV_MASK, rxd = in.ra.y;
        const int tcol = MODE == PL_TABLE && a.table_by_link ? i : rxd;      // column of this receiver in the gain table
        const float rx_pl = in.rb_.y, rx_lin = in.rb_.z, noise = in.rb_.w;
        const float sens = in.rc.x, bw_mhz = in.rc.y;
        // the interference sum in DOUBLE: each term is one rounded float product, but the running sum no longer drifts with the
        // number of same-RB links (float: 4.4e-6 of the bar's 1e-5 at 512 links on one RB, 8.5e-6 at 2048, tools/probes/crowded_rb_error.py)
        double acc = 0.0;
        int dmin = 0x7F000000;                                           // bits of the smallest squared distance met (d2 >= 0:
                                                                         // integer order == float order); 0 <=> 'math domain error'
        const bool use_lists = LISTS && IN_REGS(u) && lists_on && (unsigned)rb < (unsigned)R;
        if (skip_walk) {
        } else if (LISTS && LIKELY(use_lists)) {
            // the RB's eight slots in one read; this link's own entry out (.difference({action}), simulator.py:95) - its slot
            // came back from the counter in pass 1; the others ascending
            uint4 m = s.slots[rb];
            const unsigned sl = myslot[KEPT(u)];
            const unsigned own = LIST_EMPTY << ((sl & 1u) << 4), q = sl >> 1;
            m.x |= q == 0u ? own : 0u; m.y |= q == 1u ? own : 0u; m.z |= q == 2u ? own : 0u; m.w |= q == 3u ? own : 0u;
            unsigned v[8] = {m.x & 0xFFFFu, m.x >> 16, m.y & 0xFFFFu, m.y >> 16, m.z & 0xFFFFu, m.z >> 16, m.w & 0xFFFFu, m.w >> 16};
            sort8(v);
#define D2D_LIST_PAIR(o, j)                                                                                             \
            {                                                                                                           \
                D2D_DXY(o, j)                                                                                           \
                const float d2 = fmaf(dx, dx, dy * dy);                                                                 \
                float g;                                                                                                \
                if (MODE == PL_TABLE) g = gtab[(size_t)(s.aux[j] & 0xFFFFFF) * a.table_pitch + tcol];                                \
                else { g = pair_gain<MODE>(d2, POWLAW ? s.expo[j] : make_float2(-1.0f, 0.0f), a.pow_k); if (!NF_ONLY) dmin = min(dmin, __float_as_int(d2)); } \
                if (MODE == PL_SHADOW && d2 > a.shadow_d0sq) g *= shadow_factor(a, genv, (int)(j), i, 0u);              \
                acc += (double)((o).z * g);                              /* simulator.py:97-101, linear mW */            \
            }
            if (MODE == PL_INV_SQUARE || MODE == PL_POWER || MODE == PL_POWK) {
                // Empty slots are clamped onto the stand-in tuple at link[N] (zero power, 1e18 m away): fmaf(0, g, acc) leaves acc
                // untouched, so the steps need no predication and their tuple reads go out together - four, then two, then one,
                // each batch only if some lane of the wave still has a member (sorted: a lane's members are a prefix).
                unsigned jj[LIST_SLOTS - 1];
#pragma unroll
                for (int k = 0; k < LIST_SLOTS - 1; ++k) jj[k] = min(v[k], (unsigned)N);
                {
                    const float4 o0 = s.link[jj[0]], o1 = s.link[jj[1]], o2 = s.link[jj[2]], o3 = s.link[jj[3]];
                    D2D_LIST_PAIR(o0, jj[0]) D2D_LIST_PAIR(o1, jj[1]) D2D_LIST_PAIR(o2, jj[2]) D2D_LIST_PAIR(o3, jj[3])
                    asm volatile("" ::"v"(o0.w), "v"(o1.w), "v"(o2.w), "v"(o3.w));       // .w kept live: ds_read_b128 (4 LDS cycles), not b96 (8)
                }
                if (__builtin_amdgcn_ballot_w64(v[4] != LIST_EMPTY) != 0ull) {
                    const float4 o4 = s.link[jj[4]], o5 = s.link[jj[5]];
                    D2D_LIST_PAIR(o4, jj[4]) D2D_LIST_PAIR(o5, jj[5])
                    asm volatile("" ::"v"(o4.w), "v"(o5.w));
                    if (__builtin_amdgcn_ballot_w64(v[6] != LIST_EMPTY) != 0ull) {
                        const float4 o6 = s.link[jj[6]];
                        D2D_LIST_PAIR(o6, jj[6])
                        asm volatile("" ::"v"(o6.w));
                    }
                }
            } else {
#pragma unroll
                for (int k = 0; k < LIST_SLOTS - 1; ++k) {
                    const unsigned j = v[k];
                    if (j == LIST_EMPTY) break;
                    const float4 o = s.link[j];
                    D2D_LIST_PAIR(o, j)
                }
            }
#undef D2D_LIST_PAIR
        } else if (LIKELY(use_masks)) {
            unsigned live = s.summ[rb];                                  // non-empty words of this RB, ascending
            const int iw = i >> 5;
            const unsigned self = 1u << (i & 31);
            const unsigned* mrow = s.mask + rb;                          // word w of this RB: mrow[w * R]
            if (cfg_walk == 1) {
                // Flattened walk: a lane either fetches its next non-empty word or consumes one member; one loop
                unsigned bits = 0u;
                int jbase = 0;
                while (true) {
                    if (bits == 0u) {
                        if (live == 0u) break;
                        const int w = __builtin_ctz(live);
                        live &= live - 1u;
                        bits = mrow[__umul24((unsigned)w, (unsigned)R)];
                        if (w == iw) bits &= ~self;                      // .difference({action}), simulator.py:95
                        jbase = w << 5;
                    }
                    if (bits != 0u) {
                        const int j = jbase + __builtin_ctz(bits);
                        bits &= bits - 1u;
                        const float4 o = s.link[j];
                        D2D_DXY(o, j)
                        const float d2 = fmaf(dx, dx, dy * dy);
                        float g;
                        if (MODE == PL_TABLE) g = gtab[(size_t)(s.aux[j] & 0xFFFFFF) * a.table_pitch + tcol];
                        else { g = pair_gain<MODE>(d2, POWLAW ? s.expo[j] : make_float2(-1.0f, 0.0f), a.pow_k); if (!NF_ONLY) dmin = min(dmin, __float_as_int(d2)); }
                        if (MODE == PL_SHADOW && d2 > a.shadow_d0sq) g *= shadow_factor(a, genv, j, i, 0u);
                        acc += (double)(o.z * g);                        // simulator.py:97-101, linear mW
                    }
                }
            } else {
                // Nested walk: non-empty words (summary bits) outside, members of the word inside.  The NEXT word is
                // requested before the members of the current one are consumed, so its LDS round trip overlaps theirs.
                if (live) {
                    int w = __builtin_ctz(live);
                    live &= live - 1u;
                    const unsigned R4 = (unsigned)R * 4u;               // byte pitch of a mask word row: address = one v_mad_u32_u24
                    const unsigned char* mrow_b = reinterpret_cast<const unsigned char*>(mrow);
                    unsigned bits = *reinterpret_cast<const unsigned*>(mrow_b + __umul24((unsigned)w, R4));
                    while (true) {
                        const bool more = live != 0u;
                        const int wn = more ? __builtin_ctz(live) : w;
                        const unsigned bits_n = *reinterpret_cast<const unsigned*>(mrow_b + __umul24((unsigned)wn, R4));   // prefetch (a re-read when !more)
                        if (w == iw) bits &= ~self;                      // .difference({action}), simulator.py:95
                        while (bits) {
                            const int j = (w << 5) + __builtin_ctz(bits);
                            bits &= bits - 1u;
                            const float4 o = s.link[j];
                            D2D_DXY(o, j)
                            const float d2 = fmaf(dx, dx, dy * dy);
                            float g;
                            if (MODE == PL_TABLE) g = gtab[(size_t)(s.aux[j] & 0xFFFFFF) * a.table_pitch + tcol];
                            else { g = pair_gain<MODE>(d2, POWLAW ? s.expo[j] : make_float2(-1.0f, 0.0f), a.pow_k); if (!NF_ONLY) dmin = min(dmin, __float_as_int(d2)); }
                            if (MODE == PL_SHADOW && d2 > a.shadow_d0sq) g *= shadow_factor(a, genv, j, i, 0u);
                            acc += (double)(o.z * g);                    // simulator.py:97-101, linear mW
                            asm volatile("" ::"v"(o.w));                 // .w kept live: the tuple comes by ds_read_b128 (4 LDS cycles), not b96 (8)
                        }
                        if (!more) break;
                        live &= live - 1u;
                        w = wn;
                        bits = bits_n;
                    }
                }
            }
        } else {
#pragma unroll 4
            for (int j = 0; j < N; ++j) {
                const float4 o = s.link[j];                              // same address in every lane: LDS broadcast
                const bool same = (__float_as_int(o.w) == rb) & (j != i);
                D2D_DXY(o, j)
                const float d2 = fmaf(dx, dx, dy * dy);
                float g;
                if (MODE == PL_TABLE) g = same ? gtab[(size_t)(s.aux[j] & 0xFFFFFF) * a.table_pitch + tcol] : 0.0f;
                else { g = pair_gain<MODE>(d2, POWLAW ? s.expo[j] : make_float2(-1.0f, 0.0f), a.pow_k); if (!NF_ONLY) dmin = same ? min(dmin, __float_as_int(d2)) : dmin; }
                if (MODE == PL_SHADOW && same && d2 > a.shadow_d0sq) g *= shadow_factor(a, genv, j, i, 0u);
                acc += same ? (double)(o.z * g) : 0.0;
            }
        }

        if (D2D_STEP_ABLATE && a.dbg && acc == -1.0) my_flags |= 1 << 30;   // (never true) pins the stamp behind the walk
        STAMP(5);
        if (ABL(16384)) {                    // diagnostic: 64 extra VALU instructions per wave - is the kernel VALU bound?
            float x0 = me.x, x1 = me.y;
#pragma unroll
            for (int k = 0; k < 32; ++k) { x0 = fmaf(x0, 1.0001f, 0.5f); x1 = fmaf(x1, 0.9999f, 0.25f); }
            if (x0 + x1 == 1.2345f) my_flags |= 1 << 29;
        }
        if (ABL(524288)) {                   // diagnostic: 64 extra full-rate VALU instructions per wave that cannot be packed (the FMA
                                             // chains above compile to 32 dependent v_pk_fma_f32): four independent integer chains
            unsigned y0 = (unsigned)i, y1 = (unsigned)rb, y2 = y0 + 7u, y3 = y1 + 9u;
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                asm volatile("v_xor_b32 %0, %0, %1" : "+v"(y0) : "v"(y1));
                asm volatile("v_add_u32 %0, %0, %1" : "+v"(y1) : "v"(y2));
                asm volatile("v_xor_b32 %0, %0, %1" : "+v"(y2) : "v"(y3));
                asm volatile("v_add_u32 %0, %0, %1" : "+v"(y3) : "v"(y0));
            }
            if ((y0 ^ y1 ^ y2 ^ y3) == 0x12345679u) my_flags |= 1 << 26;
        }
        if (ABL(32768)) {                    // diagnostic: 64 extra SALU instructions per wave (wave-uniform integer chain)
            int sx = __builtin_amdgcn_readfirstlane(rb);
#pragma unroll
            for (int k = 0; k < 32; ++k) { sx = sx * 3 + k; sx ^= sx >> 3; asm volatile("" : "+s"(sx)); }
            if (sx == 0x12345678) my_flags |= 1 << 28;
        }
        if (ABL(65536)) {                    // diagnostic: 8 extra random 16-byte LDS reads per lane (the walk's tuple reads)
            float lx = 0.0f;
            unsigned jj = (unsigned)i;
#pragma unroll
            for (int k = 0; k < 8; ++k) { jj = (jj * 73u + 19u) & (unsigned)(N - 1); const float4 o = s.link[jj]; lx += o.x; }
            if (lx == 1.2345f) my_flags |= 1 << 27;
        }
        if (ABL(131072)) {                   // diagnostic: 8 extra LDS atomics per lane on random mask words
            unsigned jj = (unsigned)i;
#pragma unroll
            for (int k = 0; k < 8; ++k) { jj = (jj * 73u + 19u) & 1023u; atomicOr(&s.mask[jj], 0u); }
        }
        if (ABL(262144)) {                   // diagnostic: 4 extra 4-byte global stores per lane (into this link's own result slots)
            const unsigned o4x = fresh((row + (unsigned)i) * 4u);
            *at(a.rate, o4x) = 0.0f; *at(a.cap, o4x) = 0.0f; *at(a.snr_db, o4x) = 0.0f; *at(a.sinr_db, o4x) = 0.0f;
        }
#undef D2D_DXY
        // own link: simulator.py:93
        const float dx = XPOS ? coord_diff(me.x, rx.x, in.plo.x, rxlo.x) : me.x - rx.x;
        const float dy = XPOS ? coord_diff(me.y, rx.y, in.plo.y, rxlo.y) : me.y - rx.y;
        const float d2 = fmaf(dx, dx, dy * dy);
        float g;
        if (MODE == PL_TABLE) g = gtab[(size_t)(a.table_by_link ? i : txd) * a.table_pitch + tcol];
        else { g = pair_gain<MODE>(d2, in.hh, a.pow_k); if (!NF_ONLY) dmin = min(dmin, __float_as_int(d2)); }
        float sig = me.z * g * rx_pl * rx_lin;                           // mW at the receiver, with rx gains
        float sig_snr = sig;
        if (MODE == PL_SHADOW && d2 > a.shadow_d0sq) {
            sig_snr = sig * shadow_factor(a, genv, i, i, 1u);            // simulator.py:114: a second, independent draw
            sig *= shadow_factor(a, genv, i, i, 0u);                     // simulator.py:93
        }
        const float accf = (float)acc;
        // interferers: no rx gains (simulator.py:100); the fma every kernel's `accf * rx_pl + noise` contracted to, spelled out
        const float sinr_lin = precise_div(sig, fmaf(accf, rx_pl, noise));
        // dB = 10 log10 x = 3.0103 log2 x, log2 on the transcendental unit (v_log_f32, 1 ulp): abs error < 6e-6 dB
        // at 80 dB and < 1e-6 dB near 0 dB, inside the 1e-5 * max(|ref|, 1) bar with an order of magnitude to spare
        const float sinr_db = 3.01029995663981195f * __builtin_amdgcn_logf(sinr_lin);         // simulator.py:106-107
        const float snr_db = 3.01029995663981195f * __builtin_amdgcn_logf(precise_div(sig_snr, noise));   // simulator.py:115
        // log2(1 + x) without losing small x: log2(u) * x / (u - 1), u = fl(1 + x)
        const float u1p = 1.0f + sinr_lin, um1 = u1p - 1.0f;
        // both arms unconditional (the divisor is made harmless first), so the choice is a v_cndmask, not a divergent branch
        const float sh_big = __builtin_amdgcn_logf(u1p) * fast_div(sinr_lin, um1 == 0.0f ? 1.0f : um1);
        const float sh = um1 == 0.0f ? sinr_lin * 1.44269504088896340736f : sh_big;
        const bool ok = sinr_db > sens;                                  // simulator.py:123,149
        const float rate = ok ? sh : 0.0f;
        const float cap = ok ? bw_mhz * sh : 0.0f;                       // simulator.py:150-151

        if (!ABL(8)) {
            const unsigned o4 = fresh((row + (unsigned)i) * 4u);
            ST(at(a.sinr_db, o4), sinr_db);
            ST(at(a.snr_db, o4), snr_db);
            ST(at(a.rate, o4), rate);
            ST(at(a.cap, o4), cap);
        }
        if (cfg_write_table && !ABL(16)) {                                 // obs_fn.py:57-60
            const unsigned o4t = fresh((row + (unsigned)i) * 4u);
            float2* t = reinterpret_cast<float2*>(at(a.table, (o4t << 2) + (o4t << 1)));      // 24 bytes per link, no v_mul_lo
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
        if (!FULL && (HOT == 2 || a.fuse_obs)) {
            float2* t = reinterpret_cast<float2*>(s.tflat + 6 * i);
            t[0] = make_float2(me.x, me.y);
            t[1] = rx;
            t[2] = make_float2(sinr_db, snr_db);
        }
        // staged only for the reward pass that reads them (reward_fn.py): 2 -> own sinr, sh; 3 -> sinr, sh
        if (cfg_reward_fn >= 2) { s.sinr[i] = sinr_db; s.sh[i] = sh; }
        if (cfg_reward_fn == 1) {
            // SystemCapacityRewardFunction's -1 rule (reward_fn.py:29-41), from this link's side: I am a non-D2D
            // link whose capacity is <= min_capacity and some D2D link shares my RB.  Masks / tuples of ALL links
            // were published by the barrier before this pass, so no further synchronisation is needed here.
            if (UNLIKELY(type != LINK_SIDELINK && cap <= a.reward_param)) {
                bool hit = false;
                if (LISTS && use_lists) {
                    const unsigned short* mem = reinterpret_cast<const unsigned short*>(s.slots) + (unsigned)rb * LIST_SLOTS;
                    for (int k = 0; k < LIST_SLOTS; ++k) {
                        const unsigned j = mem[k];
                        if (j != LIST_EMPTY && j != (unsigned)i) hit |= ((a.side_words[j >> 5] >> (j & 31u)) & 1u) != 0u;
                    }
                } else if (use_masks) {
                    COLD_LOOP
                    for (int w = 0; w < W; ++w)
                        hit |= (s.mask[(unsigned)w * (unsigned)R + (unsigned)rb] & a.side_words[w]) != 0u;
                } else {
                    COLD_LOOP
                    for (int k = 0; k < N; ++k)
                        hit |= (k != i) & ((HOT ? (a.rec_a[k].x >> D2D_REC_TYPE_SHIFT) & D2D_REC_TYPE_MASK : s.aux[k] >> 24) == LINK_SIDELINK) &
                               (__float_as_int(s.link[k].w) == rb);
                }
                violated |= hit;
            }
        }
        cap_part += cap;
        // inverse-square gains: a zero distance in the walk shows up as 1/0 = inf in the accumulator (one test instead of a
        // min per interferer); the own link and the other path-loss modes track the smallest d2 itself
        if (NF_ONLY) {
            // 1 / d^2 gains: a zero distance (own link: signal inf; an interferer: accumulator inf) always ends in a
            // non-finite SINR, so the common case is one compare and the cause is sorted out behind it
            if (UNLIKELY(!(fabsf(sinr_db) <= 3.0e38f))) {
                my_flags |= FLAG_NON_FINITE;
                if (d2 == 0.0f || !(accf <= 3.0e38f)) my_flags |= FLAG_ZERO_DISTANCE;
            }
        } else {
            if (dmin == 0) my_flags |= FLAG_ZERO_DISTANCE;
            if (!(fabsf(sinr_db) <= 3.0e38f)) my_flags |= FLAG_NON_FINITE;
        }
    }
    if (my_flags) atomicOr(&s.flags[0], my_flags);
    STAMP(6);

    // ---- pass 3: reward
    if (FULL && cfg_reward_fn != 3) {
        // No barrier: every wave publishes its part in LDS, takes a ticket, and the wave that draws the last ticket (all other
        // waves' LDS operations precede their ticket in LDS order) finishes the env.
        const int lane = tid & 63;
        if (cfg_reward_fn == 2) {                                                          // reward_fn.py:52-57
#pragma unroll
            for (int u = 0; u < KEEP; ++u) { const int i = tid + u * TPE; *at(a.reward, fresh((row + (unsigned)i) * 4u)) = s.sinr[i] >= a.reward_param ? s.sh[i] : -1.0f; }
        }
        int ticket = 0;
        if (cfg_reward_fn == 1) {
            // the wave's sum into ITS slot of red[16] (no atomic: nobody else writes it); the wave that draws the last ticket adds the
            // slots in index order, exactly as the barrier path below does - one reduction, the same bits, whichever path and
            // whatever order the waves arrive in.  (A 32.32 fixed-point atomic total until round 6: see csrc/d2d_rollout.hip.)
            const float wsum = wave_sum(cap_part);
            if (lane == 0) __hip_atomic_store(&s.red[tid >> 6], wsum, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (violated) atomicOr(&s.flags[1], 1);
        }
        // consume the prefetched word here, at the end: an empty asm that names the register keeps the load alive and costs
        // no instruction (the wait for it lands here, long after it has arrived)
        asm volatile("" ::"v"(pf));
        // release: this wave's LDS atomics above are ordered before its ticket (the LDS queue is in order; the fence keeps the
        // compiler from sinking them below it); acquire: the last wave's reads below stay behind its own ticket
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        if (lane == 0) ticket = atomicAdd(&s.flags[2], 1);
        ticket = __builtin_amdgcn_readfirstlane(ticket);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        if (ticket == (TPE >> 6) - 1) {
            if (cfg_reward_fn == 1) {
                // SystemCapacityRewardFunction, reward_fn.py:27-44: mean capacity, or -1 for everyone on a violation
                float total = 0.0f;
                for (int w = 0; w < (TPE + 255) >> 8; ++w) {                   // (atomic loads: LDS reads that cannot be hoisted above the ticket)
                    const float v0 = __hip_atomic_load(&s.red[4 * w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    const float v1 = __hip_atomic_load(&s.red[4 * w + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    const float v2 = __hip_atomic_load(&s.red[4 * w + 2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    const float v3 = __hip_atomic_load(&s.red[4 * w + 3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    total += (v0 + v1) + (v2 + v3);
                }
                const int viol = atomicOr(&s.flags[1], 0);
                const float r = (viol & 1) ? -1.0f : total * a.inv_n;          // (an inf / NaN capacity rides through the float sum by itself)
                // N = LPT * blockDim is a multiple of 64: the row goes out as 16-byte stores
                if (a.reward_env) {                                            // D2D_REWARD_PER_ENV: the scalar once, not N copies
                    if (lane == 0) a.reward_env[b] = r;
                } else {
                    const f32x4 r4 = {r, r, r, r};
                    for (int k = lane * 4; k < N; k += 256) ST(reinterpret_cast<f32x4*>(at(a.reward, fresh((row + (unsigned)k) * 4u))), r4);
                }
            }
            if (lane == 0) a.env_flags[b] = atomicOr(&s.flags[0], 0);
        }
        STAMP(7);
        return;
    }
    if (cfg_reward_fn == 1) {
        // SystemCapacityRewardFunction, reward_fn.py:27-44: mean capacity, or -1 for everyone if any link reported
        // a violation above.  One barrier: wave partial sums + the violation flag.
        const float wsum = wave_sum(cap_part);
        if (active && (lt & 63) == 0) s.red[lt >> 6] = wsum;
        if (violated) atomicOr(&s.flags[1], 1);
        __syncthreads();
        if (active) {
            const float4* red4 = reinterpret_cast<const float4*>(s.red);       // 16 slots, zero-padded: fixed-order sum
            float total = 0.0f;
            for (int w = 0; w < (TPE + 255) >> 8; ++w) { const float4 v = red4[w]; total += (v.x + v.y) + (v.z + v.w); }
            const float r = s.flags[1] ? -1.0f : total * a.inv_n;
            if (a.reward_env) { if (lt == 0) a.reward_env[b] = r; }
            else { FOR_MY_LINKS(u, i) *at(a.reward, fresh((row + (unsigned)i) * 4u)) = r; }
        }
    } else if (cfg_reward_fn == 2) {
        // ShannonRewardFunction, reward_fn.py:52-57
        FOR_MY_LINKS(u, i) *at(a.reward, fresh((row + (unsigned)i) * 4u)) = s.sinr[i] >= a.reward_param ? s.sh[i] : -1.0f;
        __syncthreads();
    } else if (cfg_reward_fn == 3) {
        // CueSinrShannonRewardFunction, reward_fn.py:65-78
        __syncthreads();
        FOR_MY_LINKS(u, i) {
            const int rbi = IN_REGS(u) ? __float_as_int(me0[KEPT(u)].w) : __float_as_int(s.link[i].w);
            bool bad = false;
            if (masks_on && (unsigned)rbi < (unsigned)R) {
                unsigned live = s.summ[rbi];
                while (live) {
                    const int w = __builtin_ctz(live);
                    live &= live - 1u;
                    unsigned word = s.mask[(unsigned)w * (unsigned)R + (unsigned)rbi] & ~s.side[w];   // non-sidelink members
                    if (w == (i >> 5)) word &= ~(1u << (i & 31));
                    while (word) {
                        const int j = (w << 5) + __builtin_ctz(word);
                        word &= word - 1u;
                        bad |= s.sinr[j] < a.reward_param;
                    }
                }
            } else {
                COLD_LOOP
                for (int j = 0; j < N; ++j)
                    bad |= (j != i) & ((s.aux[j] >> 24) != LINK_SIDELINK) & (__float_as_int(s.link[j].w) == rbi) &
                           (s.sinr[j] < a.reward_param);
            }
            *at(a.reward, fresh((row + (unsigned)i) * 4u)) = bad ? -1.0f : s.sh[i];
        }
        __syncthreads();
    } else {
        __syncthreads();
    }

    asm volatile("" ::"v"(pf));                    // consume the prefetched word (late)
    if (active && lt == 0) a.env_flags[b] = s.flags[0];
    STAMP(7);

    // ---- pass 4 (small N only): LinearObs expansion of this workgroup's envs, obs_fn.py:43-53.  Every thread of the
    // workgroup streams 16-byte (or 8-byte, odd N) stores over the contiguous [N][6N] block of each env; tflat was
    // published by the reward pass's barrier.  Same mapping as csrc/d2d_obs.hip (bit-identical output).
    if (!FULL && (HOT == 2 || a.fuse_obs)) {
        const unsigned T = blockDim.x, q_per_row = a.obs_q_per_row, total = (unsigned)N * q_per_row;
        const unsigned row_floats = 6u * (unsigned)N;
        if (HOT == 2 || a.fuse_obs == 4) {
            // The obs blocks of this workgroup's envs are ONE contiguous region of n_el * N * q_per_row float4: a pass of the
            // workgroup is T consecutive float4 of it (every lane stores in every pass but the last; at N = 50 a row is 75 float4,
            // so a row-aligned pass would leave 31 of 256 lanes idle).  A lane's position (env, row, column) is carried as
            // (LDS byte address of the env's T, 6 * row, float column) and ADVANCED by the pass's constant step with two
            // conditional wraps - no division, no per-pass pointer rebuild, one 32-bit byte offset against an SGPR base for the
            // store.  The two ds_read_b64 of pass t + 1 are issued before the store of pass t (the wave is alone on its SIMD at
            // 256 threads per CU: nothing else hides the LDS round trip).
            //   source of output column f (even) in row i: 6i + f if f < 6 (the agent's own six values), f - 6 if f < 6i + 6
            //   (links before the agent shift by one slot), else f - obs_fn.py:43-53, same mapping as csrc/d2d_obs.hip.
            const unsigned n_el = min((unsigned)a.epw, (unsigned)a.B - blockIdx.x * (unsigned)a.epw);
            const unsigned region = n_el * total;                              // float4 of the whole region
            const unsigned full = region / T;                                  // passes in which every lane stores
            const unsigned passes = full + (region - full * T != 0u ? 1u : 0u);
            // the pass step T float4 = de envs + dr rows + dq columns (de = 0 unless an env is smaller than a pass)
            const unsigned de = T / total, rem_e = T - de * total;
            const unsigned dr = (unsigned)(((unsigned long long)rem_e * a.obs_q_magic) >> 40), dq = rem_e - dr * q_per_row;
            const unsigned step_f = 4u * dq, step_head = 6u * dr, step_lds = de * a.lds.env_bytes;
            unsigned char* const out_base = reinterpret_cast<unsigned char*>(a.obs + (size_t)(blockIdx.x * (unsigned)a.epw) * N * row_floats);
            // All workgroups are resident at once and reach this point together; walked from its start by everyone, the chip's
            // concurrent stores sit one fixed stride apart.  Every workgroup starts at its own pass and wraps around (two segments).
            const unsigned p0 = a.obs_rotate ? (blockIdx.x * (unsigned)a.obs_rotate) % passes : 0u;
            struct Cur { unsigned lds, head, f; };
            const auto locate = [&](unsigned idx) {                           // once per segment: the only divisions
                const unsigned el = idx / total, in_env = idx - el * total;
                const unsigned i = (unsigned)(((unsigned long long)in_env * a.obs_q_magic) >> 40), q = in_env - i * q_per_row;
                Cur c; c.lds = el * a.lds.env_bytes + a.lds.tflat; c.head = 6u * i; c.f = 4u * q;
                return c;
            };
            const auto advance = [&](Cur c) {
                c.f += step_f;
                const bool cw = c.f >= row_floats;
                c.f -= cw ? row_floats : 0u;
                c.head += step_head + (cw ? 6u : 0u);
                const bool rw = c.head >= row_floats;
                c.head -= rw ? row_floats : 0u;
                c.lds += step_lds + (rw ? a.lds.env_bytes : 0u);
                return c;
            };
            const auto fetch = [&](const Cur& c) {
                const unsigned f1 = c.f + 2u, lim = c.head + 6u;
                const unsigned s0 = c.f < 6u ? c.head + c.f : (c.f < lim ? c.f - 6u : c.f);
                const unsigned s1 = f1 < 6u ? c.head + f1 : (f1 < lim ? f1 - 6u : f1);
                const f32x2 lo = *reinterpret_cast<const f32x2*>(smem_raw + c.lds + s0 * 4u);
                const f32x2 hi = *reinterpret_cast<const f32x2*>(smem_raw + c.lds + s1 * 4u);
                const f32x4 v = {lo.x, lo.y, hi.x, hi.y};
                return v;
            };
            const auto segment = [&](unsigned begin, unsigned end) {           // whole passes [begin, end): no predication
                if (begin >= end) return;
                unsigned off = (begin * T + (unsigned)tid) * 16u;
                Cur c = locate(begin * T + (unsigned)tid);
                f32x4 v = fetch(c);
#pragma unroll 2
                for (unsigned p = begin + 1u; p < end; ++p) {
                    c = advance(c);
                    const f32x4 vn = fetch(c);
                    __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(out_base + off));
                    off += T * 16u;
                    v = vn;
                }
                __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(out_base + off));
            };
            segment(p0 < full ? p0 : full, full);
            if (passes != full) {                                              // the region's tail: part of the lanes
                const unsigned idx = full * T + (unsigned)tid;
                if (idx < region) __builtin_nontemporal_store(fetch(locate(idx)), reinterpret_cast<f32x4*>(out_base + (size_t)idx * 16u));
            }
            segment(0u, p0 < full ? p0 : full);
        } else if (HOT != 2) {
            for (int el = 0; el < a.epw; ++el) {
                const int be = blockIdx.x * a.epw + el;
                if (be >= a.B) break;
                const float* t_flat = reinterpret_cast<const float*>(smem_raw + (unsigned)el * a.lds.env_bytes + a.lds.tflat);
                float* out = a.obs + (size_t)be * N * row_floats;
#pragma unroll 2
                for (unsigned idx = tid; idx < total; idx += T) {
                    const unsigned i = (unsigned)(((unsigned long long)idx * a.obs_q_magic) >> 40);   // idx / q_per_row
                    const unsigned q = idx - i * q_per_row;
                    if (a.fuse_obs == 4) {
                        const unsigned f = q * 4u;
                        const f32x2 lo = reinterpret_cast<const f32x2*>(t_flat)[obs_src_col(f, i) >> 1];
                        const f32x2 hi = reinterpret_cast<const f32x2*>(t_flat)[obs_src_col(f + 2u, i) >> 1];
                        const f32x4 v = {lo.x, lo.y, hi.x, hi.y};
                        __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(out + (size_t)idx * 4));
                    } else {
                        const f32x2 v = reinterpret_cast<const f32x2*>(t_flat)[obs_src_col(q * 2u, i) >> 1];
                        __builtin_nontemporal_store(v, reinterpret_cast<f32x2*>(out + (size_t)idx * 2));
                    }
                }
            }
        }
    }
    STAMP(8);
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

// Per-link position rows (tx_x, tx_y, rx_x, rx_y): the link -> device gather done ONCE per reset / link change
// (simulator.py:61-75 only moves devices in reset()), so that the step kernel's prologue is a single coalesced 16-byte
// load per link instead of a dependent link table -> device position double hop.
__global__ __launch_bounds__(256) void link_positions_kernel(const float* __restrict__ px, const float* __restrict__ py,
                                                            const float* __restrict__ lx, const float* __restrict__ ly,
                                                            const int4* __restrict__ rec_a, int B, int N, int D,
                                                            float4* __restrict__ lpos, float4* __restrict__ lpos_lo) {
    const size_t gid = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (gid >= (size_t)B * N) return;
    const int b = (int)(gid / N), i = (int)(gid - (size_t)b * N);
    const int4 ra = rec_a[i];
    const size_t base = (size_t)b * D;
    const int txd = ra.x & D2D_REC_TXDEV_MASK, rxd = ra.y;
    lpos[gid] = make_float4(px[base + txd], py[base + txd], px[base + rxd], py[base + rxd]);
    if (lpos_lo) lpos_lo[gid] = make_float4(lx[base + txd], ly[base + txd], lx[base + rxd], ly[base + rxd]);   // exact positions: the low parts
}

hipError_t launch_link_positions(const float* pos_x, const float* pos_y, const float* lo_x, const float* lo_y, const int4* rec_a,
                                 int B, int N, int D, float4* lpos, float4* lpos_lo, hipStream_t stream) {
    const size_t total = (size_t)B * N;
    if (total == 0) return hipSuccess;
    hipLaunchKernelGGL(link_positions_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, pos_x, pos_y,
                       lo_x, lo_y, rec_a, B, N, D, lpos, lpos_lo);
    return hipGetLastError();
}

hipError_t launch_step(const StepArgs& a, PlMode mode, int block_threads, hipStream_t stream) {
    const size_t lds = (size_t)a.lds.env_bytes * (size_t)a.epw;
    dim3 grid((unsigned)((a.B + a.epw - 1) / a.epw)), block(block_threads);
    hipError_t err = hipSuccess;
    const int lpt = a.lpt;
    const bool full = lpt > 0 && a.epw == 1 && a.N == lpt * a.tpe && block_threads == a.tpe && !a.fuse_obs;
    const bool lists = a.walk == 2 && lpt > 0 && a.reward_fn != 3;
    const bool hot = a.action_mode == 0 && a.col_mode == 0 && a.n_fixed == 0 && a.act_stride == a.N && a.reward_fn == 1 &&
                     (a.walk == 0 || lists) && a.prefetch_envs > 0 && (a.ablate & 8191) == 0 &&
                     a.mask_words > 0 && (mode == PL_INV_SQUARE || mode == PL_POWER || mode == PL_POWK) && !a.lpos_lo;
    const int hot_opt = (a.rec_uniform ? OPT_SREC : 0) | (a.nt_results ? OPT_NT : 0);
    const int xp = a.lpos_lo ? OPT_XPOS : 0;                   // float64 positions (d2d_set_positions_f64): the coord_diff kernels
    if (a.rollout) return launch_rollout(a, mode, (a.N % 64 ? (hot_opt & OPT_NT) | OPT_PAD : hot_opt) | xp, block_threads, stream);
    // the rollout configuration with member lists has a kernel of its own (d2d_rollout.hip; chosen by run_step)
    // level 2: several small envs per workgroup with the LinearObs expansion fused (BASELINE config 2), a fixed prefix
    // (traffic-model CUEs) allowed
    const bool hot2 = a.action_mode == 0 && a.col_mode == 0 && a.act_stride > 0 && a.reward_fn == 1 && a.write_table &&
                      a.walk == 0 && a.prefetch_envs > 0 && (a.ablate & 8191) == 0 && a.mask_words > 0 &&
                      a.fuse_obs == 4 && a.obs_q_per_row > 0 && !a.lpos_lo &&
                      (mode == PL_INV_SQUARE || mode == PL_POWER || mode == PL_POWK);
#define D2D_LAUNCH_1(...)                                                                                \
    do {                                                                                                 \
        if (lds > 48 * 1024)                                                                             \
            err = hipFuncSetAttribute(reinterpret_cast<const void*>(&step_kernel<__VA_ARGS__>),          \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);             \
        if (err == hipSuccess) {                                                                         \
            hipLaunchKernelGGL((step_kernel<__VA_ARGS__>), grid, block, lds, stream, a);                 \
            err = hipGetLastError();                                                                     \
        }                                                                                                \
    } while (0)
#define D2D_LAUNCH_HOT1(M)                                                                               \
    do {                                                                                                 \
        switch (hot_opt) {                                                                               \
            case 0: D2D_LAUNCH_1(M, 1, true, 1, 0); break;                                               \
            case OPT_SREC: D2D_LAUNCH_1(M, 1, true, 1, OPT_SREC); break;                                 \
            case OPT_NT: D2D_LAUNCH_1(M, 1, true, 1, OPT_NT); break;                                     \
            default: D2D_LAUNCH_1(M, 1, true, 1, OPT_SREC | OPT_NT); break;                              \
        }                                                                                                \
    } while (0)
#define D2D_LAUNCH_L(M, L)                                                                               \
    do {                                                                                                 \
        if (lpt == 2 && full) D2D_LAUNCH_1(M, 2, true, 0, L);                                            \
        else if (lpt == 2) D2D_LAUNCH_1(M, 2, false, 0, L);                                              \
        else if (lpt == 1 && full) D2D_LAUNCH_1(M, 1, true, 0, L);                                       \
        else if (lpt == 1) D2D_LAUNCH_1(M, 1, false, 0, L);                                              \
        else D2D_LAUNCH_1(M, 0, false, 0, (L) & OPT_XPOS);                                               \
    } while (0)
#define D2D_LAUNCH_COLD(M)                                                                               \
    do {                                                                                                 \
        if (xp) { if (lists) D2D_LAUNCH_L(M, OPT_LISTS | OPT_XPOS); else D2D_LAUNCH_L(M, OPT_XPOS); }    \
        else if (lists) D2D_LAUNCH_L(M, OPT_LISTS);                                                      \
        else D2D_LAUNCH_L(M, 0);                                                                         \
    } while (0)
#define D2D_LAUNCH(M)                                                                                    \
    do {                                                                                                 \
        if (lpt == 1 && full && hot) D2D_LAUNCH_HOT1(M);                                                 \
        else if (lpt == 1 && !full && hot2) D2D_LAUNCH_1(M, 1, false, 2, 0);                             \
        else D2D_LAUNCH_COLD(M);                                                                         \
    } while (0)
    switch (mode) {
        case PL_INV_SQUARE: D2D_LAUNCH(PL_INV_SQUARE); break;
        case PL_POWER: D2D_LAUNCH(PL_POWER); break;
        case PL_POWK: D2D_LAUNCH(PL_POWK); break;
        case PL_TABLE: D2D_LAUNCH_COLD(PL_TABLE); break;          // the rollout specialisations exist for the power laws only
        case PL_SHADOW: D2D_LAUNCH_COLD(PL_SHADOW); break;
    }
#undef D2D_LAUNCH
#undef D2D_LAUNCH_COLD
#undef D2D_LAUNCH_HOT1
#undef D2D_LAUNCH_L
#undef D2D_LAUNCH_1
    return err;
}

}  // namespace d2d
