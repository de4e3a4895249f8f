// C-ABI layer of libd2d_hip.so (see include/d2d_hip.h).  Owns the SoA state of B environments in HBM and
// enqueues the step / obs kernels on one HIP stream.  No exceptions cross the boundary; no CPU fallback exists:
// if HIP is unusable every call fails with D2D_ERR_HIP.
#include "../../include/d2d_hip_diag.h"
#include "d2d_internal.h"

#include <dlfcn.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <new>
#include <stdexcept>
#include <string>
#include <vector>

namespace {

thread_local std::string g_last_error;

int fail(int code, const std::string& msg) {
    g_last_error = msg;
    return code;
}

#define HIP_TRY(expr)                                                                               \
    do {                                                                                            \
        hipError_t _e = (expr);                                                                     \
        if (_e != hipSuccess)                                                                       \
            return fail(D2D_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e));           \
    } while (0)

// Every entry point that may touch HIP makes the handle's GPU current for the calling thread for the duration of
// the call and puts the caller's device back on exit: a process that holds handles or torch tensors on several GPUs
// must not find its current device changed behind its back (torch reads it through hipGetDevice).
struct DeviceGuard {
    int prev = -1;
    hipError_t err = hipSuccess;
    explicit DeviceGuard(int want) {
        err = hipGetDevice(&prev);
        if (err != hipSuccess) { prev = -1; return; }
        if (prev != want) err = hipSetDevice(want); else prev = -1;     // nothing to restore
    }
    ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
    DeviceGuard(const DeviceGuard&) = delete;
    DeviceGuard& operator=(const DeviceGuard&) = delete;
};
#define USE_DEVICE(h)                                                                                \
    DeviceGuard device_guard_((h)->cfg.device_ordinal);                                              \
    if (device_guard_.err != hipSuccess)                                                             \
        return fail(D2D_ERR_HIP, std::string("hipSetDevice: ") + hipGetErrorString(device_guard_.err))

struct Buffer {
    void* ptr = nullptr;
    size_t bytes = 0;     // capacity
    bool owned = false;
};

struct EventPair {
    hipEvent_t start, stop;
    int kernel;
};

}  // namespace

struct d2d_handle {
    d2d_config cfg;
    int B = 0, D = 0, N = 0, Nmax = 0;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    Buffer buf[D2D_BUF_COUNT];
    // device-side tables
    float4* rec = nullptr;          // per-link records: 3 rows of Nmax x 16 B (d2d_internal.h), see refresh_tables
    float2* rec_h = nullptr;        // per-link (head, tail) of -exponent / 2, see refresh_tables
    int* rec_grp = nullptr;         // per group of 64 links: the group's record in one 64-byte row (rec_uniform only), see refresh_tables
    int* act_cols = nullptr;        // [Nmax] action column per link (arbitrary fixed sets)
    unsigned* side_words = nullptr; // [ceil(Nmax / 32)] sidelink membership bits, rebuilt with the records
    float4* lpos = nullptr;         // [B, Nmax] per-link (tx_x, tx_y, rx_x, rx_y), see refresh_link_positions
    bool lpos_dirty = true;
    // exact positions (d2d_set_positions_f64): coordinate = hi + lo; hi lives in POS_X / POS_Y (and lpos), the low parts here
    float* pos_lo = nullptr;        // [2][B, D]: x low parts, then y low parts (allocated by the first float64 upload)
    float4* lpos_lo = nullptr;      // [B, Nmax] low parts of the per-link rows
    bool have_lo = false;           // some coordinate has a non-zero low part: the step runs the OPT_XPOS kernels
    unsigned long long* dbg = nullptr;   // diagnostic builds only
    bool rec_uniform = false;       // records identical within every aligned group of 64 links (refresh_tables)
    bool rec_uniform128 = false;    // ... and within every aligned group of 128: the rollout kernel's two links per thread
    bool std_layout = false;        // link i < C is (cue i -> mbs), link C + k is (due 2k -> due 2k+1): d2d_reset_positions writes lpos itself
    float* gain_table = nullptr;
    size_t gain_elems = 0;
    int table_per_env = 0;
    int table_links = 0;            // > 0: gain_table is [B or 1][n][n] by (tx LINK, rx LINK) of that link list (d2d_set_path_loss_link_table)
    unsigned* status = nullptr;
    // host-side copies used to derive the device columns
    std::vector<double> eirp_off, rx_off, noise, sens, bw, a_tx, a_rx, expo;
    std::vector<int> host_tx, host_rx, host_type;   // host copy of the link table
    std::vector<int> fixed_rb, fixed_pwr;            // per link; fixed_rb[i] == INT32_MIN <=> agent-driven
    int n_fixed = 0;
    int col_mode = 0;                                // 0: fixed links form a prefix of the link list
    bool have_dev = false, have_pl = false, have_links = false, have_pos = false, tables_dirty = true;
    d2d::PlMode mode = d2d::PL_INV_SQUARE;
    int reward_fn = D2D_REWARD_SYSTEM_CAPACITY;
    float reward_param = 0.0f;
    int obs_mode = D2D_OBS_LINEAR;
    int bucketing = 1;
    int export_actions = 1;                          // d2d_step writes the decoded (rb, pwr) to D2D_BUF_RB / D2D_BUF_PWR
    int pow_k = 0;                                   // PL_POWK: the integer every link transmitter's exponent lies within 1/2 of (refresh_tables)
    int obs_f64 = 0;                                 // D2D_BUF_OBS holds float64 (d2d_set_obs_dtype)
    int reward_layout = D2D_REWARD_PER_AGENT;        // SystemCapacity: [B,N] rows or one scalar per env (D2D_BUF_REWARD_ENV)
    int tune_rows = 0, tune_nt = 1, tune_xcd = 1, tune_block = 0, tune_variant = 0, tune_step_threads = 0, tune_stagger = 0;
    int num_cus = 0;
    int tune_step_prefetch = -1;                     // action prefetch distance in envs: -1 auto, 0 off
    int tune_step_obs_rotate = -1;                   // fused expansion start-phase multiplier: -1 auto (29), 0 off
    int tune_step_nt = -1, tune_step_srec = -1;      // nontemporal result stores / scalar record loads: -1 auto, 0 off, 1 on (if legal)
    int tune_step_epw = 0, tune_step_block = 0, tune_step_fuse = -1, tune_step_ablate = 0, tune_step_walk = -1, tune_step_lpt = -1;
    // d2d_step_host: packed device block + pinned host mirrors
    void* host_out_dev = nullptr; size_t host_out_bytes = 0;
    void* host_out_pinned = nullptr;
    int32_t* host_in_dev = nullptr; int32_t* host_in_pinned = nullptr; size_t host_in_bytes = 0;
    // RCCL communicator (d2d_comm_init)
    void* comm = nullptr;
    int comm_world = 0, comm_rank = 0;
    bool comm_used = false;                          // a d2d_allgather has been enqueued on the communicator
    unsigned long long env_offset = 0;
    double shadow_chi = 0, shadow_d0 = 0;
    unsigned long long shadow_seed = 0, shadow_step = 0;
    unsigned char* fixed_mask_dev = nullptr;   // [D] + pad
    float* fixed_xy_dev = nullptr;             // [D,2]
    // profiling
    bool prof = false;
    std::vector<EventPair> events;
    size_t events_used = 0;
    double acc_ms[2] = {0, 0};
    int64_t launches[2] = {0, 0};
    std::vector<float> samples[2];                   // per-launch durations since the last reset (d2d_profile_median), at most 65536 each
};

namespace {

size_t active_bytes(const d2d_handle* h, int which, int n_links) {
    const size_t B = h->B, D = h->D, N = n_links;
    switch (which) {
        case D2D_BUF_POS_X: case D2D_BUF_POS_Y: return B * D * 4;
        case D2D_BUF_OBS_TABLE: return B * N * 6 * 4;
        case D2D_BUF_OBS: return B * N * 6 * N * (h->obs_f64 ? 8 : 4);
        case D2D_BUF_ENV_FLAGS: case D2D_BUF_REWARD_ENV: return B * 4;
        case D2D_BUF_LINK_POS: return B * N * 16;
        default: return B * N * 4;
    }
}

int ensure_buffer(d2d_handle* h, int which, void** out) {
    if (which == D2D_BUF_LINK_POS) { *out = h->lpos; return D2D_OK; }      // library-owned, sized at d2d_create
    Buffer& bf = h->buf[which];
    const size_t need = active_bytes(h, which, which == D2D_BUF_OBS ? h->N : h->Nmax);
    if (bf.ptr && bf.bytes >= need) { *out = bf.ptr; return D2D_OK; }
    if (bf.ptr && !bf.owned)
        return fail(D2D_ERR_INVALID, "bound buffer " + std::to_string(which) + " too small: have " +
                                         std::to_string(bf.bytes) + " need " + std::to_string(need));
    if (bf.ptr) { HIP_TRY(hipStreamSynchronize(h->stream)); HIP_TRY(hipFree(bf.ptr)); bf.ptr = nullptr; }
    HIP_TRY(hipMalloc(&bf.ptr, need ? need : 4));
    bf.bytes = need; bf.owned = true;
    if (which == D2D_BUF_ENV_FLAGS) HIP_TRY(hipMemsetAsync(bf.ptr, 0, need, h->stream));
    *out = bf.ptr;
    return D2D_OK;
}

int refresh_tables(d2d_handle* h) {
    if (!h->tables_dirty) return D2D_OK;
    if (!h->have_dev) return fail(D2D_ERR_STATE, "d2d_set_device_table has not been called");
    if (!h->have_pl) return fail(D2D_ERR_STATE, "no path-loss model set (d2d_set_path_loss_*)");
    const int D = h->D;
    std::vector<float> cols((size_t)7 * D);
    bool all_two = true;
    for (int d = 0; d < D; ++d) {
        const double a_tx = h->mode == d2d::PL_TABLE ? 0.0 : h->a_tx[d];
        const double a_rx = h->mode == d2d::PL_TABLE ? 0.0 : h->a_rx[d];
        cols[0 * D + d] = (float)std::pow(10.0, (h->eirp_off[d] - a_tx) / 10.0);
        cols[1 * D + d] = (float)std::pow(10.0, -a_rx / 10.0);
        cols[2 * D + d] = (float)std::pow(10.0, h->rx_off[d] / 10.0);
        cols[3 * D + d] = (float)std::pow(10.0, h->noise[d] / 10.0);
        cols[4 * D + d] = (float)h->sens[d];
        cols[5 * D + d] = (float)(1e-6 * h->bw[d]);
        cols[6 * D + d] = h->mode == d2d::PL_TABLE ? 2.0f : (float)h->expo[d];
        if (h->mode != d2d::PL_TABLE && h->expo[d] != 2.0) all_two = false;
        // The kernels work in linear float32 (mW and plain gain factors): a link-budget constant beyond about +-300 dB
        // (COST-Hata's mobile-height correction applied to a receiving antenna above ~100 m, path_loss.py:105-112) leaves that range
        // and would surface as inf / NaN results.  Refused here, by name, instead.
        for (int c = 0; c < 4; ++c) {
            const float v = cols[c * D + d];
            if (!(v >= 1.0e-30f && v <= 1.0e30f))
                return fail(D2D_ERR_UNSUPPORTED, "device " + std::to_string(d) + ": link-budget constant " + std::to_string(c) +
                                                 " (tx_lin, rx_pl, rx_lin, noise_mw) is outside the float32 linear range 1e-30 .. 1e30 (a term beyond +-300 dB)");
        }
    }
    if (h->mode != d2d::PL_TABLE && h->mode != d2d::PL_SHADOW) {
        h->mode = all_two ? d2d::PL_INV_SQUARE : d2d::PL_POWER;
        // PL_POWK: the exponents of the transmitters of the CURRENT links all lie within 1/2 of one integer k in 1 .. 8 (COST-Hata's
        // 3.6 / 4.375, a log-distance ple of 3.5: k = 4) - then (d^2)^(-n/2) = (d^2)^(-k/2) (d^2)^phi with |phi| <= 1/4, and the
        // kernels evaluate it with reciprocals, products and one short exp2(phi log2 d^2) (pow_k_gains); any other mix of exponents
        // keeps the general split (pow_neg_half)
        h->pow_k = 0;
        if (!all_two && h->N > 0) {
            const int k0 = (int)std::lround(h->expo[h->host_tx[0]]);
            for (int k : {k0, k0 + 1, k0 - 1}) {              // (link 0's exponent may sit at the edge of the band the others share)
                bool ok = k >= 1 && k <= 8;
                for (int i = 0; i < h->N && ok; ++i) ok = std::fabs(h->expo[h->host_tx[i]] - (double)k) <= 0.5;
                if (ok) { h->mode = d2d::PL_POWK; h->pow_k = k; break; }
            }
        }
    }
    // Per-link records, 3 rows of [Nmax] x 16 B (layout: d2d_internal.h).  tx-side columns by the link's tx device,
    // rx-side by its rx device, so the kernel reads them coalesced by link index with no link -> device -> column
    // double hop; a link with a fixed action carries (rb, pwr) here, the others their column in the action array.
    const int N = h->N, S = h->Nmax;
    std::vector<float> rec((size_t)3 * S * 4, 0.0f);
    int32_t* ra = reinterpret_cast<int32_t*>(rec.data());
    float* rb = rec.data() + (size_t)S * 4;
    float* rc = rec.data() + (size_t)2 * S * 4;
    const int levels[4] = {0, h->cfg.pwr_levels_cue, h->cfg.pwr_levels_mbs, h->cfg.pwr_levels_due};   // by d2d_link_type
    int col = 0;
    bool prefix = true;               // are the fixed links exactly the first n_fixed links?
    for (int i = 0; i < N; ++i) {
        const int t = h->host_tx[i], r = h->host_rx[i];
        const bool fixed = h->fixed_rb[i] != INT32_MIN;
        ra[4 * i + 0] = t | (h->host_type[i] << D2D_REC_TYPE_SHIFT) | (fixed ? D2D_REC_FIXED_BIT : 0);
        ra[4 * i + 1] = r;
        const uint32_t P = (uint32_t)levels[h->host_type[i]];
        // action decode by multiply-high: M = ceil(2^32 / P) is exact for act < 2^32 / P (error term M P - 2^32 < P); the
        // bound travels with it, capped at 2^24 - 1 (q * P then fits the 24-bit multiplier).  P < 2: hardware divide.
        const uint32_t M = P >= 2 ? (uint32_t)(((1ull << 32) + P - 1) / P) : 0u;
        // ... and at R * P - 1, the largest action that names an RB inside [0, R): the rollout kernel (d2d_rollout.hip) tests
        // `act <= bound` once and then knows both that the multiply-high quotient is exact and that it is a valid RB; an action
        // beyond the bound takes the division arm everywhere, which is exact for any value
        const uint32_t lim = P >= 2 ? (uint32_t)std::min<uint64_t>(std::min<uint64_t>(0xFFFFFFFFull / P, (1u << 24) - 1),
                                                                   (uint64_t)h->cfg.num_rbs * P - 1) : 0u;
        ra[4 * i + 2] = fixed ? h->fixed_rb[i] : (int32_t)M;
        ra[4 * i + 3] = fixed ? h->fixed_pwr[i] : (int32_t)lim;
        if (fixed != (i < h->n_fixed)) prefix = false;
        const uint32_t packed = (P & 0xFFFFu) | ((uint32_t)(fixed ? 0 : col++) << 16);
        std::memcpy(&rc[4 * i + 3], &packed, 4);
        rb[4 * i + 0] = cols[0 * D + t];    // tx_lin
        rb[4 * i + 1] = cols[1 * D + r];    // rx_pl
        rb[4 * i + 2] = cols[2 * D + r];    // rx_lin
        rb[4 * i + 3] = cols[3 * D + r];    // noise_mw
        rc[4 * i + 0] = cols[4 * D + r];    // sens_db
        rc[4 * i + 1] = cols[5 * D + t];    // bw_mhz
        rc[4 * i + 2] = cols[6 * D + t];    // exponent
    }
    // -exponent / 2 of every link's transmitter as head + tail: the head keeps the 12 leading bits (its product with a binary
    // exponent is exact in the kernel), the tail the rest of the DOUBLE - the pair carries ~36 bits, so that the exponent's own
    // rounding (6e-8 as a float, amplified by ln(d^2) e / 2 in (d^2)^(-e/2): 1.2e-6 at 300 m, e = 3.5) leaves the result
    std::vector<float> rech((size_t)2 * S, 0.0f);
    for (int i = 0; i < N; ++i) {
        if (h->mode == d2d::PL_POWK) {                        // (phi, 0): the exponent's distance from k, halved
            rech[2 * i] = (float)(-0.5 * (h->expo[h->host_tx[i]] - (double)h->pow_k));
            rech[2 * i + 1] = 0.0f;
            continue;
        }
        const double hd = h->mode == d2d::PL_TABLE ? -1.0 : -0.5 * h->expo[h->host_tx[i]];
        float head = (float)hd;
        uint32_t hb;
        std::memcpy(&hb, &head, 4); hb &= 0xFFFFF000u; std::memcpy(&head, &hb, 4);
        rech[2 * i] = head;
        rech[2 * i + 1] = (float)(hd - (double)head);
    }
    HIP_TRY(hipMemcpyAsync(h->rec_h, rech.data(), rech.size() * 4, hipMemcpyHostToDevice, h->stream));
    std::vector<int32_t> cols_host((size_t)S, 0);
    for (int i = 0; i < N; ++i) {
        uint32_t packed;
        std::memcpy(&packed, &rc[4 * i + 3], 4);
        cols_host[i] = (int32_t)(packed >> 16);
    }
    std::vector<uint32_t> side_host((size_t)(S + 31) / 32 + 1, 0u);
    for (int i = 0; i < N; ++i)
        if (h->host_type[i] == D2D_SIDELINK) side_host[i >> 5] |= 1u << (i & 31);
    HIP_TRY(hipMemcpyAsync(h->side_words, side_host.data(), side_host.size() * 4, hipMemcpyHostToDevice, h->stream));
    HIP_TRY(hipMemcpyAsync(h->act_cols, cols_host.data(), cols_host.size() * 4, hipMemcpyHostToDevice, h->stream));
    HIP_TRY(hipMemcpyAsync(h->rec, rec.data(), rec.size() * 4, hipMemcpyHostToDevice, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));   // rec is a stack-lifetime host buffer
    h->col_mode = prefix ? 0 : 1;
    const int C = h->cfg.num_cues;
    bool std_layout = N == C + (D - 1 - C) / 2 && D == 1 + C + 2 * (N - C);
    for (int i = 0; i < N && std_layout; ++i)
        std_layout = i < C ? (h->host_tx[i] == i + 1 && h->host_rx[i] == 0)
                           : (h->host_tx[i] == C + 1 + 2 * (i - C) && h->host_rx[i] == C + 2 + 2 * (i - C));
    h->std_layout = std_layout;
    // Are the records of every aligned group of 64 links identical, device ids and action column aside?  (Homogeneous
    // device classes with the class boundary on a multiple of 64: BASELINE configs 3-5.)  The rollout kernel then reads
    // them with one scalar load per wave instead of 64 x 48 bytes per wave.
    bool uniform = N > 0 && N % 64 == 0;
    for (int i = 0; i < N && uniform; ++i) {
        const int g = i & ~63;
        uniform = (ra[4 * i] >> D2D_REC_TYPE_SHIFT) == (ra[4 * g] >> D2D_REC_TYPE_SHIFT) && ra[4 * i + 2] == ra[4 * g + 2] &&
                  ra[4 * i + 3] == ra[4 * g + 3] && std::memcmp(&rb[4 * i], &rb[4 * g], 16) == 0 &&
                  std::memcmp(&rc[4 * i], &rc[4 * g], 12) == 0;
        uint32_t pi, pg;
        std::memcpy(&pi, &rc[4 * i + 3], 4); std::memcpy(&pg, &rc[4 * g + 3], 4);
        uniform = uniform && (pi & 0xFFFFu) == (pg & 0xFFFFu) && rech[2 * i] == rech[2 * g] && rech[2 * i + 1] == rech[2 * g + 1];
    }
    if (uniform) {
        // ... and the rollout kernel takes a group's whole record with ONE s_load_dwordx16 from a 64-byte row
        std::vector<int32_t> grp((size_t)(N / 64) * 16, 0);
        for (int g = 0; g < N / 64; ++g) {
            const int i = g * 64;
            int32_t* row = grp.data() + (size_t)g * 16;
            row[0] = ra[4 * i + 2]; row[1] = ra[4 * i + 3]; row[2] = ra[4 * i] & ~D2D_REC_TXDEV_MASK; row[3] = 0;
            std::memcpy(row + 4, &rb[4 * i], 16);
            std::memcpy(row + 8, &rc[4 * i], 16);
            std::memcpy(row + 12, &rech[2 * i], 8);
        }
        HIP_TRY(hipMemcpyAsync(h->rec_grp, grp.data(), grp.size() * 4, hipMemcpyHostToDevice, h->stream));
        HIP_TRY(hipStreamSynchronize(h->stream));
    }
    h->rec_uniform = uniform;
    bool uniform128 = uniform && N % 128 == 0;
    for (int g = 0; g + 1 < N / 64 && uniform128; g += 2) {
        const int i = g * 64, j = i + 64;
        uint32_t pi, pj;
        std::memcpy(&pi, &rc[4 * i + 3], 4); std::memcpy(&pj, &rc[4 * j + 3], 4);
        uniform128 = (ra[4 * i] >> D2D_REC_TYPE_SHIFT) == (ra[4 * j] >> D2D_REC_TYPE_SHIFT) && ra[4 * i + 2] == ra[4 * j + 2] &&
                     ra[4 * i + 3] == ra[4 * j + 3] && std::memcmp(&rb[4 * i], &rb[4 * j], 16) == 0 &&
                     std::memcmp(&rc[4 * i], &rc[4 * j], 12) == 0 && (pi & 0xFFFFu) == (pj & 0xFFFFu) &&
                     rech[2 * i] == rech[2 * j] && rech[2 * i + 1] == rech[2 * j + 1];
    }
    h->rec_uniform128 = uniform128;
    h->tables_dirty = false;
    h->lpos_dirty = true;                       // the link -> device map may have changed
    return D2D_OK;
}

// (Re)build the per-link position rows from POS_X / POS_Y through the link records (one small kernel, only after a
// reset / set_positions / set_links - simulator.py:61-75 is the only place the reference moves devices).
int refresh_link_positions(d2d_handle* h, const float* px, const float* py) {
    if (!h->lpos_dirty) return D2D_OK;
    const size_t bd = (size_t)h->B * h->D;
    if (h->have_lo && !h->lpos_lo) HIP_TRY(hipMalloc(&h->lpos_lo, (size_t)h->B * h->Nmax * 16));
    HIP_TRY(d2d::launch_link_positions(px, py, h->have_lo ? h->pos_lo : nullptr, h->have_lo ? h->pos_lo + bd : nullptr,
                                       reinterpret_cast<const int4*>(h->rec), h->B, h->N, h->D, h->lpos, h->have_lo ? h->lpos_lo : nullptr, h->stream));
    h->lpos_dirty = false;
    return D2D_OK;
}

int drain_events(d2d_handle* h);

int record_start(d2d_handle* h, int kernel, EventPair** out) {
    *out = nullptr;
    if (!h->prof) return D2D_OK;
    if (h->events_used >= 8192) {              // long profiled runs: fold what is pending instead of growing the pool
        int rc = drain_events(h);
        if (rc) return rc;
    }
    if (h->events_used == h->events.size()) {
        EventPair ep;
        HIP_TRY(hipEventCreate(&ep.start));
        HIP_TRY(hipEventCreate(&ep.stop));
        h->events.push_back(ep);
    }
    EventPair* ep = &h->events[h->events_used++];
    ep->kernel = kernel;
    HIP_TRY(hipEventRecord(ep->start, h->stream));
    *out = ep;
    return D2D_OK;
}

int drain_events(d2d_handle* h) {
    if (h->events_used == 0) return D2D_OK;
    HIP_TRY(hipStreamSynchronize(h->stream));
    for (size_t k = 0; k < h->events_used; ++k) {
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, h->events[k].start, h->events[k].stop));
        h->acc_ms[h->events[k].kernel] += ms;
        h->launches[h->events[k].kernel] += 1;
        if (h->samples[h->events[k].kernel].size() < 65536) h->samples[h->events[k].kernel].push_back(ms);
    }
    h->events_used = 0;
    return D2D_OK;
}

// Where one step writes its results: the handle's D2D_BUF_* buffers (d2d_step / d2d_step_rb_pwr) or the packed block
// of d2d_step_host.
struct OutPtrs {
    int* rb = nullptr; int* pwr = nullptr;
    float *sinr = nullptr, *snr = nullptr, *rate = nullptr, *cap = nullptr, *reward = nullptr, *table = nullptr, *obs = nullptr;
    int* env_flags = nullptr;
};

// LinearObs expansion launch geometry for a [B, N, 6] table (shared by the step path and d2d_expand_table).
void make_obs_args(const d2d_handle* h, int B, int N, const float* table, float* obs, int out_f64, d2d::ObsArgs* out) {
    d2d::ObsArgs o;
    std::memset(&o, 0, sizeof(o));
    o.B = B; o.N = N;
    o.out_f64 = out_f64;
    if (out_f64 && h->tune_variant != 3) {
        // float64 block (d2d_set_obs_dtype): flat slabs of double2 pieces - one aligned float2 of T each, 3N per row, any N
        o.vec = 2;
        o.q_per_row = (unsigned)(3 * N);
        o.q_magic = ((1ull << 40) + o.q_per_row - 1) / o.q_per_row;
        o.block = h->tune_block > 0 ? h->tune_block : 1024;
        int passes = h->tune_rows > 0 ? h->tune_rows : 2;
        if (passes > 4) passes = 4;
        o.rows_per_wg = passes;
        const unsigned total = (unsigned)N * o.q_per_row, slab = (unsigned)passes * (unsigned)o.block;
        o.chunks = (int)((total + slab - 1) / slab);
        o.xcd_remap = (h->tune_xcd > 0 && B % 8 == 0) ? 1 : 0;
        o.nontemporal = h->tune_nt;
        o.variant = 2;
        o.table = table; o.obs = obs;
        *out = o;
        return;
    }
    o.vec = (6 * N) % 4 == 0 ? 4 : 2;
    o.q_per_row = (unsigned)(6 * N / o.vec);
    o.q_magic = ((1ull << 40) + o.q_per_row - 1) / o.q_per_row;
    // Launch geometry.  Round 4 (tools/probes/obs_policy_geometry.py, MI355X, N = 512): FLAT slabs - the env's [N][6N] block as one
    // flat array of float4, a 1024-thread workgroup writing two consecutive 16 KB pieces of it whatever the row length - the
    // shape of the fastest fill of the probe family: 7.21 TB/s against 7.01 for the row-aligned shape of rounds 1-3 (768 threads
    // = one row's float4 count, two 12 KB rows per workgroup) in interleaved rounds on one box; 3 or 4 pieces, 768- or 512-thread
    // slabs and every scope-bit store policy are slower.  Rounds 1-3 (tools/tune_obs.py): among row-aligned shapes the fastest
    // is the one where every thread issues exactly TWO 16-B stores (6.96 TB/s vs 5.9 for 96-KiB slabs and 5.6 for 786-KiB slabs):
    // small slabs dispatched in order keep the chip-wide write front nearly sequential in address, and a workgroup never
    // outlives its neighbours.  D2D_TUNE_OBS_VARIANT = 3 keeps the row-aligned kernel (A/B, and 8-byte rows when 6N % 4 != 0).
    const bool flat = o.vec == 4 && (h->tune_variant == 0 || h->tune_variant == 2) && !out_f64;
    int block = h->tune_block;
    if (block <= 0) {
        block = flat ? 1024 : (int)((o.q_per_row + 63) / 64) * 64;
        if (block < 256) block = 256;
        if (block > 1024) block = 1024;
    }
    int rows = h->tune_rows;
    if (rows <= 0) {
        rows = flat ? 2 : (int)((2u * (unsigned)block + o.q_per_row / 2) / o.q_per_row);
        if (rows < 1) rows = 1;
    }
    if (rows > N && !flat) rows = N;
    o.rows_per_wg = rows;
    o.chunks = (N + rows - 1) / rows;
    if (flat) {                                          // `rows` = consecutive pieces of `block` float4 per workgroup (at most 4)
        if (rows > 4) rows = 4;
        o.rows_per_wg = rows;
        const unsigned total = (unsigned)N * o.q_per_row, slab = (unsigned)rows * (unsigned)block;
        o.chunks = (int)((total + slab - 1) / slab);
    }
    o.xcd_remap = (h->tune_xcd > 0 && B % (8 * h->tune_xcd) == 0) ? h->tune_xcd : 0;   // envs interleaved per XCD
    o.nontemporal = h->tune_nt;
    o.block = block;
    // tune_variant 2 names the flat kernel, whose grid is sized in flat chunks: when the flat shape does not apply (8-byte rows, a
    // float64 block's float32 consumers) the row-aligned default runs on the row-aligned geometry computed above
    o.variant = flat ? 2 : (h->tune_variant == 2 ? 0 : h->tune_variant);
    o.stagger = h->tune_stagger;
    o.table = table;
    o.obs = obs;
    *out = o;
}

int run_step(d2d_handle* h, int action_mode, const int32_t* a0, const int32_t* a1, const OutPtrs* redirect) {
    if (!h->have_links) return fail(D2D_ERR_STATE, "d2d_set_links has not been called");
    if (!h->have_pos) return fail(D2D_ERR_STATE, "positions not set (d2d_set_positions / upload POS_X,POS_Y)");
    int rc = refresh_tables(h);
    if (rc) return rc;
    const int N = h->N, D = h->D;
    if (N == 0) return fail(D2D_ERR_INVALID, "no links: the reference divides by len(actions) (reward_fn.py:42)");

    d2d::StepArgs s;
    std::memset(&s, 0, sizeof(s));
    s.B = h->B; s.N = N; s.R = h->cfg.num_rbs; s.D = D;
    s.action_mode = action_mode;
    s.act_stride = action_mode == 0 ? N - h->n_fixed : N;
    s.col_mode = h->col_mode; s.n_fixed = h->n_fixed;
    s.inv_n = 1.0f / (float)N;
    if ((size_t)h->B * (size_t)N * 24 >= (1ull << 32))
        return fail(D2D_ERR_UNSUPPORTED, "envs x links per GPU must stay below 2^32 / 24 (32-bit byte offsets in the step kernel)");
    s.reward_fn = h->reward_fn; s.reward_param = h->reward_param;
    s.write_table = h->obs_mode != D2D_OBS_NONE;
    // nontemporal result stores: never with LinearObs (the expansion kernel reads the table right behind this launch).  In the
    // generic kernels no consistent gain (profiles/r3_ab_*): off unless asked for.  The rollout kernel takes them by itself
    // (below): interleaved in one process at 4096 x 512, obs-less 17.98 -> 17.51 us, compact table 25.11 -> 24.76
    // (profiles/r5_rollout_nt_and_links_per_thread.jsonl)
    s.nt_results = h->tune_step_nt > 0 && h->obs_mode != D2D_OBS_LINEAR;
    s.rec_uniform = h->rec_uniform && h->tune_step_srec != 0;
    s.pow_k = h->pow_k;
    s.ablate = h->tune_step_ablate;
    s.dbg = nullptr;
#if defined(D2D_STEP_ABLATE) && D2D_STEP_ABLATE
    if (h->tune_step_ablate & 8192) {          // phase stamps: [B workgroups][16 waves][16] u64 (tools/phase_times.py)
        if (!h->dbg) HIP_TRY(hipMalloc(&h->dbg, (size_t)h->B * 16 * 16 * 8));
        s.dbg = h->dbg;
    }
#endif
    // masks cover N <= 1024; beyond that the member lists (eight slots per RB) - but only where they can hold the env: with more
    // than four links per RB on average a ninth link on some RB is the rule, the list build is wasted and the workgroup
    // sweeps all pairs anyway (N > 8 R: by pigeonhole), so those shapes go straight to the sweep
    const bool lists_can_help = (long long)N <= 4ll * h->cfg.num_rbs;
    // float64 positions uploaded as (hi, lo) pairs (d2d_set_positions_f64) with a non-zero low part somewhere: the OPT_XPOS kernels
    const int xpos = h->have_lo ? 1 : 0;
    // The rollout kernel (d2d_rollout.hip) serves: raw agent actions for every link or for all but a prefix with fixed actions, any of
    // the three rewards, one env per workgroup (64 ... 1024 links: a multiple of 64, or padded to the next one; no
    // fused expansion), a power-law path loss.  Round 5: it is the faster one in every obs mode (same box, 4096 x 512, r4 HEAD ->
    // rollout: obs-less 21.3 -> 19.3 us, compact table 27.7 -> 25.8, with the decoded planes 28.3 -> 27.3,
    // profiles/r5_ab_rollout_kernel.jsonl; further since), so wherever it applies the lists are the default.
    const bool will_fuse = h->obs_mode == D2D_OBS_LINEAR && !h->obs_f64 && (h->tune_step_fuse >= 0 ? h->tune_step_fuse != 0 : N <= 128);
    const bool rollout_cfg = action_mode == 0 && (h->n_fixed == 0 || (h->n_fixed < N && h->col_mode == 0)) && h->bucketing &&   // fixed links: a prefix
                             (h->reward_fn == D2D_REWARD_SYSTEM_CAPACITY || h->reward_fn == D2D_REWARD_SHANNON ||
                              h->reward_fn == D2D_REWARD_CUE_SINR_SHANNON) &&
                             !will_fuse && (h->mode == d2d::PL_INV_SQUARE || h->mode == d2d::PL_POWER || h->mode == d2d::PL_POWK) && (h->tune_step_ablate & ~8192) == 0 &&
                             h->tune_step_prefetch != 0 && h->tune_step_threads == 0 && h->tune_step_epw <= 1 && h->tune_step_block == 0 &&
                             (N % 64 == 0 || N > 64) && N <= 1024;       // (no multiple of 64: padded; measured from 80 links up)
    const bool lists_pay = lists_can_help && (N > 1024 || h->obs_mode == D2D_OBS_NONE || rollout_cfg);
    s.walk = h->tune_step_walk >= 0 ? h->tune_step_walk : (lists_pay ? 2 : 0);

    // ---- launch geometry.  tpe threads per env (one per link up to 1024), epw envs per workgroup: small envs share a
    // workgroup (N = 50: four 64-thread envs in 256 threads), and for small N the LinearObs expansion runs inside the
    // same launch, streamed by all threads of the workgroup - two launches of a few microseconds each are bound by
    // launch latency, not by HBM.
    // lpt = links per thread held in registers (1 or 2; 0 = strided beyond 2 x 1024 links); tpe = threads per env
    int lpt = h->tune_step_lpt > 0 ? h->tune_step_lpt : 1;
    int tpe;
    if (h->tune_step_threads > 0) {
        tpe = h->tune_step_threads;
        lpt = N <= tpe ? 1 : (N <= 2 * tpe ? 2 : 0);
    } else {
        tpe = (((N + lpt - 1) / lpt + 63) / 64) * 64;
        if (tpe > 1024) { lpt = 2; tpe = (((N + 1) / 2 + 63) / 64) * 64; }
        if (tpe > 1024) { lpt = 0; tpe = 1024; }
    }
    if (tpe > 1024) tpe = 1024;
    if (tpe < 64) tpe = 64;
    int fuse = 0;
    if (h->obs_mode == D2D_OBS_LINEAR) {
        const bool want = h->tune_step_fuse >= 0 ? h->tune_step_fuse != 0 : N <= 128;
        if (want && !h->obs_f64) fuse = (6 * N) % 4 == 0 ? 4 : 2;       // the fused expansion writes float32 only
    }
    // per-RB membership masks: u32 words, every link of the env in some thread's registers, N <= 1024 (the 32-bit summary
    // word names up to 32 mask words)
    // per-RB member lists (walk 2): any N whose links sit in registers; an env that overflows a list falls back to the masks
    // when they exist, else to the all-pairs sweep
    int lists = h->bucketing && s.walk == 2 && lpt > 0 && s.reward_fn != D2D_REWARD_CUE_SINR_SHANNON;
    // (the rollout kernel has its own way with CueSinrShannon; the generic kernels' lists have none)
    const bool rollout_wanted = h->bucketing && s.walk == 2 && rollout_cfg && h->col_mode == 0;
    if (s.walk == 2 && !lists && !rollout_wanted) s.walk = 0;
    int W = 0;
    if (h->bucketing && lpt > 0 && N <= 1024) {
        W = (N + 31) / 32;
        if (d2d::step_lds_bytes_per_env(N, s.R, W, fuse, lpt, s.reward_fn, (int)h->mode, lists, xpos) > 96 * 1024) W = 0;
    }
    // masks that do not fit (thousands of RBs): the lists are 20 bytes per RB instead of 4 per RB and 32 links - take them when
    // nobody chose a search variant
    if (W == 0 && !lists && lists_can_help && h->tune_step_walk < 0 && h->bucketing && lpt > 0 && s.reward_fn != D2D_REWARD_CUE_SINR_SHANNON) { lists = 1; s.walk = 2; }
    if (lists && d2d::step_lds_bytes_per_env(N, s.R, W, fuse, lpt, s.reward_fn, (int)h->mode, lists, xpos) > 96 * 1024) { lists = 0; s.walk = 0; }
    s.lpt = lpt;
    d2d::step_lds_layout(N, s.R, W, fuse, lpt, s.reward_fn, (int)h->mode, lists, xpos, &s.lds);
    // The rollout kernel (d2d_rollout.hip): raw agent actions for every link, SystemCapacity, one env per workgroup, a power-law
    // path loss, member lists wanted.  Its own LDS layout: no masks, 17 KB per env at 512 links on 256 RBs.
    if (rollout_wanted) {
        // One link per thread, or two ADJACENT ones (links 2t and 2t + 1, N / 2 threads per env): every per-wave instruction - the
        // scalar record load, barriers, ballots, the wave reduction, the ticket - is paid once per 128 links, half as many waves are
        // launched, and a thread's two results are one 8-byte element of every plane and 48 contiguous bytes of the table.  Two
        // wherever the device classes fill aligned groups of 128 links (one scalar record load serves the wave; per-lane records
        // for two links push the kernel past 64 VGPRs).  Same box, 4096 x 512: obs-less 20.2 -> 19.0 us, compact table 26.7 -> 24.5
        // (profiles/r5_table_rows_through_lds.jsonl).
        // (Other exponents - the power-law kernel, twice the arithmetic per pair - gain nothing from two: COST-Hata obs-less 32.0 us
        // with one link per thread, 33.4 with two; table 37.3 / 37.8.)
        int rl = h->tune_step_lpt > 0 ? h->tune_step_lpt : (h->rec_uniform128 && s.rec_uniform && h->mode == d2d::PL_INV_SQUARE ? 2 : 1);
        if (rl != 2 || N % 128 != 0) rl = 1;
        if (rl == 2 && !h->rec_uniform128) s.rec_uniform = false;     // forced by the tuning key on other records: per-lane records
        if (h->n_fixed > 0) { s.rec_uniform = false; rl = 1; }         // fixed actions live in per-link records: one link per thread
        if (s.reward_fn == D2D_REWARD_CUE_SINR_SHANNON) rl = 1;        // its second look at the RB's members: one link per thread
        if (xpos) rl = 1;                                              // exact positions: one link per thread (a fourth 16-byte row per link)
        {
            d2d::StepLds rlds;
            d2d::rollout_lds_layout(N, s.R, (int)h->mode, s.reward_fn, xpos, &rlds);
            if (rlds.env_bytes <= 64 * 1024) {
                s.rollout = 1; s.lds = rlds;
                lpt = rl; tpe = ((N / rl + 63) / 64) * 64; W = 0; s.lpt = lpt;
                if (h->tune_step_nt < 0 && h->obs_mode != D2D_OBS_LINEAR) s.nt_results = 1;      // auto: on, see above
            }
        }
    }
    const size_t env_lds = s.lds.env_bytes;
    if (env_lds > 160 * 1024) return fail(D2D_ERR_UNSUPPORTED, "links per env exceed the LDS staging capacity");
    int epw = s.rollout ? 1 : h->tune_step_epw;
    if (epw <= 0) epw = tpe >= 256 ? 1 : 256 / tpe;
    if (epw > h->B) epw = h->B;
    while (epw > 1 && ((size_t)epw * env_lds > 64 * 1024 || epw * tpe > 1024)) --epw;
    int block = h->tune_step_block > 0 ? h->tune_step_block : epw * tpe;
    if (fuse && h->tune_step_block <= 0 && block < 256) block = 256;      // more store streams per env for the obs phase
    if (block < epw * tpe) block = epw * tpe;
    if (block > 1024) return fail(D2D_ERR_INVALID, "step workgroup exceeds 1024 threads");
    s.tpe = tpe; s.epw = epw; s.mask_words = W; s.fuse_obs = fuse;
    {
        // action prefetch distance = the envs resident on the chip at once (LDS: 160 KB / CU, threads: 2048 / CU), as a
        // multiple of 8 workgroups so that the prefetching and the consuming workgroup share an XCD (and its L2)
        int per_cu = (int)((160 * 1024) / ((size_t)epw * env_lds ? (size_t)epw * env_lds : 1));
        if (per_cu > 2048 / block) per_cu = 2048 / block;
        if (per_cu < 1) per_cu = 1;
        int dist = h->tune_step_prefetch >= 0 ? h->tune_step_prefetch : per_cu * h->num_cus * epw;
        dist -= dist % (8 * epw);
        s.prefetch_envs = dist;
    }
    s.tpe_magic = ((1u << 20) + (unsigned)tpe - 1) / (unsigned)tpe;
    if (fuse) {
        s.obs_rotate = h->tune_step_obs_rotate >= 0 ? h->tune_step_obs_rotate : 29;
        s.obs_q_per_row = (unsigned)(6 * N / fuse);
        s.obs_q_magic = ((1ull << 40) + s.obs_q_per_row - 1) / s.obs_q_per_row;
    }

    void* p = nullptr;
#define GET(which, field, type)                                  \
    do {                                                         \
        rc = ensure_buffer(h, which, &p);                        \
        if (rc) return rc;                                       \
        s.field = reinterpret_cast<type>(p);                     \
    } while (0)
    if (action_mode == 0) {
        if (s.act_stride == 0) s.actions = nullptr;              // every link is fixed: nothing to read
        else if (a0) s.actions = a0; else GET(D2D_BUF_ACTIONS, actions, const int*);
    } else {
        if (a0) s.rb_in = a0; else GET(D2D_BUF_RB, rb_in, const int*);
        if (a1) s.pwr_in = a1; else GET(D2D_BUF_PWR, pwr_in, const int*);
    }
    rc = ensure_buffer(h, D2D_BUF_POS_X, &p);
    if (rc) return rc;
    const float* px = static_cast<const float*>(p);
    rc = ensure_buffer(h, D2D_BUF_POS_Y, &p);
    if (rc) return rc;
    const float* py = static_cast<const float*>(p);
    if (redirect) {
        s.rb_out = redirect->rb; s.pwr_out = redirect->pwr;        // decoded / fixed values as the kernel used them
        s.sinr_db = redirect->sinr; s.snr_db = redirect->snr; s.rate = redirect->rate; s.cap = redirect->cap;
        s.env_flags = redirect->env_flags;
        if (s.reward_fn != D2D_REWARD_NONE) s.reward = redirect->reward;      // d2d_step_host: always the [B,N] rows
        if (s.write_table) s.table = redirect->table;
        if (h->obs_mode == D2D_OBS_LINEAR) s.obs = redirect->obs;
    } else {
        if (action_mode == 0 && h->export_actions) {
            GET(D2D_BUF_RB, rb_out, int*);
            GET(D2D_BUF_PWR, pwr_out, int*);
        }
        GET(D2D_BUF_SINR_DB, sinr_db, float*);
        GET(D2D_BUF_SNR_DB, snr_db, float*);
        GET(D2D_BUF_RATE_BPS, rate, float*);
        GET(D2D_BUF_CAPACITY, cap, float*);
        GET(D2D_BUF_ENV_FLAGS, env_flags, int*);
        if (s.reward_fn == D2D_REWARD_SYSTEM_CAPACITY && h->reward_layout == D2D_REWARD_PER_ENV) GET(D2D_BUF_REWARD_ENV, reward_env, float*);
        else if (s.reward_fn != D2D_REWARD_NONE) GET(D2D_BUF_REWARD, reward, float*);
        if (s.write_table) GET(D2D_BUF_OBS_TABLE, table, float*);
        if (h->obs_mode == D2D_OBS_LINEAR) GET(D2D_BUF_OBS, obs, float*);
    }
#undef GET
    rc = refresh_link_positions(h, px, py);
    if (rc) return rc;
    const int S = h->Nmax;
    s.rec_a = reinterpret_cast<const int4*>(h->rec);
    s.rec_b = h->rec + S;
    s.rec_c = h->rec + 2 * (size_t)S;
    s.rec_h = h->rec_h;
    s.rec_grp = h->rec_grp;
    s.lpos = h->lpos;
    s.lpos_lo = xpos ? h->lpos_lo : nullptr;
    s.act_cols = h->act_cols;
    s.side_words = h->side_words;
    s.gain_table = h->gain_table;
    s.table_by_link = h->table_links > 0;
    s.table_pitch = h->table_links > 0 ? h->table_links : D;
    s.table_env_stride = h->table_per_env ? (long long)s.table_pitch * s.table_pitch : 0;
    s.env_offset = h->env_offset;
    if (h->mode == d2d::PL_SHADOW) {
        s.shadow_chi = (float)h->shadow_chi;
        s.shadow_d0sq = (float)(h->shadow_d0 * h->shadow_d0);
        s.shadow_seed_lo = (unsigned)(h->shadow_seed & 0xFFFFFFFFull);
        s.shadow_seed_hi = (unsigned)(h->shadow_seed >> 32);
        s.shadow_step = (unsigned)h->shadow_step++;
    }

    EventPair* ep = nullptr;
    rc = record_start(h, 0, &ep);
    if (rc) return rc;
    HIP_TRY(d2d::launch_step(s, h->mode, block, h->stream));
    if (ep) HIP_TRY(hipEventRecord(ep->stop, h->stream));

    if (h->obs_mode == D2D_OBS_LINEAR && !fuse) {
        d2d::ObsArgs o;
        make_obs_args(h, h->B, N, s.table, s.obs, redirect ? 0 : h->obs_f64, &o);      // d2d_step_host's packed block is float32
        rc = record_start(h, 1, &ep);
        if (rc) return rc;
        HIP_TRY(d2d::launch_obs_expand(o, h->stream));
        if (ep) HIP_TRY(hipEventRecord(ep->stop, h->stream));
    }
    return D2D_OK;
}

// ---- RCCL, loaded on first use ------------------------------------------------------------------
struct Id128 { char bytes[D2D_UNIQUE_ID_BYTES]; };     // ncclUniqueId: passed BY VALUE to ncclCommInitRank
struct Rccl {
    void* lib = nullptr;
    int (*GetUniqueId)(void*) = nullptr;
    int (*CommInitRank)(void**, int, Id128, int) = nullptr;
    int (*CommDestroy)(void*) = nullptr;
    int (*AllGather)(const void*, void*, size_t, int, void*, hipStream_t) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
};
Rccl g_rccl;

int load_rccl() {
    if (g_rccl.lib) return D2D_OK;
    const char* override_path = std::getenv("D2D_RCCL_LIBRARY");
    const char* names[] = {override_path, "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void* lib = nullptr;
    // a process that already has RCCL mapped (torch.distributed) must share that copy, not load a second one
    for (const char* n : names) if (n && !lib) lib = dlopen(n, RTLD_NOW | RTLD_NOLOAD);
    for (const char* n : names) if (n && !lib) lib = dlopen(n, RTLD_NOW | RTLD_LOCAL);
    if (!lib) {
        const char* why = dlerror();              // one call: dlerror() clears the message it returns
        return fail(D2D_ERR_UNSUPPORTED, std::string("librccl not found: ") + (why ? why : "?"));
    }
    g_rccl.GetUniqueId = reinterpret_cast<decltype(g_rccl.GetUniqueId)>(dlsym(lib, "ncclGetUniqueId"));
    g_rccl.CommInitRank = reinterpret_cast<decltype(g_rccl.CommInitRank)>(dlsym(lib, "ncclCommInitRank"));
    g_rccl.CommDestroy = reinterpret_cast<decltype(g_rccl.CommDestroy)>(dlsym(lib, "ncclCommDestroy"));
    g_rccl.AllGather = reinterpret_cast<decltype(g_rccl.AllGather)>(dlsym(lib, "ncclAllGather"));
    g_rccl.GetErrorString = reinterpret_cast<decltype(g_rccl.GetErrorString)>(dlsym(lib, "ncclGetErrorString"));
    if (!g_rccl.GetUniqueId || !g_rccl.CommInitRank || !g_rccl.CommDestroy || !g_rccl.AllGather)
        return fail(D2D_ERR_UNSUPPORTED, "librccl lacks ncclGetUniqueId / ncclCommInitRank / ncclCommDestroy / ncclAllGather");
    g_rccl.lib = lib;
    return D2D_OK;
}

int rccl_fail(const char* what, int code) {
    return fail(D2D_ERR_HIP, std::string(what) + ": " + (g_rccl.GetErrorString ? g_rccl.GetErrorString(code) : "rccl error") +
                                 " (" + std::to_string(code) + ")");
}

// No C++ exception crosses the C ABI (Errors: the reference raises Python exceptions only - d2d_env.py:100,
// simulator.py:56,74 - and a C or ctypes caller cannot catch ours): every entry point below is a function-try-block whose
// handlers turn std::bad_alloc (host vectors sized by caller input) into D2D_ERR_NO_MEMORY and anything else into D2D_ERR_STATE.
int caught(int code, const char* what) noexcept {
    try { g_last_error = what; } catch (...) { }         // the message itself may not fit: the code still goes back
    return code;
}
#define D2D_CATCH                                                                                          \
    catch (const std::bad_alloc&) { return caught(D2D_ERR_NO_MEMORY, "out of host memory"); }              \
    catch (const std::exception& e) { return caught(D2D_ERR_STATE, e.what()); }                            \
    catch (...) { return caught(D2D_ERR_STATE, "unknown C++ exception"); }

}  // namespace

extern "C" {

int d2d_abi_version(void) { return D2D_ABI_VERSION; }

const char* d2d_last_error(void) { return g_last_error.c_str(); }

int d2d_create(const d2d_config* cfg, d2d_handle** out) try {
    if (!cfg || !out) return fail(D2D_ERR_INVALID, "null argument");
    *out = nullptr;
#if defined(D2D_TEST_HOOKS) && D2D_TEST_HOOKS      // tests/test_sanitizers_cpu.py only: does the function-try-block hold?
    if (cfg->num_envs == -12345) throw std::bad_alloc();
    if (cfg->num_envs == -12346) throw std::runtime_error("test hook");
    if (cfg->num_envs == -12347) throw 42;
#endif
    if (cfg->abi_version != D2D_ABI_VERSION) return fail(D2D_ERR_INVALID, "abi_version mismatch");
    if (cfg->num_envs < 1 || cfg->num_rbs < 1 || cfg->num_cues < 0 || cfg->num_due_pairs < 0)
        return fail(D2D_ERR_INVALID, "num_envs/num_rbs must be >= 1 and device counts >= 0");
    if (cfg->pwr_levels_due < 1 || cfg->pwr_levels_cue < 1 || cfg->pwr_levels_mbs < 1)
        return fail(D2D_ERR_INVALID, "power level counts must be >= 1");
    if (cfg->pwr_levels_due > 65535 || cfg->pwr_levels_cue > 65535 || cfg->pwr_levels_mbs > 65535)
        return fail(D2D_ERR_INVALID, "power level counts must be <= 65535 (16 bits of the link record)");
    int nmax = cfg->max_links > 0 ? cfg->max_links : cfg->num_cues + cfg->num_due_pairs;
    if (nmax < 1 || nmax > D2D_MAX_LINKS)
        return fail(D2D_ERR_INVALID, "max_links must be in [1, " + std::to_string(D2D_MAX_LINKS) + "]");
    int count = 0;
    HIP_TRY(hipGetDeviceCount(&count));
    if (cfg->device_ordinal < 0 || cfg->device_ordinal >= count)
        return fail(D2D_ERR_HIP, "device_ordinal out of range (no usable GPU?)");
    DeviceGuard device_guard_(cfg->device_ordinal);
    if (device_guard_.err != hipSuccess) return fail(D2D_ERR_HIP, std::string("hipSetDevice: ") + hipGetErrorString(device_guard_.err));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, cfg->device_ordinal));
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(D2D_ERR_UNSUPPORTED, std::string("libd2d_hip is built for gfx950 only, found ") + prop.gcnArchName);

    d2d_handle* h = new (std::nothrow) d2d_handle();
    if (!h) return fail(D2D_ERR_NO_MEMORY, "out of host memory");
    h->cfg = *cfg;
    h->B = cfg->num_envs;
    h->D = 1 + cfg->num_cues + 2 * cfg->num_due_pairs;
    h->Nmax = nmax;
    h->N = 0;
    h->num_cus = prop.multiProcessorCount;
#define CREATE_TRY(expr)                                                                       \
    do {                                                                                       \
        hipError_t _e = (expr);                                                                \
        if (_e != hipSuccess) {                                                                \
            d2d_destroy(h);                                                                    \
            return fail(D2D_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e));      \
        }                                                                                      \
    } while (0)
    CREATE_TRY(hipStreamCreateWithFlags(&h->own_stream, hipStreamNonBlocking));
    h->stream = h->own_stream;
    CREATE_TRY(hipMalloc(&h->rec, (size_t)3 * h->Nmax * 16));
    CREATE_TRY(hipMalloc(&h->rec_h, (size_t)h->Nmax * 8));
    CREATE_TRY(hipMalloc(&h->rec_grp, ((size_t)h->Nmax / 64 + 1) * 64));
    CREATE_TRY(hipMalloc(&h->lpos, (size_t)h->B * h->Nmax * 16));
    CREATE_TRY(hipMalloc(&h->act_cols, (size_t)h->Nmax * 4));
    CREATE_TRY(hipMalloc(&h->side_words, ((size_t)(h->Nmax + 31) / 32 + 1) * 4));
    CREATE_TRY(hipMalloc(&h->status, 4));
    CREATE_TRY(hipMemset(h->status, 0, 4));
#undef CREATE_TRY
    *out = h;
    return D2D_OK;
} D2D_CATCH

int d2d_destroy(d2d_handle* h) try {
    if (!h) return D2D_OK;
    DeviceGuard device_guard_(h->cfg.device_ordinal);
    if (h->stream) hipStreamSynchronize(h->stream);
    if (h->comm && g_rccl.CommDestroy) {
        if (h->comm_used) (void)hipDeviceSynchronize();
        g_rccl.CommDestroy(h->comm);
    }
    for (auto& ep : h->events) { hipEventDestroy(ep.start); hipEventDestroy(ep.stop); }
    for (auto& bf : h->buf)
        if (bf.ptr && bf.owned) hipFree(bf.ptr);
    if (h->rec) hipFree(h->rec);
    if (h->rec_h) hipFree(h->rec_h);
    if (h->rec_grp) hipFree(h->rec_grp);
    if (h->lpos) hipFree(h->lpos);
    if (h->lpos_lo) hipFree(h->lpos_lo);
    if (h->pos_lo) hipFree(h->pos_lo);
    if (h->dbg) hipFree(h->dbg);
    if (h->act_cols) hipFree(h->act_cols);
    if (h->side_words) hipFree(h->side_words);
    if (h->host_out_dev) hipFree(h->host_out_dev);
    if (h->host_out_pinned) hipHostFree(h->host_out_pinned);
    if (h->host_in_dev) hipFree(h->host_in_dev);
    if (h->host_in_pinned) hipHostFree(h->host_in_pinned);
    if (h->gain_table) hipFree(h->gain_table);
    if (h->status) hipFree(h->status);
    if (h->fixed_mask_dev) hipFree(h->fixed_mask_dev);
    if (h->fixed_xy_dev) hipFree(h->fixed_xy_dev);
    if (h->own_stream) hipStreamDestroy(h->own_stream);
    delete h;
    return D2D_OK;
} D2D_CATCH

int d2d_set_stream(d2d_handle* h, void* hip_stream) try {
    if (!h) return fail(D2D_ERR_INVALID, "null handle");
    USE_DEVICE(h);
    HIP_TRY(hipStreamSynchronize(h->stream));
    // NULL is a real stream (the device's legacy default stream, which is what torch's default stream is): work
    // enqueued there is ordered with the caller's own kernels and copies.  Only the sentinel selects the private one.
    h->stream = hip_stream == D2D_STREAM_PRIVATE ? h->own_stream : reinterpret_cast<hipStream_t>(hip_stream);
    return D2D_OK;
} D2D_CATCH

int d2d_synchronize(d2d_handle* h) try {
    if (!h) return fail(D2D_ERR_INVALID, "null handle");
    USE_DEVICE(h);
    HIP_TRY(hipStreamSynchronize(h->stream));
    return D2D_OK;
} D2D_CATCH

int d2d_set_device_table(d2d_handle* h, int32_t n_dev, const double* eirp_off_db, const double* rx_off_db,
                         const double* noise_dbm, const double* sens_dbm, const double* bw_hz) try {
    if (!h || !eirp_off_db || !rx_off_db || !noise_dbm || !sens_dbm || !bw_hz) return fail(D2D_ERR_INVALID, "null argument");
    if (n_dev != h->D) return fail(D2D_ERR_INVALID, "n_dev must be 1 + num_cues + 2*num_due_pairs = " + std::to_string(h->D));
    h->eirp_off.assign(eirp_off_db, eirp_off_db + n_dev);
    h->rx_off.assign(rx_off_db, rx_off_db + n_dev);
    h->noise.assign(noise_dbm, noise_dbm + n_dev);
    h->sens.assign(sens_dbm, sens_dbm + n_dev);
    h->bw.assign(bw_hz, bw_hz + n_dev);
    h->have_dev = true; h->tables_dirty = true;
    return D2D_OK;
} D2D_CATCH

int d2d_set_path_loss_power_law(d2d_handle* h, int32_t n_dev, const double* a_tx_db, const double* a_rx_db,
                                const double* exponent) try {
    if (!h || !a_tx_db || !a_rx_db || !exponent) return fail(D2D_ERR_INVALID, "null argument");
    if (n_dev != h->D) return fail(D2D_ERR_INVALID, "n_dev must be " + std::to_string(h->D));
    for (int d = 0; d < n_dev; ++d)
        if (!(exponent[d] > 0.0) || !std::isfinite(a_tx_db[d]) || !std::isfinite(a_rx_db[d]))
            return fail(D2D_ERR_INVALID, "path-loss exponent must be > 0 and constants finite");
    h->a_tx.assign(a_tx_db, a_tx_db + n_dev);
    h->a_rx.assign(a_rx_db, a_rx_db + n_dev);
    h->expo.assign(exponent, exponent + n_dev);
    h->mode = d2d::PL_POWER;   // refined to PL_INV_SQUARE in refresh_tables when every exponent is 2
    h->have_pl = true; h->tables_dirty = true;
    return D2D_OK;
} D2D_CATCH

int d2d_set_path_loss_shadowing(d2d_handle* h, int32_t n_dev, const double* a_tx_db, const double* a_rx_db,
                                const double* exponent, double d0_m, double chi_db, uint64_t seed) try {
    if (!(d0_m >= 0.0) || !(chi_db >= 0.0)) return fail(D2D_ERR_INVALID, "d0_m and chi_db must be >= 0");
    int rc = d2d_set_path_loss_power_law(h, n_dev, a_tx_db, a_rx_db, exponent);
    if (rc) return rc;
    h->mode = d2d::PL_SHADOW;
    h->shadow_d0 = d0_m; h->shadow_chi = chi_db; h->shadow_seed = seed; h->shadow_step = 0;
    return D2D_OK;
} D2D_CATCH

namespace {
// dB -> linear gain (in DOUBLE, rounded once: a float32 dB value is off by up to 3.8e-6 dB near 100 dB, which alone is most of
// the 1e-5 bar) of `elems` table entries into the handle's device table, through two pinned staging blocks of at most 1 Mi
// floats each: the host never holds a second copy of the caller's table (9.7 GB for a per-env table at BASELINE config 3
// sizes before round 5), and the conversion of one chunk overlaps the copy of the other.
// From the first byte a new table overwrites until it is complete, the handle has NO table: a failed allocation or copy must not
// leave PL_TABLE selected over a null or half-written one (the next step then fails with D2D_ERR_STATE instead of faulting).
void gain_table_invalid(d2d_handle* h) {
    if (h->mode == d2d::PL_TABLE) h->have_pl = false;
    h->table_links = 0;
    h->tables_dirty = true;
}

int reserve_gain_table(d2d_handle* h, size_t elems) {
    HIP_TRY(hipStreamSynchronize(h->stream));
    gain_table_invalid(h);
    if (h->gain_elems < elems) {
        if (h->gain_table) HIP_TRY(hipFree(h->gain_table));
        h->gain_table = nullptr; h->gain_elems = 0;
        HIP_TRY(hipMalloc(&h->gain_table, elems * 4));
        h->gain_elems = elems;
    }
    return D2D_OK;
}

int upload_gain_table(d2d_handle* h, const double* pl_db, size_t elems) {
    int rc0 = reserve_gain_table(h, elems);
    if (rc0) return rc0;
    const size_t chunk = std::min<size_t>(elems, (size_t)1 << 20);
    float* stage[2] = {nullptr, nullptr};
    hipEvent_t done[2] = {nullptr, nullptr};
    hipError_t err = hipSuccess;
    for (int k = 0; k < 2 && err == hipSuccess; ++k) {
        err = hipHostMalloc(reinterpret_cast<void**>(&stage[k]), chunk * 4, hipHostMallocDefault);
        if (err == hipSuccess) err = hipEventCreateWithFlags(&done[k], hipEventDisableTiming);
    }
    size_t at = 0;
    for (int c = 0; at < elems && err == hipSuccess; ++c) {
        const int k = c & 1;
        const size_t n = std::min(chunk, elems - at);
        if (c >= 2) err = hipEventSynchronize(done[k]);                // the copy that last read this block has finished
        if (err != hipSuccess) break;
        for (size_t e = 0; e < n; ++e) stage[k][e] = (float)std::pow(10.0, -pl_db[at + e] / 10.0);
        err = hipMemcpyAsync(h->gain_table + at, stage[k], n * 4, hipMemcpyHostToDevice, h->stream);
        if (err == hipSuccess) err = hipEventRecord(done[k], h->stream);
        at += n;
    }
    if (err == hipSuccess) err = hipStreamSynchronize(h->stream);
    for (int k = 0; k < 2; ++k) {                                       // every exit releases what was allocated
        if (done[k]) (void)hipEventDestroy(done[k]);
        if (stage[k]) (void)hipHostFree(stage[k]);
    }
    if (err != hipSuccess) return fail(D2D_ERR_HIP, std::string("path-loss table upload: ") + hipGetErrorString(err));
    return D2D_OK;
}
}  // namespace

int d2d_set_path_loss_table(d2d_handle* h, const double* pl_db, int32_t per_env) try {
    if (!h || !pl_db) return fail(D2D_ERR_INVALID, "null argument");
    USE_DEVICE(h);
    const size_t elems = (size_t)h->D * h->D * (per_env ? (size_t)h->B : 1);
    int rc = upload_gain_table(h, pl_db, elems);
    if (rc) return rc;
    h->table_per_env = per_env ? 1 : 0;
    h->table_links = 0;
    h->mode = d2d::PL_TABLE;
    h->have_pl = true; h->tables_dirty = true;
    return D2D_OK;
} D2D_CATCH

int d2d_set_path_loss_link_table_dev(d2d_handle* h, const void* pl_db_dev, int32_t dtype, int32_t n_links, int32_t per_env) try {
    if (!h || !pl_db_dev) return fail(D2D_ERR_INVALID, "null argument");
    if (dtype != D2D_F32 && dtype != D2D_F64) return fail(D2D_ERR_INVALID, "dtype must be D2D_F32 or D2D_F64");
    USE_DEVICE(h);
    if (!h->have_links) return fail(D2D_ERR_STATE, "d2d_set_links first: the table is indexed by its link list");
    if (n_links != h->N || n_links < 1) return fail(D2D_ERR_INVALID, "n_links must be the length of the current link list (" + std::to_string(h->N) + ")");
    const size_t elems = (size_t)n_links * n_links * (per_env ? (size_t)h->B : 1);
    int rc = reserve_gain_table(h, elems);
    if (rc) return rc;
    HIP_TRY(d2d::launch_gain_from_db(pl_db_dev, dtype == D2D_F64, elems, h->gain_table, h->num_cus, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));                 // the caller's table may be freed when this returns
    h->table_per_env = per_env ? 1 : 0;
    h->table_links = n_links;
    h->mode = d2d::PL_TABLE;
    h->have_pl = true; h->tables_dirty = true;
    return D2D_OK;
} D2D_CATCH

int d2d_set_path_loss_link_table(d2d_handle* h, const double* pl_db, int32_t n_links, int32_t per_env) try {
    if (!h || !pl_db) return fail(D2D_ERR_INVALID, "null argument");
    USE_DEVICE(h);
    if (!h->have_links) return fail(D2D_ERR_STATE, "d2d_set_links first: the table is indexed by its link list");
    if (n_links != h->N || n_links < 1) return fail(D2D_ERR_INVALID, "n_links must be the length of the current link list (" + std::to_string(h->N) + ")");
    const size_t elems = (size_t)n_links * n_links * (per_env ? (size_t)h->B : 1);
    int rc = upload_gain_table(h, pl_db, elems);
    if (rc) return rc;
    h->table_per_env = per_env ? 1 : 0;
    h->table_links = n_links;
    h->mode = d2d::PL_TABLE;
    h->have_pl = true; h->tables_dirty = true;
    return D2D_OK;
} D2D_CATCH

int d2d_set_links(d2d_handle* h, int32_t n_links, const int32_t* tx_dev, const int32_t* rx_dev, const int32_t* link_type) try {
    if (!h) return fail(D2D_ERR_INVALID, "null handle");
    USE_DEVICE(h);
    if (n_links < 0 || n_links > h->Nmax) return fail(D2D_ERR_INVALID, "n_links must be in [0, max_links]");
    if (n_links > 0 && (!tx_dev || !rx_dev || !link_type)) return fail(D2D_ERR_INVALID, "null argument");
    for (int i = 0; i < n_links; ++i) {
        if (tx_dev[i] < 0 || tx_dev[i] >= h->D || rx_dev[i] < 0 || rx_dev[i] >= h->D)
            return fail(D2D_ERR_INVALID, "link " + std::to_string(i) + ": device index out of range");
        if (link_type[i] < D2D_UPLINK || link_type[i] > D2D_SIDELINK)
            return fail(D2D_ERR_INVALID, "link " + std::to_string(i) + ": link_type must be 1, 2 or 3");
    }
    HIP_TRY(hipStreamSynchronize(h->stream));
    h->N = n_links;
    h->host_tx.assign(tx_dev, tx_dev + n_links);
    h->host_rx.assign(rx_dev, rx_dev + n_links);
    h->host_type.assign(link_type, link_type + n_links);
    h->fixed_rb.assign((size_t)n_links, INT32_MIN);
    h->fixed_pwr.assign((size_t)n_links, 0);
    h->n_fixed = 0;
    h->tables_dirty = true;          // the per-link records follow the link table (uploaded by the next step)
    h->lpos_dirty = true;
    h->have_links = true;
    if (h->table_links) {            // a table indexed by the OLD link list: gone with it
        h->table_links = 0;
        if (h->mode == d2d::PL_TABLE) h->have_pl = false;
    }
    return D2D_OK;
} D2D_CATCH

int d2d_set_fixed_actions(d2d_handle* h, int32_t n_fixed, const int32_t* link_idx, const int32_t* rb, const int32_t* pwr_dbm) try {
    if (!h) return fail(D2D_ERR_INVALID, "null handle");
    if (!h->have_links) return fail(D2D_ERR_STATE, "d2d_set_links first: fixed actions refer to its link list");
    if (n_fixed < 0 || n_fixed > h->N) return fail(D2D_ERR_INVALID, "n_fixed must be in [0, n_links]");
    if (n_fixed > 0 && (!link_idx || !rb || !pwr_dbm)) return fail(D2D_ERR_INVALID, "null argument");
    std::vector<int> frb((size_t)h->N, INT32_MIN), fpw((size_t)h->N, 0);
    for (int k = 0; k < n_fixed; ++k) {
        const int i = link_idx[k];
        if (i < 0 || i >= h->N) return fail(D2D_ERR_INVALID, "fixed action " + std::to_string(k) + ": link index out of range");
        if (frb[i] != INT32_MIN) return fail(D2D_ERR_INVALID, "fixed action " + std::to_string(k) + ": link listed twice");
        if (rb[k] == INT32_MIN) return fail(D2D_ERR_INVALID, "rb out of range");
        frb[i] = rb[k]; fpw[i] = pwr_dbm[k];
    }
    h->fixed_rb.swap(frb); h->fixed_pwr.swap(fpw);
    h->n_fixed = n_fixed;
    h->tables_dirty = true;
    return D2D_OK;
} D2D_CATCH

int d2d_positions_changed(d2d_handle* h) try {
    if (!h) return fail(D2D_ERR_INVALID, "null handle");
    h->lpos_dirty = true;
    h->have_lo = false;            // whoever wrote POS_X / POS_Y wrote float32 coordinates
    return D2D_OK;
} D2D_CATCH

int d2d_set_reward(d2d_handle* h, int32_t reward_fn, float param) try {
    if (!h) return fail(D2D_ERR_INVALID, "null handle");
    USE_DEVICE(h);
    if (reward_fn < D2D_REWARD_NONE || reward_fn > D2D_REWARD_CUE_SINR_SHANNON) return fail(D2D_ERR_INVALID, "unknown reward_fn");
    h->reward_fn = reward_fn; h->reward_param = param;
    return D2D_OK;
} D2D_CATCH

int d2d_set_reward_layout(d2d_handle* h, int32_t layout) try {
    if (!h) return fail(D2D_ERR_INVALID, "null handle");
    if (layout != D2D_REWARD_PER_AGENT && layout != D2D_REWARD_PER_ENV) return fail(D2D_ERR_INVALID, "unknown reward layout");
    h->reward_layout = layout;
    return D2D_OK;
} D2D_CATCH

int d2d_set_obs_dtype(d2d_handle* h, int32_t dtype) try {
    if (!h) return fail(D2D_ERR_INVALID, "null handle");
    if (dtype != D2D_F32 && dtype != D2D_F64) return fail(D2D_ERR_INVALID, "obs dtype must be D2D_F32 or D2D_F64");
    USE_DEVICE(h);
    if ((dtype == D2D_F64) != (h->obs_f64 != 0)) {
        Buffer& bf = h->buf[D2D_BUF_OBS];                             // an owned block of the other width is dropped; a bound one is re-checked at the next step
        if (bf.ptr && bf.owned) { HIP_TRY(hipStreamSynchronize(h->stream)); HIP_TRY(hipFree(bf.ptr)); bf.ptr = nullptr; bf.bytes = 0; }
    }
    h->obs_f64 = dtype == D2D_F64;
    return D2D_OK;
} D2D_CATCH

int d2d_set_obs_mode(d2d_handle* h, int32_t obs_mode) try {
    if (!h) return fail(D2D_ERR_INVALID, "null handle");
    USE_DEVICE(h);
    if (obs_mode < D2D_OBS_NONE || obs_mode > D2D_OBS_LINEAR) return fail(D2D_ERR_INVALID, "unknown obs_mode");
    h->obs_mode = obs_mode;
    return D2D_OK;
} D2D_CATCH

int d2d_set_export_actions(d2d_handle* h, int32_t enabled) try {
    if (!h) return fail(D2D_ERR_INVALID, "null handle");
    h->export_actions = enabled ? 1 : 0;
    return D2D_OK;
} D2D_CATCH

int d2d_set_bucketing(d2d_handle* h, int32_t enabled) try {
    if (!h) return fail(D2D_ERR_INVALID, "null handle");
    USE_DEVICE(h);
    h->bucketing = enabled ? 1 : 0;
    return D2D_OK;
} D2D_CATCH

#if defined(D2D_STEP_ABLATE) && D2D_STEP_ABLATE
// diagnostic builds only (not in include/d2d_hip.h): copy the phase stamps of the last step to the host
extern "C" int d2d_debug_stamps(d2d_handle* h, void* host, size_t bytes) try {
    if (!h || !h->dbg) return fail(D2D_ERR_STATE, "no stamps: set D2D_TUNE_STEP_ABLATE bit 8192 and step first");
    USE_DEVICE(h);
    HIP_TRY(hipStreamSynchronize(h->stream));
    if (!host || bytes > (size_t)h->B * 16 * 16 * 8) return fail(D2D_ERR_INVALID, "stamps: at most B x 16 waves x 16 stamps x 8 bytes");
    HIP_TRY(hipMemcpy(host, h->dbg, bytes, hipMemcpyDeviceToHost));
    return D2D_OK;
} D2D_CATCH
#endif

int d2d_set_tuning(d2d_handle* h, int32_t key, int32_t value) try {
    if (!h) return fail(D2D_ERR_INVALID, "null handle");
    USE_DEVICE(h);
#if !(defined(D2D_DIAG) && D2D_DIAG)
    // the A/B shapes and the ablation switch of d2d_hip_diag.h exist in diagnostic builds only
    const bool diag_key = key == D2D_TUNE_OBS_VARIANT || key == D2D_TUNE_OBS_STAGGER || key == D2D_TUNE_STEP_ABLATE;
    const bool diag_value = (key == D2D_TUNE_OBS_NONTEMPORAL && value > 1) || (key == D2D_TUNE_STEP_WALK && value == 1);
    if ((diag_key && value != 0) || diag_value)
        return fail(D2D_ERR_UNSUPPORTED, "this tuning key / value needs the diagnostic build (D2D_BUILD_DIAG=1 python -m gym_d2d_amd.build; include/d2d_hip_diag.h)");
#endif
    switch (key) {
        case D2D_TUNE_OBS_ROWS_PER_WG: h->tune_rows = value; break;
        case D2D_TUNE_OBS_NONTEMPORAL:
            if (value < 0 || value > 5) return fail(D2D_ERR_INVALID, "obs store policy must be in [0, 5]");
            h->tune_nt = value;
            break;
        case D2D_TUNE_OBS_XCD_REMAP: h->tune_xcd = value < 0 ? 0 : value; break;
        case D2D_TUNE_OBS_VARIANT: h->tune_variant = value; break;
        case D2D_TUNE_OBS_STAGGER:
            if (value < 0 || value > 64) return fail(D2D_ERR_INVALID, "stagger must be in [0, 64]");
            h->tune_stagger = value;
            break;
        case D2D_TUNE_STEP_THREADS:
            if (value != 0 && (value < 64 || value > 1024 || value % 64)) return fail(D2D_ERR_INVALID, "threads must be a multiple of 64 in [64,1024]");
            h->tune_step_threads = value;
            break;
        case D2D_TUNE_OBS_BLOCK:
            if (value != 0 && (value < 64 || value > 1024 || value % 64)) return fail(D2D_ERR_INVALID, "block must be a multiple of 64 in [64,1024]");
            h->tune_block = value;
            break;
        case D2D_TUNE_STEP_ENVS_PER_WG:
            if (value < 0 || value > 16) return fail(D2D_ERR_INVALID, "envs per workgroup must be in [0,16]");
            h->tune_step_epw = value;
            break;
        case D2D_TUNE_STEP_BLOCK:
            if (value != 0 && (value < 64 || value > 1024 || value % 64)) return fail(D2D_ERR_INVALID, "block must be a multiple of 64 in [64,1024]");
            h->tune_step_block = value;
            break;
        case D2D_TUNE_STEP_PREFETCH:
            if (value < -1) return fail(D2D_ERR_INVALID, "prefetch distance must be >= -1");
            h->tune_step_prefetch = value;
            break;
        case D2D_TUNE_STEP_LPT:
            if (value != -1 && value != 1 && value != 2) return fail(D2D_ERR_INVALID, "links per thread must be -1 (auto), 1 or 2");
            h->tune_step_lpt = value;
            break;
        case D2D_TUNE_STEP_WALK:
            if (value < -1 || value > 2) return fail(D2D_ERR_INVALID, "walk must be -1, 0, 1 or 2");
            h->tune_step_walk = value;
            break;
        case D2D_TUNE_STEP_ABLATE:
#if defined(D2D_STEP_ABLATE) && D2D_STEP_ABLATE
            h->tune_step_ablate = value;
            break;
#else
            if (value != 0) return fail(D2D_ERR_UNSUPPORTED, "D2D_TUNE_STEP_ABLATE needs the diagnostic build (D2D_BUILD_DIAG=1 python -m gym_d2d_amd.build)");
            break;
#endif
        case D2D_TUNE_STEP_OBS_ROTATE:
            if (value < -1 || value > 65535) return fail(D2D_ERR_INVALID, "obs_rotate must be in [-1, 65535]");
            h->tune_step_obs_rotate = value;
            break;
        case D2D_TUNE_STEP_NT_RESULTS:
            if (value < -1 || value > 1) return fail(D2D_ERR_INVALID, "nt_results must be -1, 0 or 1");
            h->tune_step_nt = value;
            break;
        case D2D_TUNE_STEP_SCALAR_RECORDS:
            if (value < -1 || value > 1) return fail(D2D_ERR_INVALID, "scalar_records must be -1, 0 or 1");
            h->tune_step_srec = value;
            break;
        case D2D_TUNE_STEP_FUSE_OBS:
            if (value < -1 || value > 1) return fail(D2D_ERR_INVALID, "fuse_obs must be -1, 0 or 1");
            h->tune_step_fuse = value;
            break;
        default: return fail(D2D_ERR_INVALID, "unknown tuning key");
    }
    return D2D_OK;
} D2D_CATCH

int d2d_get_buffer(d2d_handle* h, int32_t which, void** dev_ptr, size_t* bytes) try {
    if (!h || !dev_ptr) return fail(D2D_ERR_INVALID, "null argument");
    USE_DEVICE(h);
    if (which < 0 || which >= D2D_BUF_COUNT) return fail(D2D_ERR_INVALID, "unknown buffer");
    if (which == D2D_BUF_OBS && h->N == 0) return fail(D2D_ERR_STATE, "set links before asking for the obs buffer");
    int rc;
    if (which == D2D_BUF_LINK_POS) {
        // the rows follow POS_X / POS_Y lazily (refresh_link_positions): bring them up to date for the reader
        if (!h->have_links || !h->have_pos) return fail(D2D_ERR_STATE, "link positions need d2d_set_links and positions first");
        rc = refresh_tables(h);
        if (rc) return rc;
        void *px = nullptr, *py = nullptr;
        rc = ensure_buffer(h, D2D_BUF_POS_X, &px);
        if (rc) return rc;
        rc = ensure_buffer(h, D2D_BUF_POS_Y, &py);
        if (rc) return rc;
        rc = refresh_link_positions(h, static_cast<const float*>(px), static_cast<const float*>(py));
        if (rc) return rc;
    }
    rc = ensure_buffer(h, which, dev_ptr);
    if (rc) return rc;
    if (bytes) *bytes = active_bytes(h, which, h->N ? h->N : h->Nmax);
    return D2D_OK;
} D2D_CATCH

int d2d_bind_buffer(d2d_handle* h, int32_t which, void* dev_ptr, size_t bytes) try {
    if (!h) return fail(D2D_ERR_INVALID, "null handle");
    USE_DEVICE(h);
    if (which < 0 || which >= D2D_BUF_COUNT) return fail(D2D_ERR_INVALID, "unknown buffer");
    if (which == D2D_BUF_LINK_POS) return fail(D2D_ERR_INVALID, "D2D_BUF_LINK_POS is derived from POS_X / POS_Y by the library: read-only");
    Buffer& bf = h->buf[which];
    HIP_TRY(hipStreamSynchronize(h->stream));
    if (bf.ptr && bf.owned) HIP_TRY(hipFree(bf.ptr));
    bf.ptr = dev_ptr; bf.bytes = dev_ptr ? bytes : 0; bf.owned = false;
    if (which == D2D_BUF_POS_X || which == D2D_BUF_POS_Y) {
        h->have_pos = h->buf[D2D_BUF_POS_X].ptr && h->buf[D2D_BUF_POS_Y].ptr;
        h->lpos_dirty = true;
        h->have_lo = false;        // float32 coordinates from here on (d2d_set_positions_f64 raises it again behind its own uploads)
    }
    return D2D_OK;
} D2D_CATCH

int d2d_upload(d2d_handle* h, int32_t which, const void* host_src, size_t bytes, size_t dst_offset) try {
    if (!h || !host_src) return fail(D2D_ERR_INVALID, "null argument");
    USE_DEVICE(h);
    if (which < 0 || which >= D2D_BUF_COUNT) return fail(D2D_ERR_INVALID, "unknown buffer");
    if (which == D2D_BUF_LINK_POS) return fail(D2D_ERR_INVALID, "D2D_BUF_LINK_POS is derived from POS_X / POS_Y by the library: read-only");
    void* p = nullptr;
    int rc = ensure_buffer(h, which, &p);
    if (rc) return rc;
    if (dst_offset + bytes > h->buf[which].bytes) return fail(D2D_ERR_INVALID, "upload out of range");
    HIP_TRY(hipMemcpyAsync(static_cast<char*>(p) + dst_offset, host_src, bytes, hipMemcpyHostToDevice, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    if (which == D2D_BUF_POS_X || which == D2D_BUF_POS_Y) {
        h->have_pos = h->buf[D2D_BUF_POS_X].ptr && h->buf[D2D_BUF_POS_Y].ptr;
        h->lpos_dirty = true;
        h->have_lo = false;        // float32 coordinates from here on (d2d_set_positions_f64 raises it again behind its own uploads)
    }
    return D2D_OK;
} D2D_CATCH

int d2d_download(d2d_handle* h, int32_t which, void* host_dst, size_t bytes, size_t src_offset) try {
    if (!h || !host_dst) return fail(D2D_ERR_INVALID, "null argument");
    USE_DEVICE(h);
    if (which < 0 || which >= D2D_BUF_COUNT) return fail(D2D_ERR_INVALID, "unknown buffer");
    Buffer bf = h->buf[which];
    if (which == D2D_BUF_LINK_POS) {
        size_t n = 0;
        int rc = d2d_get_buffer(h, which, &bf.ptr, &n);
        if (rc) return rc;
        bf.bytes = n;
    }
    if (!bf.ptr) return fail(D2D_ERR_STATE, "buffer has never been written");
    if (src_offset + bytes > bf.bytes) return fail(D2D_ERR_INVALID, "download out of range");
    HIP_TRY(hipMemcpyAsync(host_dst, static_cast<const char*>(bf.ptr) + src_offset, bytes, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    return D2D_OK;
} D2D_CATCH

int d2d_set_positions(d2d_handle* h, const float* x, const float* y, int32_t env_begin, int32_t env_count) try {
    if (!h || !x || !y) return fail(D2D_ERR_INVALID, "null argument");
    if (env_begin < 0 || env_count < 0 || env_begin + env_count > h->B) return fail(D2D_ERR_INVALID, "env range out of bounds");
    const size_t off = (size_t)env_begin * h->D * 4, bytes = (size_t)env_count * h->D * 4;
    const bool keep_lo = h->have_lo && env_count < h->B;     // other envs hold float64 positions: this range's low parts become zero
    int rc = d2d_upload(h, D2D_BUF_POS_X, x, bytes, off);
    if (rc) return rc;
    rc = d2d_upload(h, D2D_BUF_POS_Y, y, bytes, off);
    if (rc) return rc;
    if (keep_lo) {
        USE_DEVICE(h);
        const size_t bd = (size_t)h->B * h->D;
        HIP_TRY(hipMemsetAsync(h->pos_lo + (size_t)env_begin * h->D, 0, bytes, h->stream));
        HIP_TRY(hipMemsetAsync(h->pos_lo + bd + (size_t)env_begin * h->D, 0, bytes, h->stream));
        h->have_lo = true;
    }
    return D2D_OK;
} D2D_CATCH

int d2d_set_positions_f64(d2d_handle* h, const double* x, const double* y, int32_t env_begin, int32_t env_count) try {
    if (!h || !x || !y) return fail(D2D_ERR_INVALID, "null argument");
    if (env_begin < 0 || env_count < 0 || env_begin + env_count > h->B) return fail(D2D_ERR_INVALID, "env range out of bounds");
    USE_DEVICE(h);
    // coordinate = hi + lo: hi the nearest float32 (what POS_X / POS_Y, the obs table and every float32 consumer see), lo the
    // float32 nearest to the remainder - together 48 bits of the double
    const size_t n = (size_t)env_count * h->D, bd = (size_t)h->B * h->D;
    std::vector<float> hi(2 * n), lo(2 * n);
    bool any_lo = false;
    for (int c = 0; c < 2; ++c) {
        const double* src = c ? y : x;
        for (size_t k = 0; k < n; ++k) {
            const float a = (float)src[k];
            const float b = std::isfinite(a) ? (float)(src[k] - (double)a) : 0.0f;
            hi[c * n + k] = a; lo[c * n + k] = b;
            any_lo |= b != 0.0f;
        }
    }
    const bool had_lo = h->have_lo;                         // (the float32 uploads below clear the flag)
    const size_t off = (size_t)env_begin * h->D * 4;
    int rc = d2d_upload(h, D2D_BUF_POS_X, hi.data(), n * 4, off);
    if (rc) return rc;
    rc = d2d_upload(h, D2D_BUF_POS_Y, hi.data() + n, n * 4, off);
    if (rc) return rc;
    const bool keep = had_lo && env_count < h->B;           // envs outside this range hold low parts of their own
    if (!any_lo && !keep) return D2D_OK;                    // every coordinate is a float32 value: the float32 kernels serve it, same bits
    if (!h->pos_lo) {
        HIP_TRY(hipMalloc(&h->pos_lo, 2 * bd * 4));
        HIP_TRY(hipMemsetAsync(h->pos_lo, 0, 2 * bd * 4, h->stream));
    } else if (!had_lo) HIP_TRY(hipMemsetAsync(h->pos_lo, 0, 2 * bd * 4, h->stream));      // stale low parts of an earlier layout
    HIP_TRY(hipMemcpyAsync(h->pos_lo + (size_t)env_begin * h->D, lo.data(), n * 4, hipMemcpyHostToDevice, h->stream));
    HIP_TRY(hipMemcpyAsync(h->pos_lo + bd + (size_t)env_begin * h->D, lo.data() + n, n * 4, hipMemcpyHostToDevice, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));               // lo is a stack-lifetime host buffer
    h->have_lo = true;
    h->lpos_dirty = true;
    return D2D_OK;
} D2D_CATCH

int d2d_set_env_offset(d2d_handle* h, uint64_t first_env) try {
    if (!h) return fail(D2D_ERR_INVALID, "null handle");
    USE_DEVICE(h);
    h->env_offset = first_env;
    return D2D_OK;
} D2D_CATCH

int d2d_reset_positions(d2d_handle* h, uint64_t seed, uint64_t episode, const uint8_t* fixed_mask, const float* fixed_xy) try {
    if (!h) return fail(D2D_ERR_INVALID, "null handle");
    USE_DEVICE(h);
    if ((fixed_mask == nullptr) != (fixed_xy == nullptr)) return fail(D2D_ERR_INVALID, "fixed_mask and fixed_xy go together");
    void *px = nullptr, *py = nullptr;
    int rc = ensure_buffer(h, D2D_BUF_POS_X, &px);
    if (rc) return rc;
    rc = ensure_buffer(h, D2D_BUF_POS_Y, &py);
    if (rc) return rc;
    const unsigned char* m = nullptr;
    const float* xy = nullptr;
    if (fixed_mask) {
        if (!h->fixed_mask_dev) {
            HIP_TRY(hipMalloc(&h->fixed_mask_dev, (size_t)h->D));
            HIP_TRY(hipMalloc(&h->fixed_xy_dev, (size_t)h->D * 8));
        }
        HIP_TRY(hipMemcpyAsync(h->fixed_mask_dev, fixed_mask, (size_t)h->D, hipMemcpyHostToDevice, h->stream));
        HIP_TRY(hipMemcpyAsync(h->fixed_xy_dev, fixed_xy, (size_t)h->D * 8, hipMemcpyHostToDevice, h->stream));
        HIP_TRY(hipStreamSynchronize(h->stream));   // caller's host arrays may go away
        m = h->fixed_mask_dev; xy = h->fixed_xy_dev;
    }
    // with the standard link list (every uplink and every sidelink, in device order) the sampler writes the per-link
    // position rows itself; any other list is gathered from POS_X / POS_Y before the next step
    const bool rows_here = !h->tables_dirty && h->std_layout && h->lpos && h->N > 0;
    HIP_TRY(d2d::launch_reset(h->B, h->D, h->cfg.num_cues, h->cfg.cell_radius_m, h->cfg.d2d_radius_m, seed, episode,
                              h->env_offset, m, xy, static_cast<float*>(px), static_cast<float*>(py),
                              rows_here ? h->lpos : nullptr, h->N, h->stream));
    h->have_pos = true;
    h->lpos_dirty = !rows_here;
    h->have_lo = false;            // the sampler draws float32 coordinates
    return D2D_OK;
} D2D_CATCH

int d2d_step(d2d_handle* h, const int32_t* actions_dev) try {
    if (!h) return fail(D2D_ERR_INVALID, "null handle");
    USE_DEVICE(h);
    if (!actions_dev && !h->buf[D2D_BUF_ACTIONS].ptr && (h->N == 0 || h->n_fixed < h->N))
        return fail(D2D_ERR_STATE, "no actions: pass a pointer or fill D2D_BUF_ACTIONS");
    return run_step(h, 0, actions_dev, nullptr, nullptr);
} D2D_CATCH

int d2d_step_rb_pwr(d2d_handle* h, const int32_t* rb_dev, const int32_t* pwr_dev) try {
    if (!h) return fail(D2D_ERR_INVALID, "null handle");
    USE_DEVICE(h);
    if ((!rb_dev && !h->buf[D2D_BUF_RB].ptr) || (!pwr_dev && !h->buf[D2D_BUF_PWR].ptr))
        return fail(D2D_ERR_STATE, "no rb/pwr: pass pointers or fill D2D_BUF_RB / D2D_BUF_PWR");
    return run_step(h, 1, rb_dev, pwr_dev, nullptr);
} D2D_CATCH

int d2d_expand_table(d2d_handle* h, const float* table_dev, int32_t n_envs, int32_t n_links, float* obs_dev) try {
    if (!h || !table_dev || !obs_dev) return fail(D2D_ERR_INVALID, "null argument");
    if (n_envs < 1 || n_links < 1 || n_links > D2D_MAX_LINKS) return fail(D2D_ERR_INVALID, "n_envs >= 1 and 1 <= n_links <= D2D_MAX_LINKS");
    USE_DEVICE(h);
    d2d::ObsArgs o;
    make_obs_args(h, n_envs, n_links, table_dev, obs_dev, 0, &o);
    EventPair* ep = nullptr;
    int rc = record_start(h, 1, &ep);
    if (rc) return rc;
    HIP_TRY(d2d::launch_obs_expand(o, h->stream));
    if (ep) HIP_TRY(hipEventRecord(ep->stop, h->stream));
    return D2D_OK;
} D2D_CATCH

int d2d_step_host(d2d_handle* h, const int32_t* rb_host, const int32_t* pwr_host, const void** out_host, d2d_host_layout* layout) try {
    if (!h || !rb_host || !pwr_host || !out_host || !layout) return fail(D2D_ERR_INVALID, "null argument");
    USE_DEVICE(h);
    if (!h->have_links) return fail(D2D_ERR_STATE, "d2d_set_links has not been called");
    const size_t bn = (size_t)h->B * h->N;
    if (bn == 0) return fail(D2D_ERR_INVALID, "no links: the reference divides by len(actions) (reward_fn.py:42)");
    d2d_host_layout L;
    L.sinr_db = 0; L.snr_db = bn * 4; L.rate_bps = 2 * bn * 4; L.capacity = 3 * bn * 4; L.reward = 4 * bn * 4;
    L.rb = 5 * bn * 4; L.pwr = 6 * bn * 4; L.obs_table = 7 * bn * 4; L.env_flags = 13 * bn * 4;
    L.obs = (L.env_flags + (size_t)h->B * 4 + 15) & ~(size_t)15;
    L.total_bytes = L.obs + (h->obs_mode == D2D_OBS_LINEAR ? bn * 6 * (size_t)h->N * 4 : 0);
    if (h->host_out_bytes < L.total_bytes) {
        HIP_TRY(hipStreamSynchronize(h->stream));
        if (h->host_out_dev) HIP_TRY(hipFree(h->host_out_dev));
        if (h->host_out_pinned) HIP_TRY(hipHostFree(h->host_out_pinned));
        h->host_out_dev = nullptr; h->host_out_pinned = nullptr; h->host_out_bytes = 0;
        HIP_TRY(hipMalloc(&h->host_out_dev, L.total_bytes));
        HIP_TRY(hipHostMalloc(&h->host_out_pinned, L.total_bytes, hipHostMallocDefault));
        h->host_out_bytes = L.total_bytes;
    }
    if (h->host_in_bytes < 2 * bn * 4) {
        HIP_TRY(hipStreamSynchronize(h->stream));
        if (h->host_in_dev) HIP_TRY(hipFree(h->host_in_dev));
        if (h->host_in_pinned) HIP_TRY(hipHostFree(h->host_in_pinned));
        h->host_in_dev = nullptr; h->host_in_pinned = nullptr; h->host_in_bytes = 0;
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&h->host_in_dev), 2 * bn * 4));
        HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&h->host_in_pinned), 2 * bn * 4, hipHostMallocDefault));
        h->host_in_bytes = 2 * bn * 4;
    }
    std::memcpy(h->host_in_pinned, rb_host, bn * 4);
    std::memcpy(h->host_in_pinned + bn, pwr_host, bn * 4);
    // Small results (the reference's default env: 75 KB): the kernels read (rb, pwr) from and write every result to the PINNED
    // HOST blocks directly over PCIe - no copy commands, one launch (or two) and one synchronisation.  Larger ones keep the
    // staged device block and the two DMA copies (a 6 MB obs block written store by store over PCIe would crawl).
    static const size_t zero_copy_limit = [] { const char* e = std::getenv("D2D_STEP_HOST_ZERO_COPY_BYTES"); return e ? (size_t)std::atoll(e) : (size_t)(256 * 1024); }();
    const bool zero_copy = L.total_bytes <= zero_copy_limit;
    if (!zero_copy) HIP_TRY(hipMemcpyAsync(h->host_in_dev, h->host_in_pinned, 2 * bn * 4, hipMemcpyHostToDevice, h->stream));
    char* base = static_cast<char*>(zero_copy ? h->host_out_pinned : h->host_out_dev);
    OutPtrs o;
    o.sinr = reinterpret_cast<float*>(base + L.sinr_db); o.snr = reinterpret_cast<float*>(base + L.snr_db);
    o.rate = reinterpret_cast<float*>(base + L.rate_bps); o.cap = reinterpret_cast<float*>(base + L.capacity);
    o.reward = reinterpret_cast<float*>(base + L.reward);
    o.rb = reinterpret_cast<int*>(base + L.rb); o.pwr = reinterpret_cast<int*>(base + L.pwr);
    o.table = reinterpret_cast<float*>(base + L.obs_table);
    o.env_flags = reinterpret_cast<int*>(base + L.env_flags);
    o.obs = reinterpret_cast<float*>(base + L.obs);
    const int32_t* in = zero_copy ? h->host_in_pinned : h->host_in_dev;
    int rc = run_step(h, 1, in, in + bn, &o);
    if (rc) return rc;
    if (!zero_copy) HIP_TRY(hipMemcpyAsync(h->host_out_pinned, h->host_out_dev, L.total_bytes, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    *out_host = h->host_out_pinned;
    *layout = L;
    return D2D_OK;
} D2D_CATCH

int d2d_comm_unique_id(void* id_out) try {
    if (!id_out) return fail(D2D_ERR_INVALID, "null argument");
    int rc = load_rccl();
    if (rc) return rc;
    const int e = g_rccl.GetUniqueId(id_out);
    return e ? rccl_fail("ncclGetUniqueId", e) : D2D_OK;
} D2D_CATCH

int d2d_comm_init(d2d_handle* h, int32_t world_size, int32_t rank, const void* unique_id) try {
    if (!h || !unique_id) return fail(D2D_ERR_INVALID, "null argument");
    if (world_size < 1 || rank < 0 || rank >= world_size) return fail(D2D_ERR_INVALID, "rank must be in [0, world_size)");
    USE_DEVICE(h);
    int rc = load_rccl();
    if (rc) return rc;
    if (h->comm) {
        // a collective of the old communicator may still be in flight on whatever stream the caller gave d2d_allgather (a side
        // stream that may be gone by now): drain the device, a rare call can afford it
        if (h->comm_used) HIP_TRY(hipDeviceSynchronize()); else HIP_TRY(hipStreamSynchronize(h->stream));
        g_rccl.CommDestroy(h->comm);
        h->comm = nullptr; h->comm_used = false;
    }
    Id128 id;
    std::memcpy(id.bytes, unique_id, sizeof(id.bytes));
    const int e = g_rccl.CommInitRank(&h->comm, world_size, id, rank);
    if (e) { h->comm = nullptr; return rccl_fail("ncclCommInitRank", e); }
    h->comm_world = world_size; h->comm_rank = rank;
    return D2D_OK;
} D2D_CATCH

int d2d_comm_destroy(d2d_handle* h) try {
    if (!h) return fail(D2D_ERR_INVALID, "null handle");
    USE_DEVICE(h);
    if (h->comm) {
        if (h->comm_used) HIP_TRY(hipDeviceSynchronize()); else HIP_TRY(hipStreamSynchronize(h->stream));
        g_rccl.CommDestroy(h->comm);
        h->comm = nullptr; h->comm_used = false;
    }
    return D2D_OK;
} D2D_CATCH

int d2d_allgather(d2d_handle* h, const void* send_dev, void* recv_dev, size_t bytes_per_rank, void* hip_stream) try {
    if (!h || !send_dev || !recv_dev) return fail(D2D_ERR_INVALID, "null argument");
    if (!h->comm) return fail(D2D_ERR_STATE, "d2d_comm_init has not been called");
    USE_DEVICE(h);
    hipStream_t stream = hip_stream ? reinterpret_cast<hipStream_t>(hip_stream) : h->stream;
    h->comm_used = true;
    const int e = g_rccl.AllGather(send_dev, recv_dev, bytes_per_rank, /* ncclInt8 */ 0, h->comm, stream);
    return e ? rccl_fail("ncclAllGather", e) : D2D_OK;
} D2D_CATCH

int d2d_status_flags(d2d_handle* h, uint32_t* flags) try {
    if (!h || !flags) return fail(D2D_ERR_INVALID, "null argument");
    USE_DEVICE(h);
    if (!h->buf[D2D_BUF_ENV_FLAGS].ptr) { *flags = 0; return D2D_OK; }
    HIP_TRY(d2d::launch_flags_or(static_cast<const int*>(h->buf[D2D_BUF_ENV_FLAGS].ptr), h->B, h->status, h->stream));
    HIP_TRY(hipMemcpyAsync(flags, h->status, 4, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    return D2D_OK;
} D2D_CATCH

int d2d_profile_enable(d2d_handle* h, int32_t enabled) try {
    if (!h) return fail(D2D_ERR_INVALID, "null handle");
    USE_DEVICE(h);
    int rc = drain_events(h);
    if (rc) return rc;
    h->prof = enabled != 0;
    return D2D_OK;
} D2D_CATCH

int d2d_profile_read(d2d_handle* h, int32_t kernel, double* total_ms, int64_t* launches) try {
    if (!h || kernel < 0 || kernel > 1) return fail(D2D_ERR_INVALID, "bad argument");
    USE_DEVICE(h);
    int rc = drain_events(h);
    if (rc) return rc;
    if (total_ms) *total_ms = h->acc_ms[kernel];
    if (launches) *launches = h->launches[kernel];
    return D2D_OK;
} D2D_CATCH

int d2d_profile_reset(d2d_handle* h) try {
    if (!h) return fail(D2D_ERR_INVALID, "null handle");
    USE_DEVICE(h);
    int rc = drain_events(h);
    if (rc) return rc;
    h->acc_ms[0] = h->acc_ms[1] = 0; h->launches[0] = h->launches[1] = 0;
    h->samples[0].clear(); h->samples[1].clear();
    return D2D_OK;
} D2D_CATCH

int d2d_profile_median(d2d_handle* h, int32_t kernel, double* median_ms) try {
    if (!h || !median_ms || kernel < 0 || kernel > 1) return fail(D2D_ERR_INVALID, "bad argument");
    USE_DEVICE(h);
    int rc = drain_events(h);
    if (rc) return rc;
    std::vector<float> v = h->samples[kernel];
    if (v.empty()) { *median_ms = 0.0; return D2D_OK; }
    std::nth_element(v.begin(), v.begin() + v.size() / 2, v.end());
    *median_ms = v[v.size() / 2];
    return D2D_OK;
} D2D_CATCH

}  // extern "C"
