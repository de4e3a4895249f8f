// C-ABI layer of libd2d_hip.so (see include/d2d_hip.h).  Owns the SoA state of B environments in HBM and
// enqueues the step / obs kernels on one HIP stream.  No exceptions cross the boundary; no CPU fallback exists:
// if HIP is unusable every call fails with D2D_ERR_HIP.
#include "../../include/d2d_hip.h"
#include "d2d_internal.h"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
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

// Every entry point that may touch HIP first makes the handle's GPU current for the calling thread: with one
// process per GPU this is a no-op, but a caller that juggles several devices must not redirect our launches.
#define USE_DEVICE(h) HIP_TRY(hipSetDevice((h)->cfg.device_ordinal))

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
    float* dev_cols = nullptr;      // per-link constants, 7 arrays x Nmax (see refresh_tables)
    int* link_tab = nullptr;        // 3 x Nmax
    float* pow10_tab = nullptr;     // 128
    float* gain_table = nullptr;
    size_t gain_elems = 0;
    int table_per_env = 0;
    unsigned* status = nullptr;
    // host-side copies used to derive the device columns
    std::vector<double> eirp_off, rx_off, noise, sens, bw, a_tx, a_rx, expo;
    std::vector<int> host_tx, host_rx;   // host copy of the link table
    bool have_dev = false, have_pl = false, have_links = false, have_pos = false, tables_dirty = true;
    d2d::PlMode mode = d2d::PL_INV_SQUARE;
    int reward_fn = D2D_REWARD_SYSTEM_CAPACITY;
    float reward_param = 0.0f;
    int obs_mode = D2D_OBS_LINEAR;
    int bucketing = 1;
    int tune_rows = 0, tune_nt = 1, tune_xcd = 1, tune_block = 0, tune_variant = 0, tune_step_threads = 0;
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
};

namespace {

size_t active_bytes(const d2d_handle* h, int which, int n_links) {
    const size_t B = h->B, D = h->D, N = n_links;
    switch (which) {
        case D2D_BUF_POS_X: case D2D_BUF_POS_Y: return B * D * 4;
        case D2D_BUF_OBS_TABLE: return B * N * 6 * 4;
        case D2D_BUF_OBS: return B * N * 6 * N * 4;
        case D2D_BUF_ENV_FLAGS: return B * 4;
        default: return B * N * 4;
    }
}

int ensure_buffer(d2d_handle* h, int which, void** out) {
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
    }
    if (h->mode != d2d::PL_TABLE && h->mode != d2d::PL_SHADOW) h->mode = all_two ? d2d::PL_INV_SQUARE : d2d::PL_POWER;
    // flatten to per-link arrays [7][Nmax]: tx-side columns by the link's tx device, rx-side by its rx device, so
    // the kernel reads them coalesced by link index with no link -> device -> column double hop
    const int N = h->N, S = h->Nmax;
    std::vector<float> lk((size_t)7 * S, 0.0f);
    for (int i = 0; i < N; ++i) {
        const int t = h->host_tx[i], r = h->host_rx[i];
        lk[0 * S + i] = cols[0 * D + t];    // tx_lin
        lk[1 * S + i] = cols[1 * D + r];    // rx_pl
        lk[2 * S + i] = cols[2 * D + r];    // rx_lin
        lk[3 * S + i] = cols[3 * D + r];    // noise_mw
        lk[4 * S + i] = cols[4 * D + r];    // sens_db
        lk[5 * S + i] = cols[5 * D + t];    // bw_mhz
        lk[6 * S + i] = cols[6 * D + t];    // exponent
    }
    HIP_TRY(hipMemcpyAsync(h->dev_cols, lk.data(), lk.size() * 4, hipMemcpyHostToDevice, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));   // lk is a stack-lifetime host buffer
    h->tables_dirty = false;
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
    }
    h->events_used = 0;
    return D2D_OK;
}

int run_step(d2d_handle* h, int action_mode, const int32_t* a0, const int32_t* a1) {
    if (!h->have_links) return fail(D2D_ERR_STATE, "d2d_set_links has not been called");
    if (!h->have_pos) return fail(D2D_ERR_STATE, "positions not set (d2d_set_positions / upload POS_X,POS_Y)");
    int rc = refresh_tables(h);
    if (rc) return rc;
    const int N = h->N, D = h->D;
    if (N == 0) return fail(D2D_ERR_INVALID, "no links: the reference divides by len(actions) (reward_fn.py:42)");

    d2d::StepArgs s;
    std::memset(&s, 0, sizeof(s));
    s.B = h->B; s.N = N; s.R = h->cfg.num_rbs; s.D = D;
    int W = 0;
    if (h->bucketing) {
        W = (N + 63) / 64;
        if (d2d::step_lds_bytes(N, s.R, W) > 96 * 1024) W = 0;
    }
    s.mask_words = W;
    s.action_mode = action_mode;
    s.p_due = h->cfg.pwr_levels_due; s.p_cue = h->cfg.pwr_levels_cue; s.p_mbs = h->cfg.pwr_levels_mbs;
    auto magic = [](int P) -> unsigned long long { return P < 512 ? ((1ull << 40) + (unsigned)P - 1) / (unsigned)P : 0ull; };
    s.m_due = magic(s.p_due); s.m_cue = magic(s.p_cue); s.m_mbs = magic(s.p_mbs);
    s.threads = h->tune_step_threads;
    s.reward_fn = h->reward_fn; s.reward_param = h->reward_param;
    s.write_table = h->obs_mode != D2D_OBS_NONE;
    void* p = nullptr;
#define GET(which, field, type)                                  \
    do {                                                         \
        rc = ensure_buffer(h, which, &p);                        \
        if (rc) return rc;                                       \
        s.field = reinterpret_cast<type>(p);                     \
    } while (0)
    if (action_mode == 0) {
        if (a0) s.actions = a0; else GET(D2D_BUF_ACTIONS, actions, const int*);
        GET(D2D_BUF_RB, rb_out, int*);
        GET(D2D_BUF_PWR, pwr_out, int*);
    } else {
        if (a0) s.rb_in = a0; else GET(D2D_BUF_RB, rb_in, const int*);
        if (a1) s.pwr_in = a1; else GET(D2D_BUF_PWR, pwr_in, const int*);
    }
    GET(D2D_BUF_POS_X, pos_x, const float*);
    GET(D2D_BUF_POS_Y, pos_y, const float*);
    GET(D2D_BUF_SINR_DB, sinr_db, float*);
    GET(D2D_BUF_SNR_DB, snr_db, float*);
    GET(D2D_BUF_RATE_BPS, rate, float*);
    GET(D2D_BUF_CAPACITY, cap, float*);
    GET(D2D_BUF_ENV_FLAGS, env_flags, int*);
    if (s.reward_fn != D2D_REWARD_NONE) GET(D2D_BUF_REWARD, reward, float*);
    if (s.write_table) GET(D2D_BUF_OBS_TABLE, table, float*);
    s.link_tx = h->link_tab; s.link_rx = h->link_tab + h->Nmax; s.link_type = h->link_tab + 2 * h->Nmax;
    const int S = h->Nmax;
    s.lk_tx_lin = h->dev_cols; s.lk_rx_pl = h->dev_cols + S; s.lk_rx_lin = h->dev_cols + 2 * S;
    s.lk_noise_mw = h->dev_cols + 3 * S; s.lk_sens_db = h->dev_cols + 4 * S; s.lk_bw_mhz = h->dev_cols + 5 * S;
    s.lk_exp = h->dev_cols + 6 * S;
    s.pow10_tab = h->pow10_tab;
    s.gain_table = h->gain_table;
    s.table_env_stride = h->table_per_env ? (long long)D * D : 0;
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
    HIP_TRY(d2d::launch_step(s, h->mode, h->stream));
    if (ep) HIP_TRY(hipEventRecord(ep->stop, h->stream));

    if (h->obs_mode == D2D_OBS_LINEAR) {
        d2d::ObsArgs o;
        std::memset(&o, 0, sizeof(o));
        o.B = h->B; o.N = N;
        o.vec = (6 * N) % 4 == 0 ? 4 : 2;
        o.q_per_row = (unsigned)(6 * N / o.vec);
        o.q_magic = ((1ull << 40) + o.q_per_row - 1) / o.q_per_row;
        // Launch geometry (tools/tune_obs.py, MI355X, N = 512): the fastest shape is the one where every thread
        // issues exactly TWO 16-B stores - block = one row's float4 count (768), 2 rows per workgroup: 6.96 TB/s vs
        // 5.9 TB/s for 96-KiB slabs and 5.6 TB/s for 786-KiB slabs.  Small slabs dispatched in order keep the
        // chip-wide write front nearly sequential in address, and a workgroup never outlives its neighbours.
        int block = h->tune_block;
        if (block <= 0) {
            block = (int)((o.q_per_row + 63) / 64) * 64;
            if (block < 256) block = 256;
            if (block > 1024) block = 1024;
        }
        int rows = h->tune_rows;
        if (rows <= 0) {
            rows = (int)((2u * (unsigned)block + o.q_per_row / 2) / o.q_per_row);
            if (rows < 1) rows = 1;
        }
        if (rows > N) rows = N;
        o.rows_per_wg = rows;
        o.chunks = (N + rows - 1) / rows;
        o.xcd_remap = (h->tune_xcd > 0 && h->B % (8 * h->tune_xcd) == 0) ? h->tune_xcd : 0;   // envs interleaved per XCD
        o.nontemporal = h->tune_nt;
        o.block = block;
        o.variant = h->tune_variant;
        o.table = s.table;
        rc = ensure_buffer(h, D2D_BUF_OBS, &p);
        if (rc) return rc;
        o.obs = reinterpret_cast<float*>(p);
        rc = record_start(h, 1, &ep);
        if (rc) return rc;
        HIP_TRY(d2d::launch_obs_expand(o, h->stream));
        if (ep) HIP_TRY(hipEventRecord(ep->stop, h->stream));
    }
#undef GET
    return D2D_OK;
}

}  // namespace

extern "C" {

int d2d_abi_version(void) { return D2D_ABI_VERSION; }

const char* d2d_last_error(void) { return g_last_error.c_str(); }

int d2d_create(const d2d_config* cfg, d2d_handle** out) {
    if (!cfg || !out) return fail(D2D_ERR_INVALID, "null argument");
    *out = nullptr;
    if (cfg->abi_version != D2D_ABI_VERSION) return fail(D2D_ERR_INVALID, "abi_version mismatch");
    if (cfg->num_envs < 1 || cfg->num_rbs < 1 || cfg->num_cues < 0 || cfg->num_due_pairs < 0)
        return fail(D2D_ERR_INVALID, "num_envs/num_rbs must be >= 1 and device counts >= 0");
    if (cfg->pwr_levels_due < 1 || cfg->pwr_levels_cue < 1 || cfg->pwr_levels_mbs < 1)
        return fail(D2D_ERR_INVALID, "power level counts must be >= 1");
    int nmax = cfg->max_links > 0 ? cfg->max_links : cfg->num_cues + cfg->num_due_pairs;
    if (nmax < 1 || nmax > D2D_MAX_LINKS)
        return fail(D2D_ERR_INVALID, "max_links must be in [1, " + std::to_string(D2D_MAX_LINKS) + "]");
    int count = 0;
    HIP_TRY(hipGetDeviceCount(&count));
    if (cfg->device_ordinal < 0 || cfg->device_ordinal >= count)
        return fail(D2D_ERR_HIP, "device_ordinal out of range (no usable GPU?)");
    HIP_TRY(hipSetDevice(cfg->device_ordinal));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, cfg->device_ordinal));
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(D2D_ERR_UNSUPPORTED, std::string("libd2d_hip is built for gfx950 only, found ") + prop.gcnArchName);

    d2d_handle* h = new (std::nothrow) d2d_handle();
    if (!h) return fail(D2D_ERR_INVALID, "out of host memory");
    h->cfg = *cfg;
    h->B = cfg->num_envs;
    h->D = 1 + cfg->num_cues + 2 * cfg->num_due_pairs;
    h->Nmax = nmax;
    h->N = 0;
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
    CREATE_TRY(hipMalloc(&h->dev_cols, (size_t)7 * h->Nmax * 4));
    CREATE_TRY(hipMalloc(&h->link_tab, (size_t)3 * h->Nmax * 4));
    CREATE_TRY(hipMalloc(&h->pow10_tab, 128 * 4));
    CREATE_TRY(hipMalloc(&h->status, 4));
    CREATE_TRY(hipMemset(h->status, 0, 4));
    float tab[128];
    for (int p = 0; p < 128; ++p) tab[p] = (float)std::pow(10.0, p / 10.0);
    CREATE_TRY(hipMemcpy(h->pow10_tab, tab, sizeof(tab), hipMemcpyHostToDevice));
#undef CREATE_TRY
    *out = h;
    return D2D_OK;
}

int d2d_destroy(d2d_handle* h) {
    if (!h) return D2D_OK;
    hipSetDevice(h->cfg.device_ordinal);
    if (h->stream) hipStreamSynchronize(h->stream);
    for (auto& ep : h->events) { hipEventDestroy(ep.start); hipEventDestroy(ep.stop); }
    for (auto& bf : h->buf)
        if (bf.ptr && bf.owned) hipFree(bf.ptr);
    if (h->dev_cols) hipFree(h->dev_cols);
    if (h->link_tab) hipFree(h->link_tab);
    if (h->pow10_tab) hipFree(h->pow10_tab);
    if (h->gain_table) hipFree(h->gain_table);
    if (h->status) hipFree(h->status);
    if (h->fixed_mask_dev) hipFree(h->fixed_mask_dev);
    if (h->fixed_xy_dev) hipFree(h->fixed_xy_dev);
    if (h->own_stream) hipStreamDestroy(h->own_stream);
    delete h;
    return D2D_OK;
}

int d2d_set_stream(d2d_handle* h, void* hip_stream) {
    if (!h) return fail(D2D_ERR_INVALID, "null handle");
    USE_DEVICE(h);
    HIP_TRY(hipStreamSynchronize(h->stream));
    // NULL is a real stream (the device's legacy default stream, which is what torch's default stream is): work
    // enqueued there is ordered with the caller's own kernels and copies.  Only the sentinel selects the private one.
    h->stream = hip_stream == D2D_STREAM_PRIVATE ? h->own_stream : reinterpret_cast<hipStream_t>(hip_stream);
    return D2D_OK;
}

int d2d_synchronize(d2d_handle* h) {
    if (!h) return fail(D2D_ERR_INVALID, "null handle");
    USE_DEVICE(h);
    HIP_TRY(hipStreamSynchronize(h->stream));
    return D2D_OK;
}

int d2d_set_device_table(d2d_handle* h, int32_t n_dev, const double* eirp_off_db, const double* rx_off_db,
                         const double* noise_dbm, const double* sens_dbm, const double* bw_hz) {
    if (!h || !eirp_off_db || !rx_off_db || !noise_dbm || !sens_dbm || !bw_hz) return fail(D2D_ERR_INVALID, "null argument");
    if (n_dev != h->D) return fail(D2D_ERR_INVALID, "n_dev must be 1 + num_cues + 2*num_due_pairs = " + std::to_string(h->D));
    h->eirp_off.assign(eirp_off_db, eirp_off_db + n_dev);
    h->rx_off.assign(rx_off_db, rx_off_db + n_dev);
    h->noise.assign(noise_dbm, noise_dbm + n_dev);
    h->sens.assign(sens_dbm, sens_dbm + n_dev);
    h->bw.assign(bw_hz, bw_hz + n_dev);
    h->have_dev = true; h->tables_dirty = true;
    return D2D_OK;
}

int d2d_set_path_loss_power_law(d2d_handle* h, int32_t n_dev, const double* a_tx_db, const double* a_rx_db,
                                const double* exponent) {
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
}

int d2d_set_path_loss_shadowing(d2d_handle* h, int32_t n_dev, const double* a_tx_db, const double* a_rx_db,
                                const double* exponent, double d0_m, double chi_db, uint64_t seed) {
    if (!(d0_m >= 0.0) || !(chi_db >= 0.0)) return fail(D2D_ERR_INVALID, "d0_m and chi_db must be >= 0");
    int rc = d2d_set_path_loss_power_law(h, n_dev, a_tx_db, a_rx_db, exponent);
    if (rc) return rc;
    h->mode = d2d::PL_SHADOW;
    h->shadow_d0 = d0_m; h->shadow_chi = chi_db; h->shadow_seed = seed; h->shadow_step = 0;
    return D2D_OK;
}

int d2d_set_path_loss_table(d2d_handle* h, const float* pl_db, int32_t per_env) {
    if (!h || !pl_db) return fail(D2D_ERR_INVALID, "null argument");
    USE_DEVICE(h);
    const size_t elems = (size_t)h->D * h->D * (per_env ? (size_t)h->B : 1);
    std::vector<float> lin(elems);
    for (size_t k = 0; k < elems; ++k) lin[k] = (float)std::pow(10.0, -(double)pl_db[k] / 10.0);
    HIP_TRY(hipStreamSynchronize(h->stream));
    if (h->gain_elems < elems) {
        if (h->gain_table) HIP_TRY(hipFree(h->gain_table));
        h->gain_table = nullptr; h->gain_elems = 0;
        HIP_TRY(hipMalloc(&h->gain_table, elems * 4));
        h->gain_elems = elems;
    }
    HIP_TRY(hipMemcpy(h->gain_table, lin.data(), elems * 4, hipMemcpyHostToDevice));
    h->table_per_env = per_env ? 1 : 0;
    h->mode = d2d::PL_TABLE;
    h->have_pl = true; h->tables_dirty = true;
    return D2D_OK;
}

int d2d_set_links(d2d_handle* h, int32_t n_links, const int32_t* tx_dev, const int32_t* rx_dev, const int32_t* link_type) {
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
    if (n_links > 0) {
        HIP_TRY(hipMemcpy(h->link_tab, tx_dev, (size_t)n_links * 4, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(h->link_tab + h->Nmax, rx_dev, (size_t)n_links * 4, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(h->link_tab + 2 * h->Nmax, link_type, (size_t)n_links * 4, hipMemcpyHostToDevice));
    }
    h->N = n_links;
    h->host_tx.assign(tx_dev, tx_dev + n_links);
    h->host_rx.assign(rx_dev, rx_dev + n_links);
    h->tables_dirty = true;          // per-link constant arrays follow the link table
    h->have_links = true;
    return D2D_OK;
}

int d2d_set_reward(d2d_handle* h, int32_t reward_fn, float param) {
    if (!h) return fail(D2D_ERR_INVALID, "null handle");
    USE_DEVICE(h);
    if (reward_fn < D2D_REWARD_NONE || reward_fn > D2D_REWARD_CUE_SINR_SHANNON) return fail(D2D_ERR_INVALID, "unknown reward_fn");
    h->reward_fn = reward_fn; h->reward_param = param;
    return D2D_OK;
}

int d2d_set_obs_mode(d2d_handle* h, int32_t obs_mode) {
    if (!h) return fail(D2D_ERR_INVALID, "null handle");
    USE_DEVICE(h);
    if (obs_mode < D2D_OBS_NONE || obs_mode > D2D_OBS_LINEAR) return fail(D2D_ERR_INVALID, "unknown obs_mode");
    h->obs_mode = obs_mode;
    return D2D_OK;
}

int d2d_set_bucketing(d2d_handle* h, int32_t enabled) {
    if (!h) return fail(D2D_ERR_INVALID, "null handle");
    USE_DEVICE(h);
    h->bucketing = enabled ? 1 : 0;
    return D2D_OK;
}

int d2d_set_tuning(d2d_handle* h, int32_t key, int32_t value) {
    if (!h) return fail(D2D_ERR_INVALID, "null handle");
    USE_DEVICE(h);
    switch (key) {
        case D2D_TUNE_OBS_ROWS_PER_WG: h->tune_rows = value; break;
        case D2D_TUNE_OBS_NONTEMPORAL: h->tune_nt = value ? 1 : 0; break;
        case D2D_TUNE_OBS_XCD_REMAP: h->tune_xcd = value < 0 ? 0 : value; break;
        case D2D_TUNE_OBS_VARIANT: h->tune_variant = value; break;
        case D2D_TUNE_STEP_THREADS:
            if (value != 0 && (value < 64 || value > 1024 || value % 64)) return fail(D2D_ERR_INVALID, "threads must be a multiple of 64 in [64,1024]");
            h->tune_step_threads = value;
            break;
        case D2D_TUNE_OBS_BLOCK:
            if (value != 0 && (value < 64 || value > 1024 || value % 64)) return fail(D2D_ERR_INVALID, "block must be a multiple of 64 in [64,1024]");
            h->tune_block = value;
            break;
        default: return fail(D2D_ERR_INVALID, "unknown tuning key");
    }
    return D2D_OK;
}

int d2d_get_buffer(d2d_handle* h, int32_t which, void** dev_ptr, size_t* bytes) {
    if (!h || !dev_ptr) return fail(D2D_ERR_INVALID, "null argument");
    USE_DEVICE(h);
    if (which < 0 || which >= D2D_BUF_COUNT) return fail(D2D_ERR_INVALID, "unknown buffer");
    if (which == D2D_BUF_OBS && h->N == 0) return fail(D2D_ERR_STATE, "set links before asking for the obs buffer");
    int rc = ensure_buffer(h, which, dev_ptr);
    if (rc) return rc;
    if (bytes) *bytes = active_bytes(h, which, h->N ? h->N : h->Nmax);
    return D2D_OK;
}

int d2d_bind_buffer(d2d_handle* h, int32_t which, void* dev_ptr, size_t bytes) {
    if (!h) return fail(D2D_ERR_INVALID, "null handle");
    USE_DEVICE(h);
    if (which < 0 || which >= D2D_BUF_COUNT) return fail(D2D_ERR_INVALID, "unknown buffer");
    Buffer& bf = h->buf[which];
    HIP_TRY(hipStreamSynchronize(h->stream));
    if (bf.ptr && bf.owned) HIP_TRY(hipFree(bf.ptr));
    bf.ptr = dev_ptr; bf.bytes = dev_ptr ? bytes : 0; bf.owned = false;
    if (which == D2D_BUF_POS_X || which == D2D_BUF_POS_Y)
        h->have_pos = h->buf[D2D_BUF_POS_X].ptr && h->buf[D2D_BUF_POS_Y].ptr;
    return D2D_OK;
}

int d2d_upload(d2d_handle* h, int32_t which, const void* host_src, size_t bytes, size_t dst_offset) {
    if (!h || !host_src) return fail(D2D_ERR_INVALID, "null argument");
    USE_DEVICE(h);
    if (which < 0 || which >= D2D_BUF_COUNT) return fail(D2D_ERR_INVALID, "unknown buffer");
    void* p = nullptr;
    int rc = ensure_buffer(h, which, &p);
    if (rc) return rc;
    if (dst_offset + bytes > h->buf[which].bytes) return fail(D2D_ERR_INVALID, "upload out of range");
    HIP_TRY(hipMemcpyAsync(static_cast<char*>(p) + dst_offset, host_src, bytes, hipMemcpyHostToDevice, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    if (which == D2D_BUF_POS_X || which == D2D_BUF_POS_Y)
        h->have_pos = h->buf[D2D_BUF_POS_X].ptr && h->buf[D2D_BUF_POS_Y].ptr;
    return D2D_OK;
}

int d2d_download(d2d_handle* h, int32_t which, void* host_dst, size_t bytes, size_t src_offset) {
    if (!h || !host_dst) return fail(D2D_ERR_INVALID, "null argument");
    USE_DEVICE(h);
    if (which < 0 || which >= D2D_BUF_COUNT) return fail(D2D_ERR_INVALID, "unknown buffer");
    const Buffer& bf = h->buf[which];
    if (!bf.ptr) return fail(D2D_ERR_STATE, "buffer has never been written");
    if (src_offset + bytes > bf.bytes) return fail(D2D_ERR_INVALID, "download out of range");
    HIP_TRY(hipMemcpyAsync(host_dst, static_cast<const char*>(bf.ptr) + src_offset, bytes, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    return D2D_OK;
}

int d2d_set_positions(d2d_handle* h, const float* x, const float* y, int32_t env_begin, int32_t env_count) {
    if (!h || !x || !y) return fail(D2D_ERR_INVALID, "null argument");
    if (env_begin < 0 || env_count < 0 || env_begin + env_count > h->B) return fail(D2D_ERR_INVALID, "env range out of bounds");
    const size_t off = (size_t)env_begin * h->D * 4, bytes = (size_t)env_count * h->D * 4;
    int rc = d2d_upload(h, D2D_BUF_POS_X, x, bytes, off);
    if (rc) return rc;
    return d2d_upload(h, D2D_BUF_POS_Y, y, bytes, off);
}

int d2d_set_env_offset(d2d_handle* h, uint64_t first_env) {
    if (!h) return fail(D2D_ERR_INVALID, "null handle");
    USE_DEVICE(h);
    h->env_offset = first_env;
    return D2D_OK;
}

int d2d_reset_positions(d2d_handle* h, uint64_t seed, uint64_t episode, const uint8_t* fixed_mask, const float* fixed_xy) {
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
    HIP_TRY(d2d::launch_reset(h->B, h->D, h->cfg.num_cues, h->cfg.cell_radius_m, h->cfg.d2d_radius_m, seed, episode,
                              h->env_offset, m, xy, static_cast<float*>(px), static_cast<float*>(py), h->stream));
    h->have_pos = true;
    return D2D_OK;
}

int d2d_step(d2d_handle* h, const int32_t* actions_dev) {
    if (!h) return fail(D2D_ERR_INVALID, "null handle");
    USE_DEVICE(h);
    if (!actions_dev && !h->buf[D2D_BUF_ACTIONS].ptr) return fail(D2D_ERR_STATE, "no actions: pass a pointer or fill D2D_BUF_ACTIONS");
    return run_step(h, 0, actions_dev, nullptr);
}

int d2d_step_rb_pwr(d2d_handle* h, const int32_t* rb_dev, const int32_t* pwr_dev) {
    if (!h) return fail(D2D_ERR_INVALID, "null handle");
    USE_DEVICE(h);
    if ((!rb_dev && !h->buf[D2D_BUF_RB].ptr) || (!pwr_dev && !h->buf[D2D_BUF_PWR].ptr))
        return fail(D2D_ERR_STATE, "no rb/pwr: pass pointers or fill D2D_BUF_RB / D2D_BUF_PWR");
    return run_step(h, 1, rb_dev, pwr_dev);
}

int d2d_status_flags(d2d_handle* h, uint32_t* flags) {
    if (!h || !flags) return fail(D2D_ERR_INVALID, "null argument");
    USE_DEVICE(h);
    if (!h->buf[D2D_BUF_ENV_FLAGS].ptr) { *flags = 0; return D2D_OK; }
    HIP_TRY(d2d::launch_flags_or(static_cast<const int*>(h->buf[D2D_BUF_ENV_FLAGS].ptr), h->B, h->status, h->stream));
    HIP_TRY(hipMemcpyAsync(flags, h->status, 4, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    return D2D_OK;
}

int d2d_profile_enable(d2d_handle* h, int32_t enabled) {
    if (!h) return fail(D2D_ERR_INVALID, "null handle");
    USE_DEVICE(h);
    int rc = drain_events(h);
    if (rc) return rc;
    h->prof = enabled != 0;
    return D2D_OK;
}

int d2d_profile_read(d2d_handle* h, int32_t kernel, double* total_ms, int64_t* launches) {
    if (!h || kernel < 0 || kernel > 1) return fail(D2D_ERR_INVALID, "bad argument");
    USE_DEVICE(h);
    int rc = drain_events(h);
    if (rc) return rc;
    if (total_ms) *total_ms = h->acc_ms[kernel];
    if (launches) *launches = h->launches[kernel];
    return D2D_OK;
}

int d2d_profile_reset(d2d_handle* h) {
    if (!h) return fail(D2D_ERR_INVALID, "null handle");
    USE_DEVICE(h);
    int rc = drain_events(h);
    if (rc) return rc;
    h->acc_ms[0] = h->acc_ms[1] = 0; h->launches[0] = h->launches[1] = 0;
    return D2D_OK;
}

int d2d_probe_write_bandwidth(d2d_handle* h, size_t bytes, int32_t iters, double* gb_per_s) {
    if (!h || !gb_per_s || iters < 1 || bytes < 16) return fail(D2D_ERR_INVALID, "bad argument");
    USE_DEVICE(h);
    float* tmp = nullptr;
    HIP_TRY(hipMalloc(&tmp, bytes));
    hipEvent_t e0, e1;
    HIP_TRY(hipEventCreate(&e0));
    HIP_TRY(hipEventCreate(&e1));
    HIP_TRY(d2d::launch_fill(tmp, bytes / 16, 1.0f, h->stream));   // warm-up / page touch
    HIP_TRY(hipEventRecord(e0, h->stream));
    for (int k = 0; k < iters; ++k) HIP_TRY(d2d::launch_fill(tmp, bytes / 16, (float)k, h->stream));
    HIP_TRY(hipEventRecord(e1, h->stream));
    HIP_TRY(hipEventSynchronize(e1));
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
    hipEventDestroy(e0); hipEventDestroy(e1);
    HIP_TRY(hipFree(tmp));
    *gb_per_s = (double)(bytes / 16 * 16) * iters / (ms * 1e-3) / 1e9;
    return D2D_OK;
}

}  // extern "C"
