// Internal launch interface between the C-ABI layer (d2d_capi.hip) and the kernels.  gfx950 only.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace d2d {

// Path-loss evaluation variants the step kernel is compiled for.
enum PlMode : int {
    PL_INV_SQUARE = 0,   // every exponent == 2 (LogDistance default / FreeSpace): gain = k / d^2, one v_rcp
    PL_POWER = 1,        // per-tx exponent (ple != 2, COST-Hata): gain = k * (d^2)^(-e/2)
    PL_TABLE = 2,        // host-evaluated [D,D] (or [B,D,D]) linear gain table
    PL_SHADOW = 3        // PL_POWER + log-normal shadowing beyond d0, fresh Philox Gaussian per evaluation
};

struct StepArgs {
    // geometry
    int B, N, R, D;
    int mask_words;          // ceil(N/64): u64 words per RB membership mask (0 -> all-pairs path)
    int action_mode;         // 0: raw int actions (a // P, a % P)   1: explicit rb / pwr
    int p_due, p_cue, p_mbs; // power levels per link type (d2d_env.py:31-35)
    unsigned long long m_due, m_cue, m_mbs;   // ceil(2^40 / P) division magics (0 -> use the hardware divide)
    int threads;             // threads per workgroup (0 -> one per link, rounded up to a wave, max 1024)
    int reward_fn;
    float reward_param;
    int write_table;
    // inputs
    const int* actions;      // [B,N]
    const int* rb_in;        // [B,N]
    const int* pwr_in;       // [B,N]
    const int* link_tx;      // [N] device index
    const int* link_rx;      // [N]
    const int* link_type;    // [N] 1 uplink, 2 downlink, 3 sidelink
    const float* pos_x;      // [B,D]
    const float* pos_y;      // [B,D]
    // per-link constants [N], flattened on the host from the per-device columns whenever links or tables change
    const float* lk_tx_lin;    // 10^((eirp_off - a_tx)/10) of the tx device: EIRP offset + tx side of the path-loss constant
    const float* lk_rx_pl;     // 10^(-a_rx/10) of the rx device: rx side of the path-loss constant (signal AND interference)
    const float* lk_rx_lin;    // 10^(rx_off/10): rx antenna/body/cable terms (signal only, simulator.py:93 vs :100)
    const float* lk_noise_mw;  // 10^(thermal_noise_dBm/10) of the rx device
    const float* lk_sens_db;   // rx_sensitivity_dBm of the rx device
    const float* lk_bw_mhz;    // 1e-6 * RB bandwidth (Hz) of the tx device
    const float* lk_exp;       // path-loss exponent of the tx device
    const float* pow10_tab;    // [128] 10^(p/10) for integer p dBm
    const float* gain_table;   // PL_TABLE: linear gain [D,D] (tx major)
    long long table_env_stride; // 0 or D*D
    // PL_SHADOW (ShadowingPathLoss, path_loss.py:69-81)
    float shadow_chi;        // std of the shadowing term, dB
    float shadow_d0sq;       // (close-in reference distance)^2
    unsigned shadow_seed_lo, shadow_seed_hi, shadow_step;
    unsigned long long env_offset;
    // outputs
    int* rb_out;             // [B,N] (nullable)
    int* pwr_out;
    float* sinr_db;
    float* snr_db;
    float* rate;
    float* cap;
    float* reward;           // [B,N] (nullable when reward_fn == 0)
    float* table;            // [B,N,6]
    int* env_flags;          // [B]  (OR-reduced on demand by launch_flags_or)
};

struct ObsArgs {
    int B, N;
    int rows_per_wg;
    int chunks;              // ceil(N / rows_per_wg)
    int vec;                 // floats per store: 4 (6N % 4 == 0) or 2
    unsigned q_per_row;      // 6N / vec
    unsigned long long q_magic;  // ceil(2^40 / q_per_row)
    int xcd_remap;
    int nontemporal;
    int block;               // threads per workgroup (0 -> 256)
    int variant;             // 0: T staged in LDS   1: T read straight from global (A/B)
    const float* table;      // [B,N,6]
    float* obs;              // [B,N,6N]
};

hipError_t launch_step(const StepArgs& a, PlMode mode, hipStream_t stream);
hipError_t launch_obs_expand(const ObsArgs& a, hipStream_t stream);
hipError_t launch_fill(float* dst, size_t n_float4, float value, hipStream_t stream);
size_t step_lds_bytes(int N, int R, int mask_words);
hipError_t launch_flags_or(const int* env_flags, int B, unsigned* status, hipStream_t stream);
hipError_t launch_reset(int B, int D, int C, float cell_radius, float d2d_radius, unsigned long long seed,
                        unsigned long long episode, unsigned long long env_offset, const unsigned char* fixed_mask,
                        const float* fixed_xy, float* pos_x, float* pos_y, hipStream_t stream);

}  // namespace d2d
