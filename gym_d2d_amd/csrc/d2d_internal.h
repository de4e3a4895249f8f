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
    PL_SHADOW = 3,       // PL_POWER + log-normal shadowing beyond d0, fresh Philox Gaussian per evaluation
    PL_POWK = 4          // PL_POWER where the exponent of every link's transmitter lies within 1/2 of ONE integer k (StepArgs::pow_k;
                         // COST-Hata's 3.6 / 4.375: k = 4): gain = (d^2)^(-k/2) by reciprocals and products, times (d^2)^phi,
                         // phi = -(n - k) / 2 in [-1/4, 1/4] - three transcendentals and four products per pair instead of the
                         // general split's two and fourteen (pow_k_gains, d2d_step_device.h)
};

// Per-link record, three 16-byte rows shared by all envs and read coalesced by link index (L2-resident).  Built on
// the host from the per-device columns + the link table whenever links, tables or fixed actions change.
//   a (int4)   x: tx device | link_type << 24 | fixed << 28     y: rx device
//              z, w: fixed ? (rb, tx power dBm) : (division magic ceil(2^32 / P), largest action it is exact for; 0, 0 = divide)
//   b (float4) x: tx_lin = 10^((eirp_off - a_tx)/10)   y: rx_pl = 10^(-a_rx/10)   z: rx_lin = 10^(rx_off/10)
//              w: noise_mw = 10^(thermal_noise_dBm/10)
//   c (float4) x: rx_sensitivity_dBm   y: 1e-6 * RB bandwidth (Hz) of the tx   z: path-loss exponent of the tx (informative: the
//              kernels read the head / tail pair of -exponent / 2 from rec_h, which carries more than float precision)
//              w: (bits) P = power levels of this link's type (d2d_env.py:31-35) | action column << 16
#define D2D_REC_TXDEV_MASK 0x00FFFFFF
#define D2D_REC_TYPE_SHIFT 24
#define D2D_REC_TYPE_MASK 0xF
#define D2D_REC_FIXED_BIT (1 << 28)

// Byte offsets of one env's LDS arrays (step_lds_layout): only what the configuration reads back is allocated.
struct StepLds {
    unsigned aux, rx, sinr, sh, expo, tflat, mask, lists, pool, env_bytes;
    unsigned lo;             // exact positions (StepArgs::lpos_lo): float2 per link (+ stand-in), the low parts of (tx_x, tx_y)
};

struct StepArgs {
    // geometry
    int B, N, R, D;
    int mask_words;          // ceil(N/32): u32 words per RB membership mask (0 -> all-pairs path)
    int lpt;                 // links per thread held in registers: 1, 2, or 0 = strided
    int rollout;             // the rollout kernel (d2d_rollout.hip) serves this launch; lds = rollout_lds_layout
    StepLds lds;             // LDS layout of one env
    int action_mode;         // 0: raw int actions (a // P, a % P)   1: explicit rb / pwr
    int act_stride;          // columns of the action array(s): N - n_fixed in mode 0, N in mode 1
    int col_mode;            // 0: fixed links are exactly the first n_fixed links (column = link - n_fixed)   1: column from act_cols
    int n_fixed;
    int tpe;                 // threads per env (multiple of 64, <= 1024)
    unsigned tpe_magic;      // ceil(2^20 / tpe): tid / tpe == (tid * tpe_magic) >> 20 for tid < 1024
    float inv_n;             // 1 / N
    int epw;                 // envs per workgroup (epw * tpe <= blockDim)
    int prefetch_envs;       // software-prefetch distance for the action rows, in envs (multiple of 8 * epw; 0 = off)
    int walk;                // same-RB search: 0 mask walk, nested (words outside, members inside), 1 mask walk, flattened,
                             // 2 per-RB member lists (slot counter + eight u16 slots per RB, sorted in registers by the receiver;
                             // an env in which some RB holds more than eight links rebuilds the masks and walks them)
    int reward_fn;
    float reward_param;
    int write_table;
    int pow_k;               // PL_POWK: the common integer k of the transmitters' exponents (1 .. 8); rec_h then holds (phi, 0)
    int rec_uniform;         // every aligned group of 64 links has identical records (device ids aside): scalar record loads
    int nt_results;          // nontemporal result stores (nothing re-reads them from L2 right behind this launch)
    int ablate;              // DIAGNOSTIC builds only (-DD2D_STEP_ABLATE=1): skip parts of the kernel to time the rest
    unsigned long long* dbg; // DIAGNOSTIC builds only: [workgroup][wave][8] shader-clock stamps at the phase boundaries, or null
    // fused LinearObs expansion (small N: one launch per step instead of two); 0 = off, else floats per store (2 | 4)
    int fuse_obs;
    int obs_rotate;                  // fused expansion: workgroup w starts at step (w * obs_rotate) mod (its steps) of its (env, pass) sequence; 0 = all from the start
    unsigned obs_q_per_row;          // 6N / fuse_obs
    unsigned long long obs_q_magic;  // ceil(2^40 / obs_q_per_row)
    // inputs
    const int* actions;      // [B, act_stride]
    const int* rb_in;        // [B, N]
    const int* pwr_in;       // [B, N]
    const int4* rec_a;       // [N]
    const float4* rec_b;     // [N]
    const float4* rec_c;     // [N]
    const float2* rec_h;     // [N] head / tail of -exponent / 2 of the link's transmitter (power-law / shadowing kernels)
    const int* rec_grp;      // rec_uniform only: [N / 64][16 dwords] the record of every aligned group of 64 links in ONE 64-byte row -
                             // (magic, bound, a.x, 0 | b.xyzw | c.xyzw | h.x, h.y, 0, 0): one s_load_dwordx16 per wave (rollout kernel)
    const int* act_cols;     // [N] action column of every link (col_mode 1 only; 0 for fixed links)
    const unsigned* side_words;  // [ceil(N / 32)] bit i set <=> link i is a sidelink (host-built with the records)
    const float4* lpos;      // [B, N] (tx_x, tx_y, rx_x, rx_y) of every link, rebuilt when positions / links change
    const float4* lpos_lo;   // [B, N] or null: the LOW parts of the same four coordinates when the host uploaded float64 positions
                             // (d2d_set_positions_f64: coordinate = hi + lo, hi = float(v), lo = float(v - hi)).  Kernels compiled with
                             // OPT_XPOS form every difference as (tx_hi - rx_hi) + (tx_lo - rx_lo): exact to ~1e-7 of the DIFFERENCE
                             // where float32 absolute coordinates carry 3e-5 m at 500 m (position.py:11-12 works in Python floats)
    const float* gain_table; // PL_TABLE: linear gain, tx major: [D,D] by (tx device, rx device), or [N,N] by (tx link, rx link)
    long long table_env_stride; // 0 or pitch * pitch
    int table_by_link;       // rows / columns are link indices (d2d_set_path_loss_link_table), else device indices
    int table_pitch;         // row length: N or D
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
    float* reward_env;       // [B] SystemCapacity's one scalar per env (D2D_REWARD_PER_ENV); null = the [B,N] row above
    float* table;            // [B,N,6]
    float* obs;              // [B,N,6N] (fuse_obs only)
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
    int out_f64;             // obs is float64 [B,N,6N] (d2d_set_obs_dtype): obs_expand_f64_kernel
    int stagger;             // > 0: wave w of a workgroup sleeps w * stagger x 64 clocks before its stores (A/B)
    const float* table;      // [B,N,6]
    float* obs;              // [B,N,6N]
};

hipError_t launch_step(const StepArgs& a, PlMode mode, int block_threads, hipStream_t stream);
hipError_t launch_rollout(const StepArgs& a, PlMode mode, int opt, int block_threads, hipStream_t stream);
void rollout_lds_layout(int N, int R, int mode, int reward_fn, int xpos, StepLds* out);
hipError_t launch_obs_expand(const ObsArgs& a, hipStream_t stream);
size_t step_lds_bytes_per_env(int N, int R, int mask_words, int fuse_obs, int lpt, int reward_fn, int mode, int lists, int xpos);
void step_lds_layout(int N, int R, int mask_words, int fuse_obs, int lpt, int reward_fn, int mode, int lists, int xpos, StepLds* out);
hipError_t launch_flags_or(const int* env_flags, int B, unsigned* status, hipStream_t stream);
hipError_t launch_link_positions(const float* pos_x, const float* pos_y, const float* lo_x, const float* lo_y, const int4* rec_a,
                                 int B, int N, int D, float4* lpos, float4* lpos_lo, hipStream_t stream);
hipError_t launch_gain_from_db(const void* pl_db, int is_f64, size_t elems, float* gain, int num_cus, hipStream_t stream);
hipError_t launch_reset(int B, int D, int C, float cell_radius, float d2d_radius, unsigned long long seed,
                        unsigned long long episode, unsigned long long env_offset, const unsigned char* fixed_mask,
                        const float* fixed_xy, float* pos_x, float* pos_y, float4* lpos, int N, hipStream_t stream);

}  // namespace d2d
