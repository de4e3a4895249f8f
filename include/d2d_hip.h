/*
 * d2d_hip.h - C ABI of libd2d_hip.so: the MI355X (gfx950) implementation of GymD2D's per-step
 * SINR / interference path, batched over B independent environments.
 *
 * This is the drop-in boundary.  The reference (davidcotton/gym-d2d v0.0.3, pure Python) has no
 * FFI of its own; each entry point below names the reference interface it replaces (file:line under
 * /root/reference/src/gym_d2d) and INTEGRATION.md shows the ctypes binding a maintainer would add.
 *
 * Conventions
 *   - plain C types only; every function returns a d2d_status (0 = ok); d2d_last_error() gives text.
 *   - "host" pointers are read/written synchronously before the call returns.
 *   - "device" pointers are HIP device memory on the handle's GPU; work on them is enqueued on the
 *     handle's stream and is asynchronous unless stated.
 *   - layout: row-major, env axis outermost.  B envs, D devices/env (0 = base station, 1..C = CUEs,
 *     then DUE tx/rx interleaved - devices.py:20-25), N links/step.
 *   - a handle is not thread-safe (the reference is single-threaded too).
 */
#ifndef D2D_HIP_H
#define D2D_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define D2D_ABI_VERSION 5      /* 5: float64 positions (d2d_set_positions_f64), device-resident path-loss table (d2d_set_path_loss_link_table_dev) */
#define D2D_MAX_LINKS 2048      /* links per env the step kernel's LDS staging is sized for */
#define D2D_UNIQUE_ID_BYTES 128 /* size of an RCCL unique id (ncclUniqueId)                  */

typedef struct d2d_handle d2d_handle;

typedef enum d2d_status {
    D2D_OK = 0,
    D2D_ERR_INVALID = 1,        /* bad argument / inconsistent sizes                        */
    D2D_ERR_HIP = 2,            /* a HIP runtime call failed (text in d2d_last_error)       */
    D2D_ERR_STATE = 3,          /* call order: tables / links / positions not set yet; also any
                                   C++ exception other than bad_alloc caught at the boundary */
    D2D_ERR_UNSUPPORTED = 4,
    D2D_ERR_NO_MEMORY = 5       /* host allocation failed (std::bad_alloc caught at the boundary:
                                   no C++ exception ever crosses this ABI)                    */
} d2d_status;

/* link_type.py:4-7 */
typedef enum d2d_link_type { D2D_UPLINK = 1, D2D_DOWNLINK = 2, D2D_SIDELINK = 3 } d2d_link_type;

/* reward_fn.py:22-78 */
typedef enum d2d_reward_fn {
    D2D_REWARD_NONE = 0,
    D2D_REWARD_SYSTEM_CAPACITY = 1,   /* param = min_capacity_mbps   (reward_fn.py:23-44) */
    D2D_REWARD_SHANNON = 2,           /* param = min_sinr            (reward_fn.py:48-57) */
    D2D_REWARD_CUE_SINR_SHANNON = 3   /* param = sinr_threshold_dB   (reward_fn.py:61-78) */
} d2d_reward_fn;

/* obs_fn.py:43-61 */
typedef enum d2d_obs_mode {
    D2D_OBS_NONE = 0,
    D2D_OBS_TABLE = 1,   /* compact base table T[B,N,6] = (tx_x,tx_y,rx_x,rx_y,sinr_db,snr_db) only */
    D2D_OBS_LINEAR = 2   /* + LinearObsFunction expansion obs[B,N,6N] (reference semantics)         */
} d2d_obs_mode;

/* buffers addressable through d2d_get_buffer / d2d_bind_buffer / d2d_upload / d2d_download */
typedef enum d2d_buffer {
    D2D_BUF_POS_X = 0,      /* f32 [B,D]   device x positions (simulator.py:61-75)                  */
    D2D_BUF_POS_Y = 1,      /* f32 [B,D]                                                            */
    D2D_BUF_ACTIONS = 2,    /* i32 [B,A]   raw discrete actions (d2d_env.py:93-96); A = N minus the
                               links given fixed actions by d2d_set_fixed_actions                   */
    D2D_BUF_RB = 3,         /* i32 [B,N]   decoded resource block                                   */
    D2D_BUF_PWR = 4,        /* i32 [B,N]   decoded tx power, dBm                                    */
    D2D_BUF_SINR_DB = 5,    /* f32 [B,N]   state['sinrs_db']       (simulator.py:89-108)            */
    D2D_BUF_SNR_DB = 6,     /* f32 [B,N]   state['snrs_db']        (simulator.py:110-116)           */
    D2D_BUF_RATE_BPS = 7,   /* f32 [B,N]   state['rate_bps']       (simulator.py:118-127)           */
    D2D_BUF_CAPACITY = 8,   /* f32 [B,N]   state['capacity_mbps']  (simulator.py:144-154)           */
    D2D_BUF_REWARD = 9,     /* f32 [B,N]   per-agent reward        (reward_fn.py)                   */
    D2D_BUF_OBS_TABLE = 10, /* f32 [B,N,6]                          (obs_fn.py:55-61)               */
    D2D_BUF_OBS = 11,       /* f32 [B,N,6N] (f64 under d2d_set_obs_dtype(D2D_F64))  (obs_fn.py:43-53) */
    D2D_BUF_ENV_FLAGS = 12, /* i32 [B]     D2D_FLAG_* bits raised by the last step                  */
    D2D_BUF_LINK_POS = 13,  /* f32 [B,N,4] (tx_x,tx_y,rx_x,rx_y) of every link = columns 0-3 of the obs
                               table (obs_fn.py:57-59).  READ-ONLY, library-owned (cannot be bound or
                               uploaded): derived from POS_X / POS_Y and the link list, constant between
                               resets (simulator.py:61-75) - a learner that runs D2D_OBS_NONE takes the
                               positions from here once per episode and (sinr, snr) from the planes     */
    D2D_BUF_REWARD_ENV = 14,/* f32 [B]     SystemCapacity's one scalar per env (reward_fn.py:42-44),
                               written instead of D2D_BUF_REWARD under D2D_REWARD_PER_ENV             */
    D2D_BUF_COUNT = 15
} d2d_buffer;

#define D2D_FLAG_ZERO_DISTANCE 1u /* an interacting tx/rx pair at distance 0: the reference raises
                                     ValueError('math domain error') here (path_loss.py:66)          */
#define D2D_FLAG_RB_OUT_OF_RANGE 2u /* rb outside [0,num_rbs): accepted like the reference does
                                     (d2d_env.py:94-96); the env took the all-pairs code path       */
#define D2D_FLAG_NON_FINITE 4u    /* a non-finite SINR was produced                                  */

/* EnvConfig fields the device side needs (env_config.py:12-27) + batch geometry (new). */
typedef struct d2d_config {
    int32_t abi_version;      /* D2D_ABI_VERSION                                                    */
    int32_t device_ordinal;   /* HIP device to create the handle on                                 */
    int32_t num_envs;         /* B                                                                  */
    int32_t num_rbs;          /* env_config.py:12                                                   */
    int32_t num_cues;         /* env_config.py:13                                                   */
    int32_t num_due_pairs;    /* env_config.py:14                                                   */
    int32_t max_links;        /* capacity of the per-link buffers; 0 -> num_cues + num_due_pairs    */
    int32_t pwr_levels_due;   /* due_max - due_min + 1   (d2d_env.py:32)                            */
    int32_t pwr_levels_cue;   /* cue_max + 1             (d2d_env.py:33)                            */
    int32_t pwr_levels_mbs;   /* mbs_max + 1             (d2d_env.py:34)                            */
    float cell_radius_m;      /* env_config.py:15                                                   */
    float d2d_radius_m;       /* env_config.py:16                                                   */
} d2d_config;

/* ---- lifetime -------------------------------------------------------------------------------- */
/* Simulator.__init__ (simulator.py:54-59): allocate the SoA state for B envs on one GPU.           */
int d2d_create(const d2d_config* cfg, d2d_handle** out);
int d2d_destroy(d2d_handle* h);
const char* d2d_last_error(void);
int d2d_abi_version(void);

/* Run on a caller-supplied hipStream_t (e.g. torch's current stream) so that the library's kernels are
 * ordered with the caller's own work on that stream.  NULL is the device's default (null) stream - a
 * valid choice; D2D_STREAM_PRIVATE goes back to the handle's own non-blocking stream (the default).   */
#define D2D_STREAM_PRIVATE ((void*)(intptr_t)-1)
int d2d_set_stream(d2d_handle* h, void* hip_stream);
int d2d_synchronize(d2d_handle* h);

/* ---- static tables (host pointers, copied) --------------------------------------------------- */
/* Per-device link-budget columns, n_dev = 1 + C + 2P entries each, replacing Device.eirp_dBm /
 * rx_signal_level_dBm / thermal_noise_dBm / rx_sensitivity_dBm / rb_bandwidth_kHz
 * (device.py:51-80,85-95,130-173):
 *   eirp_off_db[d] = eirp_dBm(p) - p ; rx_off_db[d] = rx_signal_level_dBm(e, pl) - (e - pl) ;
 *   noise_dbm[d] = thermal_noise_dBm ; sens_dbm[d] = rx_sensitivity_dBm ; bw_hz[d] = RB bandwidth. */
int d2d_set_device_table(d2d_handle* h, int32_t n_dev, const double* eirp_off_db, const double* rx_off_db,
                         const double* noise_dbm, const double* sens_dbm, const double* bw_hz);

/* PathLoss plugin, native route (path_loss.py:42-66,90-123).  Any model of the form
 *   PL_dB(tx, rx) = a_tx_db[tx] + a_rx_db[rx] + 10 * exponent[tx] * log10(distance_m)
 * LogDistance/FreeSpace: a_tx = pl_constant_dB, a_rx = 0, exponent = ple.  COST-Hata: see
 * gym_d2d_amd/path_loss.py.  Arrays have n_dev entries.                                            */
int d2d_set_path_loss_power_law(d2d_handle* h, int32_t n_dev, const double* a_tx_db, const double* a_rx_db,
                                const double* exponent);

/* ShadowingPathLoss (path_loss.py:69-81): the power law above plus gauss(0, chi_db) dB on EVERY evaluation with
 * distance > d0_m - SINR signal term, each interferer term and the SNR's re-evaluation are independent
 * draws, as in the reference.  Draws come from Philox4x32-10 keyed by `seed` with counter
 * (global env, step, tx link | rx link << 16, kind); `step` counts d2d_step* calls since this call.   */
int d2d_set_path_loss_shadowing(d2d_handle* h, int32_t n_dev, const double* a_tx_db, const double* a_rx_db,
                                const double* exponent, double d0_m, double chi_db, uint64_t seed);

/* PathLoss plugin, table route for arbitrary Python subclasses (path_loss.py:12-25;
 * examples/custom_path_loss.py:8-16): pl_db[(e,) tx_dev, rx_dev] evaluated on the host once per
 * episode.  per_env = 0: one [D,D] table shared by all envs; 1: [B,D,D].  float64 dB in (as the plugin
 * returns them), converted to linear gains in double and rounded once, chunk by chunk through a bounded
 * pinned staging block (the library never holds a second copy of the table on the host); entries of pairs
 * that no link uses may hold anything (NaN included) - only (tx of link j, rx of link i) pairs are read. */
int d2d_set_path_loss_table(d2d_handle* h, const double* pl_db, int32_t per_env);
/* The same route without the dense device cube: pl_db[(e,) j, i] = PathLoss(tx of link j, rx of link i) for
 * the CURRENT link list (n_links = its length) - exactly the pairs the step reads (simulator.py:93,97-101:
 * own link j == i, interferers j != i).  [N,N] or [B,N,N] instead of [D,D] / [B,D,D] (D = 1 + C + 2P:
 * 4.3 GB instead of 9.7 GB per 4096 envs of 512 links).  A later d2d_set_links drops the table: the next
 * step then fails with D2D_ERR_STATE until a path-loss model is set again.                              */
int d2d_set_path_loss_link_table(d2d_handle* h, const double* pl_db, int32_t n_links, int32_t per_env);
/* The same table already RESIDENT ON THE DEVICE - what an array-native PathLoss plugin computes for a whole batch at once
 * (gym_d2d_amd.path_loss.ArrayPathLoss.compute(view) -> pl_db[B,N,N]; a per-object PathLoss.__call__ over a batch is
 * B x N x N Python calls, 1.1e9 at 4096 x 512).  pl_db_dev: device pointer, dtype D2D_F64 or D2D_F32 (float32 dB near 100 dB
 * carry 3.8e-6 dB of their own: prefer float64), [N,N] (per_env = 0) or [B,N,N]; converted to linear gains by one streaming
 * kernel (csrc/d2d_gain.hip: the exponential in double, rounded once) on the handle's stream - work that produced the table on
 * ANOTHER stream must have finished.  Synchronous: the caller may free the table when the call returns.  A failed call leaves
 * the handle without a path-loss table (the next step answers D2D_ERR_STATE), never with a partial one.                      */
int d2d_set_path_loss_link_table_dev(d2d_handle* h, const void* pl_db_dev, int32_t dtype, int32_t n_links, int32_t per_env);

/* Which (tx, rx) device pairs act this step and as what (Action.tx/rx/link_type, actions.py:9-15;
 * typing rule d2d_env.py:80-91).  n_links <= max_links.  Order = agent order of the outputs.       */
int d2d_set_links(d2d_handle* h, int32_t n_links, const int32_t* tx_dev, const int32_t* rx_dev,
                  const int32_t* link_type);

/* TrafficModel plugin (traffic_model.py:6-32; the hook the reference left commented out at
 * simulator.py:78): links whose (rb, tx_pwr_dBm) are NOT chosen by an agent.  link_idx[] indexes the
 * link list of the last d2d_set_links; rb[] / pwr_dbm[] are the constants UplinkTrafficModel /
 * DownlinkTrafficModel.get_traffic would produce (rb = k mod num_rbs, pwr = device max power).  They
 * are stored in the per-link records the step kernel reads anyway, so d2d_step then takes actions for
 * the remaining A = n_links - n_fixed links only: actions_dev is i32 [B, A], columns in link order
 * with the fixed links skipped.  (rb, pwr) are used as given - no a // P decode - so any power is
 * legal, as with the reference's Action(rb, pwr).  d2d_step_rb_pwr keeps the full [B, N] layout and
 * ignores the entries of fixed links.  n_fixed = 0 (or a new d2d_set_links) clears the set.          */
int d2d_set_fixed_actions(d2d_handle* h, int32_t n_fixed, const int32_t* link_idx, const int32_t* rb,
                          const int32_t* pwr_dbm);

/* RewardFunction / ObsFunction plugin selection (d2d_env.py:27-28).                                */
int d2d_set_reward(d2d_handle* h, int32_t reward_fn, float param);
int d2d_set_obs_mode(d2d_handle* h, int32_t obs_mode);
/* Element type of D2D_BUF_OBS under D2D_OBS_LINEAR.  D2D_F32 (default): the kernels' own type.  D2D_F64: the reference's
 * (LinearObsFunction builds float64 arrays, obs_fn.py:47,51) - the expansion kernel widens on the way out, so the block
 * is written once (48 N bytes per agent-step) instead of being cast by the caller afterwards (24 N written, 24 N read
 * back, 48 N written).  D2D_BUF_OBS is then f64 [B,N,6N] (twice the bytes; a bound buffer must be that large) and the
 * expansion always runs as its own launch.  d2d_step_host's packed block and d2d_expand_table stay float32.             */
typedef enum d2d_dtype { D2D_F32 = 0, D2D_F64 = 1 } d2d_dtype;
int d2d_set_obs_dtype(d2d_handle* h, int32_t dtype);
/* SystemCapacityRewardFunction hands the SAME scalar to every agent of an env (reward_fn.py:42-44).
 * D2D_REWARD_PER_AGENT (default): d2d_step writes it N times, D2D_BUF_REWARD f32 [B,N] - the dict the
 * reference returns.  D2D_REWARD_PER_ENV: written once per env to D2D_BUF_REWARD_ENV f32 [B]; D2D_BUF_REWARD
 * is then not touched (4 bytes per link and step less).  Only SystemCapacity is affected: the per-agent
 * rewards (Shannon, CueSinrShannon) and d2d_step_host always produce [B,N].                              */
typedef enum d2d_reward_layout { D2D_REWARD_PER_AGENT = 0, D2D_REWARD_PER_ENV = 1 } d2d_reward_layout;
int d2d_set_reward_layout(d2d_handle* h, int32_t layout);
/* 1 (default): same-RB interferers found through per-RB membership bitmasks in LDS
 * (Actions.get_actions_by_rb, actions.py:27-31).  0: masked all-pairs sweep.  Same results.        */
int d2d_set_bucketing(d2d_handle* h, int32_t enabled);
/* 1 (default): d2d_step writes the decoded (rb, tx power dBm) of every link to D2D_BUF_RB / D2D_BUF_PWR
 * - the `rb` / `tx_pwr_dbm` entries of D2DEnv._info (d2d_env.py:108-109).  0: a rollout that knows its
 * own actions skips those 8 bytes per link and step; the two buffers then keep their last contents.
 * d2d_step_rb_pwr (the values are the caller's) and d2d_step_host (always exported) are not affected.  */
int d2d_set_export_actions(d2d_handle* h, int32_t enabled);

/* Launch-geometry knobs (performance only, results do not change).  The A/B shapes the kernels were tuned against (other
 * store policies, the row-aligned / unstaged expansion kernels, the flattened mask walk, wave staggering) and the ablation
 * switch live in d2d_hip_diag.h and exist in diagnostic builds only; a release build answers them D2D_ERR_UNSUPPORTED.    */
typedef enum d2d_tuning {
    D2D_TUNE_OBS_ROWS_PER_WG = 0,  /* consecutive 16-KiB pieces of an env's obs block written per workgroup (1 .. 4); 0 = auto (2) */
    D2D_TUNE_OBS_NONTEMPORAL = 1,  /* store policy of the obs stream: 1 (default) nontemporal, 0 plain                */
    D2D_TUNE_OBS_XCD_REMAP = 2,    /* 1 (default): chunks of one env share an XCD                    */
    D2D_TUNE_OBS_BLOCK = 3,        /* threads per obs workgroup; 0 = auto                            */
    D2D_TUNE_STEP_THREADS = 5,     /* threads per ENV in the step kernel; 0 = auto (one per link).  Below half the link
                                      count no link sits in registers (strided kernel) and the per-RB search
                                      structures give way to the O(N^2) sweep                              */
    D2D_TUNE_STEP_ENVS_PER_WG = 6, /* envs sharing one step workgroup; 0 = auto                      */
    D2D_TUNE_STEP_BLOCK = 7,       /* threads per step workgroup (>= envs * threads/env); 0 = auto   */
    D2D_TUNE_STEP_FUSE_OBS = 8,    /* LinearObs expansion inside the step launch: 1 on, 0 off,
                                      -1 = auto (on for small N, where two launches are latency bound) */
    D2D_TUNE_STEP_WALK = 10,       /* same-RB interferer search: 0 membership masks (words / members walk), 2 per-RB member
                                      lists (an env that puts more than 8 links on one RB falls back to the masks inside
                                      the launch); -1 = auto                                                    */
    D2D_TUNE_STEP_PREFETCH = 11,   /* software-prefetch distance of the action rows, in envs: -1 = auto (the envs
                                      resident on the chip at once), 0 = off                                    */
    D2D_TUNE_STEP_LPT = 12,        /* links per thread held in registers: 1, 2 (half the waves per env; the rollout
                                      kernel: adjacent links 2t, 2t + 1, N a multiple of 128), -1 = auto            */
    D2D_TUNE_STEP_NT_RESULTS = 13, /* nontemporal result stores: 1 on (ignored with LinearObs, whose expansion kernel
                                      re-reads the table behind the step), 0 off, -1 = auto: on in the rollout kernel
                                      (d2d_rollout.hip), off elsewhere (no consistent gain measured there)         */
    D2D_TUNE_STEP_SCALAR_RECORDS = 14, /* rollout kernel: link records by one scalar load per wave when every aligned
                                      group of 64 links (128 with two links per thread) has identical records;
                                      -1 = auto (on when legal), 0 off                                           */
    D2D_TUNE_STEP_OBS_ROTATE = 15  /* fused LinearObs expansion: workgroup w starts at step (w * value) mod (its steps) of
                                      its (env, pass) store sequence and wraps, so that the concurrent stores of the
                                      resident workgroups are not one fixed stride apart; -1 = auto (29), 0 = off  */
} d2d_tuning;
int d2d_set_tuning(d2d_handle* h, int32_t key, int32_t value);

/* ---- buffers --------------------------------------------------------------------------------- */
/* Device pointer + size in bytes of a buffer (allocated on first use unless bound).                */
int d2d_get_buffer(d2d_handle* h, int32_t which, void** dev_ptr, size_t* bytes);
/* Use caller-owned device memory for a buffer (e.g. a torch tensor); never freed by the library.   */
int d2d_bind_buffer(d2d_handle* h, int32_t which, void* dev_ptr, size_t bytes);
/* Synchronous host<->device copies (ordered after queued work on the handle's stream).             */
int d2d_upload(d2d_handle* h, int32_t which, const void* host_src, size_t bytes, size_t dst_offset);
int d2d_download(d2d_handle* h, int32_t which, void* host_dst, size_t bytes, size_t src_offset);
/* Simulator.reset (simulator.py:61-75) with host-supplied positions: x,y [env_count, D] f32.       */
int d2d_set_positions(d2d_handle* h, const float* x, const float* y, int32_t env_begin, int32_t env_count);

/* The same in the REFERENCE'S OWN precision: Position holds Python floats (position.py:7-12), the samplers and a
 * device_config_file produce float64 coordinates (position.py:18-45, simulator.py:61-75, d2d_env.py:124-134).  x, y
 * [env_count, D] float64.  Each coordinate is kept as hi + lo, hi = the nearest float32 (what POS_X / POS_Y, LINK_POS, the obs
 * table and D2D_BUF_OBS hold - the float32 rounding of the reference's value), lo = the float32 nearest to the remainder; the
 * step then forms every tx - rx difference as (tx_hi - rx_hi) + (tx_lo - rx_lo), exact to ~1e-7 of the DIFFERENCE.  (float32
 * coordinates carry 3e-5 m at 500 m: a receiver 0.1 m from its transmitter is then off by 3e-4 relative - 2e-5 on sinr_db,
 * twice the 1e-5 bar, from input rounding alone.)  When every value is float32-representable the call is d2d_set_positions:
 * same kernels, same bits.  Any other way of writing positions (d2d_set_positions on ALL envs, d2d_upload / d2d_bind_buffer of
 * POS_X / POS_Y, d2d_reset_positions - whose sampler draws float32 coordinates -, d2d_positions_changed) returns the handle to
 * float32 positions; d2d_set_positions on a sub-range keeps the other envs' low parts.  Costs one more 16-byte row per link and
 * step and the one-link-per-thread kernels (4096 x 512: see DESIGN.md 4.1).                                                  */
int d2d_set_positions_f64(d2d_handle* h, const double* x, const double* y, int32_t env_begin, int32_t env_count);

/* The step kernel reads per-LINK position rows (tx_x, tx_y, rx_x, rx_y) that the library derives from
 * POS_X / POS_Y whenever it knows they changed (d2d_set_positions, d2d_upload, d2d_reset_positions,
 * d2d_bind_buffer, d2d_set_links).  A caller that writes device positions straight into a BOUND
 * buffer (its own kernel / a torch op) must say so before the next step.                             */
int d2d_positions_changed(d2d_handle* h);

/* Simulator.reset on the device (simulator.py:61-75; samplers position.py:18-45) for all B envs:
 * BS at the origin, CUEs / DUE transmitters uniform in the cell disc, DUE receivers uniform within
 * d2d_radius_m of their transmitter and re-drawn until inside the cell.  Counter-based Philox4x32-10
 * stream keyed by `seed`, counter (first_env + env, device, try, episode) - see csrc/d2d_reset.hip.
 * fixed_mask[D] / fixed_xy[D,2] (host, may be NULL): devices pinned by a device_config_file
 * (simulator.py:65-66).  Asynchronous.                                                             */
int d2d_reset_positions(d2d_handle* h, uint64_t seed, uint64_t episode, const uint8_t* fixed_mask,
                        const float* fixed_xy);
/* Global index of this handle's env 0 (multi-GPU sharding of one logical batch); default 0.        */
int d2d_set_env_offset(d2d_handle* h, uint64_t first_env);

/* ---- the hot path ---------------------------------------------------------------------------- */
/* D2DEnv.step (d2d_env.py:62-71) for all B envs: decode -> Simulator.step (simulator.py:77-154) ->
 * reward -> obs.  actions_dev: i32 [B,N] device pointer, or NULL to read D2D_BUF_ACTIONS.           */
int d2d_step(d2d_handle* h, const int32_t* actions_dev);
/* Same with (rb, tx_pwr_dBm) given explicitly - the ndarray action form (d2d_env.py:97-98).
 * NULL pointers read D2D_BUF_RB / D2D_BUF_PWR.                                                     */
int d2d_step_rb_pwr(d2d_handle* h, const int32_t* rb_dev, const int32_t* pwr_dev);
/* LinearObsFunction.get_state (obs_fn.py:43-53) on a table the CALLER holds - e.g. the compact tables a
 * learner received from other GPUs: table_dev f32 [n_envs, n_links, 6] -> obs_dev f32
 * [n_envs, n_links, 6*n_links], same kernel and bit-identical to the D2D_BUF_OBS the owning GPU
 * produced.  Independent of the handle's own B / N; asynchronous on the handle's stream.            */
int d2d_expand_table(d2d_handle* h, const float* table_dev, int32_t n_envs, int32_t n_links, float* obs_dev);

/* D2DEnv.step for host callers (the single-env drop-in, d2d_env.py:62-71): (rb, pwr) [B,N] from HOST
 * memory in, every result of the step back in ONE pinned host block - one H2D copy, the kernels, one
 * D2H copy, one synchronisation.  *out_host points at library-owned pinned memory laid out as
 * d2d_host_layout says (offsets in bytes), valid until the next call on this handle.  The D2D_BUF_*
 * output buffers are NOT updated by this entry.  (Blocks of up to 256 KB are written by the kernels straight
 * into the pinned host memory - no copy commands; larger ones are staged in device memory.)  As with d2d_step_rb_pwr, a link that carries a fixed
 * action (d2d_set_fixed_actions) keeps it: the caller's (rb, pwr) entries for that link are ignored and
 * the returned rb / pwr regions hold the values the kernel used.                                      */
typedef struct d2d_host_layout {
    size_t sinr_db, snr_db, rate_bps, capacity, reward;   /* f32 [B,N] each                          */
    size_t rb, pwr;                                       /* i32 [B,N]                               */
    size_t obs_table;                                     /* f32 [B,N,6]                             */
    size_t env_flags;                                     /* i32 [B]                                 */
    size_t obs;                                           /* f32 [B,N,6N] (obs mode LINEAR only)     */
    size_t total_bytes;
} d2d_host_layout;
int d2d_step_host(d2d_handle* h, const int32_t* rb_host, const int32_t* pwr_host, const void** out_host,
                  d2d_host_layout* layout);

/* OR of D2D_FLAG_* over all envs of the last step (synchronises the stream).                       */
int d2d_status_flags(d2d_handle* h, uint32_t* flags);

/* ---- multi-GPU: the end-of-step concat --------------------------------------- */
/* Environments are independent (simulator.py:95, reward_fn.py:42, obs_fn.py:46-51), so one process per
 * GPU steps its own env block with no exchange; the only collective is an all-gather of what a single
 * learner reads at the end of a step.  These entry points run it over RCCL (xGMI) without Python:
 * rank 0 makes an id (d2d_comm_unique_id), ships the 128 bytes to the other ranks by any means, every
 * rank calls d2d_comm_init, then d2d_allgather(send, recv, bytes, stream) enqueues ncclAllGather:
 * recv_dev holds world * bytes_per_rank, rank-major = global env order.  hip_stream = NULL runs it on the
 * handle's stream (in order with the step); a caller that wants the gather to overlap the obs expansion
 * and the next step passes its own side stream and orders it with events, as StepGatherer does.
 * librccl is dlopen()ed on first use (D2D_RCCL_LIBRARY overrides the path), so single-GPU users need no
 * RCCL at all.                                                                                          */
int d2d_comm_unique_id(void* id_out /* D2D_UNIQUE_ID_BYTES */);
int d2d_comm_init(d2d_handle* h, int32_t world_size, int32_t rank, const void* unique_id);
int d2d_comm_destroy(d2d_handle* h);
int d2d_allgather(d2d_handle* h, const void* send_dev, void* recv_dev, size_t bytes_per_rank, void* hip_stream);

/* ---- measurement ----------------------------------------------------------------------------- */
/* When enabled, every kernel launched by d2d_step is bracketed by hipEvents on the handle's stream. */
int d2d_profile_enable(d2d_handle* h, int32_t enabled);
/* kernel: 0 = step (decode+SINR+reward+table), 1 = LinearObs expansion.  Synchronises, then returns
 * accumulated device time in ms and launch count since the last reset.                             */
int d2d_profile_read(d2d_handle* h, int32_t kernel, double* total_ms, int64_t* launches);
/* Median duration (ms) of that kernel's launches since the last reset (the first 65536 of them): the figure a
 * roofline is quoted on beside the mean, which a cold first launch or a clock ramp pulls up.               */
int d2d_profile_median(d2d_handle* h, int32_t kernel, double* median_ms);
int d2d_profile_reset(d2d_handle* h);

#ifdef __cplusplus
}
#endif
#endif /* D2D_HIP_H */
