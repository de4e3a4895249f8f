/* Plain C client of include/d2d_hip.h: proves the boundary needs nothing but a C compiler and the shared library
 * (no Python, no torch, no C++).  One env, 2 CUEs + 2 DUE pairs, LogDistance; prints the SINRs as JSON.
 *   gcc -std=c99 -I include tests/c/abi_smoke.c -L gym_d2d_amd/lib -ld2d_hip -Wl,-rpath,$PWD/gym_d2d_amd/lib -lm
 */
#include <math.h>
#include <stdio.h>
#include <string.h>

#include "d2d_hip.h"

#define CK(call)                                                              \
    do {                                                                      \
        int rc_ = (call);                                                     \
        if (rc_ != D2D_OK) {                                                  \
            fprintf(stderr, "%s -> %d: %s\n", #call, rc_, d2d_last_error());  \
            return 1;                                                         \
        }                                                                     \
    } while (0)

int main(void) {
    d2d_config cfg;
    memset(&cfg, 0, sizeof cfg);
    cfg.abi_version = D2D_ABI_VERSION;
    cfg.num_envs = 1; cfg.num_rbs = 2; cfg.num_cues = 2; cfg.num_due_pairs = 2;
    cfg.pwr_levels_due = 21; cfg.pwr_levels_cue = 24; cfg.pwr_levels_mbs = 47;   /* d2d_env.py:31-35 */
    cfg.cell_radius_m = 500.0f; cfg.d2d_radius_m = 20.0f;
    d2d_handle* h = NULL;
    CK(d2d_create(&cfg, &h));

    enum { D = 7, N = 4 };
    /* device.py defaults: BS first, then UEs.  eirp offset, rx offset, thermal noise, sensitivity, RB bandwidth */
    double eirp[D], rxo[D], noise[D], sens[D], bw[D], a_tx[D], a_rx[D], ple[D];
    for (int d = 0; d < D; ++d) {
        const int bs = d == 0;
        eirp[d] = bs ? 17.5 - 2.0 - 2.0 + 2.0 : 0.0 - 3.0 - 3.0;
        rxo[d] = bs ? 17.5 - 2.0 + 2.0 : 0.0 - 3.0;
        noise[d] = bs ? -118.4 : -104.5;
        sens[d] = bs ? 2.0 - 118.4 - 7.0 : 7.0 - 104.5 - 10.0;
        bw[d] = 180000.0;
        a_tx[d] = 20.0 * log10(2.1e9) + 20.0 * log10(4.0 * 3.14159265358979323846 / 299792458.0);  /* path_loss.py:28-39 */
        a_rx[d] = 0.0; ple[d] = 2.0;
    }
    CK(d2d_set_device_table(h, D, eirp, rxo, noise, sens, bw));
    CK(d2d_set_path_loss_power_law(h, D, a_tx, a_rx, ple));
    const int32_t tx[N] = {1, 2, 3, 5}, rx[N] = {0, 0, 4, 6}, ty[N] = {D2D_UPLINK, D2D_UPLINK, D2D_SIDELINK, D2D_SIDELINK};
    CK(d2d_set_links(h, N, tx, rx, ty));
    const float x[D] = {0, 100, -200, 50, 55, -300, -310}, y[D] = {0, 50, 120, -80, -70, 10, 25};
    CK(d2d_set_positions(h, x, y, 0, 1));
    CK(d2d_set_obs_mode(h, D2D_OBS_LINEAR));
    const int32_t actions[N] = {0 * 24 + 23, 1 * 24 + 10, 0 * 21 + 20, 1 * 21 + 5};      /* rb * P + pwr */
    CK(d2d_upload(h, D2D_BUF_ACTIONS, actions, sizeof actions, 0));
    CK(d2d_step(h, NULL));
    uint32_t flags = 0;
    CK(d2d_status_flags(h, &flags));
    float sinr[N], reward[N], obs[N * 6 * N];
    CK(d2d_download(h, D2D_BUF_SINR_DB, sinr, sizeof sinr, 0));
    CK(d2d_download(h, D2D_BUF_REWARD, reward, sizeof reward, 0));
    CK(d2d_download(h, D2D_BUF_OBS, obs, sizeof obs, 0));
    /* the host-caller's transport (d2d_step_host): explicit (rb, pwr) in, every result in ONE pinned block out */
    const int32_t rb[N] = {0, 1, 0, 1}, pw[N] = {23, 10, 20, 5};
    const void* block = NULL;
    d2d_host_layout lay;
    CK(d2d_step_host(h, rb, pw, &block, &lay));
    const float* h_sinr = (const float*)((const char*)block + lay.sinr_db);
    const float* h_obs = (const float*)((const char*)block + lay.obs);
    int host_match = lay.total_bytes > lay.obs;
    for (int i = 0; i < N; ++i) host_match &= h_sinr[i] == sinr[i];
    for (int i = 0; i < N * 6 * N; ++i) host_match &= h_obs[i] == obs[i];
    /* traffic-model links (d2d_set_fixed_actions): links 0 and 1 carry their (rb, pwr) in the link records, the action
     * array shrinks to the two sidelinks - same step, same numbers */
    const int32_t fixed_idx[2] = {0, 1}, fixed_rb[2] = {0, 1}, fixed_pw[2] = {23, 10};
    CK(d2d_set_fixed_actions(h, 2, fixed_idx, fixed_rb, fixed_pw));
    const int32_t due_actions[2] = {0 * 21 + 20, 1 * 21 + 5};
    CK(d2d_upload(h, D2D_BUF_ACTIONS, due_actions, sizeof due_actions, 0));
    CK(d2d_step(h, NULL));
    float sinr2[N];
    CK(d2d_download(h, D2D_BUF_SINR_DB, sinr2, sizeof sinr2, 0));
    int fixed_match = 1;
    for (int i = 0; i < N; ++i) fixed_match &= sinr2[i] == sinr[i];
    printf("{\"flags\": %u, \"sinr_db\": [%.6f, %.6f, %.6f, %.6f], \"reward\": %.6f, \"obs_1_0\": %.3f, "
           "\"host_match\": %d, \"fixed_match\": %d}\n", flags, sinr[0], sinr[1], sinr[2], sinr[3], reward[0],
           obs[1 * 6 * N + 0], host_match, fixed_match);
    CK(d2d_destroy(h));
    return 0;
}
