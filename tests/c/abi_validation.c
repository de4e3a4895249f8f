/* Argument validation of every C-ABI entry point (include/d2d_hip.h), for the host sanitizer pass: built with
 * -fsanitize=address,undefined against a host-instrumented libd2d_hip (tests/test_sanitizers_cpu.py).  Every call below is
 * WRONG on purpose - null handles, null arrays, sizes out of range - and must come back with an error code and a message,
 * never a crash and never a sanitizer report.  Runs without a GPU: d2d_create itself then fails with D2D_ERR_HIP, which
 * is also checked; with a GPU the valid handle is used for the second half (state errors, range checks).            */
#include <stdio.h>
#include <string.h>

#include "d2d_hip.h"

static int failures = 0;
#define EXPECT_ERR(call)                                                                        \
    do {                                                                                        \
        int rc_ = (call);                                                                       \
        if (rc_ == D2D_OK || d2d_last_error()[0] == 0) {                                        \
            fprintf(stderr, "%s:%d: %s returned %d (%s)\n", __FILE__, __LINE__, #call, rc_, d2d_last_error()); \
            ++failures;                                                                         \
        }                                                                                       \
    } while (0)

int main(void) {
    double one[8] = {1, 1, 1, 1, 1, 1, 1, 1};
    float fone[8] = {0};
    int32_t ione[8] = {0};
    uint8_t mask[8] = {0};
    double rate = 0.0;
    void* p = NULL;
    size_t n = 0;
    uint32_t flags = 0;
    int64_t launches = 0;
    char id[D2D_UNIQUE_ID_BYTES];
    d2d_host_layout lay;
    const void* out = NULL;

    if (d2d_abi_version() != D2D_ABI_VERSION) { fprintf(stderr, "abi version\n"); return 2; }
    /* null handle: every entry point */
    EXPECT_ERR(d2d_set_stream(NULL, NULL));
    EXPECT_ERR(d2d_synchronize(NULL));
    EXPECT_ERR(d2d_set_device_table(NULL, 1, one, one, one, one, one));
    EXPECT_ERR(d2d_set_path_loss_power_law(NULL, 1, one, one, one));
    EXPECT_ERR(d2d_set_path_loss_shadowing(NULL, 1, one, one, one, 1.0, 1.0, 1));
    EXPECT_ERR(d2d_set_path_loss_table(NULL, one, 0));
    EXPECT_ERR(d2d_set_path_loss_link_table(NULL, one, 1, 0));
    EXPECT_ERR(d2d_set_path_loss_link_table_dev(NULL, one, D2D_F64, 1, 0));
    EXPECT_ERR(d2d_set_links(NULL, 1, ione, ione, ione));
    EXPECT_ERR(d2d_set_fixed_actions(NULL, 1, ione, ione, ione));
    EXPECT_ERR(d2d_positions_changed(NULL));
    EXPECT_ERR(d2d_set_reward(NULL, 1, 0.0f));
    EXPECT_ERR(d2d_set_reward_layout(NULL, 0));
    EXPECT_ERR(d2d_set_obs_mode(NULL, 1));
    EXPECT_ERR(d2d_set_obs_dtype(NULL, D2D_F64));
    EXPECT_ERR(d2d_set_bucketing(NULL, 1));
    EXPECT_ERR(d2d_set_export_actions(NULL, 1));
    EXPECT_ERR(d2d_set_tuning(NULL, 0, 0));
    EXPECT_ERR(d2d_get_buffer(NULL, 0, &p, &n));
    EXPECT_ERR(d2d_bind_buffer(NULL, 0, NULL, 0));
    EXPECT_ERR(d2d_upload(NULL, 0, fone, 4, 0));
    EXPECT_ERR(d2d_download(NULL, 0, fone, 4, 0));
    EXPECT_ERR(d2d_set_positions(NULL, fone, fone, 0, 1));
    EXPECT_ERR(d2d_set_positions_f64(NULL, one, one, 0, 1));
    EXPECT_ERR(d2d_reset_positions(NULL, 1, 0, mask, fone));
    EXPECT_ERR(d2d_set_env_offset(NULL, 0));
    EXPECT_ERR(d2d_step(NULL, NULL));
    EXPECT_ERR(d2d_step_rb_pwr(NULL, NULL, NULL));
    EXPECT_ERR(d2d_expand_table(NULL, fone, 1, 1, fone));
    EXPECT_ERR(d2d_step_host(NULL, ione, ione, &out, &lay));
    EXPECT_ERR(d2d_status_flags(NULL, &flags));
    EXPECT_ERR(d2d_comm_unique_id(NULL));
    EXPECT_ERR(d2d_comm_init(NULL, 1, 0, id));
    EXPECT_ERR(d2d_comm_destroy(NULL));
    EXPECT_ERR(d2d_allgather(NULL, fone, fone, 4, NULL));
    EXPECT_ERR(d2d_profile_enable(NULL, 1));
    EXPECT_ERR(d2d_profile_read(NULL, 0, &rate, &launches));
    EXPECT_ERR(d2d_profile_reset(NULL));
    EXPECT_ERR(d2d_profile_median(NULL, 0, &rate));
    if (d2d_destroy(NULL) != D2D_OK) { fprintf(stderr, "d2d_destroy(NULL) must be a no-op\n"); ++failures; }

    /* d2d_create: every field that can be wrong */
    d2d_config good, cfg;
    d2d_handle* h = NULL;
    memset(&good, 0, sizeof good);
    good.abi_version = D2D_ABI_VERSION;
    good.num_envs = 3; good.num_rbs = 2; good.num_cues = 2; good.num_due_pairs = 2;
    good.pwr_levels_due = 21; good.pwr_levels_cue = 24; good.pwr_levels_mbs = 47;
    good.cell_radius_m = 500.0f; good.d2d_radius_m = 20.0f;
    EXPECT_ERR(d2d_create(NULL, &h));
    EXPECT_ERR(d2d_create(&good, NULL));
    cfg = good; cfg.abi_version = D2D_ABI_VERSION - 1; EXPECT_ERR(d2d_create(&cfg, &h));
    cfg = good; cfg.num_envs = 0; EXPECT_ERR(d2d_create(&cfg, &h));
    cfg = good; cfg.num_rbs = 0; EXPECT_ERR(d2d_create(&cfg, &h));
    cfg = good; cfg.num_cues = -1; EXPECT_ERR(d2d_create(&cfg, &h));
    cfg = good; cfg.pwr_levels_due = 0; EXPECT_ERR(d2d_create(&cfg, &h));
    cfg = good; cfg.pwr_levels_cue = 70000; EXPECT_ERR(d2d_create(&cfg, &h));
    cfg = good; cfg.max_links = D2D_MAX_LINKS + 1; EXPECT_ERR(d2d_create(&cfg, &h));
    cfg = good; cfg.num_cues = 0; cfg.num_due_pairs = 0; EXPECT_ERR(d2d_create(&cfg, &h));
    cfg = good; cfg.device_ordinal = 1000; EXPECT_ERR(d2d_create(&cfg, &h));
#ifdef D2D_TEST_HOOKS
    /* the exception barrier (no C++ exception crosses the boundary): a library built with -DD2D_TEST_HOOKS=1
     * throws inside d2d_create on these sentinels; what arrives here must be a status code and a message */
    cfg = good; cfg.num_envs = -12345;
    { int rc_ = d2d_create(&cfg, &h); if (rc_ != D2D_ERR_NO_MEMORY || d2d_last_error()[0] == 0) { fprintf(stderr, "bad_alloc came back as %d\n", rc_); ++failures; } }
    cfg = good; cfg.num_envs = -12346;
    { int rc_ = d2d_create(&cfg, &h); if (rc_ != D2D_ERR_STATE || strstr(d2d_last_error(), "test hook") == NULL) { fprintf(stderr, "runtime_error came back as %d (%s)\n", rc_, d2d_last_error()); ++failures; } }
    cfg = good; cfg.num_envs = -12347;
    { int rc_ = d2d_create(&cfg, &h); if (rc_ != D2D_ERR_STATE || d2d_last_error()[0] == 0) { fprintf(stderr, "a foreign exception came back as %d\n", rc_); ++failures; } }
#endif
    if (h != NULL) { fprintf(stderr, "a failed d2d_create must leave *out NULL\n"); ++failures; }

    int rc = d2d_create(&good, &h);
    if (rc != D2D_OK) {
        /* no usable GPU (the CPU container): that is an error code and a message too, not a crash */
        if (rc != D2D_ERR_HIP && rc != D2D_ERR_UNSUPPORTED) { fprintf(stderr, "d2d_create without a GPU: rc %d\n", rc); ++failures; }
        if (d2d_last_error()[0] == 0) { fprintf(stderr, "d2d_create without a GPU: no message\n"); ++failures; }
        printf("{\"gpu\": 0, \"failures\": %d}\n", failures);
        return failures ? 1 : 0;
    }
    /* a live handle: call-order and range errors */
    enum { D = 7 };
    EXPECT_ERR(d2d_step(h, NULL));                                   /* nothing set yet */
    EXPECT_ERR(d2d_set_device_table(h, D - 1, one, one, one, one, one));
    EXPECT_ERR(d2d_set_device_table(h, D, NULL, one, one, one, one));
    EXPECT_ERR(d2d_set_path_loss_power_law(h, D + 1, one, one, one));
    EXPECT_ERR(d2d_set_path_loss_table(h, NULL, 0));
    EXPECT_ERR(d2d_set_path_loss_link_table_dev(h, NULL, D2D_F64, 1, 0));
    EXPECT_ERR(d2d_set_path_loss_link_table_dev(h, one, 9, 1, 0));   /* unknown dtype */
    EXPECT_ERR(d2d_set_path_loss_link_table_dev(h, one, D2D_F64, 1, 0));   /* before d2d_set_links */
    EXPECT_ERR(d2d_set_positions_f64(h, NULL, one, 0, 1));
    EXPECT_ERR(d2d_set_positions_f64(h, one, one, 0, 100000));       /* env range */
    { const int32_t tx[2] = {1, 9}, rx[2] = {0, 0}, ty[2] = {1, 1}; EXPECT_ERR(d2d_set_links(h, 2, tx, rx, ty)); }
    { const int32_t tx[2] = {1, 2}, rx[2] = {0, 0}, ty[2] = {1, 7}; EXPECT_ERR(d2d_set_links(h, 2, tx, rx, ty)); }
    EXPECT_ERR(d2d_set_links(h, 5000, ione, ione, ione));
    EXPECT_ERR(d2d_set_fixed_actions(h, 1, ione, ione, ione));       /* before d2d_set_links */
    EXPECT_ERR(d2d_set_reward(h, 9, 0.0f));
    EXPECT_ERR(d2d_set_reward_layout(h, 2));
    EXPECT_ERR(d2d_set_obs_mode(h, 3));
    EXPECT_ERR(d2d_set_obs_dtype(h, 9));
    EXPECT_ERR(d2d_set_tuning(h, 999, 0));
    EXPECT_ERR(d2d_set_tuning(h, D2D_TUNE_STEP_BLOCK, 100));
    EXPECT_ERR(d2d_get_buffer(h, D2D_BUF_COUNT, &p, &n));
    EXPECT_ERR(d2d_get_buffer(h, D2D_BUF_LINK_POS, &p, &n));         /* no links / positions yet */
    EXPECT_ERR(d2d_bind_buffer(h, -1, NULL, 0));
    EXPECT_ERR(d2d_bind_buffer(h, D2D_BUF_LINK_POS, fone, 16));
    EXPECT_ERR(d2d_upload(h, D2D_BUF_POS_X, fone, (size_t)1 << 30, 0));
    EXPECT_ERR(d2d_download(h, D2D_BUF_SINR_DB, fone, 4, 0));        /* never written */
    EXPECT_ERR(d2d_set_positions(h, fone, fone, 2, 5));
    EXPECT_ERR(d2d_reset_positions(h, 1, 0, mask, NULL));
    EXPECT_ERR(d2d_expand_table(h, NULL, 1, 1, NULL));
    EXPECT_ERR(d2d_allgather(h, fone, fone, 4, NULL));               /* no communicator */
    EXPECT_ERR(d2d_comm_init(h, 2, 5, id));
    EXPECT_ERR(d2d_profile_read(h, 7, &rate, &launches));
    EXPECT_ERR(d2d_profile_median(h, 2, &rate));
    EXPECT_ERR(d2d_set_path_loss_link_table(h, one, 2, 0));          /* before d2d_set_links */
    EXPECT_ERR(d2d_set_tuning(h, 9 /* D2D_TUNE_STEP_ABLATE, d2d_hip_diag.h */, 1));   /* release build: diagnostic keys refused */
    EXPECT_ERR(d2d_set_tuning(h, D2D_TUNE_STEP_WALK, 1));           /* the flattened walk: diagnostic builds only */
    {   /* a per-env table of 2^20 envs x 7 x 7 doubles is never read past the handle's own 3 x 7 x 7: sizes come from the handle */
        const int32_t tx[2] = {1, 2}, rx[2] = {0, 0}, ty[2] = {1, 1};
        if (d2d_set_links(h, 2, tx, rx, ty) != D2D_OK) { fprintf(stderr, "d2d_set_links failed: %s\n", d2d_last_error()); ++failures; }
        EXPECT_ERR(d2d_set_path_loss_link_table(h, one, 3, 0));      /* not the length of the link list */
    }
    if (d2d_destroy(h) != D2D_OK) { fprintf(stderr, "d2d_destroy failed\n"); ++failures; }
    printf("{\"gpu\": 1, \"failures\": %d}\n", failures);
    return failures ? 1 : 0;
}
