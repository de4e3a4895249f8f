/* CPU oracle for the GymD2D per-step path in plain C (float64).  TEST INFRASTRUCTURE ONLY - the same standing as
 * oracle/d2d_oracle.py: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may build, load or call it;
 * nothing under gym_d2d_amd/ does.
 *
 * What it is for: (1) a second, independently written checker (tests/test_oracle_c.py holds it to the NumPy oracle and,
 * through it, to the reference's golden vectors, <= 1e-12 relative); (2) a CPU baseline that is not limited by NumPy's
 * temporaries: the reference's algorithm in straight loops, one thread or every core (OpenMP over envs).
 *
 * Scope: log-distance path loss (the default model, per-call exponent), raw-action decode, SINR / SNR / rate /
 * capacity, SystemCapacity reward, compact obs table and LinearObs expansion.  Everything follows the reference's
 * dB-domain arithmetic literally; citations are file:line under /root/reference/src/gym_d2d.
 *
 * Build: gcc -O2 -fopenmp -shared -fPIC oracle/c/d2d_oracle.c -o oracle/c/libd2d_oracle_c.so -lm   (oracle/c/build.sh)
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

#define SIDELINK 3 /* link_type.py:4-7 */

static double db_to_linear(double db) { return pow(10.0, db / 10.0); }   /* conversion.py:4-13 */
static double linear_to_db(double x) { return 10.0 * log10(x); }         /* conversion.py:16-25 */

typedef struct {
    int32_t B, D, N, R;
    const double* pos;          /* [B, D, 2] */
    const int32_t* link_tx;     /* [N] device index of the transmitter */
    const int32_t* link_rx;     /* [N] */
    const int32_t* link_type;   /* [N] 1 uplink, 2 downlink, 3 sidelink */
    const int32_t* pwr_levels;  /* [N] size of the link's power alphabet (d2d_env.py:31-40) */
    const int64_t* actions;     /* [B, N] raw ints */
    const double* eirp_off_db;  /* [D] device.py:51-60 */
    const double* rx_off_db;    /* [D] device.py:62-72 */
    const double* noise_dbm;    /* [D] device.py:118-119 */
    const double* sens_dbm;     /* [D] device.py:74-80 */
    const double* bw_hz;        /* [D] device.py:85-95 */
    double ple, pl_const_db;    /* path_loss.py:28-39,65-66 */
    double min_capacity_mbps;   /* reward_fn.py:22-25 */
    /* outputs, any may be NULL */
    int64_t* rb;                /* [B, N] */
    int64_t* pwr;               /* [B, N] */
    double* sinr_db;            /* [B, N] */
    double* snr_db;
    double* rate_bps;
    double* capacity_mbps;
    double* reward;             /* [B] SystemCapacity */
    double* table;              /* [B, N, 6] */
    double* obs;                /* [B, N, 6N] */
    int32_t threads;            /* OpenMP threads over envs; <= 1: serial */
} oracle_args;

/* Python floor division / modulo of a by p > 0 (d2d_env.py:94-96) */
static void floor_divmod(int64_t a, int64_t p, int64_t* q, int64_t* r) {
    int64_t qq = a / p, rr = a - qq * p;
    if (rr < 0) { rr += p; qq -= 1; }
    *q = qq; *r = rr;
}

static void one_env(const oracle_args* a, int b, int64_t* rb, int64_t* pw, double* sinr, double* snr, double* cap) {
    const int N = a->N, D = a->D;
    const double* pos = a->pos + (size_t)b * D * 2;
    for (int i = 0; i < N; ++i)
        floor_divmod(a->actions[(size_t)b * N + i], a->pwr_levels[i], &rb[i], &pw[i]);
    for (int i = 0; i < N; ++i) {
        const int ti = a->link_tx[i], ri = a->link_rx[i];
        const double rx_x = pos[2 * ri], rx_y = pos[2 * ri + 1];
        /* simulator.py:93  received signal of the link itself */
        double dx = pos[2 * ti] - rx_x, dy = pos[2 * ti + 1] - rx_y;
        double dist = sqrt(dx * dx + dy * dy);                                        /* position.py:11-12 */
        double pl = 10.0 * a->ple * log10(dist) + a->pl_const_db;                     /* path_loss.py:65-66 */
        const double sig = (double)pw[i] + a->eirp_off_db[ti] - pl + a->rx_off_db[ri];
        /* simulator.py:95-101  interferers = the other links on the same RB, summed in mW, ascending link index */
        double sum_ix = 0.0;
        for (int j = 0; j < N; ++j) {
            if (j == i || rb[j] != rb[i]) continue;
            const int tj = a->link_tx[j];
            dx = pos[2 * tj] - rx_x; dy = pos[2 * tj + 1] - rx_y;
            dist = sqrt(dx * dx + dy * dy);
            pl = 10.0 * a->ple * log10(dist) + a->pl_const_db;
            sum_ix += db_to_linear((double)pw[j] + a->eirp_off_db[tj] - pl);
        }
        const double noise = a->noise_dbm[ri];
        sinr[i] = sig - linear_to_db(sum_ix + db_to_linear(noise));                  /* simulator.py:106-107 */
        snr[i] = sig - noise;                                                         /* simulator.py:114-115 */
        const int ok = sinr[i] > a->sens_dbm[ri];                                     /* simulator.py:123,149 */
        const double shannon = log2(1.0 + db_to_linear(sinr[i]));
        cap[i] = ok ? 1e-6 * a->bw_hz[ti] * shannon : 0.0;                            /* simulator.py:150-153 */
        if (a->rate_bps) a->rate_bps[(size_t)b * N + i] = ok ? shannon : 0.0;         /* simulator.py:124-126 */
    }
}

int d2d_oracle_step(const oracle_args* a) {
    const int B = a->B, N = a->N, D = a->D;
    if (B < 0 || N < 1 || D < 1) return 1;
    int failed = 0;
#ifdef _OPENMP
#pragma omp parallel num_threads(a->threads > 1 ? a->threads : 1) if (a->threads > 1)
#endif
    {
        int64_t* rb = (int64_t*)malloc(sizeof(int64_t) * 2 * (size_t)N);
        double* f = (double*)malloc(sizeof(double) * (3 + 6) * (size_t)N);
        if (!rb || !f) {
#ifdef _OPENMP
#pragma omp atomic write
#endif
            failed = 1;
        } else {
            int64_t* pw = rb + N;
            double *sinr = f, *snr = f + N, *cap = f + 2 * N, *t = f + 3 * N;
#ifdef _OPENMP
#pragma omp for schedule(static)
#endif
            for (int b = 0; b < B; ++b) {
                one_env(a, b, rb, pw, sinr, snr, cap);
                const size_t row = (size_t)b * N;
                if (a->rb) memcpy(a->rb + row, rb, sizeof(int64_t) * N);
                if (a->pwr) memcpy(a->pwr + row, pw, sizeof(int64_t) * N);
                if (a->sinr_db) memcpy(a->sinr_db + row, sinr, sizeof(double) * N);
                if (a->snr_db) memcpy(a->snr_db + row, snr, sizeof(double) * N);
                if (a->capacity_mbps) memcpy(a->capacity_mbps + row, cap, sizeof(double) * N);
                if (a->reward) {
                    /* reward_fn.py:27-44  mean capacity, or -1 if a non-D2D link at or below min capacity shares its RB
                     * with a D2D link */
                    double total = 0.0;
                    int violated = 0;
                    for (int i = 0; i < N; ++i) {
                        total += cap[i];
                        if (a->link_type[i] != SIDELINK && cap[i] <= a->min_capacity_mbps)
                            for (int j = 0; j < N && !violated; ++j)
                                if (j != i && rb[j] == rb[i] && a->link_type[j] == SIDELINK) violated = 1;
                    }
                    a->reward[b] = violated ? -1.0 : total / N;
                }
                if (a->table || a->obs) {
                    const double* pos = a->pos + (size_t)b * D * 2;
                    for (int i = 0; i < N; ++i) {                                     /* obs_fn.py:55-61 */
                        const int ti = a->link_tx[i], ri = a->link_rx[i];
                        t[6 * i + 0] = pos[2 * ti]; t[6 * i + 1] = pos[2 * ti + 1];
                        t[6 * i + 2] = pos[2 * ri]; t[6 * i + 3] = pos[2 * ri + 1];
                        t[6 * i + 4] = sinr[i]; t[6 * i + 5] = snr[i];
                    }
                    if (a->table) memcpy(a->table + row * 6, t, sizeof(double) * 6 * N);
                    if (a->obs) {
                        /* obs_fn.py:43-53  row of agent i = its own six values, then every other link's in link order */
                        double* out = a->obs + row * 6 * N;
                        for (int i = 0; i < N; ++i) {
                            double* o = out + (size_t)i * 6 * N;
                            memcpy(o, t + 6 * i, sizeof(double) * 6);
                            memcpy(o + 6, t, sizeof(double) * 6 * (size_t)i);
                            memcpy(o + 6 + 6 * (size_t)i, t + 6 * (i + 1), sizeof(double) * 6 * (size_t)(N - 1 - i));
                        }
                    }
                }
            }
        }
        free(rb); free(f);
    }
    return failed;
}

int d2d_oracle_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
