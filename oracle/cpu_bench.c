/*
 * oracle/cpu_bench.c — TEST INFRASTRUCTURE ONLY: the host-CPU baseline of bench.py.
 *
 * Times the scalar CPU restatement of the receive chain (oracle/dab_rx.c) with no Python in
 * the loop: one independent ensemble per thread, one thread per core, every thread decoding the
 * same resident periodic Mode-I signal (as bench.py's GPU streams do) for a fixed wall time.
 * SURVEY.md §8(d) "CPU baseline plan (2)": cores stated, FIC-only and full-MSC legs.
 *
 *   cpu_bench <threads> <seconds> <n_sub 48-CU EEP 3-A sub-channels, 0 = FIC only> <snr_db>
 *
 * Prints one JSON object.  The reference's own CPU path (libdabsdr.so, closed binary) is not
 * run; this is `"kind": "port"` in bench.py's cpu_baseline.
 */
#define _GNU_SOURCE
#include "dab_spec.h"
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

typedef struct orx orx_t;
orx_t *orx_create(int fmt, int64_t ring_len, int ti_slots);
void orx_destroy(orx_t *s);
int orx_set_subch(orx_t *s, int n, const int32_t *cfg);
void orx_push(orx_t *s, const void *iq, int64_t n);
void orx_set_wr(orx_t *s, int64_t wr);
int orx_process(orx_t *s, int n_frames, void *sync, int8_t *fic_soft, int8_t *msc_soft, uint8_t *fib, uint8_t *fib_ok,
                uint8_t *msc, uint8_t *msc_valid);

#include "dab_tx.h"
typedef dab_tx_cfg_t tx_cfg_t;

enum { PERIOD = 12, STEP = 4 };

static double now_s(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

typedef struct {
    orx_t *rx;
    int msc_bytes;
    double seconds;
    pthread_barrier_t *bar;
    long frames, fib_bad;
    double t0, t1;
} job_t;

static void *worker(void *arg)
{
    job_t *j = (job_t *)arg;
    uint8_t fib_ok[STEP * 12];
    uint8_t *msc = j->msc_bytes ? (uint8_t *)malloc((size_t)STEP * 4 * (size_t)j->msc_bytes) : NULL;
    /* acquisition and time de-interleaver fill outside the timed region */
    for (int k = 0; k < 5; k++) orx_process(j->rx, STEP, NULL, NULL, NULL, NULL, fib_ok, msc, NULL);
    pthread_barrier_wait(j->bar);
    j->t0 = now_s();
    double t = j->t0;
    while (t - j->t0 < j->seconds) {
        if (orx_process(j->rx, STEP, NULL, NULL, NULL, NULL, fib_ok, msc, NULL) != STEP) { j->fib_bad += 12 * STEP; break; }
        for (int i = 0; i < STEP * 12; i++) j->fib_bad += !fib_ok[i];
        j->frames += STEP;
        t = now_s();
    }
    j->t1 = t;
    free(msc);
    return NULL;
}

int main(int argc, char **argv)
{
    const int threads = argc > 1 ? atoi(argv[1]) : 1;
    const double seconds = argc > 2 ? atof(argv[2]) : 5.0;
    const int nsub = argc > 3 ? atoi(argv[3]) : 18;
    const double snr = argc > 4 ? atof(argv[4]) : 20.0;
    if (threads < 1 || threads > 1024 || nsub < 0 || nsub > 18) { fprintf(stderr, "bad arguments\n"); return 2; }

    tx_cfg_t c;
    memset(&c, 0, sizeof c);
    c.seed = 99; c.eid = 0x1099; c.n_frames = PERIOD; c.n_subch = nsub; c.loop = 1; c.fmt = 0;
    c.snr_db = snr; c.cfo_hz = 1234.0; c.rms = 28.0; c.tii_main = -1;
    for (int i = 0; i < nsub; i++) { c.subch[i][0] = 48 * i; c.subch[i][1] = 0; c.subch[i][2] = 3; c.subch[i][3] = 64; }
    const int mb = dab_tx_msc_bytes_per_cif(&c);
    uint8_t *iq = (uint8_t *)malloc((size_t)PERIOD * DAB_TF * 2);
    uint8_t *fib = (uint8_t *)malloc((size_t)PERIOD * 12 * 32);
    uint8_t *msc = (uint8_t *)malloc((size_t)PERIOD * 4 * (size_t)(mb > 0 ? mb : 1));
    if (dab_tx_generate(&c, iq, fib, msc)) { fprintf(stderr, "tx failed\n"); return 1; }
    /* arbitrary start position inside the frame, like the GPU streams */
    const size_t shift = 2 * 54321, total = (size_t)PERIOD * DAB_TF * 2;
    uint8_t *rot = (uint8_t *)malloc(total);
    memcpy(rot + shift, iq, total - shift);
    memcpy(rot, iq + total - shift, shift);

    job_t *jobs = (job_t *)calloc((size_t)threads, sizeof *jobs);
    pthread_t *th = (pthread_t *)calloc((size_t)threads, sizeof *th);
    pthread_barrier_t bar;
    pthread_barrier_init(&bar, NULL, (unsigned)threads);
    int32_t cfg[18 * 4];
    for (int i = 0; i < nsub; i++) memcpy(cfg + 4 * i, c.subch[i], 16);
    for (int k = 0; k < threads; k++) {
        jobs[k].rx = orx_create(0, (int64_t)PERIOD * DAB_TF, 64);
        orx_set_subch(jobs[k].rx, nsub, cfg);
        orx_push(jobs[k].rx, rot, (int64_t)PERIOD * DAB_TF);
        orx_set_wr(jobs[k].rx, (int64_t)1 << 62);          /* resident periodic ring: never underruns */
        jobs[k].msc_bytes = mb; jobs[k].seconds = seconds; jobs[k].bar = &bar;
    }
    for (int k = 0; k < threads; k++) pthread_create(&th[k], NULL, worker, &jobs[k]);
    long frames = 0, bad = 0;
    double t0 = 1e300, t1 = 0;
    for (int k = 0; k < threads; k++) {
        pthread_join(th[k], NULL);
        frames += jobs[k].frames; bad += jobs[k].fib_bad;
        if (jobs[k].t0 < t0) t0 = jobs[k].t0;
        if (jobs[k].t1 > t1) t1 = jobs[k].t1;
        orx_destroy(jobs[k].rx);
    }
    const double dt = t1 - t0, x = (double)frames * 0.096 / dt;
    printf("{\"threads\": %d, \"n_sub\": %d, \"frames\": %ld, \"seconds\": %.3f, \"x_realtime\": %.3f, \"x_realtime_per_thread\": %.3f, "
           "\"fib_crc_bad\": %ld}\n", threads, nsub, frames, dt, x, x / threads, bad);
    free(rot); free(iq); free(fib); free(msc); free(jobs); free(th);
    return bad ? 1 : 0;
}
