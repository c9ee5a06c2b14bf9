/* how fast does the legacy 24-function path run one ensemble when the host's input callback does not pace it?
 * usage: legacy_rate <file.u8 raw IQ> <repeat>   (the file is a periodic signal, see legacy_rate.py)            */
#include "../../include/dabsdr_amd.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
static float *sig; static size_t n_sig, pos; static long n_periodic, fib_err; static int level;
static void input(float buf[], uint16_t n)
{
    size_t want = 2 * (size_t)n, done = 0;
    while (done < want) { size_t c = want - done < n_sig - pos ? want - done : n_sig - pos; memcpy(buf + done, sig + pos, c * sizeof(float)); done += c; pos += c; if (pos == n_sig) pos = 0; }
}
static void ntf(dabsdrNotificationCBData_t *d, void *ctx)
{
    (void)ctx;
    if (d->nid == DABSDR_NID_PERIODIC && d->pData) { const dabsdrNtfPeriodic_t *p = d->pData; n_periodic++; fib_err += p->fibErrorCntr; level = p->syncLevel; }
}
static double now(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }
int main(int argc, char **argv)
{
    FILE *f = fopen(argv[1], "rb"); if (!f) return 2;
    fseek(f, 0, SEEK_END); long nb = ftell(f); fseek(f, 0, SEEK_SET);
    unsigned char *raw = malloc(nb); if (fread(raw, 1, nb, f) != (size_t)nb) return 2;
    n_sig = nb; sig = malloc(sizeof(float) * nb);
    for (long i = 0; i < nb; i++) sig[i] = (float)raw[i] - 128.0f;
    const long frames = argc > 2 ? atol(argv[2]) : 2000;
    dabsdrHandle_t h;
    if (dabsdrInit(&h)) return 1;
    dabsdrRegisterInputFcn(h, input); dabsdrRegisterDummyInputFcn(h, input); dabsdrRegisterNotificationCb(h, ntf, NULL);
    dabsdr(h);
    dabsdrRequest_SetPeriodicNotify(h, 1, 0);            /* every frame */
    dabsdrRequest_Tune(h, 225648);
    while (n_periodic < 20) { struct timespec ts = {0, 1000000}; nanosleep(&ts, NULL); }
    const long p0 = n_periodic; const double t0 = now();
    while (n_periodic < p0 + frames) { struct timespec ts = {0, 1000000}; nanosleep(&ts, NULL); }
    const double dt = now() - t0; const long done = n_periodic - p0;
    printf("legacy path (C host): %ld frames in %.3f s = %.0f x real-time, %.3f ms per frame, sync level %d, fib errors %ld\n", done, dt, done * 0.096 / dt, dt / done * 1e3, level, fib_err);
    dabsdrRequest_Exit(h); dabsdrDeinit(&h);
    return 0;
}
