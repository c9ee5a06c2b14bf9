/* How fast does the drop-in 24-function library (include/dabsdr_amd.h = the reference's lib/linux_x86_64/dabsdr.h) decode ONE
 * ensemble when the host's input callback does not pace it?  The reference's own worker does 188-195 x real time FIC-only and
 * 105-110 x with one 48-CU DAB+ service on one CPU thread.
 *
 *   legacy_rate <periodic.u8> <frames> [SId]
 *
 * <periodic.u8>: a sample-continuous periodic Mode-I recording (u8 IQ; bench.py / tests write it with the oracle's
 * transmitter, loop = 1).  The input callback hands it out as floats exactly as RawFileWorker + getSamples do
 * (reference: src/input/rawfileinput.cpp:690-693 float(u8 - 128); src/input/inputdevice.cpp:89-102 memcpy out of a float
 * FIFO), never blocking.  With an SId the service's primary audio component is selected the way radiocontrol.cpp does it
 * (service list -> dabsdrRequest_ServiceSelection) and the clock starts once access units arrive.
 * Frames are counted by the samples the library pulled, not by notifications.  One JSON line on stdout. */
#include "../include/dabsdr_amd.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

static float *sig;
static size_t n_sig, pos;
static volatile long long pulled;              /* complex samples handed out */
static volatile long n_periodic, fib_err, crc_ok, crc_err, n_au, au_bytes, au_concealed;
static volatile int level, have_list, selected, sel_status = -1;
static uint32_t want_sid;
static dabsdrHandle_t H;

static void input(float buf[], uint16_t n)
{
    size_t want = 2 * (size_t)n, done = 0;
    while (done < want) {
        size_t c = want - done < n_sig - pos ? want - done : n_sig - pos;
        memcpy(buf + done, sig + pos, c * sizeof(float));
        done += c; pos += c;
        if (pos == n_sig) pos = 0;
    }
    pulled += n;
}

static void ntf(dabsdrNotificationCBData_t *d, void *ctx)
{
    (void)ctx;
    if (d->nid == DABSDR_NID_PERIODIC && d->pData) {
        const dabsdrNtfPeriodic_t *p = d->pData;
        n_periodic++; fib_err += p->fibErrorCntr; crc_ok += p->mscCrcOkCntr; crc_err += p->mscCrcErrorCntr; level = p->syncLevel;
    } else if (d->nid == DABSDR_NID_SYNC_STATUS && d->pData) {
        level = ((const dabsdrNtfSyncStatus_t *)d->pData)->syncLevel;
    } else if (d->nid == DABSDR_NID_SERVICE_LIST && d->pData) {
        const dabsdrNtfServiceList_t *l = d->pData;
        for (int i = 0; i < l->numServices; i++) {
            dabsdrServiceListItem_t it;
            if (l->getServiceListItem(H, (uint8_t)i, &it) == 0 && it.sid == want_sid) have_list = 1;
        }
    } else if (d->nid == DABSDR_NID_SERVICE_SELECTION) {
        sel_status = d->status;
        if (d->status == DABSDR_NSTAT_SUCCESS) selected = 1;
    }
}

static void audio(dabsdrAudioCBData_t *a, void *ctx)
{
    (void)ctx;
    n_au++; au_bytes += a->auLen;
    if (a->header.raw & 0x80) au_concealed++;
}

static double now(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }
static void nap(long us) { struct timespec ts = {0, us * 1000}; nanosleep(&ts, NULL); }

int main(int argc, char **argv)
{
    if (argc < 3) { fprintf(stderr, "usage: %s <periodic.u8> <frames> [SId]\n", argv[0]); return 2; }
    FILE *f = fopen(argv[1], "rb");
    if (!f) return 2;
    fseek(f, 0, SEEK_END); long nb = ftell(f); fseek(f, 0, SEEK_SET);
    unsigned char *raw = malloc(nb);
    if (!raw || fread(raw, 1, nb, f) != (size_t)nb) return 2;
    fclose(f);
    n_sig = nb; sig = malloc(sizeof(float) * nb);
    for (long i = 0; i < nb; i++) sig[i] = (float)raw[i] - 128.0f;
    free(raw);
    const long frames = atol(argv[2]);
    want_sid = argc > 3 ? (uint32_t)strtoul(argv[3], NULL, 0) : 0;
    if (dabsdrInit(&H)) { fprintf(stderr, "dabsdrInit failed (no GPU?)\n"); return 1; }
    dabsdrRegisterInputFcn(H, input); dabsdrRegisterDummyInputFcn(H, input);
    dabsdrRegisterNotificationCb(H, ntf, NULL); dabsdrRegisterAudioCb(H, audio, NULL);
    dabsdr(H);
    dabsdrRequest_SetPeriodicNotify(H, 0, 0);            /* 2^0: every frame */
    dabsdrRequest_Tune(H, 225648);
    const double t_start = now();
    while (level != DABSDR_SYNC_LEVEL_FIC && now() - t_start < 30) nap(1000);
    if (level != DABSDR_SYNC_LEVEL_FIC) { printf("{\"error\": \"no FIC sync\"}\n"); return 1; }
    if (want_sid) {
        while (!have_list && now() - t_start < 30) { dabsdrRequest_GetServiceList(H); nap(5000); }
        while (!selected && now() - t_start < 30) { dabsdrRequest_ServiceSelection(H, want_sid, 0, DABSDR_ID_AUDIO_PRIMARY); nap(20000); }
        while (n_au < 30 && now() - t_start < 30) nap(1000);       /* time de-interleaver filled, super frame sync found */
        if (n_au < 30) { printf("{\"error\": \"no audio\", \"selection_status\": %d}\n", sel_status); return 1; }
    } else {
        while (n_periodic < 40 && now() - t_start < 30) nap(1000);
    }
    const long long s0 = pulled; const long e0 = fib_err, a0 = n_au, p0 = n_periodic, c0 = crc_ok, x0 = crc_err, k0 = au_concealed;
    const double t0 = now();
    while (pulled - s0 < (long long)frames * 196608) nap(500);
    const double dt = now() - t0;
    const double done = (double)(pulled - s0) / 196608.0;
    printf("{\"frames\": %.1f, \"seconds\": %.4f, \"x_realtime\": %.1f, \"ms_per_frame\": %.4f, \"sync_level\": %d, \"periodic_ntf\": %ld, "
           "\"fib_errors\": %ld, \"access_units\": %ld, \"au_crc_ok\": %ld, \"au_crc_err\": %ld, \"au_concealed\": %ld, \"service\": \"%s\"}\n",
           done, dt, done * 0.096 / dt, dt / done * 1e3, level, n_periodic - p0, fib_err - e0, n_au - a0, crc_ok - c0, crc_err - x0,
           au_concealed - k0, want_sid ? "the primary audio component of the given SId" : "none (FIC only)");
    dabsdrRequest_Exit(H); dabsdrDeinit(&H);
    return 0;
}
