/*
 * decode_rawfiles.c — the batch C ABI from plain C: decode several AbracaDABra raw recordings (.raw / .uff, u8 or s16 IQ at
 * 2.048 Msps — what src/input/rawfileinput.cpp reads and src/input/inputdevicerecorder.cpp writes) side by side on one GPU
 * and report what the FIC says.  All files must share one sample format.
 *
 *   gcc -std=c11 -O2 -I include -o decode_rawfiles examples/decode_rawfiles.c -L abracadabra_amd -l:libdabsdr_amd.so -Wl,-rpath,$PWD/abracadabra_amd
 *   ./decode_rawfiles a.raw b.raw ...
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "dabx.h"

#define FRAMES_PER_STEP 4

int main(int argc, char **argv)
{
    if (argc < 2) { fprintf(stderr, "usage: %s file.raw [more files]\n", argv[0]); return 2; }
    const int S = argc - 1;
    FILE **f = calloc((size_t)S, sizeof *f);
    int fmt = -1;
    for (int s = 0; s < S; s++) {
        f[s] = fopen(argv[s + 1], "rb");
        if (!f[s]) { perror(argv[s + 1]); return 1; }
        uint8_t head[2048];
        const int n = (int)fread(head, 1, sizeof head, f[s]);
        dabx_rawfile_info_t info;
        memset(&info, 0, sizeof info);
        int this_fmt = DABX_FMT_U8;
        long skip = 0;
        if (dabx_rawfile_probe(head, n, &info) == DABX_OK && info.has_header) { this_fmt = info.fmt; skip = (long)info.data_offset; }
        if (fmt >= 0 && this_fmt != fmt) { fprintf(stderr, "%s: sample format differs from the first file\n", argv[s + 1]); return 1; }
        fmt = this_fmt;
        fseek(f[s], skip, SEEK_SET);
    }
    const int bps = fmt == DABX_FMT_U8 ? 2 : 4;
    dabx_config_t cfg = {S, fmt, (int64_t)(2 * FRAMES_PER_STEP + 4) * DABX_FRAME_SAMPLES, FRAMES_PER_STEP, 0};
    dabx_ctx *ctx = NULL;
    int rc = dabx_create(&cfg, &ctx);
    if (rc) { fprintf(stderr, "dabx_create: %s\n", dabx_strerror(rc)); return 1; }
    /* page-locked staging: one slice per file, filled by fread, pushed as one strided copy */
    const int64_t chunk = (int64_t)FRAMES_PER_STEP * DABX_FRAME_SAMPLES;
    const size_t stride = (size_t)(chunk + DABX_FRAME_SAMPLES + 4096) * (size_t)bps;
    uint8_t *stage = dabx_alloc_pinned(stride * (size_t)S);
    if (!stage) { fprintf(stderr, "no pinned memory\n"); return 1; }
    int64_t want = chunk + DABX_FRAME_SAMPLES + 4096;      /* acquisition needs one frame more */
    long steps = 0;
    long *fib_good = calloc((size_t)S, sizeof *fib_good), *fib_bad = calloc((size_t)S, sizeof *fib_bad);
    int *eid = calloc((size_t)S, sizeof *eid);
    for (int s = 0; s < S; s++) eid[s] = -1;
    for (;;) {
        int64_t got = want;
        for (int s = 0; s < S; s++) {
            const int64_t n = (int64_t)fread(stage + (size_t)s * stride, (size_t)bps, (size_t)want, f[s]);
            if (n < got) got = n;
        }
        if (got <= 0) break;
        rc = dabx_push_all(ctx, stage, stride, got, DABX_SRC_PINNED);
        if (rc) { fprintf(stderr, "dabx_push_all: %s\n", dabx_strerror(rc)); break; }
        const int nf = dabx_frames_available(ctx);
        if (nf > 0) {
            rc = dabx_process(ctx, nf);                   /* also waits for the copy above: the staging buffer is free again */
            if (rc) { fprintf(stderr, "dabx_process: %s\n", dabx_strerror(rc)); break; }
            ++steps;
            for (int s = 0; s < S; s++) {                 /* what the FIC said in this step */
                static uint8_t fib[FRAMES_PER_STEP * DABX_FIBS_PER_FRAME * DABX_FIB_BYTES], ok[FRAMES_PER_STEP * DABX_FIBS_PER_FRAME];
                if (dabx_get_fib(ctx, s, fib, ok)) continue;
                for (int k = 0; k < nf * DABX_FIBS_PER_FRAME; k++) {
                    if (!ok[k]) { ++fib_bad[s]; continue; }
                    ++fib_good[s];
                    const uint8_t *b = fib + k * DABX_FIB_BYTES;
                    if (eid[s] < 0 && b[0] == 0x05 && (b[1] & 0x1F) == 0) eid[s] = (b[2] << 8) | b[3];    /* FIG 0/0: ensemble identifier */
                }
            }
        } else if (got < want) break;
        want = chunk;
    }
    int64_t ok = 0, bad = 0;
    dabx_get_fib_counts(ctx, &ok, &bad);
    printf("%d file(s), %ld step(s); last step: %lld FIBs with a good CRC, %lld bad\n", S, steps, (long long)ok, (long long)bad);
    for (int s = 0; s < S; s++) {
        dabx_stream_state_t st;
        dabx_get_state(ctx, s, &st);
        printf("  %s: %s, carrier offset %.1f Hz, %lld CIFs, FIBs %ld good %ld bad, EId %04X\n", argv[s + 1], st.locked ? "locked" : "no DAB signal",
               (double)st.inc * 2048000.0 / 4294967296.0, (long long)st.cif, fib_good[s], fib_bad[s], eid[s] & 0xFFFF);
        fclose(f[s]);
    }
    dabx_free_pinned(stage);
    dabx_destroy(ctx);
    free(f); free(fib_good); free(fib_bad); free(eid);
    return 0;
}
