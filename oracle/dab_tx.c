/*
 * oracle/dab_tx.c — TEST INFRASTRUCTURE ONLY.
 *
 * Seeded synthetic DAB Mode-I transmitter (ETSI EN 300 401 framing, FIC and
 * MSC coding, time/frequency interleaving, pi/4-DQPSK OFDM) producing the
 * raw-file sample formats the reference's RawFileInput reads
 * (reference: src/input/rawfileinput.cpp:640-713 — u8 with value-128, or s16).
 * It is the source of every input vector in tests/, smoke() and bench.py.
 * The reference has no transmitter; conventions follow SURVEY.md Appendix B.
 */
#include "dab_spec.h"
#include "dab_tx.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>



/* ---- deterministic PRNG (splitmix64) ---- */
static uint64_t sm64(uint64_t *s)
{
    uint64_t z = (*s += 0x9E3779B97F4A7C15ULL);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}
static double urand(uint64_t *s) { return ((sm64(s) >> 11) + 0.5) * (1.0 / 9007199254740992.0); }
static void gauss2(uint64_t *s, double *a, double *b)
{
    double u = urand(s), v = urand(s), r = sqrt(-2.0 * log(u));
    *a = r * cos(2.0 * M_PI * v);
    *b = r * sin(2.0 * M_PI * v);
}

/* ---- plain double radix-2 inverse FFT (not the receiver's FFT) ---- */
static void ifft2048(double *re, double *im, const double *tw /* cos,sin of 2*pi*k/2048 */)
{
    const int N = DAB_TU;
    for (int i = 1, j = 0; i < N; i++) {
        int bit = N >> 1;
        for (; j & bit; bit >>= 1) j ^= bit;
        j ^= bit;
        if (i < j) { double t = re[i]; re[i] = re[j]; re[j] = t; t = im[i]; im[i] = im[j]; im[j] = t; }
    }
    for (int len = 2; len <= N; len <<= 1) {
        const int st = N / len;
        for (int i = 0; i < N; i += len)
            for (int k = 0; k < len / 2; k++) {
                double wr = tw[2 * k * st], wi = tw[2 * k * st + 1];
                int a = i + k, b = i + k + len / 2;
                double tr = re[b] * wr - im[b] * wi, ti = re[b] * wi + im[b] * wr;
                re[b] = re[a] - tr; im[b] = im[a] - ti;
                re[a] += tr; im[a] += ti;
            }
    }
}

/* ---- FIG / FIB construction (EN 300 401 §5.2, §6, §8) ---- */
typedef struct { uint8_t b[30]; int len; } fig_t;

static int build_figs(const dab_tx_cfg_t *c, const dab_profile_t *prof, fig_t *figs)
{
    int n = 0;
    /* FIG 0/1 sub-channel organisation: long form (4 bytes) for EEP, short form (3 bytes,
     * table index) for UEP; as many entries per FIG as fit into 29 data bytes */
    for (int s = 0; s < c->n_subch;) {
        fig_t *f = &figs[n++];
        f->b[1] = 0x01; f->len = 2;
        while (s < c->n_subch && f->len + (c->subch[s][1] == 2 ? 3 : 4) <= 30) {
            int addr = c->subch[s][0], size = prof[s].n_cu;
            f->b[f->len++] = (uint8_t)((s << 2) | (addr >> 8));
            f->b[f->len++] = (uint8_t)(addr & 0xFF);
            if (c->subch[s][1] == 2) f->b[f->len++] = (uint8_t)(c->subch[s][2] & 0x3F);
            else {
                f->b[f->len++] = (uint8_t)(0x80 | (c->subch[s][1] << 4) | ((c->subch[s][2] - 1) << 2) | (size >> 8));
                f->b[f->len++] = (uint8_t)(size & 0xFF);
            }
            ++s;
        }
        f->b[0] = (uint8_t)(f->len - 1);
    }
    /* FIG 0/2 basic service organisation: one audio service per sub-channel; a packet-mode sub-channel
     * (cfg.packet_sub) is instead the second component of the first service */
    const int pk = c->packet_sub > 0 && c->packet_sub < c->n_subch ? c->packet_sub : 0;
    for (int s = 0; s < c->n_subch;) {
        fig_t *f = &figs[n++];
        f->b[1] = 0x02; f->len = 2;
        while (s < c->n_subch && f->len + 7 <= 30) {
            if (s == pk && pk) { ++s; continue; }
            int sid = 0x1A01 + s, two = (s == 0 && pk);
            f->b[f->len++] = (uint8_t)(sid >> 8); f->b[f->len++] = (uint8_t)sid;
            f->b[f->len++] = (uint8_t)(two ? 0x02 : 0x01);   /* number of components */
            f->b[f->len++] = (uint8_t)(c->subch[s][1] == 2 ? 0x00 : 0x3F);   /* TMId 0; ASCTy 0 (MP2) on UEP, 63 (DAB+) on EEP */
            f->b[f->len++] = (uint8_t)((s << 2) | 0x02);     /* SubChId, primary   */
            if (two) {
                int scid = 0x200 + pk;
                f->b[f->len++] = (uint8_t)(0xC0 | (scid >> 6));          /* TMId 3, SCId */
                f->b[f->len++] = (uint8_t)((scid & 0x3F) << 2);          /* secondary, no CA */
            }
            ++s;
        }
        f->b[0] = (uint8_t)(f->len - 1);
    }
    if (pk) {
        int scid = 0x200 + pk, addr = 0x155;
        { fig_t *f = &figs[n++];                              /* FIG 0/3: SCId, DG used, DSCTy 60 (MOT), SubChId, packet address */
          f->b[0] = 0x06; f->b[1] = 0x03; f->b[2] = (uint8_t)(scid >> 4); f->b[3] = (uint8_t)((scid & 0xF) << 4);
          f->b[4] = 60; f->b[5] = (uint8_t)((pk << 2) | (addr >> 8)); f->b[6] = (uint8_t)addr; f->len = 7; }
        { fig_t *f = &figs[n++];                              /* FIG 0/8: SCIdS 1 <-> SCId (long form) */
          f->b[0] = 0x06; f->b[1] = 0x08; f->b[2] = 0x1A; f->b[3] = 0x01; f->b[4] = 0x01;
          f->b[5] = (uint8_t)(0x80 | (scid >> 8)); f->b[6] = (uint8_t)scid; f->len = 7; }
        { fig_t *f = &figs[n++];                              /* FIG 0/13: SCIdS 1, one application: SPI (type 7), no data */
          f->b[0] = 0x06; f->b[1] = 0x0D; f->b[2] = 0x1A; f->b[3] = 0x01; f->b[4] = 0x11; f->b[5] = 0x00; f->b[6] = 0xE0; f->len = 7; }
    }
    { /* FIG 0/9: LTO +1 h, ECC 0xE2, international table 1 */
        fig_t *f = &figs[n++];
        f->b[0] = 0x04; f->b[1] = 0x09; f->b[2] = 0x02; f->b[3] = 0xE2; f->b[4] = 0x01; f->len = 5;
    }
    { /* FIG 0/10 date and time, long form: MJD 60587 (2024-10-04), 12:34:56.789 UTC */
        fig_t *f = &figs[n++];
        const unsigned mjd = 60587, hh = 12, mm = 34, ss = 56, ms = 789;
        const unsigned w = (mjd << 14) | (0u << 13) | (1u << 12) | (1u << 11) | (hh << 6) | mm;
        f->b[0] = 0x07; f->b[1] = 0x0A;
        f->b[2] = (uint8_t)(w >> 24); f->b[3] = (uint8_t)(w >> 16); f->b[4] = (uint8_t)(w >> 8); f->b[5] = (uint8_t)w;
        f->b[6] = (uint8_t)((ss << 2) | (ms >> 8)); f->b[7] = (uint8_t)ms; f->len = 8;
    }
    { /* FIG 1/0 ensemble label */
        fig_t *f = &figs[n++];
        char lab[17];
        memcpy(lab, "GRAFT ENS       ", 16);
        lab[10] = "0123456789ABCDEF"[(c->eid >> 12) & 15]; lab[11] = "0123456789ABCDEF"[(c->eid >> 8) & 15];
        lab[12] = "0123456789ABCDEF"[(c->eid >> 4) & 15];  lab[13] = "0123456789ABCDEF"[c->eid & 15];
        f->b[0] = 0x20 | 21; f->b[1] = 0x00; f->b[2] = (uint8_t)(c->eid >> 8); f->b[3] = (uint8_t)c->eid;
        memcpy(f->b + 4, lab, 16); f->b[20] = 0xFF; f->b[21] = 0x00; f->len = 22;
    }
    for (int s = 0; s < c->n_subch; s++) { /* FIG 1/1 programme service labels */
        if (pk && s == pk) continue;
        fig_t *f = &figs[n++];
        int sid = 0x1A01 + s;
        char lab[17];
        memcpy(lab, "SERVICE 00      ", 16);
        lab[8] = (char)('0' + s / 10); lab[9] = (char)('0' + s % 10);
        f->b[0] = 0x20 | 21; f->b[1] = 0x01; f->b[2] = (uint8_t)(sid >> 8); f->b[3] = (uint8_t)sid;
        memcpy(f->b + 4, lab, 16); f->b[20] = 0xFF; f->b[21] = 0x00; f->len = 22;
    }
    if (c->extra_figs && c->n_subch > 0) {
        const int sid = 0x1A01;
        static const uint8_t lang[] = {0x03, 0x05, 0x00, 0x09};                                   /* 0/5: SubCh 0, English          */
        static const uint8_t glob[] = {0x05, 0x08, 0x1A, 0x01, 0x00, 0x00};                       /* 0/8: SCIdS 0 <-> SubCh 0       */
        static const uint8_t apps[] = {0x08, 0x0D, 0x1A, 0x01, 0x01, 0x00, 0x42, 0x0C, 0x3C};     /* 0/13: SLS (type 2) over X-PAD  */
        static const uint8_t pty[]  = {0x05, 0x11, 0x1A, 0x01, 0x00, 0x0A};                       /* 0/17: PTy 10 (pop music)       */
        static const uint8_t asu[]  = {0x07, 0x12, 0x1A, 0x01, 0x00, 0x02, 0x01, 0x07};           /* 0/18: traffic, cluster 7       */
        static const uint8_t asw[]  = {0x05, 0x13, 0x07, 0x00, 0x02, 0x80};                       /* 0/19: cluster 7 on SubCh 0     */
        const uint8_t *src[] = {lang, glob, apps, pty, asu, asw};
        const int len[] = {sizeof lang, sizeof glob, sizeof apps, sizeof pty, sizeof asu, sizeof asw};
        (void)sid;
        for (int k = 0; k < 6; k++) { fig_t *f = &figs[n++]; memcpy(f->b, src[k], (size_t)len[k]); f->len = len[k]; }
    }
    return n;
}

/* one FIB: 30 bytes of FIGs (+0xFF end marker, zero padding) + CRC */
static void build_fib(uint8_t *fib, int with_fig00, int eid, int cif, const fig_t *figs, int nfig, int *next)
{
    int pos = 0;
    memset(fib, 0, DAB_FIB_BYTES);
    if (with_fig00) {
        fib[0] = 0x05; fib[1] = 0x00; fib[2] = (uint8_t)(eid >> 8); fib[3] = (uint8_t)eid;
        fib[4] = (uint8_t)((cif / 250) % 20); fib[5] = (uint8_t)(cif % 250);
        pos = 6;
    }
    for (int tries = 0; tries < nfig; tries++) {
        const fig_t *f = &figs[*next % nfig];
        if (pos + f->len > 30) break;
        memcpy(fib + pos, f->b, (size_t)f->len);
        pos += f->len;
        (*next)++;
    }
    if (pos < 30) fib[pos] = 0xFF;
    uint16_t crc = dab_crc16(fib, 30);
    fib[30] = (uint8_t)(crc >> 8); fib[31] = (uint8_t)crc;
}

static void bytes_to_bits(const uint8_t *by, int nbytes, uint8_t *bits)
{
    for (int i = 0; i < nbytes; i++)
        for (int b = 0; b < 8; b++) bits[8 * i + b] = (by[i] >> (7 - b)) & 1;
}

/* energy dispersal + mother code + puncturing of one codeword */
static void encode_cw(const dab_profile_t *p, const uint8_t *info_bytes, const uint8_t *prbs, uint8_t *coded,
                      uint8_t *tmp_bits, uint8_t *tmp_mother)
{
    bytes_to_bits(info_bytes, p->n_in / 8, tmp_bits);
    for (int i = 0; i < p->n_in; i++) tmp_bits[i] ^= prbs[i];
    dab_conv_encode(tmp_bits, p->n_in, tmp_mother);
    dab_puncture(p, tmp_mother, coded);
}

/* sizes the caller must provide */
int dab_tx_msc_bytes_per_cif(const dab_tx_cfg_t *c)
{
    int tot = 0;
    dab_profile_t p;
    for (int s = 0; s < c->n_subch; s++) {
        if (dab_profile_any(c->subch[s][1], c->subch[s][2], c->subch[s][3], &p)) return -1;
        tot += p.n_in / 8;
    }
    return tot;
}

/*
 * Generate n_frames transmission frames.
 *   iq        : (delay + n_frames*196608) complex samples, interleaved I,Q (u8 or s16)
 *   fib_out   : n_frames*12*32 bytes, the transmitted FIBs (incl. CRC)
 *   msc_out   : n_frames*4 logical frames x msc_bytes_per_cif payload bytes
 *               (sub-channels concatenated in cfg order), before dispersal
 * Returns 0 on success.
 */
int dab_tx_generate(const dab_tx_cfg_t *c, void *iq, uint8_t *fib_out, uint8_t *msc_out)
{
    const int NF = c->n_frames, NC = NF * DAB_CIFS;
    dab_profile_t prof[64], ficp;
    if (c->n_subch < 0 || c->n_subch > 64 || NF < 1) return -1;
    int msc_bytes = 0, off_bytes[64];
    for (int s = 0; s < c->n_subch; s++) {
        if (dab_profile_any(c->subch[s][1], c->subch[s][2], c->subch[s][3], &prof[s])) return -2;
        if (c->subch[s][0] < 0 || c->subch[s][0] + prof[s].n_cu > DAB_NCU) return -3;
        off_bytes[s] = msc_bytes;
        msc_bytes += prof[s].n_in / 8;
    }
    if (c->loop && (NC % DAB_TI_DEPTH)) return -4;
    dab_profile_fic(&ficp);

    uint64_t rng = c->seed * 0x9E3779B97F4A7C15ULL + 0x1234567ULL;
    uint64_t nrng = c->seed * 0xD1342543DE82EF95ULL + 0x7654321ULL;
    uint8_t *prbs = (uint8_t *)malloc(DAB_CIF_BITS);
    uint8_t *tmp_bits = (uint8_t *)malloc(DAB_CIF_BITS + 64);
    uint8_t *tmp_mother = (uint8_t *)malloc(4 * (DAB_CIF_BITS + 64));
    uint8_t *coded = (uint8_t *)calloc((size_t)NC, DAB_CIF_BITS);      /* logical frames, pre-interleaver */
    uint8_t *ficbits = (uint8_t *)malloc((size_t)NF * DAB_FIC_BITS);
    fig_t *figs = (fig_t *)calloc(256, sizeof(fig_t));
    double *re = (double *)malloc(sizeof(double) * DAB_TU), *im = (double *)malloc(sizeof(double) * DAB_TU);
    int16_t kofn[DAB_K];
    int8_t prsq[DAB_TU];
    uint8_t ph8[DAB_TU];
    uint8_t symbits[DAB_SYM_BITS];
    double *tw = (double *)malloc(sizeof(double) * 2 * DAB_TU);
    for (int k = 0; k < DAB_TU; k++) { tw[2 * k] = cos(2.0 * M_PI * k / DAB_TU); tw[2 * k + 1] = sin(2.0 * M_PI * k / DAB_TU); }
    dab_prbs(prbs, DAB_CIF_BITS);
    dab_freq_interleaver(kofn);
    dab_prs_quadrants(prsq);
    int nfig = build_figs(c, prof, figs), nextfig = 0;
    /* optional second ensemble (EId + 1) from frame eid2_from on: same multiplex, its own FIG 0/0 and labels */
    fig_t *figs2 = (fig_t *)calloc(256, sizeof(fig_t));
    int nfig2 = 0, nextfig2 = 0;
    if (c->eid2_from > 0) { dab_tx_cfg_t c2 = *c; c2.eid = c->eid + 1; nfig2 = build_figs(&c2, prof, figs2); }

    /* ---- FIC: 4 codewords of 3 FIBs per frame ---- */
    for (int f = 0; f < NF; f++)
        for (int cw = 0; cw < DAB_FIC_CW; cw++) {
            uint8_t *fibs = fib_out + ((size_t)f * 12 + 3 * cw) * DAB_FIB_BYTES;
            const int second = c->eid2_from > 0 && f >= c->eid2_from;
            for (int j = 0; j < 3; j++) {
                if (second) build_fib(fibs + j * DAB_FIB_BYTES, j == 0, c->eid + 1, f * 4 + cw, figs2, nfig2, &nextfig2);
                else build_fib(fibs + j * DAB_FIB_BYTES, j == 0, c->eid, f * 4 + cw, figs, nfig, &nextfig);
            }
            encode_cw(&ficp, fibs, prbs, ficbits + (size_t)f * DAB_FIC_BITS + cw * DAB_FIC_CW_BITS, tmp_bits, tmp_mother);
        }
    /* ---- MSC logical frames ---- */
    for (int r = 0; r < NC; r++)
        for (int s = 0; s < c->n_subch; s++) {
            uint8_t *pay = msc_out + (size_t)r * msc_bytes + off_bytes[s];
            int nb = prof[s].n_in / 8;
            for (int i = 0; i < nb && !c->payload_given; i += 8) {
                uint64_t x = sm64(&rng);
                for (int j = 0; j < 8 && i + j < nb; j++) pay[i + j] = (uint8_t)(x >> (8 * j));
            }
            encode_cw(&prof[s], pay, prbs, coded + (size_t)r * DAB_CIF_BITS + c->subch[s][0] * DAB_CU_BITS, tmp_bits, tmp_mother);
        }

    /* ---- modulation: the clean complex baseband signal of the whole recording ---- */
    const double amp = c->rms * DAB_TU / sqrt((double)DAB_K);
    const double nsig = (c->snr_db >= 100.0) ? 0.0 : c->rms * pow(10.0, -c->snr_db / 20.0) / sqrt(2.0);
    const double lim_lo = c->fmt ? -32768.0 : 0.0, lim_hi = c->fmt ? 32767.0 : 255.0, bias = c->fmt ? 0.0 : 128.0;
    const size_t total = (size_t)c->delay + (size_t)NF * DAB_TF;
    if (c->loop && c->sco_ppm != 0.0) return -5;
    double *sig = (double *)calloc(2 * total, sizeof(double));
    for (int f = 0; f < NF; f++) {
        double *frame = sig + 2 * ((size_t)c->delay + (size_t)f * DAB_TF);
        if (c->tii_main >= 0) {
            /* TII: carrier pairs k, k+1 with k = base + 2c + 48b carry the PRS phase of carrier k */
            static const int base[4] = {-768, -384, 1, 385};
            int word = -1, cnt = 0;
            for (int w = 0; w < 256 && word < 0; w++)
                if (__builtin_popcount(w) == 4 && cnt++ == c->tii_main) word = w;
            for (int b = 0; b < DAB_TU; b++) re[b] = im[b] = 0.0;
            for (int blk = 0; blk < 4 && word >= 0; blk++)
                for (int b = 0; b < 8; b++) {
                    if (!(word & (0x80 >> b))) continue;
                    int k = base[blk] + 2 * c->tii_sub + 48 * b, q = prsq[k & 2047];
                    static const double cq[4] = {1, 0, -1, 0};
                    for (int d = 0; d < 2; d++) { re[(k + d) & 2047] = cq[q & 3]; im[(k + d) & 2047] = cq[(q + 3) & 3]; }
                }
            ifft2048(re, im, tw);
            for (int n = 0; n < DAB_TNULL; n++) {
                int src = (n + DAB_TU - (DAB_TNULL - DAB_TU)) & (DAB_TU - 1);
                frame[2 * n] = re[src] * amp / DAB_TU; frame[2 * n + 1] = im[src] * amp / DAB_TU;
            }
        }
        for (int b = 0; b < DAB_TU; b++) ph8[b] = (uint8_t)(prsq[b] < 0 ? 0 : 2 * prsq[b]);
        for (int l = 0; l < DAB_NSYM; l++) {
            if (l > 0) {
                if (l <= DAB_FIC_SYMS) memcpy(symbits, ficbits + (size_t)f * DAB_FIC_BITS + (l - 1) * DAB_SYM_BITS, DAB_SYM_BITS);
                else {
                    int t = f * DAB_CIFS + (l - 4) / DAB_CIF_SYMS;           /* transmitted CIF index */
                    int base = ((l - 4) % DAB_CIF_SYMS) * DAB_SYM_BITS;
                    for (int i = 0; i < DAB_SYM_BITS; i++) {
                        int r = t - dab_ti_delay(base + i);
                        if (r < 0) r = c->loop ? r + NC : -1;
                        symbits[i] = (r < 0) ? 0 : coded[(size_t)r * DAB_CIF_BITS + base + i];
                    }
                }
                for (int n = 0; n < DAB_K; n++) {
                    int b = kofn[n] & 2047;
                    int y = symbits[n] ? (symbits[n + DAB_K] ? 5 : 3) : (symbits[n + DAB_K] ? 7 : 1);
                    ph8[b] = (uint8_t)((ph8[b] + y) & 7);
                }
            }
            for (int b = 0; b < DAB_TU; b++) {
                if (prsq[b] < 0) { re[b] = im[b] = 0.0; continue; }
                static const double c8[8] = {1, M_SQRT1_2, 0, -M_SQRT1_2, -1, -M_SQRT1_2, 0, M_SQRT1_2};
                re[b] = c8[ph8[b]]; im[b] = c8[(ph8[b] + 6) & 7];
            }
            ifft2048(re, im, tw);
            double *o = frame + 2 * ((size_t)DAB_TNULL + (size_t)l * DAB_TS);
            for (int n = 0; n < DAB_TS; n++) {
                int src = (n + DAB_TU - DAB_TG) & (DAB_TU - 1);
                o[2 * n] = re[src] * amp / DAB_TU; o[2 * n + 1] = im[src] * amp / DAB_TU;
            }
        }
    }
    /* ---- channel: second path (echo), then the receiver's sampling clock ---- */
    if (c->echo_db > 0.0 && c->echo_delay > 0) {
        const double a = pow(10.0, -c->echo_db / 20.0), er = a * cos(c->echo_phase), ei = a * sin(c->echo_phase);
        const size_t D = (size_t)c->echo_delay, span = (size_t)NF * DAB_TF;
        double *y = (double *)malloc(sizeof(double) * 2 * total);
        memcpy(y, sig, sizeof(double) * 2 * total);
        for (size_t n = (size_t)c->delay; n < total; n++) {
            size_t m;
            if (n - (size_t)c->delay >= D) m = n - D;
            else if (c->loop) m = n + span - D;               /* periodic signal: the echo wraps */
            else continue;
            y[2 * n] += sig[2 * m] * er - sig[2 * m + 1] * ei;
            y[2 * n + 1] += sig[2 * m] * ei + sig[2 * m + 1] * er;
        }
        free(sig); sig = y;
    }
    if (c->sco_ppm != 0.0) {
        /* recording sample n = signal at time n (1 + sco) / Fs: Kaiser-windowed sinc, 2 x 24 taps
         * (signal bandwidth 0.75 Nyquist; interpolation error below -70 dB) */
        enum { HT = 24, NPH = 1024 };
        const double eps = c->sco_ppm * 1e-6, beta = 9.0;
        double *y = (double *)calloc(2 * total, sizeof(double));
        double *tab = (double *)malloc(sizeof(double) * (NPH + 1) * 2 * HT);   /* tab[p][j]: tap k = j - HT + 1 at fraction p / NPH */
        double i0b = 1.0;
        { double t = 1.0; for (int k = 1; k < 60; k++) { t *= (beta / 2.0 / k) * (beta / 2.0 / k); i0b += t; } }
        for (int p = 0; p <= NPH; p++)
            for (int j = 0; j < 2 * HT; j++) {
                const double x = (double)(j - HT + 1) - (double)p / NPH, u = x / (double)HT;
                double w = 0.0;
                if (u > -1.0 && u < 1.0) {
                    const double z = beta * sqrt(1.0 - u * u);
                    double t = 1.0, sum = 1.0;
                    for (int q = 1; q < 60; q++) { t *= (z / 2.0 / q) * (z / 2.0 / q); sum += t; }
                    w = sum / i0b;
                }
                tab[p * 2 * HT + j] = (fabs(x) < 1e-12 ? 1.0 : sin(M_PI * x) / (M_PI * x)) * w;
            }
        for (size_t n = 0; n < total; n++) {
            const double pos = (double)n * (1.0 + eps);
            const long i0 = (long)floor(pos);
            const double fr = (pos - (double)i0) * NPH;
            const int p0 = (int)fr;
            const double a1 = fr - p0, a0 = 1.0 - a1;
            const double *t0 = tab + (size_t)p0 * 2 * HT, *t1 = t0 + 2 * HT;
            double ar = 0.0, ai = 0.0;
            for (int j = 0; j < 2 * HT; j++) {
                const long m = i0 + j - HT + 1;
                if (m < 0 || (size_t)m >= total) continue;
                const double h = a0 * t0[j] + a1 * t1[j];
                ar += sig[2 * m] * h; ai += sig[2 * m + 1] * h;
            }
            y[2 * n] = ar; y[2 * n + 1] = ai;
        }
        free(tab);
        free(sig); sig = y;
    }
    /* ---- carrier offset, noise, DC offset, quantisation ---- */
    for (size_t nabs = 0; nabs < total; nabs++) {
        double xr = sig[2 * nabs], xi = sig[2 * nabs + 1];
        if (c->cfo_hz != 0.0) {
            double a = 2.0 * M_PI * fmod(c->cfo_hz * (double)nabs / DAB_FS, 1.0);
            double cr = cos(a), ci = sin(a), t = xr * cr - xi * ci;
            xi = xr * ci + xi * cr; xr = t;
        }
        if (nsig > 0.0) { double g1, g2; gauss2(&nrng, &g1, &g2); xr += nsig * g1; xi += nsig * g2; }
        xr += c->dc_i; xi += c->dc_q;
        double qr = floor(xr + 0.5) + bias, qi = floor(xi + 0.5) + bias;
        qr = qr < lim_lo ? lim_lo : (qr > lim_hi ? lim_hi : qr);
        qi = qi < lim_lo ? lim_lo : (qi > lim_hi ? lim_hi : qi);
        if (c->fmt) { ((int16_t *)iq)[2 * nabs] = (int16_t)qr; ((int16_t *)iq)[2 * nabs + 1] = (int16_t)qi; }
        else { ((uint8_t *)iq)[2 * nabs] = (uint8_t)qr; ((uint8_t *)iq)[2 * nabs + 1] = (uint8_t)qi; }
    }
    free(sig); free(figs2);
    free(tw); free(re); free(im); free(figs); free(ficbits); free(coded);
    free(tmp_mother); free(tmp_bits); free(prbs);
    return 0;
}
