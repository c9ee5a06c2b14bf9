/*
 * oracle/dab_plus.c — TEST INFRASTRUCTURE ONLY.
 *
 * DAB+ audio super frame ENCODER (ETSI TS 102 563 §5, §6) for the synthetic transmitter:
 * fire code, access-unit table, AU CRCs and RS(120,110) parity.  The product's decoder
 * (abracadabra_amd/csrc/superframe.hpp) is checked against what this builds.  The reference
 * decodes super frames inside its closed binary (exports init_rs_char/decode_rs_char,
 * SURVEY.md §1); nothing here comes from reference source.
 */
#include "dab_spec.h"
#include <stdlib.h>
#include <string.h>

static uint8_t gexp[512], glog[256], gpoly[11];
static int ginit;

static void gf_init(void)
{
    if (ginit) return;
    unsigned x = 1;
    for (int i = 0; i < 255; i++) { gexp[i] = (uint8_t)x; glog[x] = (uint8_t)i; x <<= 1; if (x & 0x100) x ^= 0x11D; }
    for (int i = 255; i < 512; i++) gexp[i] = gexp[i - 255];
    /* g(x) = prod_{i=0..9} (x + alpha^i), gpoly[k] = coefficient of x^k */
    memset(gpoly, 0, sizeof gpoly);
    gpoly[0] = 1;
    for (int i = 0; i < 10; i++) {
        for (int k = i + 1; k > 0; k--) {
            uint8_t t = gpoly[k] ? gexp[glog[gpoly[k]] + i] : 0;   /* gpoly[k] * alpha^i */
            gpoly[k] = (uint8_t)(gpoly[k - 1] ^ t);
        }
        gpoly[0] = gpoly[0] ? gexp[glog[gpoly[0]] + i] : 0;
    }
    ginit = 1;
}

/* systematic RS(120,110): parity[0..9] = (msg(x) x^10) mod g(x), highest power first */
void dab_rs_encode_120_110(const uint8_t *msg, uint8_t *parity)
{
    gf_init();
    uint8_t r[10];
    memset(r, 0, sizeof r);
    for (int i = 0; i < 110; i++) {
        uint8_t fb = (uint8_t)(msg[i] ^ r[0]);
        for (int k = 0; k < 9; k++) r[k] = (uint8_t)(r[k + 1] ^ (fb ? gexp[glog[fb] + glog[gpoly[9 - k]]] : 0));
        r[9] = fb ? gexp[glog[fb] + glog[gpoly[0]]] : 0;
    }
    memcpy(parity, r, 10);
}

uint16_t dab_firecode(const uint8_t *d, int n)
{
    unsigned c = 0;
    for (int i = 0; i < n; i++) {
        c ^= (unsigned)d[i] << 8;
        for (int b = 0; b < 8; b++) c = (c & 0x8000) ? ((c << 1) ^ 0x782F) & 0xFFFF : (c << 1) & 0xFFFF;
    }
    return (uint16_t)c;
}

/*
 * Build one super frame of 120*s bytes (s = kbps/8) from num_aus access units.
 *   au[i], au_len[i]: AU payloads WITHOUT CRC; their lengths + 2 must exactly fill
 *   110*s - first_au_start bytes.  Returns 0 on success.
 */
int dab_superframe_build(int s, int dac_rate, int sbr, int ch_mode, int ps, int surr, const uint8_t *const *au,
                         const int *au_len, uint8_t *out)
{
    const int num_aus = dac_rate ? (sbr ? 3 : 6) : (sbr ? 2 : 4);
    const int first = dac_rate ? (sbr ? 6 : 11) : (sbr ? 5 : 8);
    int start[7], pos = first;
    for (int i = 0; i < num_aus; i++) { start[i] = pos; pos += au_len[i] + 2; }
    if (pos != 110 * s) return -1;
    memset(out, 0, (size_t)120 * s);
    out[2] = (uint8_t)((dac_rate << 6) | (sbr << 5) | (ch_mode << 4) | (ps << 3) | (surr & 7));
    for (int i = 1; i < num_aus; i++) {
        int bit = 24 + 12 * (i - 1), byte = bit >> 3;
        if (bit & 4) { out[byte] |= (uint8_t)(start[i] >> 8); out[byte + 1] = (uint8_t)start[i]; }
        else { out[byte] = (uint8_t)(start[i] >> 4); out[byte + 1] |= (uint8_t)((start[i] & 0xF) << 4); }
    }
    for (int i = 0; i < num_aus; i++) {
        memcpy(out + start[i], au[i], (size_t)au_len[i]);
        uint16_t crc = dab_crc16(out + start[i], au_len[i]);
        out[start[i] + au_len[i]] = (uint8_t)(crc >> 8); out[start[i] + au_len[i] + 1] = (uint8_t)crc;
    }
    /* the fire code covers bytes 2..10 whatever they hold (header, AU table, first AU bytes) */
    uint16_t fc = dab_firecode(out + 2, 9);
    out[0] = (uint8_t)(fc >> 8); out[1] = (uint8_t)fc;
    uint8_t msg[110], par[10];
    for (int j = 0; j < s; j++) {
        for (int k = 0; k < 110; k++) msg[k] = out[j + k * s];
        dab_rs_encode_120_110(msg, par);
        for (int k = 0; k < 10; k++) out[j + (110 + k) * s] = par[k];
    }
    return 0;
}

/* ------------------------------------------------------------------------------------------------
 * DAB+ audio super frame DECODER (ETSI TS 102 563 §5.2 audio super frame syntax, §5.3 fire code
 * x^16+x^14+x^13+x^12+x^11+x^5+x^3+x^2+x+1 over bytes 2..10, §6 RS(120,110) over GF(2^8) with
 * p(x) = x^8+x^4+x^3+x^2+1 and generator roots alpha^0..alpha^9, AU CRC = CRC-16-CCITT inverted).
 * Oracle for the GPU kernel k_superframe: same records, bit for bit.  Synchronisation is the sliding
 * window of five logical frames any receiver uses: try the window, on failure slide by one frame.
 * ------------------------------------------------------------------------------------------------ */
static uint8_t gmul(uint8_t a, uint8_t b) { return (a && b) ? gexp[glog[a] + glog[b]] : 0; }
static uint8_t gdiv(uint8_t a, uint8_t b) { return a ? gexp[glog[a] + 255 - glog[b]] : 0; }
static uint8_t gpow(int e) { return gexp[((e % 255) + 255) % 255]; }

/* cw[0..109] data, cw[110..119] parity, cw[0] = highest power.  Returns corrected bytes, -1 = uncorrectable. */
int dab_rs_decode_120_110(uint8_t *cw)
{
    gf_init();
    enum { N = 120, T2 = 10 };
    uint8_t S[T2];
    int clean = 1;
    for (int i = 0; i < T2; i++) {
        uint8_t s = 0, a = gpow(i);
        for (int k = 0; k < N; k++) s = (uint8_t)(gmul(s, a) ^ cw[k]);
        S[i] = s;
        if (s) clean = 0;
    }
    if (clean) return 0;
    uint8_t C[T2 + 1] = {1}, B[T2 + 1] = {1}, b = 1;           /* Berlekamp-Massey */
    int L = 0, m = 1;
    for (int n = 0; n < T2; n++) {
        uint8_t d = S[n];
        for (int i = 1; i <= L; i++) d ^= gmul(C[i], S[n - i]);
        if (!d) { m++; continue; }
        uint8_t Tp[T2 + 1];
        memcpy(Tp, C, sizeof Tp);
        uint8_t coef = gdiv(d, b);
        for (int i = 0; i + m <= T2; i++) C[i + m] ^= gmul(coef, B[i]);
        if (2 * L <= n) { L = n + 1 - L; memcpy(B, Tp, sizeof B); b = d; m = 1; }
        else m++;
    }
    if (L > T2 / 2) return -1;
    int pos[T2 / 2], nerr = 0;                                  /* Chien search over the 120 positions */
    for (int k = 0; k < N; k++) {
        int p = N - 1 - k;
        uint8_t v = 0;
        for (int i = 0; i <= L; i++) v ^= gmul(C[i], gpow(-p * i));
        if (!v) { if (nerr == T2 / 2) return -1; pos[nerr++] = k; }
    }
    if (nerr != L) return -1;
    uint8_t Om[T2];                                             /* Forney, first consecutive root alpha^0 */
    memset(Om, 0, sizeof Om);
    for (int i = 0; i < T2; i++)
        for (int j = 0; j <= L && j <= i; j++) Om[i] ^= gmul(S[i - j], C[j]);
    for (int e = 0; e < nerr; e++) {
        int p = N - 1 - pos[e];
        uint8_t Xinv = gpow(-p), num = 0, den = 0;
        for (int i = T2 - 1; i >= 0; i--) num = (uint8_t)(gmul(num, Xinv) ^ Om[i]);
        for (int i = 1; i <= L; i += 2) den ^= gmul(C[i], gpow(-p * (i - 1)));
        if (!den) return -1;
        cw[pos[e]] ^= gmul(gdiv(num, den), gpow(p));
    }
    return nerr;
}

/* one record per decoded super frame; layout shared with include/dabx.h (dabx_superframe_t) */
typedef struct {
    uint32_t first_frame;      /* index of its first logical frame among the frames pushed so far */
    uint8_t header;            /* dac_rate<<6 | sbr<<5 | aac_channel_mode<<4 | ps<<3 | mpeg_surround */
    uint8_t num_aus;
    uint8_t au_valid;          /* bit i: AU i has sane bounds */
    uint8_t au_ok;             /* bit i: and its CRC is good   */
    uint16_t au_start[8];      /* byte offsets into the 110 s data bytes; au_start[num_aus] = 110 s */
    uint16_t rs_corrected;     /* bytes corrected in this super frame */
    uint16_t rs_failed;        /* code words left uncorrected          */
    uint32_t pad;
} dab_sf_rec_t;                /* 32 bytes */

typedef struct {
    int s, frame_bytes, carry;         /* carry: logical frames waiting in buf (0..4) */
    uint32_t frames_seen;
    uint32_t stats[6];                 /* superframes, au_ok, au_crc_err, rs_corrected, rs_uncorrectable, sync_loss */
    int synced;
    uint8_t buf[4 * 24 * 24 + 8];      /* up to 4 frames of 3*192 bytes */
} dab_sf_state_t;

void dab_sf_init(dab_sf_state_t *st, int kbps)
{
    memset(st, 0, sizeof *st);
    st->s = kbps / 8;
    st->frame_bytes = 3 * kbps;
}

/* try the five logical frames at sf (120 s bytes, modified in place by the RS correction) */
static int sf_try(dab_sf_state_t *st, uint8_t *sf, dab_sf_rec_t *rec)
{
    const int s = st->s;
    int corrected = 0, failed = 0;
    uint8_t cw[120];
    for (int j = 0; j < s; j++) {
        for (int k = 0; k < 120; k++) cw[k] = sf[j + k * s];
        int r = dab_rs_decode_120_110(cw);
        if (r < 0) { failed++; continue; }
        if (r > 0) { corrected += r; for (int k = 0; k < 110; k++) sf[j + k * s] = cw[k]; }
    }
    if (dab_firecode(sf + 2, 9) != ((sf[0] << 8) | sf[1]) || (sf[0] == 0 && sf[1] == 0 && sf[2] == 0)) return 0;
    st->stats[0]++; st->stats[3] += (uint32_t)corrected; st->stats[4] += (uint32_t)failed;
    const int dac = (sf[2] >> 6) & 1, sbr = (sf[2] >> 5) & 1;
    const int num = dac ? (sbr ? 3 : 6) : (sbr ? 2 : 4);
    memset(rec, 0, sizeof *rec);
    rec->header = sf[2] & 0x7F;
    rec->num_aus = (uint8_t)num;
    rec->rs_corrected = (uint16_t)corrected; rec->rs_failed = (uint16_t)failed;
    int start[8];
    start[0] = dac ? (sbr ? 6 : 11) : (sbr ? 5 : 8);
    for (int i = 1; i < num; i++) {
        int bit = 24 + 12 * (i - 1), byte = bit >> 3;
        start[i] = (bit & 4) ? (((sf[byte] & 0x0F) << 8) | sf[byte + 1]) : ((sf[byte] << 4) | (sf[byte + 1] >> 4));
    }
    start[num] = 110 * s;
    for (int i = 0; i <= num; i++) rec->au_start[i] = (uint16_t)start[i];
    for (int i = 0; i < num; i++) {
        int len = start[i + 1] - start[i];
        if (start[i] < start[0] || len < 3 || start[i + 1] > 110 * s) { st->stats[2]++; continue; }
        rec->au_valid |= (uint8_t)(1 << i);
        const uint8_t *au = sf + start[i];
        if (dab_crc16(au, len - 2) == ((au[len - 2] << 8) | au[len - 1])) { rec->au_ok |= (uint8_t)(1 << i); st->stats[1]++; }
        else st->stats[2]++;
    }
    return 1;
}

/* Feed n logical frames (n * 3 kbps bytes).  recs/data receive up to max super frames (data: 110 s bytes each).
 * Returns the number decoded. */
int dab_sf_push(dab_sf_state_t *st, const uint8_t *frames, int n, dab_sf_rec_t *recs, uint8_t *data, int max)
{
    const int fb = st->frame_bytes, total = st->carry + n;
    uint8_t *seq = (uint8_t *)malloc((size_t)(total + 1) * (size_t)fb), *win = (uint8_t *)malloc((size_t)5 * (size_t)fb);
    memcpy(seq, st->buf, (size_t)st->carry * (size_t)fb);
    memcpy(seq + (size_t)st->carry * (size_t)fb, frames, (size_t)n * (size_t)fb);
    const uint32_t base = st->frames_seen - (uint32_t)st->carry;
    int i = 0, out = 0;
    while (i + 5 <= total) {
        memcpy(win, seq + (size_t)i * (size_t)fb, (size_t)5 * (size_t)fb);
        dab_sf_rec_t rec;
        if (sf_try(st, win, &rec)) {
            rec.first_frame = base + (uint32_t)i;
            if (out < max) { recs[out] = rec; memcpy(data + (size_t)out * 110u * (size_t)st->s, win, 110u * (size_t)st->s); }
            out++;
            st->synced = 1;
            i += 5;
        } else {
            if (st->synced) { st->stats[5]++; st->synced = 0; }
            i += 1;
        }
    }
    st->carry = total - i;
    memcpy(st->buf, seq + (size_t)i * (size_t)fb, (size_t)st->carry * (size_t)fb);
    st->frames_seen += (uint32_t)n;
    free(seq); free(win);
    return out < max ? out : max;
}
