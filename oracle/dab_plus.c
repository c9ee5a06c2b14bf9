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
