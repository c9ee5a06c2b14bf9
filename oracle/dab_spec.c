/*
 * oracle/dab_spec.c — TEST INFRASTRUCTURE ONLY.  See dab_spec.h for the
 * provenance statement ("parity unpinned": follows ETSI EN 300 401, the
 * reference has no source or vectors for this path).
 */
#include "dab_spec.h"
#include <string.h>
#include <stdlib.h>

/* EN 300 401 §14.6.1: PI(0)=0, PI(i) = (13 PI(i-1) + 511) mod 2048; keep the
 * values in [256,1792] except 1024, in order of generation; k = d - 1024. */
void dab_freq_interleaver(int16_t k_of_n[DAB_K])
{
    int pi = 0, n = 0;
    for (int i = 0; i < 2048; i++) {
        if (i > 0) pi = (13 * pi + 511) & 2047;
        if (pi >= 256 && pi <= 1792 && pi != 1024) k_of_n[n++] = (int16_t)(pi - 1024);
    }
}

/* EN 300 401 §14.3.2 tables 39 and 41 (Mode I). */
static const uint8_t prs_h[4][16] = {
    {0, 2, 0, 0, 0, 0, 1, 1, 2, 0, 0, 0, 2, 2, 1, 1},
    {0, 3, 2, 3, 0, 1, 3, 0, 2, 1, 2, 3, 2, 3, 3, 0},
    {0, 0, 0, 2, 0, 2, 1, 3, 2, 2, 0, 2, 2, 0, 1, 3},
    {0, 1, 2, 1, 0, 3, 3, 2, 2, 3, 2, 1, 2, 1, 3, 2},
};
/* (i, n) per block of 32 carriers: k' = -768,-736,…,-32 then 1,33,…,737 */
static const uint8_t prs_in[48][2] = {
    {0,1},{1,2},{2,0},{3,1},{0,3},{1,2},{2,2},{3,3},{0,2},{1,1},{2,2},{3,3},
    {0,1},{1,2},{2,3},{3,3},{0,2},{1,2},{2,2},{3,1},{0,1},{1,3},{2,1},{3,2},
    {0,3},{3,1},{2,1},{1,1},{0,2},{3,2},{2,1},{1,0},{0,2},{3,2},{2,3},{1,3},
    {0,0},{3,2},{2,1},{1,3},{0,3},{3,3},{2,3},{1,0},{0,3},{3,0},{2,1},{1,1},
};

void dab_prs_quadrants(int8_t q[DAB_TU])
{
    memset(q, -1, DAB_TU);
    for (int b = 0; b < 48; b++) {
        int kp = (b < 24) ? (-768 + 32 * b) : (1 + 32 * (b - 24));
        int i = prs_in[b][0], n = prs_in[b][1];
        for (int j = 0; j < 32; j++) {
            int k = kp + j;
            q[k & 2047] = (int8_t)((prs_h[i][j & 15] + n) & 3);
        }
    }
}

/* Table 29 has 8 groups of 4 flags; the first flag of every group is always
 * kept and PI further flags are added group by group in the order
 * 0,4,2,6,1,5,3,7 (three rounds: second, third, fourth flag). */
void dab_punct_vector(int pi, uint8_t v[32])
{
    static const uint8_t rank[8] = {0, 4, 2, 6, 1, 5, 3, 7};
    for (int g = 0; g < 8; g++) {
        int ones = 1;
        for (int m = 0; m < 3; m++)
            if (pi - 8 * m > rank[g]) ones++;
        for (int j = 0; j < 4; j++) v[4 * g + j] = (uint8_t)(j < ones);
    }
}

int dab_ti_delay(int i)
{
    static const uint8_t d[16] = {0, 8, 4, 12, 2, 10, 6, 14, 1, 9, 5, 13, 3, 11, 7, 15};
    return d[i & 15];
}

void dab_prbs(uint8_t *bits, int n)
{
    unsigned reg = 0x1FF; /* bit 0 = newest */
    for (int i = 0; i < n; i++) {
        unsigned b = ((reg >> 8) ^ (reg >> 4)) & 1; /* taps at delays 9 and 5 */
        reg = ((reg << 1) | b) & 0x1FF;
        bits[i] = (uint8_t)b;
    }
}

uint16_t dab_crc16(const uint8_t *data, int nbytes)
{
    unsigned crc = 0xFFFF;
    for (int i = 0; i < nbytes; i++) {
        crc ^= (unsigned)data[i] << 8;
        for (int b = 0; b < 8; b++)
            crc = (crc & 0x8000) ? ((crc << 1) ^ 0x1021) & 0xFFFF : (crc << 1) & 0xFFFF;
    }
    return (uint16_t)(~crc & 0xFFFF);
}

/* generator taps, bit 6 = delay 0 (the input) … bit 0 = delay 6 */
static const uint8_t gen[4] = {0133, 0171, 0145, 0133};

int dab_conv_output(int state, int in)
{
    /* shift register: bit 6 = input, bits 5..0 = state (bit 5 newest) */
    unsigned sr = ((unsigned)(in & 1) << 6) | (unsigned)(state & 63);
    int out = 0;
    for (int g = 0; g < 4; g++)
        out = (out << 1) | (__builtin_popcount(sr & gen[g]) & 1);
    return out;
}

void dab_conv_encode(const uint8_t *in, int nin, uint8_t *out)
{
    int state = 0;
    for (int t = 0; t < nin + 6; t++) {
        int u = (t < nin) ? (in[t] & 1) : 0;
        int o = dab_conv_output(state, u);
        out[4 * t + 0] = (uint8_t)((o >> 3) & 1);
        out[4 * t + 1] = (uint8_t)((o >> 2) & 1);
        out[4 * t + 2] = (uint8_t)((o >> 1) & 1);
        out[4 * t + 3] = (uint8_t)(o & 1);
        state = (state >> 1) | (u << 5);
    }
}

static void profile_finish(dab_profile_t *p)
{
    int blocks = 0, coded = 0;
    for (int s = 0; s < p->nseg; s++) {
        blocks += p->L[s];
        coded += p->L[s] * 4 * (8 + p->PI[s]);
    }
    p->n_in = blocks * 32;
    p->n_coded = coded + 12;
}

void dab_profile_fic(dab_profile_t *p)
{
    memset(p, 0, sizeof *p);
    p->nseg = 2;
    p->L[0] = 21; p->PI[0] = 16;
    p->L[1] = 3;  p->PI[1] = 15;
    profile_finish(p);
}

int dab_profile_eep(int option, int level, int kbps, dab_profile_t *p)
{
    memset(p, 0, sizeof *p);
    p->nseg = 2;
    if (level < 1 || level > 4 || kbps <= 0) return -1;
    if (option == 0) {
        if (kbps % 8) return -1;
        int n = kbps / 8;
        switch (level) {
        case 1: p->L[0] = 6 * n - 3; p->L[1] = 3; p->PI[0] = 24; p->PI[1] = 23; p->n_cu = 12 * n; break;
        case 2:
            if (n == 1) { p->L[0] = 5; p->L[1] = 1; p->PI[0] = 13; p->PI[1] = 12; }
            else { p->L[0] = 2 * n - 3; p->L[1] = 4 * n + 3; p->PI[0] = 14; p->PI[1] = 13; }
            p->n_cu = 8 * n; break;
        case 3: p->L[0] = 6 * n - 3; p->L[1] = 3; p->PI[0] = 8; p->PI[1] = 7; p->n_cu = 6 * n; break;
        case 4: p->L[0] = 4 * n - 3; p->L[1] = 2 * n + 3; p->PI[0] = 3; p->PI[1] = 2; p->n_cu = 4 * n; break;
        }
    } else if (option == 1) {
        static const uint8_t pi1[5] = {0, 10, 6, 4, 2};
        static const uint8_t cu[5] = {0, 27, 21, 18, 15};
        if (kbps % 32) return -1;
        int n = kbps / 32;
        p->L[0] = 24 * n - 3; p->L[1] = 3;
        p->PI[0] = pi1[level]; p->PI[1] = pi1[level] - 1;
        p->n_cu = cu[level] * n;
    } else return -1;
    profile_finish(p);
    if (p->n_cu > DAB_NCU || p->n_coded != p->n_cu * DAB_CU_BITS) return -1;
    return 0;
}

/* EN 300 401 §11.3.1 tables 31-33: unequal error protection, indexed by the 6-bit table
 * index of FIG 0/1's short form (table 8).  Columns: bit rate, protection level, L1..L4,
 * PI1..PI4, padding bits.  Every row satisfies sum(L)*32 = 24*bitrate and
 * sum(L*4*(8+PI)) + 12 + padding = 64 * sub-channel size (tests/test_oracle_spec.py). */
static const int16_t uep_table[64][11] = {
    {32, 5, 3, 4, 17, 0, 5, 3, 2, 0, 0},
    {32, 4, 3, 3, 18, 0, 11, 6, 5, 0, 0},
    {32, 3, 3, 4, 14, 3, 15, 9, 6, 8, 0},
    {32, 2, 3, 4, 14, 3, 22, 13, 8, 13, 0},
    {32, 1, 3, 5, 13, 3, 24, 17, 12, 17, 4},
    {48, 5, 4, 3, 26, 3, 5, 4, 2, 3, 0},
    {48, 4, 3, 4, 26, 3, 9, 6, 4, 6, 0},
    {48, 3, 3, 4, 26, 3, 15, 10, 6, 9, 4},
    {48, 2, 3, 4, 26, 3, 24, 14, 8, 15, 0},
    {48, 1, 3, 5, 25, 3, 24, 18, 13, 18, 0},
    {56, 5, 6, 10, 23, 3, 5, 4, 2, 3, 0},
    {56, 4, 6, 10, 23, 3, 9, 6, 4, 5, 0},
    {56, 3, 6, 12, 21, 3, 16, 7, 6, 9, 0},
    {56, 2, 6, 10, 23, 3, 23, 13, 8, 13, 8},
    {64, 5, 6, 9, 31, 2, 5, 3, 2, 3, 0},
    {64, 4, 6, 9, 33, 0, 11, 6, 5, 0, 0},
    {64, 3, 6, 12, 27, 3, 16, 8, 6, 9, 0},
    {64, 2, 6, 10, 29, 3, 23, 13, 8, 13, 8},
    {64, 1, 6, 11, 28, 3, 24, 18, 12, 18, 4},
    {80, 5, 6, 10, 41, 3, 6, 3, 2, 3, 0},
    {80, 4, 6, 10, 41, 3, 11, 6, 5, 6, 0},
    {80, 3, 6, 11, 40, 3, 16, 8, 6, 7, 0},
    {80, 2, 6, 10, 41, 3, 23, 13, 8, 13, 8},
    {80, 1, 6, 10, 41, 3, 24, 17, 12, 18, 4},
    {96, 5, 7, 9, 53, 3, 5, 4, 2, 4, 0},
    {96, 4, 7, 10, 52, 3, 9, 6, 4, 6, 0},
    {96, 3, 6, 12, 51, 3, 16, 9, 6, 10, 4},
    {96, 2, 6, 10, 53, 3, 22, 12, 9, 12, 0},
    {96, 1, 6, 13, 50, 3, 24, 18, 13, 19, 0},
    {112, 5, 14, 17, 50, 3, 5, 4, 2, 5, 0},
    {112, 4, 11, 21, 49, 3, 9, 6, 4, 8, 0},
    {112, 3, 11, 23, 47, 3, 16, 8, 6, 9, 0},
    {112, 2, 11, 21, 49, 3, 23, 12, 9, 14, 4},
    {128, 5, 12, 19, 62, 3, 5, 3, 2, 4, 0},
    {128, 4, 11, 21, 61, 3, 11, 6, 5, 7, 0},
    {128, 3, 11, 22, 60, 3, 16, 9, 6, 10, 4},
    {128, 2, 11, 21, 61, 3, 22, 12, 9, 14, 0},
    {128, 1, 11, 20, 62, 3, 24, 17, 13, 19, 8},
    {160, 5, 11, 19, 87, 3, 5, 4, 2, 4, 0},
    {160, 4, 11, 23, 83, 3, 11, 6, 5, 9, 0},
    {160, 3, 11, 24, 82, 3, 16, 8, 6, 11, 0},
    {160, 2, 11, 21, 85, 3, 22, 11, 9, 13, 0},
    {160, 1, 11, 22, 84, 3, 24, 18, 12, 19, 0},
    {192, 5, 11, 20, 110, 3, 6, 4, 2, 5, 0},
    {192, 4, 11, 22, 108, 3, 10, 6, 4, 9, 0},
    {192, 3, 11, 24, 106, 3, 16, 10, 6, 11, 0},
    {192, 2, 11, 20, 110, 3, 22, 13, 9, 13, 8},
    {192, 1, 11, 21, 109, 3, 24, 20, 13, 24, 0},
    {224, 5, 12, 22, 131, 3, 8, 6, 2, 6, 4},
    {224, 4, 12, 26, 127, 3, 12, 8, 4, 11, 0},
    {224, 3, 11, 20, 134, 3, 16, 10, 7, 9, 0},
    {224, 2, 11, 22, 132, 3, 24, 16, 10, 15, 0},
    {224, 1, 11, 24, 130, 3, 24, 20, 12, 20, 4},
    {256, 5, 11, 24, 154, 3, 6, 5, 2, 5, 0},
    {256, 4, 11, 24, 154, 3, 12, 9, 5, 10, 4},
    {256, 3, 11, 27, 151, 3, 16, 10, 7, 10, 0},
    {256, 2, 11, 22, 156, 3, 24, 14, 10, 13, 8},
    {256, 1, 11, 26, 152, 3, 24, 19, 14, 18, 4},
    {320, 5, 11, 26, 200, 3, 8, 5, 2, 6, 4},
    {320, 4, 11, 25, 201, 3, 13, 9, 5, 10, 8},
    {320, 2, 11, 26, 200, 3, 24, 17, 9, 17, 0},
    {384, 5, 11, 27, 247, 3, 8, 6, 2, 7, 0},
    {384, 3, 11, 24, 250, 3, 16, 9, 7, 10, 4},
    {384, 1, 12, 28, 245, 3, 24, 20, 14, 23, 8}
};

int dab_profile_uep(int index, dab_profile_t *p, int *kbps, int *level)
{
    memset(p, 0, sizeof *p);
    if (index < 0 || index > 63) return -1;
    const int16_t *r = uep_table[index];
    p->nseg = r[5] ? 4 : 3;
    int coded = 12 + r[10], blocks = 0;
    for (int s = 0; s < p->nseg; s++) { p->L[s] = r[2 + s]; p->PI[s] = r[6 + s]; blocks += p->L[s]; coded += p->L[s] * 4 * (8 + p->PI[s]); }
    p->n_in = 32 * blocks;
    p->n_coded = coded;                 /* includes the padding bits, which carry no information */
    p->n_cu = coded / DAB_CU_BITS;
    if (kbps) *kbps = r[0];
    if (level) *level = r[1];
    return (coded % DAB_CU_BITS) ? -1 : 0;
}

int dab_profile_any(int option, int level, int kbps, dab_profile_t *p)
{
    if (option == 2) return dab_profile_uep(level, p, NULL, NULL);
    return dab_profile_eep(option, level, kbps, p);
}

int dab_profile_stepinfo(const dab_profile_t *p, uint32_t *info)
{
    uint8_t v[32];
    int t = 0;
    uint32_t off = 0;
    for (int s = 0; s < p->nseg; s++) {
        dab_punct_vector(p->PI[s], v);
        for (int blk = 0; blk < p->L[s]; blk++)
            for (int g = 0; g < 32; g++) {           /* 32 steps of 4 = 128 mother bits */
                const uint8_t *f = v + 4 * (g & 7);
                unsigned mask = (f[0] << 3) | (f[1] << 2) | (f[2] << 1) | f[3];
                info[t++] = (off << 4) | mask;
                off += f[0] + f[1] + f[2] + f[3];
            }
    }
    for (int g = 0; g < 6; g++) { info[t++] = (off << 4) | 0xC; off += 2; } /* tail: 1100 x 6 */
    return t;
}

int dab_puncture(const dab_profile_t *p, const uint8_t *mother, uint8_t *coded)
{
    uint32_t *info = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)(p->n_in + 8));
    int nstep = dab_profile_stepinfo(p, info), n = 0;
    for (int t = 0; t < nstep; t++)
        for (int j = 0; j < 4; j++)
            if (info[t] & (8u >> j)) coded[n++] = mother[4 * t + j];
    free(info);
    return n;
}
