/*
 * oracle/dab_rx.c — TEST INFRASTRUCTURE ONLY (checker, never the product).
 *
 * Scalar CPU restatement of the DAB Mode-I receive chain that the HIP library
 * (abracadabra_amd/csrc) implements: null search, CP/PRS synchronisation,
 * 2048-point FFT, pi/4-DQPSK demapping, frequency and time de-interleaving,
 * depuncturing, K=7 Viterbi, energy de-dispersal and FIB CRC.
 *
 * PARITY UNPINNED.  The reference implements this chain only inside the closed
 * binary lib/linux_x86_64/libdabsdr.so.4.0.1 (boundary: lib/linux_x86_64/
 * dabsdr.h:397-429; caller src/radiocontrol.cpp:81-93).  There is no reference
 * source, test or golden vector to follow, and the binary is not executed.
 * The bit-level rules follow ETSI EN 300 401; the arithmetic contract (what
 * "bit-exact" means for the float stages) is fixed in DESIGN.md §3: every
 * float operation is a single IEEE-754 binary32 operation in a stated order
 * (a fused multiply-add only where fmaf() is written out: the complex
 * products), so a GPU and a CPU evaluation agree bit for bit.  Build with
 * -ffp-contract=off; -mfma makes fmaf() one instruction where the CPU has it.
 */
#include "dab_spec.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define NFFT 2048
#define BACKOFF 24          /* FFT window starts this many samples inside the guard */
#define MW 16               /* integer carrier-offset search range (kHz)            */
#define SOFT_EXP 15         /* soft-bit scale exponent, see demap()                 */
#define SOFT_MAX 31.0f      /* soft bits are limited to +-31: twice the sum of two fits a byte (the GPU packs 2 (x0 + x3)) */
#define PM_INIT (-1000000)  /* path metric of states other than 0 at trellis start  */
#define LOCK_THR 48.0f
#define EARLY_SPAN 400      /* the first path may lead the strongest one by up to this many samples ...   */
#define EARLY_THR 0.125f    /* ... if it carries at least this fraction of its power (-9 dB)               */
#define SLOPE_MAX (60 << 16)/* sampling-clock tracker: |drift| <= 60 samples per frame (~300 ppm), Q16      */
#define SCO_MIN (1 << 16)            /* de-rotate the differential product from this drift on: 1 sample per frame = 5.1 ppm */

/* ------------------------------------------------------------------ tables */
typedef struct {
    float wr[NFFT], wi[NFFT];       /* W^k = exp(-j 2 pi k / 2048)                 */
    float nhr[NFFT], nhi[NFFT];     /* NCO coarse: exp(+j 2 pi k / 2^11)           */
    float nlr[NFFT], nli[NFFT];     /* NCO fine:   exp(+j 2 pi k / 2^22)           */
    int8_t  prsq[NFFT];             /* PRS quadrant per bin, -1 unused             */
    int8_t  prsdq[NFFT];            /* quadrant(P[k]) - quadrant(P[k-1]), -1 n/a   */
    int16_t kofn[DAB_K];
    int16_t n_of_bin[NFFT];         /* frequency de-interleaver, -1 unused         */
    int16_t bin_of_pos[NFFT];       /* FFT output placement (digit reversal)       */
    int16_t pos_of_bin[NFFT];
    int16_t cfo_car[1534];          /* carriers k with k and k-1 both active       */
    uint8_t prbs[DAB_CIF_BITS];
    int ready;
} tables_t;
static tables_t T;

static void tables_init(void)
{
    if (T.ready) return;
    for (int k = 0; k < NFFT; k++) {
        double a = 2.0 * M_PI * k / 2048.0;
        T.wr[k] = (float)cos(a); T.wi[k] = (float)(-sin(a));
        T.nhr[k] = (float)cos(a); T.nhi[k] = (float)sin(a);
        double b = 2.0 * M_PI * k / 4194304.0;
        T.nlr[k] = (float)cos(b); T.nli[k] = (float)sin(b);
    }
    dab_prs_quadrants(T.prsq);
    dab_freq_interleaver(T.kofn);
    memset(T.n_of_bin, -1, sizeof T.n_of_bin);
    for (int n = 0; n < DAB_K; n++) T.n_of_bin[T.kofn[n] & 2047] = (int16_t)n;
    for (int p = 0; p < NFFT; p++) {
        int k1 = p >> 8, c = (p >> 5) & 7, f = (p >> 2) & 7, kk = p & 3;
        int b = k1 + 8 * c + 64 * f + 512 * kk;
        T.bin_of_pos[p] = (int16_t)b; T.pos_of_bin[b] = (int16_t)p;
    }
    int j = 0;
    for (int b = 0; b < NFFT; b++) T.prsdq[b] = -1;
    for (int k = -767; k <= 768; k++) {
        if (k == 0 || k == 1) continue;
        T.cfo_car[j++] = (int16_t)k;
        T.prsdq[k & 2047] = (int8_t)((T.prsq[k & 2047] - T.prsq[(k - 1) & 2047]) & 3);
    }
    dab_prbs(T.prbs, DAB_CIF_BITS);
    T.ready = 1;
}

/* ----------------------------------------------------- arithmetic contract */
/* complex products: one rounded product, then one fused multiply-add, the same two operations in the same roles as
 * the kernels' cmul / cmulc (abracadabra_amd/csrc/dabx_kernels.hip) */
static inline void cmul(float ar, float ai, float br, float bi, float *yr, float *yi)
{
    float p1 = ai * bi, p3 = ai * br;
    float r = fmaf(ar, br, -p1), i = fmaf(ar, bi, p3);
    *yr = r; *yi = i;
}
static inline void cmulc(float ar, float ai, float br, float bi, float *yr, float *yi)
{   /* a * conj(b) */
    float p1 = ai * bi, p3 = ar * bi;
    float r = fmaf(ar, br, p1), i = fmaf(ai, br, -p3);
    *yr = r; *yi = i;
}
/* multiply by exp(-j q pi/2): exact */
static inline void rotq(float xr, float xi, int q, float *yr, float *yi)
{
    switch (q & 3) {
    case 0: *yr = xr;  *yi = xi;  break;
    case 1: *yr = xi;  *yi = -xr; break;
    case 2: *yr = -xr; *yi = -xi; break;
    default:*yr = -xi; *yi = xr;  break;
    }
}

/* fixed-order sum of 256 per-thread partials: xor butterfly inside each group
 * of 64 (d = 1,2,4,8,16,32), then (W0+W1)+(W2+W3) */
static float reduce256(const float *part)
{
    float p[256], q[256];
    memcpy(p, part, sizeof p);
    for (int d = 1; d < 64; d <<= 1) {
        for (int i = 0; i < 256; i++) q[i] = p[i] + p[i ^ d];
        memcpy(p, q, sizeof p);
    }
    float s01 = p[0] + p[64], s23 = p[128] + p[192];
    return s01 + s23;
}

/* integer CORDIC: angle of (x + j y) in units of 2^-32 turn */
static const int32_t cordic_tab[28] = {
    536870912, 316933406, 167458907, 85004756, 42667331, 21354465, 10679838, 5340245, 2670163, 1335087,
    667544, 333772, 166886, 83443, 41722, 20861, 10430, 5215, 2608, 1304, 652, 326, 163, 81, 41, 20, 10, 5};
int32_t orx_cordic(int64_t y, int64_t x)
{
    if (x == 0 && y == 0) return 0;
    uint64_t ax = (uint64_t)(x < 0 ? -x : x), ay = (uint64_t)(y < 0 ? -y : y), m = ax > ay ? ax : ay;
    int sh = 0;
    while ((m >> sh) >= (1ULL << 29)) sh++;
    if (sh) { x >>= sh; y >>= sh; }             /* arithmetic shifts */
    else while ((m << 1) < (1ULL << 29)) { m <<= 1; x *= 2; y *= 2; }
    uint32_t ang = 0;
    if (x < 0) { x = -x; y = -y; ang = 0x80000000u; }
    for (int i = 0; i < 28; i++) {
        int64_t xs = x >> i, ys = y >> i;
        if (y > 0) { x += ys; y -= xs; ang += (uint32_t)cordic_tab[i]; }
        else       { x -= ys; y += xs; ang -= (uint32_t)cordic_tab[i]; }
    }
    return (int32_t)ang;
}

/* NCO: exp(+j 2 pi theta / 2^32) from the two tables */
static inline void nco(uint32_t th, float *cr, float *ci)
{
    unsigned hi = th >> 21, lo = (th >> 10) & 2047;
    cmul(T.nhr[hi], T.nhi[hi], T.nlr[lo], T.nli[lo], cr, ci);
}

/* ----------------------------------------------------------------- the FFT */
static inline void r4(float *r, float *i)
{   /* natural-order 4-point DFT */
    float p0r = r[0] + r[2], p0i = i[0] + i[2], p1r = r[0] - r[2], p1i = i[0] - i[2];
    float q0r = r[1] + r[3], q0i = i[1] + i[3], tr = r[1] - r[3], ti = i[1] - i[3];
    float q1r = ti, q1i = -tr;                           /* (u1-u3) * (-j) */
    r[0] = p0r + q0r; i[0] = p0i + q0i; r[2] = p0r - q0r; i[2] = p0i - q0i;
    r[1] = p1r + q1r; i[1] = p1i + q1i; r[3] = p1r - q1r; i[3] = p1i - q1i;
}
static inline void r8(float *r, float *i)
{   /* natural-order 8-point DFT, decimation in frequency */
    const float c8 = 0.70710678118654752440f;
    float ar[4], ai[4], br[4], bi[4];
    for (int j = 0; j < 4; j++) {
        ar[j] = r[j] + r[j + 4]; ai[j] = i[j] + i[j + 4];
        br[j] = r[j] - r[j + 4]; bi[j] = i[j] - i[j + 4];
    }
    float t0, t1;
    t0 = br[1] + bi[1]; t1 = bi[1] - br[1]; br[1] = c8 * t0; bi[1] = c8 * t1;           /* * W8^1 */
    t0 = br[2]; br[2] = bi[2]; bi[2] = -t0;                                             /* * W8^2 */
    t0 = bi[3] - br[3]; t1 = br[3] + bi[3]; br[3] = c8 * t0; bi[3] = -(c8 * t1);        /* * W8^3 */
    r4(ar, ai); r4(br, bi);
    for (int m = 0; m < 4; m++) { r[2 * m] = ar[m]; i[2 * m] = ai[m]; r[2 * m + 1] = br[m]; i[2 * m + 1] = bi[m]; }
}

/* in place; result bin T.bin_of_pos[p] ends at position p */
static void fft_pos(float *re, float *im)
{
    float vr[8], vi[8];
    for (int t = 0; t < 256; t++) {                      /* pass A: stride 256, twiddle W^(t*c) */
        for (int j = 0; j < 8; j++) { vr[j] = re[t + 256 * j]; vi[j] = im[t + 256 * j]; }
        r8(vr, vi);
        for (int c = 1; c < 8; c++) cmul(vr[c], vi[c], T.wr[t * c], T.wi[t * c], &vr[c], &vi[c]);
        for (int c = 0; c < 8; c++) { re[t + 256 * c] = vr[c]; im[t + 256 * c] = vi[c]; }
    }
    for (int t = 0; t < 256; t++) {                      /* pass B: stride 32, twiddle W^(8*b*c) */
        int base = (t >> 5) * 256 + (t & 31), b = t & 31;
        for (int j = 0; j < 8; j++) { vr[j] = re[base + 32 * j]; vi[j] = im[base + 32 * j]; }
        r8(vr, vi);
        for (int c = 1; c < 8; c++) cmul(vr[c], vi[c], T.wr[8 * b * c], T.wi[8 * b * c], &vr[c], &vi[c]);
        for (int c = 0; c < 8; c++) { re[base + 32 * c] = vr[c]; im[base + 32 * c] = vi[c]; }
    }
    for (int t = 0; t < 256; t++) {                      /* pass C: stride 4, twiddle W^(64*e*f) */
        int base = (t >> 2) * 32 + (t & 3), e = t & 3;
        for (int j = 0; j < 8; j++) { vr[j] = re[base + 4 * j]; vi[j] = im[base + 4 * j]; }
        r8(vr, vi);
        for (int f = 1; f < 8; f++) cmul(vr[f], vi[f], T.wr[64 * e * f], T.wi[64 * e * f], &vr[f], &vi[f]);
        for (int f = 0; f < 8; f++) { re[base + 4 * f] = vr[f]; im[base + 4 * f] = vi[f]; }
    }
    for (int g = 0; g < 512; g++) r4(re + 4 * g, im + 4 * g);  /* pass D: radix 4 */
}

/* natural-order convenience wrapper (tests compare with numpy.fft) */
void orx_fft(float *re, float *im)
{
    float tr[NFFT], ti[NFFT];
    tables_init();
    fft_pos(re, im);
    for (int p = 0; p < NFFT; p++) { tr[T.bin_of_pos[p]] = re[p]; ti[T.bin_of_pos[p]] = im[p]; }
    memcpy(re, tr, sizeof tr); memcpy(im, ti, sizeof ti);
}

/* ------------------------------------------------------------------ stream */
typedef struct {
    int32_t start_cu, option, level, kbps;
    dab_profile_t prof;
    uint32_t *stepinfo;
    int nsteps, out_off;
} subch_t;

typedef struct {
    int fmt;                    /* 0 = u8, 1 = s16 */
    int64_t ring_len;           /* samples */
    uint8_t *ring;
    int64_t wr;                 /* samples pushed so far */
    int64_t pos;                /* estimated start of the next frame's null symbol */
    int32_t inc;                /* carrier offset, 2^-32 turn per sample */
    int32_t locked, bad;
    int32_t slope;              /* sampling-clock drift of the recording, samples per frame in Q16 (tracked) */
    int64_t cif;                /* CIFs demodulated since lock */
    int n_subch, msc_bytes;
    subch_t sub[64];
    int8_t *ti;                 /* [ti_slots][55296] time de-interleaver ring; a whole step of
                                   frames is demodulated before any of it is decoded, so the ring
                                   is deeper than the 16 CIFs of the interleaver: >= 15 + 4*frames */
    int ti_slots;               /* power of two */
    uint32_t fic_stepinfo[DAB_FIC_CW_IN + 6];
    float spectrum[NFFT];       /* |FFT|^2 of the last frame's PRS window, natural bin order */
    float null_spectrum[NFFT];  /* same for 2048 samples centred in the null symbol */
    int soft_extra;             /* 0: the product's six-bit soft decisions (+-31); 2: eight bits (+-127), for the sensitivity
                                   comparison of tests/test_msc_sensitivity.py only */
} orx_t;

typedef struct {                /* per-frame synchronisation record (same layout as dabx_sync_rec) */
    int64_t t_sym0;             /* first sample of the PRS FFT window */
    int32_t inc;
    int32_t flags;              /* bit0 frame ok, bit1 wide search used */
    int32_t peak_idx, m_int;
    float peak, total;
    int64_t cp_re, cp_im;
    int64_t e_null, e_sig;      /* sample energy over 2048 samples of the null symbol / of the PRS */
} orx_sync_t;

static void vit_init(void);
orx_t *orx_create(int fmt, int64_t ring_len, int ti_slots)
{
    tables_init();
    vit_init();
    orx_t *s = (orx_t *)calloc(1, sizeof *s);
    s->fmt = fmt; s->ring_len = ring_len;
    s->ring = (uint8_t *)calloc((size_t)ring_len, fmt ? 4 : 2);
    s->ti_slots = ti_slots;
    s->ti = (int8_t *)calloc((size_t)ti_slots, DAB_CIF_BITS);
    dab_profile_t fp; dab_profile_fic(&fp);
    dab_profile_stepinfo(&fp, s->fic_stepinfo);
    return s;
}
void orx_destroy(orx_t *s)
{
    if (!s) return;
    for (int i = 0; i < s->n_subch; i++) free(s->sub[i].stepinfo);
    free(s->ti); free(s->ring); free(s);
}
int orx_set_subch(orx_t *s, int n, const int32_t *cfg)
{
    for (int i = 0; i < s->n_subch; i++) { free(s->sub[i].stepinfo); s->sub[i].stepinfo = NULL; }
    s->n_subch = 0; s->msc_bytes = 0;
    if (n < 0 || n > 64) return -1;
    for (int i = 0; i < n; i++) {
        subch_t *u = &s->sub[i];
        u->start_cu = cfg[4 * i]; u->option = cfg[4 * i + 1]; u->level = cfg[4 * i + 2]; u->kbps = cfg[4 * i + 3];
        if (dab_profile_any(u->option, u->level, u->kbps, &u->prof)) return -2;
        if (u->start_cu < 0 || u->start_cu + u->prof.n_cu > DAB_NCU) return -3;
        u->stepinfo = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)(u->prof.n_in + 8));
        u->nsteps = dab_profile_stepinfo(&u->prof, u->stepinfo);
        u->out_off = s->msc_bytes;
        s->msc_bytes += u->prof.n_in / 8;
    }
    s->n_subch = n;
    return s->msc_bytes;
}
void orx_push(orx_t *s, const void *iq, int64_t n)
{
    const int bps = s->fmt ? 4 : 2;
    for (int64_t i = 0; i < n; i++) {
        int64_t w = (s->wr + i) % s->ring_len;
        memcpy(s->ring + w * bps, (const uint8_t *)iq + i * bps, (size_t)bps);
    }
    s->wr += n;
}
/* resident periodic ring (benchmarks): wr = 2^62 means "never underruns, never overruns", as dabx_set_write_pos */
void orx_set_wr(orx_t *s, int64_t wr) { s->wr = wr; }
/* test knob: soft decisions of 6 (the contract, default) or 8 bits */
void orx_set_soft_bits(orx_t *s, int bits) { s->soft_extra = bits == 8 ? 2 : 0; }
void orx_get_spectrum(const orx_t *s, float *out) { memcpy(out, s->spectrum, sizeof s->spectrum); }
void orx_get_null_spectrum(const orx_t *s, float *out) { memcpy(out, s->null_spectrum, sizeof s->null_spectrum); }
void orx_get_state(const orx_t *s, int64_t *st)
{
    st[0] = s->pos; st[1] = s->inc; st[2] = s->locked; st[3] = s->cif; st[4] = s->bad; st[5] = s->wr; st[6] = s->slope;
}

static inline void sample(const orx_t *s, int64_t n, int32_t *i, int32_t *q)
{
    int64_t w = n % s->ring_len;
    if (w < 0) w += s->ring_len;
    if (s->fmt) { const int16_t *p = (const int16_t *)s->ring + 2 * w; *i = p[0]; *q = p[1]; }
    else { const uint8_t *p = s->ring + 2 * w; *i = (int32_t)p[0] - 128; *q = (int32_t)p[1] - 128; }
}

/* -------- null search: 64-sample block energies, 41-block moving sum -------- */
#define NS_BLOCKS 3072
#define NS_WIN 41
static int null_search(const orx_t *s, int64_t from, int64_t *null_start)
{
    uint64_t E[NS_BLOCKS + NS_WIN];
    uint64_t tot = 0;
    for (int b = 0; b < NS_BLOCKS + NS_WIN; b++) {
        /* 64 x the variance of the block: a DC offset of the recording (RTL-SDR dongles have one) must not fill
         * the null symbol.  Exact integers: 64 sum |x|^2 - |sum x|^2 */
        uint64_t e = 0;
        int64_t si = 0, sq = 0;
        for (int n = 0; n < 64; n++) {
            int32_t i, q; sample(s, from + 64 * (int64_t)b + n, &i, &q);
            e += (uint64_t)((int64_t)i * i + (int64_t)q * q);
            si += i; sq += q;
        }
        E[b] = 64 * e - (uint64_t)(si * si) - (uint64_t)(sq * sq);
        if (b < NS_BLOCKS) tot += E[b];
    }
    uint64_t best = ~0ULL; int bb = 0;
    for (int b = 0; b < NS_BLOCKS; b++) {
        uint64_t m = 0;
        for (int j = 0; j < NS_WIN; j++) m += E[b + j];
        if (m < best) { best = m; bb = b; }
    }
    /* the quietest window must be at least 2.5 dB (a factor 9/16) below the average window: a null symbol still
     * shows at 0 dB SNR (-3 dB), the minimum of 3072 windows of pure noise stays within 0.3 dB of the average */
    if (!(best * 16 * NS_BLOCKS < tot * NS_WIN * 9)) { *null_start = from; return 0; }
    /* the null ends where two consecutive blocks rise above the midpoint between the null's level and the
     * average block energy; the frame starts one null length before that edge */
    const uint64_t mid = best * NS_BLOCKS + tot * NS_WIN;         /* x 2 NS_WIN NS_BLOCKS */
    int edge = bb + NS_WIN;
    for (int b = bb; b + 1 < NS_BLOCKS + NS_WIN; b++)
        if (E[b] * 2 * NS_WIN * NS_BLOCKS > mid && E[b + 1] * 2 * NS_WIN * NS_BLOCKS > mid) { edge = b; break; }
    *null_start = from + 64 * (int64_t)edge - DAB_TNULL;
    return 1;
}

/* window of 2048 samples starting at w0, de-rotated by -inc with phase 0 at
 * sample `ref`; thread t handles n = t + 256 j with a step-256 recurrence */
static void load_window(const orx_t *s, int64_t w0, int64_t ref, int32_t inc, float *re, float *im)
{
    uint32_t dth = (uint32_t)(-(int64_t)inc);
    float sr, si;
    nco(dth * 256u, &sr, &si);
    for (int t = 0; t < 256; t++) {
        float cr, ci;
        nco(dth * (uint32_t)(w0 - ref + t), &cr, &ci);
        for (int j = 0; j < 8; j++) {
            int32_t i, q; sample(s, w0 + t + 256 * j, &i, &q);
            if (j) cmul(cr, ci, sr, si, &cr, &ci);
            cmul((float)i, (float)q, cr, ci, &re[t + 256 * j], &im[t + 256 * j]);
        }
    }
}

static void sync_pass(const orx_t *s, int64_t pos_f, int32_t inc0, int wide, orx_sync_t *rec, float *spectrum, float *null_spectrum);

/* One frame: a pass at the predicted frame start and, when the PRS turns out to sit more than RESYNC_THR samples away
 * from where the window expected it (acquisition of a recording with a sampling-clock error, several frames per step),
 * a second pass at the corrected position, so that the guard-interval correlation looks at guard intervals. */
#define RESYNC_THR 32
#define RELOCK_TOL 512      /* a null symbol further than this from where the flywheel expects it is another frame phase */
static void sync_frame(const orx_t *s, int64_t pos_f, int32_t inc0, int wide, orx_sync_t *rec, float *spectrum, float *null_spectrum)
{
    sync_pass(s, pos_f, inc0, wide, rec, spectrum, null_spectrum);
    int64_t m = rec->t_sym0 - (pos_f + DAB_TNULL + DAB_TG - BACKOFF);
    if (m > RESYNC_THR || m < -RESYNC_THR) sync_pass(s, pos_f + m, inc0, wide, rec, spectrum, null_spectrum);
}

static void sync_pass(const orx_t *s, int64_t pos_f, int32_t inc0, int wide, orx_sync_t *rec, float *spectrum, float *null_spectrum)
{
    /* 1. guard-interval correlation over PRS + 3 FIC symbols (exact integers) */
    int64_t cre = 0, cim = 0;
    for (int sy = 0; sy < 4; sy++) {
        int64_t g = pos_f + DAB_TNULL + (int64_t)sy * DAB_TS;
        for (int n = 48; n < 456; n++) {
            int32_t i1, q1, i2, q2;
            sample(s, g + n, &i1, &q1); sample(s, g + n + DAB_TU, &i2, &q2);
            cre += (int64_t)i1 * i2 + (int64_t)q1 * q2;
            cim += (int64_t)q1 * i2 - (int64_t)i1 * q2;
        }
    }
    int64_t e_null = 0, e_sig = 0;
    for (int n = 0; n < DAB_TU; n++) {
        int32_t i, q;
        sample(s, pos_f + 128 + n, &i, &q); e_null += (int64_t)i * i + (int64_t)q * q;
        sample(s, pos_f + DAB_TNULL + DAB_TG + n, &i, &q); e_sig += (int64_t)i * i + (int64_t)q * q;
    }
    int32_t A = orx_cordic(cim, cre);
    int32_t inc_meas = (int32_t)((-(int64_t)A) >> 11);   /* |.| <= 2^20 */
    int32_t inc;
    if (wide) inc = inc_meas;
    else {
        int32_t d = (int32_t)(((uint32_t)(inc_meas - inc0) + (1u << 20)) & ((1u << 21) - 1)) - (1 << 20);
        inc = inc0 + d;
    }
    float xr[NFFT], xi[NFFT], hr[NFFT], hi[NFFT], part[256];
    int64_t w0 = pos_f + DAB_TNULL + DAB_TG - BACKOFF;
    int m_best = 0;
    load_window(s, w0, w0, inc, xr, xi);
    fft_pos(xr, xi);
    if (wide) {
        /* 2. integer carrier offset: differential PRS correlation in frequency */
        float best = -1.0f;
        for (int m = -MW; m <= MW; m++) {
            float pr[256], pi[256];
            for (int t = 0; t < 256; t++) {
                float ar = 0.0f, ai = 0.0f;
                for (int i = 0; i < 6; i++) {
                    int j = t + 256 * i;
                    if (j >= 1534) break;
                    int k = T.cfo_car[j], b1 = (k + m) & 2047, b0 = (k + m - 1) & 2047;
                    int p1 = T.pos_of_bin[b1], p0 = T.pos_of_bin[b0];
                    float dr, di, er, ei;
                    cmulc(xr[p1], xi[p1], xr[p0], xi[p0], &dr, &di);
                    rotq(dr, di, T.prsdq[k & 2047], &er, &ei);
                    ar = ar + er; ai = ai + ei;
                }
                pr[t] = ar; pi[t] = ai;
            }
            float cr = reduce256(pr), ci = reduce256(pi);
            float c0 = cr * cr, c1 = ci * ci, cm = c0 + c1;
            if (cm > best) { best = cm; m_best = m; }
        }
        inc = inc_meas + m_best * (1 << 21);
        load_window(s, w0, w0, inc, xr, xi);
        fft_pos(xr, xi);
    }
    if (spectrum)
        for (int p = 0; p < NFFT; p++) {
            float a = xr[p] * xr[p], b = xi[p] * xi[p];
            spectrum[T.bin_of_pos[p]] = a + b;
        }
    /* 3. fine timing: impulse response from the PRS */
    for (int p = 0; p < NFFT; p++) {
        int b = T.bin_of_pos[p];
        float rr = 0.0f, ri = 0.0f;
        if (T.prsq[b] >= 0) rotq(xr[p], xi[p], T.prsq[b], &rr, &ri);
        hr[b] = rr; hi[b] = -ri;                       /* conj(R) in natural order */
    }
    fft_pos(hr, hi);
    float peak = -1.0f; int pidx = 0;
    for (int t = 0; t < 256; t++) {
        float acc = 0.0f;
        for (int e = 0; e < 8; e++) {
            int p = 8 * t + e;
            float a = hr[p] * hr[p], b = hi[p] * hi[p], m2 = a + b;
            acc = acc + m2;
            int n = T.bin_of_pos[p];
            if (m2 > peak || (m2 == peak && n < pidx)) { peak = m2; pidx = n; }
        }
        part[t] = acc;
    }
    float total = reduce256(part);
    /* the FFT window follows the FIRST significant path, not the strongest: a weaker path that arrives earlier would
     * otherwise leak its next symbol into the window.  Among the taps up to EARLY_SPAN samples before the peak the
     * earliest one with at least EARLY_THR of the peak's power wins. */
    {
        float thr = peak * EARLY_THR;
        int dmax = 0;
        for (int p = 0; p < NFFT; p++) {
            int d = (pidx - T.bin_of_pos[p]) & 2047;
            if (d == 0 || d > EARLY_SPAN) continue;
            float a = hr[p] * hr[p], b = hi[p] * hi[p], m2 = a + b;
            if (m2 >= thr && d > dmax) dmax = d;
        }
        pidx = (pidx - dmax) & 2047;
    }
    int delta = pidx >= 1024 ? pidx - 2048 : pidx;
    rec->t_sym0 = w0 + delta - BACKOFF;
    rec->inc = inc;
    rec->flags = ((total > 0.0f && peak * 2048.0f >= LOCK_THR * total) ? 1 : 0) | (wide ? 2 : 0);   /* (silence is not a phase reference symbol) */
    rec->peak_idx = pidx; rec->m_int = m_best;
    rec->peak = peak; rec->total = total;
    rec->cp_re = cre; rec->cp_im = cim;
    rec->e_null = e_null; rec->e_sig = e_sig;
    if (null_spectrum) {
        int64_t n0 = pos_f + (DAB_TNULL - DAB_TU) / 2;
        load_window(s, n0, n0, inc, xr, xi);
        fft_pos(xr, xi);
        for (int p = 0; p < NFFT; p++) {
            float a = xr[p] * xr[p], b = xi[p] * xi[p];
            null_spectrum[T.bin_of_pos[p]] = a + b;
        }
    }
}

/* demodulate the 76 symbols of one frame: FIC soft bits to fic[9216], MSC soft
 * bits into the time de-interleaver ring slots (cif0 + c) & 15 */
static void demod_frame(orx_t *s, const orx_sync_t *rec, int64_t cif0, int8_t *fic, int32_t slope)
{
    float pr[NFFT], pi[NFFT], xr[NFFT], xi[NFFT], yr[NFFT], yi[NFFT], part[256];
    /* Sampling-clock offset: with the windows at their nominal spacing a recording whose clock is off by eps sees every
     * symbol eps * TS samples later in its window than the one before, i.e. a phase of -2 pi k eps TS / TU on carrier k
     * in the differential product (34 degrees at the band edge for 100 ppm).  From SCO_MIN (about 5 ppm) on the product
     * is turned back with the tracked drift: dth = slope * 319 / 768 is that phase per carrier in 2^-32 turns
     * (2^32 * TS / (TU * 65536 * TF) = 319 / 768), the factor of position 8 t + e is R[t] * S[e], carrier
     * = b0(t) + m(e) as the FFT leaves its bins (tables_init). */
    static const int m_e[8] = {0, 512, -1024, -512, 64, 576, -960, -448};
    const int comp = slope >= SCO_MIN || slope <= -SCO_MIN;
    const int32_t dth = (int32_t)(((int64_t)slope * 319) / 768);
    float rtr[256], rti[256], smr[8], smi[8];
    if (comp) {
        for (int t = 0; t < 256; t++) nco((uint32_t)(((t >> 5) + 8 * ((t >> 2) & 7) + 128 * (t & 3)) * dth), &rtr[t], &rti[t]);
        for (int e = 0; e < 8; e++) nco((uint32_t)(m_e[e] * dth), &smr[e], &smi[e]);
    }
    /* The symbol windows of a frame keep their nominal spacing: moving a window by a whole sample between two symbols
     * would put a phase step of 2 pi k / 2048 into the differential product.  A sampling-clock error of 100 ppm moves
     * the last symbol by 20 samples against the first, which the BACKOFF of the window inside the guard absorbs. */
    for (int l = 0; l < DAB_NSYM; l++) {
        load_window(s, rec->t_sym0 + (int64_t)l * DAB_TS, rec->t_sym0, rec->inc, xr, xi);
        fft_pos(xr, xi);
        if (l > 0) {
            for (int t = 0; t < 256; t++) {
                float acc = 0.0f;
                for (int e = 0; e < 8; e++) {
                    int p = 8 * t + e;
                    if (T.n_of_bin[T.bin_of_pos[p]] < 0) continue;
                    cmulc(xr[p], xi[p], pr[p], pi[p], &yr[p], &yi[p]);
                    if (comp) {
                        cmul(yr[p], yi[p], smr[e], smi[e], &yr[p], &yi[p]);
                        cmul(yr[p], yi[p], rtr[t], rti[t], &yr[p], &yi[p]);
                    }
                    float a = fabsf(yr[p]) + fabsf(yi[p]);
                    acc = acc + a;
                }
                part[t] = acc;
            }
            float S = reduce256(part), g = 0.0f;
            if (S > 0.0f && S < INFINITY) { int E; frexpf(S, &E); g = ldexpf(1.0f, SOFT_EXP + s->soft_extra - E); }
            const float smax = s->soft_extra ? 127.0f : SOFT_MAX;
            int8_t *dst;
            if (l <= DAB_FIC_SYMS) dst = fic + (l - 1) * DAB_SYM_BITS;
            else dst = s->ti + (size_t)((cif0 + (l - 4) / DAB_CIF_SYMS) & (s->ti_slots - 1)) * DAB_CIF_BITS + ((l - 4) % DAB_CIF_SYMS) * DAB_SYM_BITS;
            for (int p = 0; p < NFFT; p++) {
                int n = T.n_of_bin[T.bin_of_pos[p]];
                if (n < 0) continue;
                float a = rintf(yr[p] * g), b = rintf(yi[p] * g);
                a = a > smax ? smax : (a < -smax ? -smax : a);
                b = b > smax ? smax : (b < -smax ? -smax : b);
                dst[n] = (int8_t)a; dst[n + DAB_K] = (int8_t)b;
            }
        }
        memcpy(pr, xr, sizeof pr); memcpy(pi, xi, sizeof pi);
    }
}

/* ------------------------------------------------------------------ Viterbi
 * x4: 4 soft values per step (0 where punctured), positive = bit 0.
 * States: bit 5 newest … bit 0 oldest.  Metric = correlation, maximised.
 * Tie rule: keep the predecessor whose oldest bit equals the new input bit.
 *
 * The textbook recursion, organised so that the scalar code is a fair CPU baseline
 * (bench.py cpu_baseline): the encoder outputs of the two predecessors of every state are
 * tabulated once, the 16 branch metrics of a step are built from 8 sums and their negations,
 * and the scratch buffers are per-thread and grow-only (no malloc per codeword). */
static uint8_t vit_c[32];                       /* encoder output of the transition 2j -> j (input bit 0) */
static int vit_ready;
static void vit_init(void)
{
    if (vit_ready) return;
    for (int j = 0; j < 32; j++) {
        /* butterfly j: predecessors 2j, 2j+1 -> successors j (input 0) and j+32 (input 1).  All four
         * generators tap the newest and the oldest register bit, so the four branch words are c, ~c, ~c, c */
        int c = dab_conv_output(2 * j, 0);
        if (dab_conv_output(2 * j + 1, 0) != (c ^ 15) || dab_conv_output(2 * j, 1) != (c ^ 15) || dab_conv_output(2 * j + 1, 1) != c) abort();
        vit_c[j] = (uint8_t)c;
    }
    __atomic_store_n(&vit_ready, 1, __ATOMIC_RELEASE);
}
static __thread uint64_t *vit_dec;
static __thread int vit_dec_cap;
static __thread int8_t *cw_x4;
static __thread uint8_t *cw_bits;
static __thread int cw_cap;

void orx_viterbi(const int8_t *x4, int nsteps, uint8_t *bits)
{
    int32_t pm[64], nm[64], m[32];
    uint8_t tk[64];
    vit_init();
    if (nsteps > vit_dec_cap) {
        free(vit_dec);
        vit_dec = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)nsteps);
        vit_dec_cap = nsteps;
    }
    uint64_t *dec = vit_dec;
    for (int s = 0; s < 64; s++) pm[s] = PM_INIT;
    pm[0] = 0;
    for (int t = 0; t < nsteps; t++) {
        const int8_t *x = x4 + 4 * t;
        int32_t bm[16];
        for (int c = 0; c < 8; c++) {           /* bm[c] = sum over j of (bit j of c set ? -x[j] : x[j]); bm[15 - c] = -bm[c] */
            int32_t v = x[0] + ((c & 4) ? -x[1] : x[1]) + ((c & 2) ? -x[2] : x[2]) + ((c & 1) ? -x[3] : x[3]);
            bm[c] = v; bm[15 - c] = -v;
        }
        for (int j = 0; j < 32; j++) m[j] = bm[vit_c[j]];
        for (int j = 0; j < 32; j++) {
            /* successor j (input 0): keeps 2j, receives 2j+1; successor j+32 (input 1): keeps 2j+1, receives 2j */
            int32_t a = pm[2 * j], b = pm[2 * j + 1], mj = m[j];
            int32_t k0 = a + mj, r0 = b - mj, k1 = b + mj, r1 = a - mj;
            tk[j] = (uint8_t)(r0 > k0); nm[j] = r0 > k0 ? r0 : k0;
            tk[j + 32] = (uint8_t)(r1 > k1); nm[j + 32] = r1 > k1 ? r1 : k1;
        }
        memcpy(pm, nm, sizeof pm);
        uint64_t d = 0;
        for (int g = 0; g < 8; g++) {           /* eight 0/1 bytes -> eight bits */
            uint64_t w;
            memcpy(&w, tk + 8 * g, 8);
            d |= ((w * 0x0102040810204080ULL) >> 56) << (8 * g);
        }
        dec[t] = d;
    }
    int st = 0;                                         /* terminated trellis */
    for (int t = nsteps - 1; t >= 0; t--) {
        int u = st >> 5;
        bits[t] = (uint8_t)u;
        int own = ((st << 1) & 63) | u;
        st = ((dec[t] >> st) & 1) ? (own ^ 1) : own;
    }
}

/* depuncture one codeword through a caller-supplied soft-bit getter */
typedef int8_t (*soft_fn)(const void *ctx, int i);
static void decode_cw(const uint32_t *info, int nsteps, int n_in, soft_fn get, const void *ctx, uint8_t *out_bytes)
{
    if (nsteps > cw_cap) {
        free(cw_x4); free(cw_bits);
        cw_x4 = (int8_t *)malloc((size_t)nsteps * 4);
        cw_bits = (uint8_t *)malloc((size_t)nsteps);
        cw_cap = nsteps;
    }
    int8_t *x4 = cw_x4;
    uint8_t *bits = cw_bits;
    memset(x4, 0, (size_t)nsteps * 4);
    for (int t = 0; t < nsteps; t++) {
        int off = (int)(info[t] >> 4), k = 0;
        for (int j = 0; j < 4; j++)
            if (info[t] & (8u >> j)) x4[4 * t + j] = get(ctx, off + k++);
    }
    orx_viterbi(x4, nsteps, bits);
    for (int i = 0; i < n_in / 8; i++) {
        unsigned v = 0;
        for (int b = 0; b < 8; b++) v = (v << 1) | (unsigned)(bits[8 * i + b] ^ T.prbs[8 * i + b]);
        out_bytes[i] = (uint8_t)v;
    }
}

typedef struct { const int8_t *p; } lin_ctx;
static int8_t get_lin(const void *c, int i) { return ((const lin_ctx *)c)->p[i]; }
typedef struct { const int8_t *ti; int64_t r; int base; int mask; } ti_ctx;
static int8_t get_ti(const void *c, int i)
{
    const ti_ctx *x = (const ti_ctx *)c;
    return x->ti[(size_t)((x->r + dab_ti_delay(i)) & x->mask) * DAB_CIF_BITS + x->base + i];
}

/* stage-level entry for tests: decode a punctured codeword given linear soft bits.
 * kind 0: FIC codeword; otherwise EEP (option, level, kbps) */
int orx_decode_linear(int kind, int option, int level, int kbps, const int8_t *soft, uint8_t *out_bytes)
{
    tables_init();
    dab_profile_t p;
    if (kind == 0) dab_profile_fic(&p); else if (dab_profile_any(option, level, kbps, &p)) return -1;
    uint32_t *info = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)(p.n_in + 8));
    int ns = dab_profile_stepinfo(&p, info);
    lin_ctx c = {soft};
    decode_cw(info, ns, p.n_in, get_lin, &c, out_bytes);
    free(info);
    return p.n_in / 8;
}

/*
 * Process n_frames frames.  Output arrays (any may be NULL):
 *   sync      [n_frames]                    synchronisation records
 *   fic_soft  [n_frames][9216]              int8
 *   msc_soft  [n_frames][4][55296]          int8, the CIFs demodulated in this call
 *   fib       [n_frames][12][32], fib_ok [n_frames][12]
 *   msc       [n_frames][4][msc_bytes], msc_valid [n_frames][4]
 * Returns number of frames processed, or <0 when the ring does not hold enough samples.
 */
int orx_process(orx_t *s, int n_frames, orx_sync_t *sync, int8_t *fic_soft, int8_t *msc_soft,
                uint8_t *fib, uint8_t *fib_ok, uint8_t *msc, uint8_t *msc_valid)
{
    int wide = !s->locked;
    /* a locked receiver whose last frame came without a phase reference symbol looks for the null symbol again before it
     * goes on (one more frame of samples, like an acquisition): a recording that loops or was cut jumps to another frame
     * phase, and waiting for four bad frames before searching would lose them all (the reference loses one frame at the
     * wrap of a looped file: SURVEY.md App. A.3) */
    const int suspect = s->locked && s->bad > 0;
    int64_t need = s->pos + (int64_t)(n_frames + ((wide || suspect) ? 1 : 0)) * DAB_TF + 4096;
    if (need > s->wr) return -1;
    if (15 + 4 * n_frames > s->ti_slots) return -3;
    if (s->wr < ((int64_t)1 << 62) && s->wr - s->pos > s->ring_len) return -2;
    if (suspect) {
        int64_t ns;
        if (null_search(s, s->pos, &ns)) {
            /* where the null symbol is against where the flywheel expects it, to the nearest whole frame */
            int64_t d = (ns - s->pos) % DAB_TF;
            if (d < 0) d += DAB_TF;
            if (d >= DAB_TF / 2) d -= DAB_TF;
            if (d > RELOCK_TOL || d < -RELOCK_TOL) {        /* another frame phase: a new acquisition from there */
                s->pos = ns; s->cif = 0; s->locked = 0;
                wide = 1;
            }
        }
    } else if (wide) {
        int64_t ns;
        if (!null_search(s, s->pos, &ns)) {
            s->pos += (int64_t)n_frames * DAB_TF;
            if (sync) memset(sync, 0, sizeof(orx_sync_t) * (size_t)n_frames);
            if (fib_ok) memset(fib_ok, 0, (size_t)n_frames * 12);
            if (msc_valid) memset(msc_valid, 0, (size_t)n_frames * 4);
            return 0;
        }
        s->pos = ns; s->cif = 0;
    }
    int8_t *ficbuf = (int8_t *)malloc(DAB_FIC_BITS);
    orx_sync_t rec;
    int nbad = s->bad;
    if (wide) s->slope = 0;
    const int64_t pos0 = s->pos;
    int64_t e_first = 0, e_last = 0;
    int f_first = -1, f_last = -1;
    for (int f = 0; f < n_frames; f++) {
        /* frame f is expected where the tracked sampling-clock drift puts it */
        const int64_t pos_f = pos0 + (int64_t)f * DAB_TF + (((int64_t)f * s->slope) >> 16);
        sync_frame(s, pos_f, s->inc, wide, &rec, f == n_frames - 1 ? s->spectrum : NULL, f == n_frames - 1 ? s->null_spectrum : NULL);
        if (sync) sync[f] = rec;
        int64_t cif0 = s->cif + 4 * (int64_t)f;
        demod_frame(s, &rec, cif0, ficbuf, s->slope);
        if (fic_soft) memcpy(fic_soft + (size_t)f * DAB_FIC_BITS, ficbuf, DAB_FIC_BITS);
        if (msc_soft)
            for (int c = 0; c < 4; c++)
                memcpy(msc_soft + ((size_t)f * 4 + c) * DAB_CIF_BITS, s->ti + (size_t)((cif0 + c) & (s->ti_slots - 1)) * DAB_CIF_BITS, DAB_CIF_BITS);
        for (int cw = 0; cw < DAB_FIC_CW; cw++) {
            uint8_t out[96];
            lin_ctx lc = {ficbuf + cw * DAB_FIC_CW_BITS};
            decode_cw(s->fic_stepinfo, DAB_FIC_CW_IN + 6, DAB_FIC_CW_IN, get_lin, &lc, out);
            for (int j = 0; j < 3; j++) {
                const uint8_t *fb = out + 32 * j;
                int ok = dab_crc16(fb, 30) == (uint16_t)((fb[30] << 8) | fb[31]);
                if (fib) memcpy(fib + ((size_t)f * 12 + 3 * cw + j) * 32, fb, 32);
                if (fib_ok) fib_ok[(size_t)f * 12 + 3 * cw + j] = (uint8_t)ok;
            }
        }
        for (int c = 0; c < 4; c++) {
            int64_t r = cif0 + c - 15;                  /* logical frame completed by this CIF */
            if (msc_valid) msc_valid[(size_t)f * 4 + c] = (uint8_t)(r >= 0);
            if (!msc) continue;
            uint8_t *o = msc + ((size_t)f * 4 + c) * (size_t)s->msc_bytes;
            if (r < 0) { memset(o, 0, (size_t)s->msc_bytes); continue; }
            for (int k = 0; k < s->n_subch; k++) {
                subch_t *u = &s->sub[k];
                ti_ctx tc = {s->ti, r, u->start_cu * DAB_CU_BITS, s->ti_slots - 1};
                decode_cw(u->stepinfo, u->nsteps, u->prof.n_in, get_ti, &tc, o + u->out_off);
            }
        }
        nbad = (rec.flags & 1) ? 0 : nbad + 1;
        if (rec.flags & 1) {                                /* timing error of a good frame against its prediction */
            int64_t e = rec.t_sym0 + BACKOFF - DAB_TG - DAB_TNULL - pos_f;
            if (f_first < 0) { f_first = f; e_first = e; }
            f_last = f; e_last = e;
        }
        if (f == n_frames - 1) {
            /* sampling-clock tracker (first order, gain 1/4).  Tracking: the error of frame f has built up over f + 1
             * frames since the last measured frame start.  Acquisition: the drift between two good frames of the step. */
            int32_t sl = s->slope;
            if (!wide && f_last >= 0) sl += (int32_t)(((e_last * 65536) / (f_last + 1)) / 4);
            else if (wide && f_last > f_first) sl = (int32_t)(((e_last - e_first) * 65536) / (f_last - f_first));
            if (sl > SLOPE_MAX) sl = SLOPE_MAX;
            if (sl < -SLOPE_MAX) sl = -SLOPE_MAX;
            s->slope = sl;
            s->pos = rec.t_sym0 + BACKOFF - DAB_TG - DAB_TNULL + DAB_TF + (sl >> 16);
            s->inc = rec.inc;
        }
    }
    s->cif += 4 * (int64_t)n_frames;
    s->bad = nbad;
    /* acquisition succeeds only if its last frame was good; lock drops after 4 bad frames in a row */
    s->locked = wide ? (nbad == 0) : (nbad < 4);
    free(ficbuf);
    return n_frames;
}
