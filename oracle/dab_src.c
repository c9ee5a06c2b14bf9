/*
 * oracle/dab_src.c — TEST INFRASTRUCTURE ONLY (checker, never the product).
 *
 * CPU restatement of the reference's sample-rate converters in front of the dabsdr input FIFO
 * (reference: src/input/inputdevicesrc.h:78-150, src/input/inputdevicesrc.cpp:33-47, 109-200, 233-316), the
 * pre-stage that SDR devices delivering 4.096 Msps (Airspy, src/input/airspyinput.cpp:278) or another rate
 * (SoapySDR, src/input/soapysdrinput.cpp:665) run before the PHY chain.  This is the one row of the hot path
 * whose reference arithmetic exists as source, so every function below follows the cited lines operation by
 * operation (same float operations in the same order; the reference file itself cannot be compiled here: it
 * includes <QDebug>).  Written as index arithmetic on a linear history instead of the reference's doubled
 * circular buffer — the values read and the order they are combined in are the same.
 *
 * Arithmetic contract: one IEEE binary32 operation per C operator, no contraction (-ffp-contract=off).  The
 * reference's own builds may contract a*b+c into an FMA on some platforms; tests state this tolerance.
 */
#include <math.h>
#include <stdint.h>
#include <string.h>

/* inputdevicesrc.h:39-40 */
#define LEVEL_ATTACK 5e-5
#define LEVEL_RELEASE 5e-2

/* ------------------------------------------------------------------ half-band decimator 4096 kHz -> 2048 kHz
 * inputdevicesrc.h:96-109: order 42 (43 taps), every other coefficient zero, symmetric; the 12 distinct values: */
#define DS2_TAPS 43
static const float ds2_coef[12] = {
    0.000223158782894952853123604619156594708f,  -0.00070774549637065342286290636764078954f,  0.001735782601167994458266075064045708132f,
    -0.003619832275410410967614316390950079949f, 0.006788741778432844271862212082169207861f,  -0.01183550169261274320753329902800032869f,
    0.019680477383812611941182879604639310855f,  -0.032073581325677551212560700832909788005f, 0.053382280107447499517547839786857366562f,
    -0.099631117404426483563639749263529665768f, 0.316099577146216947909351802081800997257f,  0.5f};

typedef struct {
    float hi[DS2_TAPS - 1], hq[DS2_TAPS - 1];   /* the 42 input samples before the current one, oldest first */
    float level, catt, crel;
} osrc_ds2_t;

void osrc_ds2_reset(osrc_ds2_t *s)
{   /* inputdevicesrc.cpp:80-81 (constants), :94-105 (reset) */
    memset(s, 0, sizeof *s);
    s->catt = (float)(1 - exp(-1 / (LEVEL_ATTACK * 2048e3)));
    s->crel = (float)(1 - exp(-1 / (LEVEL_RELEASE * 2048e3)));
}

/* in: n_in floats = n_in / 2 complex samples at 4096 kHz; out: n_in / 2 floats = n_in / 4 complex samples.
 * Returns the number of complex output samples... the reference returns numInDataIQ / 2 with numInDataIQ counting
 * complex INPUT samples (inputdevicesrc.cpp:109-200: the loop runs numInDataIQ / 2 times and consumes two complex
 * samples per pass).  Here n_cplx = number of complex input samples (even). */
int osrc_ds2_process(osrc_ds2_t *s, const float *in, int n_cplx, float *out)
{
    float level = s->level;
    for (int n = 0; n < n_cplx / 2; n++) {
        /* history x[-42..-1] followed by the new even sample x[0] (inputdevicesrc.cpp:118-126) */
        const float xi0 = in[4 * n], xq0 = in[4 * n + 1];
        float accI = 0, accQ = 0;
        for (int c = 0; c < (DS2_TAPS + 1) / 4; c++) {     /* :131-139: (x[-42 + 2c] + x[-2c]) * coef[c], c = 0..10 */
            const float oi = s->hi[2 * c], oq = s->hq[2 * c];                            /* x[-42 + 2c] */
            const float ni = c ? s->hi[DS2_TAPS - 1 - 2 * c] : xi0, nq = c ? s->hq[DS2_TAPS - 1 - 2 * c] : xq0;   /* x[-2c] */
            float t = oi + ni;
            t = t * ds2_coef[c];
            accI = accI + t;
            t = oq + nq;
            t = t * ds2_coef[c];
            accQ = accQ + t;
        }
        {   /* :141-142: centre tap, x[-21] * 0.5 */
            float t = s->hi[21] * ds2_coef[11];
            accI = accI + t;
            t = s->hq[21] * ds2_coef[11];
            accQ = accQ + t;
        }
        out[2 * n] = accI; out[2 * n + 1] = accQ;                                         /* :144-145 */
        /* the odd sample only enters the delay line (:154-165) and drives the level detector (:167-173) */
        const float xi1 = in[4 * n + 2], xq1 = in[4 * n + 3];
        float abs2 = xi1 * xi1;
        float q2 = xq1 * xq1;
        abs2 = abs2 + q2;
        const float c = abs2 > level ? s->catt : s->crel;
        {
            float a = c * abs2, b = c * level;
            a = a + level;
            level = a - b;                                                                /* c * abs2 + level - c * level */
        }
        memmove(s->hi, s->hi + 2, sizeof(float) * (DS2_TAPS - 3));
        memmove(s->hq, s->hq + 2, sizeof(float) * (DS2_TAPS - 3));
        s->hi[DS2_TAPS - 3] = xi0; s->hq[DS2_TAPS - 3] = xq0;
        s->hi[DS2_TAPS - 2] = xi1; s->hq[DS2_TAPS - 2] = xq1;
    }
    s->level = level;                                                                     /* :196-197 */
    return n_cplx / 2;
}
float osrc_ds2_level(const osrc_ds2_t *s) { return s->level; }

/* ------------------------------------------------------------------ transposed Farrow resampler, any rate -> 2048 kHz
 * inputdevicesrc.h:142-149: 6 polynomials of 4 coefficients */
#define FW_M 4
#define FW_N 6
static const float fw_coef[FW_N][FW_M] = {
    {0.001667349914006070960362f, 0.032712194697834547085780f, -0.146457831613232558609639f, 0.004040531324696360060411f},
    {-0.103347648141097675500433f, -0.244367915078825215235980f, 0.233146907266815583970043f, 0.243745693669456003904727f},
    {0.123959393981824803065983f, 0.873574620095563081356715f, 0.586104518066954516264389f, -0.711183949124104208827646f},
    {0.873450879200179830519346f, 0.039931348783534291457809f, -1.514110581690161660972649f, 0.711183949124100878158572f},
    {0.114518381640217964401174f, -0.923204505507555395205088f, 0.952958408884428287421997f, -0.243745693669455892882425f},
    {-0.104194288733202022889657f, 0.243880907754789599817258f, -0.134525637544989168370435f, -0.004040531324696002707375f},
};

typedef struct {
    float mu, R;
    float xi[FW_M], xq[FW_M];
    float yi[FW_N], yq[FW_N];
    float level, catt, crel;
} osrc_farrow_t;

void osrc_farrow_reset(osrc_farrow_t *s, float in_rate)
{   /* inputdevicesrc.cpp:206-231 */
    memset(s, 0, sizeof *s);
    s->R = (float)(2048e3 / in_rate);
    s->catt = (float)(1 - exp(-1 / (LEVEL_ATTACK * in_rate)));
    s->crel = (float)(1 - exp(-1 / (LEVEL_RELEASE * in_rate)));
}

/* in: n_cplx complex samples; returns the number of complex output samples written (inputdevicesrc.cpp:233-316).
 * mu_out / dump_out (optional, n_cplx entries each): the value of mu every input sample is integrated with and
 * whether an output was dumped just before it — the data-independent schedule the GPU kernel is given. */
int osrc_farrow_process(osrc_farrow_t *s, const float *in, int n_cplx, float *out, float *mu_out, uint8_t *dump_out)
{
    float level = s->level;
    int n_out = 0;
    for (int k = 0; k < n_cplx; k++) {
        s->mu = s->mu - s->R;                                    /* :241 */
        int dump = 0;
        if (s->mu < 0) {                                         /* :242 dump condition */
            dump = 1;
            s->mu = s->mu + 1.0f;
            for (int n = 0; n < FW_N; n++) {                     /* :246-257 */
                float accI = 0, accQ = 0;
                for (int m = 0; m < FW_M; m++) {
                    float t = s->xi[m] * fw_coef[n][m];
                    accI = accI + t;
                    t = s->xq[m] * fw_coef[n][m];
                    accQ = accQ + t;
                }
                s->yi[n] = s->yi[n] + accI;
                s->yq[n] = s->yq[n] + accQ;
            }
            out[2 * n_out] = s->R * s->yi[0];                    /* :258-260 */
            out[2 * n_out + 1] = s->R * s->yq[0];
            n_out++;
            memmove(s->yi, s->yi + 1, sizeof(float) * (FW_N - 1));   /* :262-266 */
            memmove(s->yq, s->yq + 1, sizeof(float) * (FW_N - 1));
            s->yi[FW_N - 1] = 0; s->yq[FW_N - 1] = 0;
            memset(s->xi, 0, sizeof s->xi); memset(s->xq, 0, sizeof s->xq);   /* :268-273 */
        }
        float inI = in[2 * k], inQ = in[2 * k + 1];
        if (mu_out) mu_out[k] = s->mu;
        if (dump_out) dump_out[k] = (uint8_t)dump;
        {   /* :282-292 level detector */
            float abs2 = inI * inI, q2 = inQ * inQ;
            abs2 = abs2 + q2;
            const float c = abs2 > level ? s->catt : s->crel;
            float a = c * abs2, b = c * level;
            a = a + level;
            level = a - b;
        }
        s->xi[0] = s->xi[0] + inI;                               /* :295-296 */
        s->xq[0] = s->xq[0] + inQ;
        for (int m = 1; m < FW_M; m++) {                         /* :298-305: in * mu^m, accumulated */
            inI = inI * s->mu;
            inQ = inQ * s->mu;
            s->xi[m] = s->xi[m] + inI;
            s->xq[m] = s->xq[m] + inQ;
        }
    }
    s->level = level;
    return n_out;
}
float osrc_farrow_level(const osrc_farrow_t *s) { return s->level; }
int osrc_sizeof_ds2(void) { return (int)sizeof(osrc_ds2_t); }
int osrc_sizeof_farrow(void) { return (int)sizeof(osrc_farrow_t); }

/* ------------------------------------------------------------------ pass-through (2048 kHz in): inputdevicesrc.cpp:314-345
 * copies the samples and runs the level detector on every one of them (:330-341), constants of :316-317 */
float osrc_passthrough_level(float level, const float *in, int n_cplx)
{
    const float catt = (float)(1 - exp(-1 / (LEVEL_ATTACK * 2048e3))), crel = (float)(1 - exp(-1 / (LEVEL_RELEASE * 2048e3)));
    for (int n = 0; n < n_cplx; n++) {
        float abs2 = in[2 * n] * in[2 * n];
        const float q2 = in[2 * n + 1] * in[2 * n + 1];
        abs2 = abs2 + q2;
        const float c = abs2 > level ? catt : crel;
        float a = c * abs2;
        const float b = c * level;
        a = a + level;
        level = a - b;
    }
    return level;
}
