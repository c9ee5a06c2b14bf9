// dabx_kernels.hip — CDNA4 (gfx950) kernels of the DAB Mode-I PHY decode chain.
//
// Replaces the sample-by-sample worker loop inside the reference's closed
// dabsdr library (reference: lib/linux_x86_64/dabsdr.h:397 `dabsdr()`;
// observed stage order in SURVEY.md §3.3) with four batched stages over
// (stream, frame):
//   k_null_search  frame acquisition on raw sample energy            (integer)
//   k_sync         guard-interval CFO + PRS impulse-response timing  (integer + f32)
//   k_demod        2048-point FFT, pi/4-DQPSK demap, frequency and time de-interleave -> int8
//   k_viterbi      depuncturing gather, K=7 Viterbi, de-dispersal
//   k_finish       FIB CRC-16 and per-stream tracking state
// No MFMA GEMM anywhere (nothing here is a contraction): the FFT is bound by vector issue and its LDS exchanges, the Viterbi by VALU
// issue.  The matrix core appears once, as a sign-combination engine: six v_mfma_i32_4x4x4_16b_i8 per 24 trellis steps
// form the branch metrics of k_viterbi so that the vector ALU is left with three instructions per step.
//
// Arithmetic contract: every float operation below is one IEEE
// binary32 operation in a fixed order — a fused multiply-add only where it is
// written out (the complex products); the file is compiled with
// -ffp-contract=off so results equal the CPU checker bit for bit.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "dabx_dev.h"

namespace {

constexpr int TF = 196608, TNULL = 2656, TS = 2552, TU = 2048, TG = 504;
constexpr int NSYM = 76, SYMBITS = 3072, NCAR = 1536;
constexpr int FICBITS = 9216, CIFBITS = 55296;
constexpr int BACKOFF = 24, CFO_RANGE = 16, SOFT_EXP = 15, PM_INIT = -1000000;
constexpr float SOFT_MAX = 31.0f;          // soft bits are limited to +-31: twice the sum of two fits a byte (k_viterbi packs 2 (x0 + x3))
constexpr float LOCK_THR = 48.0f;
constexpr int EARLY_SPAN = 400;            // the first path may lead the strongest one by up to this many samples ...
constexpr float EARLY_THR = 0.125f;        // ... if it carries at least this fraction of its power (-9 dB)
constexpr int RESYNC_THR = 32;             // PRS further than this from where the window expected it: second pass
constexpr int RELOCK_TOL = 512;            // a null symbol further than this from where the flywheel expects it is another frame phase
constexpr int SLOPE_MAX = 60 << 16;        // sampling-clock tracker: |drift| <= 60 samples per frame (~300 ppm), Q16
constexpr int SCO_MIN = 1 << 16;           // de-rotate the differential product from this drift on: 1 sample per frame = 5.1 ppm
constexpr int TI_SEG = CIFBITS / 16;   // bytes per residue class in a residue-major MSC row

// A complex value is one pair of adjacent vector registers, so that gfx950's packed binary32 instructions (v_pk_add_f32,
// v_pk_mul_f32, v_pk_fma_f32: two independent IEEE operations per issue slot) do both components at once.  The arithmetic
// contract is untouched: every helper below performs exactly the operations, in exactly the roles, of the scalar formulas it
// replaces (oracle/dab_rx.c has them written out) — the packed forms only choose, with their op_sel / neg modifiers, which
// half of a pair feeds which half of the result.  k_demod is bound by vector-ALU issue (profiles/r03*): written with scalar
// components the compiler spent 23 % of the symbol loop on moves and sign flips to line operands up; 509 -> 3xx instructions.
typedef float cf __attribute__((ext_vector_type(2)));          // .x = real part, .y = imaginary part

// complex products: one rounded product, then one fused multiply-add per component —
// the same two operations, in the same roles, in the CPU checker (oracle/dab_rx.c: fmaf)
__device__ __forceinline__ cf cmul(cf a, cf b)
{
    cf p, d;      // p = (a.i b.i, a.i b.r);  d = (fma(a.r, b.r, -p.x), fma(a.r, b.i, p.y))
    // (one asm statement: between two of them the compiler, which cannot see what they hold, puts a wait state)
    asm("v_pk_mul_f32 %1, %2, %3 op_sel:[1,1] op_sel_hi:[1,0]\n\t"
        "v_pk_fma_f32 %0, %2, %3, %1 op_sel:[0,0,0] op_sel_hi:[0,1,1] neg_lo:[0,0,1]" : "=v"(d), "=&v"(p) : "v"(a), "v"(b));
    return d;
}
__device__ __forceinline__ cf cmulc(cf a, cf b)   // a * conj(b)
{
    cf p, d;      // p = (a.i b.i, a.r b.i);  d = (fma(a.r, b.r, p.x), fma(a.i, b.r, -p.y))
    asm("v_pk_mul_f32 %1, %2, %3 op_sel:[1,1] op_sel_hi:[0,1]\n\t"
        "v_pk_fma_f32 %0, %2, %3, %1 op_sel:[0,0,0] op_sel_hi:[1,0,1] neg_hi:[0,0,1]" : "=v"(d), "=&v"(p) : "v"(a), "v"(b));
    return d;
}
// a + (-j) b = (a.r + b.i, a.i - b.r) and a - (-j) b = (a.r - b.i, a.i + b.r): one packed add each, the halves of b swapped by
// op_sel and one of them negated (x + (-y) is x - y in IEEE arithmetic)
__device__ __forceinline__ cf add_mj(cf a, cf b) { cf d; asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(d) : "v"(a), "v"(b)); return d; }
__device__ __forceinline__ cf sub_mj(cf a, cf b) { cf d; asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(d) : "v"(a), "v"(b)); return d; }
#include "dabx_cplx.inc"
// v[1..7] *= w[0..6] (the twiddles of one FFT pass) as one asm statement
__device__ __forceinline__ void cmul7(cf v[8], const cf w[7])
{
    cf p, q;
    asm(DABX_CMUL7_INPLACE
        : [v0] "+v"(v[1]), [v1] "+v"(v[2]), [v2] "+v"(v[3]), [v3] "+v"(v[4]), [v4] "+v"(v[5]), [v5] "+v"(v[6]), [v6] "+v"(v[7]), [p] "=&v"(p), [q] "=&v"(q)
        : [w0] "v"(w[0]), [w1] "v"(w[1]), [w2] "v"(w[2]), [w3] "v"(w[3]), [w4] "v"(w[4]), [w5] "v"(w[5]), [w6] "v"(w[6]));
}
// y[e] = a[e] conj(b[e]), e < 8
__device__ __forceinline__ void cmulc8(cf y[8], const cf a[8], const cf b[8])
{
    cf p, q;
    asm(DABX_CMULC8
        : [y0] "=&v"(y[0]), [y1] "=&v"(y[1]), [y2] "=&v"(y[2]), [y3] "=&v"(y[3]), [y4] "=&v"(y[4]), [y5] "=&v"(y[5]), [y6] "=&v"(y[6]), [y7] "=&v"(y[7]),
          [p] "=&v"(p), [q] "=&v"(q)
        : [a0] "v"(a[0]), [a1] "v"(a[1]), [a2] "v"(a[2]), [a3] "v"(a[3]), [a4] "v"(a[4]), [a5] "v"(a[5]), [a6] "v"(a[6]), [a7] "v"(a[7]),
          [b0] "v"(b[0]), [b1] "v"(b[1]), [b2] "v"(b[2]), [b3] "v"(b[3]), [b4] "v"(b[4]), [b5] "v"(b[5]), [b6] "v"(b[6]), [b7] "v"(b[7]));
}
// x[j] = x[j] rot_j with rot_0 = rot, rot_j = rot_{j-1} step (the window's de-rotation: a recurrence over the eight samples of a thread)
__device__ __forceinline__ void derotate8(cf x[8], cf rot, cf step)
{
    cf p, q;
    asm(DABX_DEROTATE8
        : [x0] "+v"(x[0]), [x1] "+v"(x[1]), [x2] "+v"(x[2]), [x3] "+v"(x[3]), [x4] "+v"(x[4]), [x5] "+v"(x[5]), [x6] "+v"(x[6]), [x7] "+v"(x[7]),
          [rot] "+v"(rot), [p] "=&v"(p), [q] "=&v"(q)
        : [step] "v"(step));
}
__device__ __forceinline__ cf rotq(cf x, int q)   // x * exp(-j q pi/2), exact
{
    switch (q & 3) {
    case 0: return x;
    case 1: return {x.y, -x.x};
    case 2: return {-x.x, -x.y};
    default: return {-x.y, x.x};
    }
}

// ---- 2048-point FFT: radix 8-8-8-4 decimation in frequency, 256 threads, 8 points each.
__device__ __forceinline__ void r4(cf &u0, cf &u1, cf &u2, cf &u3)
{
    const cf p0 = u0 + u2, p1 = u0 - u2, q0 = u1 + u3, t = u1 - u3;
    u0 = p0 + q0; u2 = p0 - q0;
    u1 = add_mj(p1, t); u3 = sub_mj(p1, t);          // p1 +- (-j) t
}
__device__ __forceinline__ void r8(cf v[8])
{
    const float c8 = 0.70710678118654752440f;
    cf a0 = v[0] + v[4], b0 = v[0] - v[4], a1 = v[1] + v[5], b1 = v[1] - v[5];
    cf a2 = v[2] + v[6], b2 = v[2] - v[6], a3 = v[3] + v[7], b3 = v[3] - v[7];
    // b1 (1 - j) / sqrt 2 = c8 (r + i, i - r);  b3 (-1 - j) / sqrt 2 = (c8 (i - r), -(c8 (r + i)))
    b1 = (cf){c8, c8} * add_mj(b1, b1);
    { const cf m = sub_mj(b3, b3); b3 = (cf){c8, c8} * (cf){-m.x, -m.y}; }        // m = (r - i, i + r)
    r4(a0, a1, a2, a3);
    // r4 of (b0, b1, (-j) b2, b3) with the rotation of b2 folded into its first butterflies
    const cf p0 = add_mj(b0, b2), p1 = sub_mj(b0, b2), q0 = b1 + b3, t = b1 - b3;
    v[0] = a0; v[2] = a1; v[4] = a2; v[6] = a3;
    v[1] = p0 + q0; v[5] = p0 - q0;
    v[3] = add_mj(p1, t); v[7] = sub_mj(p1, t);
}

// LDS index with four pad slots (32 bytes) per 32 complex values: rows of the stride-4 pass
// then start 8 banks apart, so a 32-lane ds_read_b64 group (8 rows x 4 values) covers all
// 64 banks once
__device__ __forceinline__ int pad(int i) { return i + ((i >> 5) << 2); }
constexpr int FFT_LDS = 2048 + 256;

struct Twiddles { cf a[7], b[7], c[7]; };          // per-thread constants of passes A, B, C

__device__ __forceinline__ void load_twiddles(Twiddles &tw, const float2 *__restrict__ W, int t)
{
#pragma unroll
    for (int c = 1; c < 8; ++c) {
        float2 x = W[t * c], y = W[8 * (t & 31) * c], z = W[64 * (t & 3) * c];
        tw.a[c - 1] = {x.x, x.y}; tw.b[c - 1] = {y.x, y.y}; tw.c[c - 1] = {z.x, z.y};
    }
}

// Twiddles of passes B and C for all threads of a workgroup: B depends on t & 31, C on t & 3.  k_demod keeps
// them in LDS (252 values) instead of 28 VGPRs per thread: that is what lets four of its waves share a SIMD.
constexpr int TWL = 7 * 32 + 7 * 4;
__device__ __forceinline__ void load_twiddles_lds(cf a[7], float2 *twl, const float2 *__restrict__ W, int t)
{
#pragma unroll
    for (int c = 1; c < 8; ++c) { float2 x = W[t * c]; a[c - 1] = {x.x, x.y}; }
    if (t < 7 * 32) { const int b = t / 7, c = t % 7 + 1; twl[t] = W[8 * b * c]; }
    if (t < 7 * 4) { const int e = t / 7, c = t % 7 + 1; twl[7 * 32 + t] = W[64 * e * c]; }
    __syncthreads();
}

// in: v[j] = x[t + 256 j]; out: v[e] = X[bin_of_pos(8 t + e)].  buf: FFT_LDS float2.
// The caller must __syncthreads() before the next use of buf (and before it reads buf itself: the last exchange is wave-local).
// twa/twb/twc: the thread's twiddles of the three radix-8 passes (registers), or, when twl is given,
// passes B and C read theirs from the LDS table of load_twiddles_lds().
template <bool SYNC = true>
__device__ __forceinline__ void fft2048_core(cf v[8], float2 *buf, int t, const cf *twa, const cf *twb, const cf *twc, const float2 *twl)
{
    r8(v);
    cmul7(v, twa);
#pragma unroll
    for (int c = 0; c < 8; ++c) buf[pad(t + 256 * c)] = make_float2(v[c].x, v[c].y);
    if (SYNC) __syncthreads();                                   // (SYNC = false: timing probe only)
    int base = (t >> 5) * 256 + (t & 31);
#pragma unroll
    for (int j = 0; j < 8; ++j) { float2 x = buf[pad(base + 32 * j)]; v[j] = {x.x, x.y}; }
    r8(v);
    if (twl) {
        cf w[7];
#pragma unroll
        for (int c = 0; c < 7; ++c) { const float2 x = twl[(t & 31) * 7 + c]; w[c] = {x.x, x.y}; }
        cmul7(v, w);
    } else cmul7(v, twb);
#pragma unroll
    for (int c = 0; c < 8; ++c) buf[pad(base + 32 * c)] = make_float2(v[c].x, v[c].y);
    if (SYNC) __syncthreads();                                   // (SYNC = false: timing probe only)
    base = (t >> 2) * 32 + (t & 3);
#pragma unroll
    for (int j = 0; j < 8; ++j) { float2 x = buf[pad(base + 4 * j)]; v[j] = {x.x, x.y}; }
    r8(v);
    if (twc) cmul7(v, twc);                          // (registers win over the LDS table when both are given)
    else {
        cf w[7];
#pragma unroll
        for (int c = 0; c < 7; ++c) { const float2 x = twl[7 * 32 + (t & 3) * 7 + c]; w[c] = {x.x, x.y}; }
        cmul7(v, w);
    }
    // The last exchange stays inside a quad of lanes: row t >> 2 of the buffer is written and read by lanes 4 (t >> 2) .. + 3 only,
    // which run in lockstep in one wave.  So it needs no workgroup barrier (the LDS serves a wave's operations in order), and it
    // may lay the row out differently from the exchange before it: the 16-byte slots (two values) of a row are stored with slot
    // bit 0 flipped in the lanes with bit 4 set.  Read as they lie, the four 16-byte reads of a thread cost two LDS cycles each:
    // ds_read_b128 serves lanes {0-3, 12-15, 20-27} together (MI355X_MICROARCH.md, LDS), and rows 0 and 6, 3 and 5 start on the
    // same 16-byte column.  With the flip every group covers the 16 columns once (64 of 212 conflict cycles per symbol gone).
#ifdef DABX_PROBE_EXCH_NOFLIP
    const int flip = 0;                                          // timing probe
#else
    const int flip = (t >> 4) & 1;
#endif
    {
        const int b3 = (t >> 2) * 32 + 2 * (((t >> 1) & 1) ^ flip) + (t & 1);
#pragma unroll
        for (int c = 0; c < 8; ++c) buf[pad(b3 + 4 * c)] = make_float2(v[c].x, v[c].y);
    }
#ifdef DABX_PROBE_EXCH_BARRIER
    __syncthreads();                                             // timing probe
#else
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#endif
    {   // positions 8t .. 8t+7 are contiguous (never across a pad): four 16-byte reads; slot e lies at e ^ flip = e +- flip
        const float4 *p = reinterpret_cast<const float4 *>(buf + pad(8 * t));
        const float4 *pe = p + flip, *po = p - flip;
        { float4 x = pe[0]; v[0] = {x.x, x.y}; v[1] = {x.z, x.w}; }
        { float4 x = po[1]; v[2] = {x.x, x.y}; v[3] = {x.z, x.w}; }
        { float4 x = pe[2]; v[4] = {x.x, x.y}; v[5] = {x.z, x.w}; }
        { float4 x = po[3]; v[6] = {x.x, x.y}; v[7] = {x.z, x.w}; }
    }
    r4(v[0], v[1], v[2], v[3]);
    r4(v[4], v[5], v[6], v[7]);
}
__device__ __forceinline__ void fft2048(cf v[8], float2 *buf, int t, const Twiddles &tw)
{
    fft2048_core(v, buf, t, tw.a, tw.b, tw.c, nullptr);
}

// x + x[lane ^ d] for d = 1, 2, 4, 8, 16, 32 in that order (the xor butterfly of the arithmetic contract) without a trip
// through the LDS crossbar: DPP for 1, 2, 4, 8; for 16 and 32 gfx950's v_permlane16_swap / v_permlane32_swap exchange
// the odd rows (upper half) of one copy with the even rows (lower half) of another, after which copy A + copy B is
// x + x[lane ^ 16] (x + x[lane ^ 32]) in every lane.
__device__ __forceinline__ float wave_xor_sum(float x)
{
#define DABX_DPP(v, ctrl) __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), ctrl, 0xf, 0xf, false))
    x = x + DABX_DPP(x, 0xB1);                                   // quad_perm [1,0,3,2]
    x = x + DABX_DPP(x, 0x4E);                                   // quad_perm [2,3,0,1]
    {                                                            // lane ^ 4: banks 0, 2 read lane + 4, banks 1, 3 read lane - 4
        int o = __builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x104, 0xf, 0x5, false);          // row_shl:4
        o = __builtin_amdgcn_update_dpp(o, __float_as_int(x), 0x114, 0xf, 0xa, false);              // row_shr:4
        x = x + __int_as_float(o);
    }
    x = x + DABX_DPP(x, 0x128);                                  // row_ror:8
#undef DABX_DPP
    {
        const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
        x = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    }
    {
        const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
        x = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    }
    return x;
}

// fixed-order block sum (256 threads): xor butterfly in each wave, then (W0+W1)+(W2+W3)
__device__ __forceinline__ float reduce256(float x, float *red /*4 floats*/, int t)
{
    x = wave_xor_sum(x);
    __syncthreads();
    if ((t & 63) == 0) red[t >> 6] = x;
    __syncthreads();
    float s01 = red[0] + red[1], s23 = red[2] + red[3];
    return s01 + s23;
}

__device__ __forceinline__ cf nco(const DevTables &T, uint32_t th)
{
    float2 h = T.nco_hi[th >> 21], l = T.nco_lo[(th >> 10) & 2047];
    return cmul((cf){h.x, h.y}, (cf){l.x, l.y});
}

template <int FMT>
__device__ __forceinline__ void sample(const uint8_t *ring, int64_t idx, int &i, int &q)
{
    if (FMT == 0) { uint16_t u = reinterpret_cast<const uint16_t *>(ring)[idx]; i = (int)(u & 0xff) - 128; q = (int)(u >> 8) - 128; }
    else { uint32_t u = reinterpret_cast<const uint32_t *>(ring)[idx]; i = (int16_t)(u & 0xffff); q = (int16_t)(u >> 16); }
}

// the same in two halves, so that a batch of loads can be in flight with one register per sample
template <int FMT>
__device__ __forceinline__ uint32_t sample_raw(const uint8_t *ring, uint32_t idx)
{
    if (FMT == 0) return reinterpret_cast<const uint16_t *>(ring)[idx];
    return reinterpret_cast<const uint32_t *>(ring)[idx];
}
template <int FMT>
__device__ __forceinline__ void sample_unpack(uint32_t u, int &i, int &q)
{
    if (FMT == 0) { i = (int)(u & 0xff) - 128; q = (int)(u >> 8) - 128; }
    else { i = (int16_t)(u & 0xffff); q = (int16_t)(u >> 16); }
}

__device__ __forceinline__ int64_t wrap(int64_t n, int64_t len)
{
    int64_t w = n % len;
    return w < 0 ? w + len : w;
}

// one complex sample as floats: u8 pairs are offset binary (value - 128), s16 pairs signed; both conversions are exact
template <int FMT>
__device__ __forceinline__ cf sample_f(const uint8_t *ring, uint32_t idx)
{
    if (FMT == 0) {
        const uint32_t u = reinterpret_cast<const uint16_t *>(ring)[idx];
        // v_cvt_f32_ubyte0 / ubyte1 straight from the loaded pair, then ONE packed subtraction (written out: left to itself the
        // compiler subtracts in the integer domain first — exact as well, but two byte extractions and two adds more per sample)
        float a, b;
        asm("v_cvt_f32_ubyte0 %0, %2\n\tv_cvt_f32_ubyte1 %1, %2" : "=&v"(a), "=v"(b) : "v"(u));      // (a must not share u's register: u is read again)
        return (cf){a, b} + (cf){-128.0f, -128.0f};
    }
    const uint32_t u = reinterpret_cast<const uint32_t *>(ring)[idx];
    return {(float)(int16_t)(u & 0xffff), (float)(int16_t)(u >> 16)};
}

// 2048-sample window starting at ring index `widx` (already wrapped, wave-uniform), de-rotated
// by -inc with phase zero `phase_off` samples before the window.  The head of the ring is mirrored behind its end
// (DABX_RING_MIRROR samples, kept by the host side), so the window never wraps: the eight loads of a thread are a
// scalar base + 2 t + an immediate — no per-sample address arithmetic.
// In two halves for k_demod: the eight loads of the NEXT symbol's window travel while this symbol's FFT runs
// (window_issue), and are turned into samples at the top of the next iteration (window_finish; the empty asm statement pins the
// conversion there — left alone the compiler converts right behind the loads and the wave waits out their latency).
template <int FMT>
__device__ __forceinline__ void window_issue(uint32_t raw[8], const uint8_t *ring, int64_t widx, int t)
{
    const uint8_t *base = ring + widx * (FMT == 0 ? 2 : 4);
#pragma unroll
    for (int j = 0; j < 8; ++j)
        raw[j] = FMT == 0 ? (uint32_t)reinterpret_cast<const uint16_t *>(base)[t + 256 * j] : reinterpret_cast<const uint32_t *>(base)[t + 256 * j];
}
template <int FMT>
__device__ __forceinline__ void window_finish(cf v[8], uint32_t raw[8], const DevTables &T, uint32_t phase_off, int32_t inc, int t)
{
    const uint32_t dth = (uint32_t)(-(int64_t)inc);
    const cf step = nco(T, dth * 256u);
    const cf rot = nco(T, dth * (phase_off + (uint32_t)t));
    asm volatile("" : "+v"(raw[0]), "+v"(raw[1]), "+v"(raw[2]), "+v"(raw[3]), "+v"(raw[4]), "+v"(raw[5]), "+v"(raw[6]), "+v"(raw[7]));
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const uint32_t u = raw[j];
        if (FMT == 0) {
            float a, b;
            asm("v_cvt_f32_ubyte0 %0, %2\n\tv_cvt_f32_ubyte1 %1, %2" : "=&v"(a), "=v"(b) : "v"(u));
            v[j] = (cf){a, b} + (cf){-128.0f, -128.0f};
        } else v[j] = {(float)(int16_t)(u & 0xffff), (float)(int16_t)(u >> 16)};
    }
    derotate8(v, rot, step);
}

// both halves back to back (k_sync, k_null_search): the phase table reads of window_finish overlap the sample loads
template <int FMT>
__device__ __forceinline__ void load_window(cf v[8], const DevTables &T, const uint8_t *ring, int64_t ring_len,
                                            int64_t widx, uint32_t phase_off, int32_t inc, int t)
{
#ifdef DABX_PROBE_NOLOAD
    const uint32_t dth = (uint32_t)(-(int64_t)inc);
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = {(float)((t + 7 * j) & 31) - 16.0f, (float)((t * 3 + j) & 31) - 16.0f};   // timing probe: no memory
    derotate8(v, nco(T, dth * (phase_off + (uint32_t)t)), nco(T, dth * 256u));
#else
    uint32_t raw[8];
    window_issue<FMT>(raw, ring, widx, t);
    window_finish<FMT>(v, raw, T, phase_off, inc, t);
#endif
}

// the eight output bins of thread t (bin_of_pos[8 t .. 8 t + 7]) in one 16-byte load
__device__ __forceinline__ void load_bins(const DevTables &T, int t, int b[8])
{
    const uint4 w = *reinterpret_cast<const uint4 *>(T.bin_of_pos + 8 * t);
    b[0] = w.x & 0xffff; b[1] = w.x >> 16; b[2] = w.y & 0xffff; b[3] = w.y >> 16;
    b[4] = w.z & 0xffff; b[5] = w.z >> 16; b[6] = w.w & 0xffff; b[7] = w.w >> 16;
}

// integer CORDIC, angle of (x + j y) in 2^-32 turns.  The table, round(atan(2^-i) / (2 pi) * 2^32), is spelled out so that the
// unrolled loop carries literals: read from memory it was 28 dependent round trips on the one lane every other phase waits for.
__device__ __forceinline__ int32_t cordic(int64_t y, int64_t x)
{
    constexpr uint32_t tab[28] = {536870912, 316933406, 167458907, 85004756, 42667331, 21354465, 10679838, 5340245,
                                  2670163,   1335087,   667544,    333772,   166886,   83443,    41722,    20861,
                                  10430,     5215,      2608,      1304,     652,      326,      163,      81,
                                  41,        20,        10,        5};
    if (x == 0 && y == 0) return 0;
    uint64_t ax = (uint64_t)(x < 0 ? -x : x), ay = (uint64_t)(y < 0 ? -y : y), m = ax > ay ? ax : ay;
    int sh = 0;
    while ((m >> sh) >= (1ULL << 29)) sh++;
    if (sh) { x >>= sh; y >>= sh; }
    else while ((m << 1) < (1ULL << 29)) { m <<= 1; x *= 2; y *= 2; }
    uint32_t ang = 0;
    if (x < 0) { x = -x; y = -y; ang = 0x80000000u; }
#pragma unroll
    for (int i = 0; i < 28; i++) {
        int64_t xs = x >> i, ys = y >> i;
        if (y > 0) { x += ys; y -= xs; ang += tab[i]; }
        else       { x -= ys; y += xs; ang -= tab[i]; }
    }
    return (int32_t)ang;
}

}  // namespace

// ------------------------------------------------------------------ null search
// One workgroup per stream; acts only on streams that are not locked.
constexpr int NS_BLOCKS = 3072, NS_WIN = 41;

template <int FMT>
__global__ __launch_bounds__(256) void k_null_search(DevCtx C)
{
    const int s = blockIdx.x, t = threadIdx.x;
    DevState &st = C.state[s];
    // a locked stream whose last frame came without a phase reference symbol looks for the null symbol again (a recording that
    // loops or was cut jumps to another frame phase; waiting for four bad frames first would lose them all)
    const bool suspect = st.locked && st.bad > 0;
    if (st.locked && !suspect) return;
    __shared__ uint64_t E[NS_BLOCKS + NS_WIN];
    __shared__ uint64_t redv[256];
    __shared__ int redi[256];
    const uint8_t *ring = C.ring + (size_t)s * C.ring_bytes;
    const int64_t from = wrap(st.pos, C.ring_len);
    for (int b = t; b < NS_BLOCKS + NS_WIN; b += 256) {
        // 64 x the variance of the block (exact integers): a DC offset of the recording must not fill the null symbol
        uint64_t e = 0;
        int64_t si = 0, sq = 0;
        int64_t idx = from + 64 * (int64_t)b;
        idx = idx >= C.ring_len ? idx - C.ring_len : idx;
        idx = idx >= C.ring_len ? idx - C.ring_len : idx;
        for (int n = 0; n < 64; ++n) {
            int i, q;
            sample<FMT>(ring, idx, i, q);
            e += (uint64_t)((int64_t)i * i + (int64_t)q * q);
            si += i; sq += q;
            if (++idx >= C.ring_len) idx -= C.ring_len;
        }
        E[b] = 64 * e - (uint64_t)(si * si) - (uint64_t)(sq * sq);
    }
    __syncthreads();
    uint64_t tot = 0, best = ~0ULL;
    int bb = 0;
    for (int b = t; b < NS_BLOCKS; b += 256) {
        uint64_t m = 0;
        for (int j = 0; j < NS_WIN; ++j) m += E[b + j];
        if (m < best) { best = m; bb = b; }
        tot += E[b];
    }
    redv[t] = best; redi[t] = bb;
    __syncthreads();
    for (int d = 128; d > 0; d >>= 1) {
        if (t < d) {
            uint64_t o = redv[t + d]; int oi = redi[t + d];
            if (o < redv[t] || (o == redv[t] && oi < redi[t])) { redv[t] = o; redi[t] = oi; }
        }
        __syncthreads();
    }
    best = redv[0]; bb = redi[0];
    __syncthreads();
    redv[t] = tot;
    __syncthreads();
    for (int d = 128; d > 0; d >>= 1) { if (t < d) redv[t] += redv[t + d]; __syncthreads(); }
    tot = redv[0];
    if (t == 0) {
        // quietest window at least 2.5 dB below the average window; the null ends where two consecutive blocks rise
        // above the midpoint between the null's level and the average block energy
        if (!(best * 16 * NS_BLOCKS < tot * NS_WIN * 9)) { if (!suspect) st.acq_fail = 1; return; }
        const uint64_t mid = best * NS_BLOCKS + tot * NS_WIN;
        int edge = bb + NS_WIN;
        for (int b = bb; b + 1 < NS_BLOCKS + NS_WIN; ++b)
            if (E[b] * 2 * NS_WIN * NS_BLOCKS > mid && E[b + 1] * 2 * NS_WIN * NS_BLOCKS > mid) { edge = b; break; }
        const int64_t ns = st.pos + 64 * (int64_t)edge - TNULL;
        if (suspect) {
            // where the null symbol is against where the flywheel expects it, to the nearest whole frame; within RELOCK_TOL the
            // flywheel carries on, beyond it this is another frame phase: a new acquisition from there
            int64_t d = (ns - st.pos) % TF;
            if (d < 0) d += TF;
            if (d >= TF / 2) d -= TF;
            if (d <= RELOCK_TOL && d >= -RELOCK_TOL) return;
            st.locked = 0;
        }
        st.pos = ns;
        st.cif = 0;
        st.acq_fail = 0;
    }
}

// ------------------------------------------------------------------------- sync
// One workgroup per (stream, frame): guard correlation -> fractional CFO, PRS FFT,
// optional integer-CFO search, impulse response peak -> window position.
template <int FMT>
__global__ __launch_bounds__(256, 4) void k_sync(DevCtx C, int n_frames)
{
    const int s = blockIdx.x / n_frames, f = blockIdx.x % n_frames, t = threadIdx.x;
    const DevState &st = C.state[s];
    DevSync &rec = C.sync[(size_t)s * C.max_frames + f];
    if (st.acq_fail) {
        if (t == 0) { DevSync z = {}; rec = z; }
        return;
    }
    __shared__ __attribute__((aligned(16))) float2 buf[FFT_LDS];
    __shared__ float2 nat[TU];
    __shared__ float red[4];
    __shared__ int64_t red64[16];
    __shared__ int32_t sh_inc;
    __shared__ float pk_v[4];
    __shared__ int pk_i[4];
    __shared__ int64_t sh_m;
    __shared__ int sh_d[4];
    const DevTables &T = C.tab;
    const uint8_t *ring = C.ring + (size_t)s * C.ring_bytes;
    const int wide = !st.locked;
    // frame f is expected where the tracked sampling-clock drift puts it (no drift is known while acquiring)
    int64_t pos_f = st.pos + (int64_t)f * TF + (wide ? 0 : (((int64_t)f * st.slope) >> 16));
    // twiddles of passes B and C in LDS, like k_demod: at <= 128 registers four workgroups share a CU instead of two, and this
    // kernel is one long chain of dependent phases per frame — latency, not throughput
    __shared__ float2 twl[TWL];
    cf twa[7];
    load_twiddles_lds(twa, twl, T.W, t);
    cf v[8];
    int32_t inc = 0;

    // A pass at the predicted frame start and, when the PRS turns out to sit more than RESYNC_THR samples from where
    // the window expected it, a second pass at the corrected position (the guard correlation must look at guards).
    for (int pass = 0; pass < 2; ++pass) {
    // 1. guard-interval correlation over the PRS and the three FIC symbols
    int64_t cre = 0, cim = 0, en = 0, es = 0;
    {
        const int64_t g0 = wrap(pos_f + TNULL, C.ring_len);
        // both loops are unrolled by hand with every load issued before the first use: left as loops the compiler waits for
        // each pair of samples in turn, fifteen exposed round trips to memory in a kernel that is latency from end to end
        // (32-bit ring offsets here: a ring is far below 2^31 samples, dabx_create checks it, and a 64-bit address per load
        // in flight does not fit the 128 registers)
        const uint32_t L = (uint32_t)C.ring_len, p0 = (uint32_t)wrap(pos_f, C.ring_len), g32 = (uint32_t)g0;
        uint32_t ga[7], gb[7], ea[8], eb[8];
#pragma unroll
        for (int it = 0; it < 7; ++it) {
            int k = t + 256 * it;
            if (k >= 4 * 408) k = 4 * 408 - 1;              // the tail lanes of the last round re-read one sample; masked below
            uint32_t sy = (uint32_t)k / 408u, n = 48u + (uint32_t)k % 408u;
            uint32_t a = g32 + sy * TS + n;
            if (a >= L) a -= L;
            uint32_t b = a + TU;
            if (b >= L) b -= L;
            ga[it] = sample_raw<FMT>(ring, a); gb[it] = sample_raw<FMT>(ring, b);
        }
        // sample energy inside the null symbol and inside the PRS (SNR estimate for the host)
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            uint32_t n = t + 256 * it;
            uint32_t a = p0 + 128 + n, b = p0 + TNULL + TG + n;
            if (a >= L) a -= L;
            if (b >= L) b -= L;
            ea[it] = sample_raw<FMT>(ring, a); eb[it] = sample_raw<FMT>(ring, b);
        }
#pragma unroll
        for (int it = 0; it < 7; ++it) {
            if (t + 256 * it < 4 * 408) {
                int i1, q1, i2, q2;
                sample_unpack<FMT>(ga[it], i1, q1); sample_unpack<FMT>(gb[it], i2, q2);
                cre += (int64_t)i1 * i2 + (int64_t)q1 * q2;
                cim += (int64_t)q1 * i2 - (int64_t)i1 * q2;
            }
        }
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            int i1, q1, i2, q2;
            sample_unpack<FMT>(ea[it], i1, q1); sample_unpack<FMT>(eb[it], i2, q2);
            en += (int64_t)i1 * i1 + (int64_t)q1 * q1;
            es += (int64_t)i2 * i2 + (int64_t)q2 * q2;
        }
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            cre += __shfl_xor(cre, d, 64); cim += __shfl_xor(cim, d, 64);
            en += __shfl_xor(en, d, 64); es += __shfl_xor(es, d, 64);
        }
        if ((t & 63) == 0) { red64[4 * (t >> 6)] = cre; red64[4 * (t >> 6) + 1] = cim; red64[4 * (t >> 6) + 2] = en; red64[4 * (t >> 6) + 3] = es; }
        __syncthreads();
        cre = red64[0] + red64[4] + red64[8] + red64[12];
        cim = red64[1] + red64[5] + red64[9] + red64[13];
        en = red64[2] + red64[6] + red64[10] + red64[14];
        es = red64[3] + red64[7] + red64[11] + red64[15];
    }
    if (t == 0) {
        int32_t A = cordic(cim, cre);
        int32_t inc_meas = (int32_t)((-(int64_t)A) >> 11);
        int32_t inc = inc_meas;
        if (!wide) {
            int32_t d = (int32_t)(((uint32_t)(inc_meas - st.inc) + (1u << 20)) & ((1u << 21) - 1)) - (1 << 20);
            inc = st.inc + d;
        }
        sh_inc = inc;
    }
    __syncthreads();
    inc = sh_inc;

    const int64_t w0 = pos_f + TNULL + TG - BACKOFF;
    const int64_t w0i = wrap(w0, C.ring_len);
    load_window<FMT>(v, T, ring, C.ring_len, w0i, 0u, inc, t);
    fft2048_core(v, buf, t, twa, nullptr, nullptr, twl);
    int m_best = 0;
    if (wide) {
        // spectrum to natural order for the shifted differential correlation
#pragma unroll
        for (int e = 0; e < 8; ++e) nat[T.bin_of_pos[8 * t + e]] = make_float2(v[e].x, v[e].y);
        __syncthreads();
        float best = -1.0f;
        for (int m = -CFO_RANGE; m <= CFO_RANGE; ++m) {
            float ar = 0.0f, ai = 0.0f;
            for (int i = 0; i < 6; ++i) {
                int j = t + 256 * i;
                if (j >= 1534) break;
                int k = T.cfo_car[j];
                float2 x1 = nat[(k + m) & 2047], x0 = nat[(k + m - 1) & 2047];
                cf d = cmulc((cf){x1.x, x1.y}, (cf){x0.x, x0.y});
                cf e = rotq(d, T.prs_dq[k & 2047]);
                ar = ar + e.x; ai = ai + e.y;
            }
            float cr = reduce256(ar, red, t);
            float ci = reduce256(ai, red, t);
            float c0 = cr * cr, c1 = ci * ci, cm = c0 + c1;
            if (cm > best) { best = cm; m_best = m; }
        }
        inc = inc + m_best * (1 << 21);
        __syncthreads();
        load_window<FMT>(v, T, ring, C.ring_len, w0i, 0u, inc, t);
        fft2048_core(v, buf, t, twa, nullptr, nullptr, twl);
    }
    // optional signal spectrum of the last frame (reference: dabsdrSpectrumCBFunc_t, dabsdr.h:393):
    // linear power of the un-normalised 2048-point FFT, natural bin order (a second pass overwrites the first)
    if (C.spectrum && f == n_frames - 1) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float a = v[e].x * v[e].x, b = v[e].y * v[e].y;
            C.spectrum[(size_t)s * TU + T.bin_of_pos[8 * t + e]] = a + b;
        }
    }
    // 3. conj(X * conj(P)) in natural order, second FFT -> impulse response
    __syncthreads();
    int bins[8], pq[8];
    load_bins(T, t, bins);
#pragma unroll
    for (int e = 0; e < 8; ++e) pq[e] = T.prs_q[bins[e]];         // all eight in flight before the first is used
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        cf r = {0.0f, 0.0f};
        if (pq[e] >= 0) r = rotq(v[e], pq[e]);
        nat[bins[e]] = make_float2(r.x, -r.y);
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 8; ++j) { float2 x = nat[t + 256 * j]; v[j] = {x.x, x.y}; }
    __syncthreads();
    fft2048_core(v, buf, t, twa, nullptr, nullptr, twl);
    float acc = 0.0f, peak = -1.0f;
    int pidx = 0;
    load_bins(T, t, bins);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        float a = v[e].x * v[e].x, b = v[e].y * v[e].y, m2 = a + b;
        acc = acc + m2;
        int n = bins[e];
        if (m2 > peak || (m2 == peak && n < pidx)) { peak = m2; pidx = n; }
    }
    float total = reduce256(acc, red, t);
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        float op = __shfl_xor(peak, d, 64); int oi = __shfl_xor(pidx, d, 64);
        if (op > peak || (op == peak && oi < pidx)) { peak = op; pidx = oi; }
    }
    if ((t & 63) == 0) { pk_v[t >> 6] = peak; pk_i[t >> 6] = pidx; }
    __syncthreads();
    for (int w = 0; w < 4; ++w)                                  // every thread: the block's peak
        if (pk_v[w] > peak || (pk_v[w] == peak && pk_i[w] < pidx)) { peak = pk_v[w]; pidx = pk_i[w]; }
    // the FFT window follows the FIRST significant path, not the strongest: among the taps up to EARLY_SPAN samples
    // before the peak the earliest one with at least EARLY_THR of the peak's power wins
    {
        const float thr = peak * EARLY_THR;
        int dmax = 0;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int d = (pidx - bins[e]) & 2047;
            const float a = v[e].x * v[e].x, b = v[e].y * v[e].y, m2 = a + b;
            if (d != 0 && d <= EARLY_SPAN && m2 >= thr && d > dmax) dmax = d;
        }
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) dmax = max(dmax, __shfl_xor(dmax, d, 64));
        __syncthreads();
        if ((t & 63) == 0) sh_d[t >> 6] = dmax;
        __syncthreads();
        dmax = max(max(sh_d[0], sh_d[1]), max(sh_d[2], sh_d[3]));
        pidx = (pidx - dmax) & 2047;
    }
    if (t == 0) {
        int delta = pidx >= 1024 ? pidx - 2048 : pidx;
        sh_m = (int64_t)delta - BACKOFF;                         // how far the PRS sits from where the window expected it
        DevSync r;
        r.t_sym0 = w0 + delta - BACKOFF;
        r.inc = inc;
        r.flags = ((total > 0.0f && peak * 2048.0f >= LOCK_THR * total) ? 1 : 0) | (wide ? 2 : 0);      // (silence is not a phase reference symbol)
        r.peak_idx = pidx; r.m_int = m_best;
        r.peak = peak; r.total = total;
        r.cp_re = cre; r.cp_im = cim;
        r.e_null = en; r.e_sig = es;
        rec = r;
    }
    __syncthreads();
    const int64_t m = sh_m;
    if (pass == 1 || (m <= RESYNC_THR && m >= -RESYNC_THR)) break;
    pos_f += m;
    __syncthreads();
    }   // pass
    // optional spectrum of the null symbol (TII carriers; reference: DABSDR_SPECT_NULL / dabsdrNtfTii_t):
    // 2048 samples centred in the 2656-sample null symbol, de-rotated like the PRS window
    // (every frame of the step: the host's noise estimate is per frame)
    if (C.null_spectrum) {
        __syncthreads();
        const int64_t n0 = pos_f + (TNULL - TU) / 2;
        load_window<FMT>(v, T, ring, C.ring_len, wrap(n0, C.ring_len), 0u, inc, t);
        fft2048_core(v, buf, t, twa, nullptr, nullptr, twl);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float a = v[e].x * v[e].x, b = v[e].y * v[e].y;
            C.null_spectrum[((size_t)s * C.max_frames + f) * TU + T.bin_of_pos[8 * t + e]] = a + b;
        }
    }
}

// ------------------------------------------------------------------------ demod
// One workgroup per (stream, frame, group of 19 symbols).  The previous symbol's
// spectrum stays in registers (same thread owns the same bins in every symbol).
// FFT output slot e of thread t holds bin b0(t) + 64 (e >> 2) + 512 (e & 3) with b0(t) in 0..447 (dabx_spec.hpp: bin_of_pos(8 t + e)):
// the slots whose 448 bins all are carriers (1..768 or 1280..2047) need no mask in the demapper's sum
constexpr bool slot_always_carrier(int e)
{
    const int lo = 64 * (e >> 2) + 512 * (e & 3), hi = lo + 447;
    return (lo >= 1 && hi <= 768) || (lo >= 1280 && hi <= 2047);
}
static_assert(!slot_always_carrier(0) && slot_always_carrier(3) && slot_always_carrier(4) && slot_always_carrier(7), "bin_of_pos");
constexpr int DEMOD_GROUPS = 4, DEMOD_GSYMS = 19;          // (2 x 38 and 1 x 76 symbols per workgroup run at the same speed)

// SCO: the variant that de-rotates the differential product for streams whose sampling clock is off (below).  Both
// variants are launched over the whole grid; a workgroup whose stream belongs to the other one leaves at once.  (One
// kernel holding both symbol loops costs the common loop a spilled twiddle and 6 % of its time.)
template <int FMT, bool SCO>
__global__ __launch_bounds__(256, 4) void k_demod(DevCtx C, int n_frames)
{
    const int g = blockIdx.x % DEMOD_GROUPS;
    const int sf = blockIdx.x / DEMOD_GROUPS;
    const int s = sf / n_frames, f = sf % n_frames, t = threadIdx.x;
    const DevState &st = C.state[s];
    if (st.acq_fail) return;
    const DevSync rec = C.sync[(size_t)s * C.max_frames + f];
    const DevTables &T = C.tab;
    __shared__ __attribute__((aligned(16))) float2 buf[FFT_LDS];
    __shared__ float red[4];
    __shared__ __attribute__((aligned(16))) uint16_t soft[NCAR + 8];    // one (Re, Im) soft-bit pair per carrier + one dummy slot
    const uint8_t *ring = C.ring + (size_t)s * C.ring_bytes;
    int8_t *fic = C.fic_soft + ((size_t)s * C.max_frames + f) * FICBITS;
    int8_t *ti = C.ti + (size_t)s * C.ti_slots * CIFBITS;
    const int64_t cif0 = st.cif + 4 * (int64_t)f;

    __shared__ float2 twl[TWL];
    cf twa[7], twc[7];                                           // passes A and C: registers; pass B: the LDS table (the registers end here)
    load_twiddles_lds(twa, twl, T.W, t);
#pragma unroll
    for (int c = 1; c < 8; ++c) { const float2 z = T.W[64 * (t & 3) * c]; twc[c - 1] = {z.x, z.y}; }
#ifdef DABX_PROBE_STAGGER
    // timing probe: workgroups in odd slots of their CU (HW_ID.TG_ID) start late, so that their LDS phases meet the others' VALU phases
    if ((__builtin_amdgcn_s_getreg(((4 - 1) << 11) | (16 << 6) | 4) & 1)) __builtin_amdgcn_s_sleep(DABX_PROBE_STAGGER);
#endif
    // where the soft-bit pair of each FFT output position goes in the staging buffer: the frequency de-interleaver index n
    // itself for FIC symbols ([0]), its residue-major place (n & 15) * 96 + (n >> 4) for MSC symbols ([1]); the bins
    // outside the 1536 carriers all go to ONE dummy slot (no branch around the store; lanes storing to one dword do not conflict,
    // a slot per lane did: 15 of the scatter's 135 conflict cycles per symbol), used[] masks them out of the sum
    __shared__ __attribute__((aligned(16))) uint16_t dst_l[2][TU];         // BYTE offsets into soft[] (2 x the index): one 16-byte read per symbol and thread
    uint32_t used = 0;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int n = T.n_of_bin[T.bin_of_pos[8 * t + e]];
        used |= (n >= 0 ? 1u : 0u) << e;
        dst_l[0][8 * t + e] = static_cast<uint16_t>(2 * (n < 0 ? NCAR : n));
        dst_l[1][8 * t + e] = static_cast<uint16_t>(2 * (n < 0 ? NCAR : (n & 15) * (NCAR / 16) + (n >> 4)));
    }

    // Sampling-clock offset: the windows keep their nominal spacing, so a recording whose clock is off by eps sees every
    // symbol eps * TS samples later in its window than the one before: a phase of -2 pi k eps TS / TU on carrier k in the
    // differential product (34 degrees at the band edge for 100 ppm).  From SCO_MIN on it is turned back with the tracked
    // drift: dth = slope * 319 / 768 is that phase per carrier in 2^-32 turns; the carrier of FFT output 8 t + e is
    // b0(t) + m(e), so the factor is rt (per thread) times sm[e] (uniform: scalar registers).
    const int32_t slope = (rec.flags & 2) ? 0 : st.slope;        // an acquisition step starts without a drift estimate
    const bool sco = slope >= SCO_MIN || slope <= -SCO_MIN;      // workgroup-uniform
    if (sco != SCO) return;
    __shared__ float2 rt_l[SCO ? 256 : 1];                       // the threads' factors: kept in LDS, the registers are all taken
    cf sm[8];
    if (SCO) {
        const int32_t dth = (int32_t)(((int64_t)slope * 319) / 768);
        const int m_e[8] = {0, 512, -1024, -512, 64, 576, -960, -448};
        const cf rt = nco(T, (uint32_t)(((t >> 5) + 8 * ((t >> 2) & 7) + 128 * (t & 3)) * dth));
        rt_l[t] = make_float2(rt.x, rt.y);                       // read back by the same thread only
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const cf x = nco(T, (uint32_t)(m_e[e] * dth));
            sm[e] = {__int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(x.x))), __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(x.y)))};
        }
    }

    const int l_first = g * DEMOD_GSYMS;                 // first symbol to demap (0 = PRS: reference only)
    const int l_ref = l_first == 0 ? 0 : l_first - 1;    // symbol whose spectrum seeds the differential
    const int l_last = l_first + DEMOD_GSYMS - 1;
    int64_t widx = wrap(rec.t_sym0 + (int64_t)l_ref * TS, C.ring_len);
    cf prev[8], v[8];
    uint32_t raw[8];
    window_issue<FMT>(raw, ring, widx, t);
    for (int l = l_ref; l <= l_last; ++l) {
        window_finish<FMT>(v, raw, T, (uint32_t)(l * TS), rec.inc, t);
        widx += TS;
        if (widx >= C.ring_len) widx -= C.ring_len;
        if (l < l_last) window_issue<FMT>(raw, ring, widx, t);           // the next symbol's samples travel during this symbol's FFT
#ifdef DABX_PROBE_DEMOD_NOBARRIER
        fft2048_core<false>(v, buf, t, twa, nullptr, twc, twl);
#else
        fft2048_core(v, buf, t, twa, nullptr, twc, twl);
#endif
        if (l > l_ref) {
            cf y[8];
            float acc = 0.0f;
            const uint4 nidx4 = *reinterpret_cast<const uint4 *>(dst_l[l <= 3 ? 0 : 1] + 8 * t);      // eight byte offsets, two per dword
            const uint32_t nidx2[4] = {nidx4.x, nidx4.y, nidx4.z, nidx4.w};
            cmulc8(y, v, prev);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                if constexpr (SCO) { const float2 r = rt_l[t]; y[e] = cmul(cmul(y[e], sm[e]), (cf){r.x, r.y}); }
                const float a = fabsf(y[e].x) + fabsf(y[e].y);
                if (slot_always_carrier(e)) acc = acc + a;               // (three of the eight slots: no select)
                else acc = acc + ((used >> e) & 1u ? a : 0.0f);          // adding +0 leaves the sum as it is
            }
            float S = reduce256(acc, red, t), gsc = 0.0f;
            if (S > 0.0f && S < __builtin_inff()) { int E; frexpf(S, &E); gsc = ldexpf(1.0f, SOFT_EXP - E); }
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                // rint(y * gsc) clamped to +-31: gsc is a power of two, so the product is exact and adding 1.5 * 2^23 rounds it
                // to the nearest integer (ties to even) into the low bits of the float — the same value as rintf(), one (packed)
                // FMA for both components.  The bit pattern is 0x4B400000 + n: clamped as an integer between 0x4B400000 -+ 31 its
                // low byte IS the soft bit (two's complement), so no subtraction and no masking: one byte permute packs the pair.
                const cf tq = __builtin_elementwise_fma(y[e], (cf){gsc, gsc}, (cf){12582912.0f, 12582912.0f});
                const int lo = 0x4B400000 - (int)SOFT_MAX, hi = 0x4B400000 + (int)SOFT_MAX;
                const int qa = min(max(__float_as_int(tq.x), lo), hi), qb = min(max(__float_as_int(tq.y), lo), hi);
                const uint32_t off = (e & 1) ? nidx2[e >> 1] >> 16 : nidx2[e >> 1] & 0xffffu;
                *reinterpret_cast<uint16_t *>(reinterpret_cast<char *>(soft) + off) = (uint16_t)__builtin_amdgcn_perm((uint32_t)qb, (uint32_t)qa, 0x0c0c0400u);
            }
            __syncthreads();
            if (t < NCAR / 16) {
                // 16 carriers per thread: 32 bytes of pairs -> 16 Re bytes and 16 Im bytes
                const uint4 p0 = reinterpret_cast<const uint4 *>(soft)[2 * t], p1 = reinterpret_cast<const uint4 *>(soft)[2 * t + 1];
                uint4 re, im;
                re.x = __builtin_amdgcn_perm(p0.y, p0.x, 0x06040200); im.x = __builtin_amdgcn_perm(p0.y, p0.x, 0x07050301);
                re.y = __builtin_amdgcn_perm(p0.w, p0.z, 0x06040200); im.y = __builtin_amdgcn_perm(p0.w, p0.z, 0x07050301);
                re.z = __builtin_amdgcn_perm(p1.y, p1.x, 0x06040200); im.z = __builtin_amdgcn_perm(p1.y, p1.x, 0x07050301);
                re.w = __builtin_amdgcn_perm(p1.w, p1.z, 0x06040200); im.w = __builtin_amdgcn_perm(p1.w, p1.z, 0x07050301);
                int8_t *dst;
                int imoff;
                if (l <= 3) {                                            // FIC: bit n = Re of carrier n, bit n + 1536 = Im
                    dst = fic + (l - 1) * SYMBITS + 16 * t;
                    imoff = NCAR;
                } else {
                    // residue-major row: bit b of the CIF lives at (b & 15) * TI_SEG + (b >> 4): 16 segments of 192 bytes
                    // per symbol, each Re (96 bytes) then Im (96 bytes); the staging buffer is in that order.  The time de-interleaver
                    // is applied HERE: residue class q of CIF c belongs to logical frame c - bitrev4(q) (EN 300 401 §12) and is filed
                    // in that frame's row, so the decoder finds a whole logical frame in ONE row and the places of a codeword's bits no
                    // longer depend on the frame (k_viterbi reads them from a table made once per profile).
                    const int res = t / 6, part = t % 6;
                    const int64_t frame = cif0 + (l - 4) / 18 - (int64_t)(__builtin_bitreverse32((uint32_t)res) >> 28);
                    // (32-bit arithmetic: at most 256 rows of 55296 bytes)
                    dst = ti + ((uint32_t)(frame & (C.ti_slots - 1)) * (uint32_t)CIFBITS + (uint32_t)(res * TI_SEG + ((l - 4) % 18) * (SYMBITS / 16) + 16 * part));
                    imoff = NCAR / 16;
                }
                *reinterpret_cast<uint4 *>(dst) = re;
                *reinterpret_cast<uint4 *>(dst + imoff) = im;
            }
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) prev[e] = v[e];
        __syncthreads();
    }
}

// ---------------------------------------------------------------------- Viterbi
// One wave per codeword, one trellis state per lane.  The lane<->state map
// rotates every step so that the two candidates of a butterfly always sit in
// lanes that differ by one fixed xor vector of the cycle {1,2,7,8,16,32}:
// four of the six exchanges are a DPP modifier of the max, two go through the LDS
// crossbar (ds_swizzle / ds_bpermute).  LDS MEMORY holds what is not exchanged: the
// block's soft values (the A rows of the MFMAs) and the ring of decision words the merge
// test and the traceback walk.
namespace {

// (the xor vectors of the six phases: 1, 2, 7, 8, 16, 32 — quad_perm, quad_perm, row_half_mirror, row_ror:8 as DPP
// modifiers of the max, swap-16 and xor-32 through ds_swizzle / ds_bpermute; see tools/gen_acs32.py)

// coordinates of a lane in the basis XV: a0 = b0^b2, a1 = b1^b2, a2..a5 = b2..b5
__device__ __forceinline__ int lane_coord(int lane, int k)
{
    int b2 = (lane >> 2) & 1;
    if (k == 0) return (lane & 1) ^ b2;
    if (k == 1) return ((lane >> 1) & 1) ^ b2;
    return (lane >> k) & 1;
}

// mother code output nibble (x0 in bit 3) for state (bit 5 newest) and input 0
__device__ __forceinline__ int conv_out0(int state)
{
    const int g[4] = {0133 & 63, 0171 & 63, 0145 & 63, 0133 & 63};
    int o = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) o = (o << 1) | (__popc(state & g[k]) & 1);
    return o;
}

// MSC rows are stored residue-major (TI_SEG = 55296/16 bytes per residue class of the bit index) and per LOGICAL frame: k_demod
// files residue class q of CIF c in the row of frame c - bitrev4(q), where its 192 bytes per symbol are contiguous (written from the
// other side — one row per CIF, the classes picked from 16 rows here — the decoder needed a table of 16 row offsets per codeword and nine
// address instructions per gathered step; written de-interleaved in natural bit order the writer would store single bytes 16 apart).
// So a codeword's bit i sits at a fixed offset from the codeword's base, (i & 15) * TI_SEG + (i >> 4), and the four offsets and the byte
// mask of every trellis step come from a table made once per profile (dabx_spec.hpp: step_gather; linear rows — FIC, stage tests — have
// their own variant): five words per step and lane, each from an array of its own (contiguous across the wave), read a round ahead.
constexpr int VIT_INFO_PAD = 256;                    // = dabx::kStepInfoPad: empty entries behind the end of each array
// The soft values of a step as the A row the matrix core wants: (2 x0, 2 x1, 2 x2, 2 x3), one byte each, zero where punctured
// (|x| <= 31, so twice a value fits its byte).  A step keeps a prefix of its four bits; all four bytes are loaded unconditionally
// (valid addresses: the punctured ones point at the step's first bit), shifted left by one inside their bytes, and the punctured
// ones are masked off.  Three phases, each a loop half or more apart so that none waits for the one before:
//   gather_load   the step's record -> the four offsets and the mask
//   gather_bytes  the four byte loads, each into the register that held its offset
//   gather_finish the A row
// (d16 byte loads that would drop the bytes into the halves of two registers were tried: on a GPU with SRAM ECC — this one — a d16
// load clears the other half of its destination, so the bytes still have to be collected by instructions.)
struct Gather { uint32_t b0, b1, b2, b3, m; };
__device__ __forceinline__ void gather_load(Gather &g, const char *info, uint32_t toff, uint32_t arr_bytes)
{
    // info, arr_bytes: wave-uniform — five scalar bases, one 32-bit lane offset
    g.b0 = *reinterpret_cast<const uint32_t *>(info + toff);
    g.b1 = *reinterpret_cast<const uint32_t *>((info + arr_bytes) + toff);
    g.b2 = *reinterpret_cast<const uint32_t *>((info + 2 * (size_t)arr_bytes) + toff);
    g.b3 = *reinterpret_cast<const uint32_t *>((info + 3 * (size_t)arr_bytes) + toff);
    g.m = *reinterpret_cast<const uint32_t *>((info + 4 * (size_t)arr_bytes) + toff);
}
__device__ __forceinline__ void gather_bytes(Gather &g, const uint8_t *base)
{
    g.b0 = base[g.b0]; g.b1 = base[g.b1]; g.b2 = base[g.b2]; g.b3 = base[g.b3];
}
// One asm statement: it pins the place (left alone, the compiler collects the bytes right behind the loads and the wave waits out
// their whole latency there) and takes five instructions where the compiler's shifts and ors take six.  Every byte holds a value of
// -31..31: shifted left by one inside its byte it is twice that value; 0xFE per kept byte clears the neighbour's sign bit that
// came in from below, 0 per punctured byte clears the byte.
__device__ __forceinline__ int gather_finish(const Gather &g)
{
    uint32_t lo, hi;
    asm volatile("v_lshl_or_b32 %[lo], %[b1], 8, %[b0]\n\t"
                 "v_lshl_or_b32 %[hi], %[b3], 8, %[b2]\n\t"
                 "v_lshl_or_b32 %[lo], %[hi], 16, %[lo]\n\t"
                 "v_add_u32 %[lo], %[lo], %[lo]\n\t"                 // << 1 as an add: full rate, the shift is not
                 "v_and_b32 %[lo], %[lo], %[m]"
                 : [lo] "=&v"(lo), [hi] "=&v"(hi) : [b0] "v"(g.b0), [b1] "v"(g.b1), [b2] "v"(g.b2), [b3] "v"(g.b3), [m] "v"(g.m) : "memory");
    return (int)lo;
}

// a wave-uniform 64-bit value into scalar registers (the compiler cannot prove the uniformity of what is computed from a loaded
// work item and otherwise re-reads it from vector registers at every use; applied to OFFSETS: a pointer rebuilt from integers
// loses its address space and its loads become flat ones with 64-bit vector addresses)
__device__ __forceinline__ size_t scalar_u64(size_t v)
{
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v), hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32));
    return ((size_t)hi << 32) | lo;
}

// ---- add-compare-select, hand scheduled (text generated by tools/gen_acs32.py).
// Path metrics are scaled by 128; the low seven bits of a lane's value Q are a field that starts every group of six
// steps at 63 and records the keep(1)/receive(0) history of the lane's SURVIVOR — it travels with the path through the
// max.  Per step, 3 VALU issues and ONE number from the matrix core, X = 128 M + (1 << ph)
// (M = +-x0 +-x1 +-x2 +-x3, the branch metric of the lane's state; 128 M = (2 x) . (+-64), the tag is the MFMA's C operand):
//   S = Q - X                     sent to the butterfly partner: the field goes down by 1 << ph
//   K = Q + X                     kept: the field goes up by 1 << ph
//   Q' = max(K, S of the butterfly partner lane)
// After the six steps of a group (phases 0..5) the field has gone from 63 to 2 x (the six tags of ITS survivor); it is
// shifted into the lane's decision word (v_alignbit) and set back to 63 (v_and_or).  On the way it stays within 0..126
// (down: 63 - (2^ph - 1) >= 2^ph; up: 63 + 63), so it never touches the metric, and a metric tie keeps the own path,
// exactly as the textbook rule: the kept candidate's field is larger by at least 2 (tests/lane_model.py).
// Range: |metric| <= 27654 steps x 4 x 31 x 128 < 2^31.
// X comes from the matrix core: six v_mfma_i32_4x4x4_16b_i8 per 24 steps (one per phase, its four result rows are the
// four groups of a decision word, see tools/gen_acs32.py), otherwise idle, while the vector ALU is what bounds this
// kernel.  The A operands are the packed soft values of the chunk, staged in LDS memory by the lanes that gathered
// them: lane l reads the row of group l mod 4.
#include "dabx_acs32.inc"
#define DABX_ACS_IN                                                                                                 \
    : [va] "v"(va), [ad] "v"(lane_x32), [wa] "v"(wa), [k0] "v"(sk[0]), [k1] "v"(sk[1]), [k2] "v"(sk[2]),            \
      [k3] "v"(sk[3]), [k4] "v"(sk[4]), [k5] "v"(sk[5]), [m128] "s"(-128)                                           \
    : "memory", DABX_ACS_CLOBBER
// four groups = 24 steps = one decision word, written byte by byte to the LDS address wa; va = LDS byte address of the lane's
// A row: the chunk's first dword + 24 (lane & 3).  SECOND: the second chunk of a loop iteration, with the first chunk's va and wa
// (its rows lie 96 bytes further, its decision word one row of 256 bytes further: immediate offsets).
template <bool SECOND = false>
__device__ __forceinline__ void acs24(int &pm, const int *sk, uint32_t va, int lane_x32, uint32_t wa)
{
    int S, K, D;
    if (SECOND) asm volatile(DABX_ACS24B_TEXT : [pm] "+v"(pm), [S] "=&v"(S), [K] "=&v"(K), [D] "=&v"(D) DABX_ACS_IN);
    else asm volatile(DABX_ACS24_TEXT : [pm] "+v"(pm), [S] "=&v"(S), [K] "=&v"(K), [D] "=&v"(D) DABX_ACS_IN);
}
// one group of six steps (the tail of a codeword): only row 0 of the results is used
__device__ __forceinline__ void acs6(int &pm, const int *sk, uint32_t va, int lane_x32, uint32_t &bits)
{
    int S, K, D;
    const uint32_t wa = 0;
    asm volatile(DABX_ACS6_TEXT : [pm] "+v"(pm), [bits] "+v"(bits), [S] "=&v"(S), [K] "=&v"(K), [D] "=&v"(D) DABX_ACS_IN);
}
#undef DABX_ACS_IN

constexpr int VIT_BLK = 48;          // trellis steps per loop iteration = two chunks of 24 (one decision word each)
constexpr int VIT_XS = 192;          // staging ring of A rows: four iterations of 48 steps
constexpr int VIT_XS_WORDS = VIT_XS;
constexpr int VIT_RING = 16;         // decision words (24 steps each) a wave keeps in LDS: 4 KB
constexpr int VIT_UNIT = 8;          // words decoded (or, without a merge, spilled) at a time: 192 steps = 6 output words

// Walk 96 steps (the four decision words wd[0..3], lane = basis coordinate) backwards from position A; writes the
// three output words of block b96.  The position A at the end of a group IS the group's six decoded bits
// (bit q = step 6 g + q), the position six steps earlier is A ^ ~tags.  Only the low six bits of A matter
// (v_readlane and the bit reversal ignore the rest), so neither the tags are masked nor the complement trimmed.
__device__ __forceinline__ uint32_t walk96(const uint32_t wd[4], uint32_t A, uint32_t *out32, const uint32_t *__restrict__ prbs32, int b96,
                                           int lane)
{
    uint32_t o[3] = {0u, 0u, 0u};                                // o[k]: step 96 b + 32 k + j at bit 31 - j
#pragma unroll
    for (int grp = 15; grp >= 0; --grp) {
        const uint32_t rev = __builtin_bitreverse32(A) >> 26;                 // step 6 grp + q at bit 5 - q
        const int f = 6 * grp, k = f >> 5, off = f & 31;
        if (off + 6 <= 32) o[k] |= rev << (26 - off);
        else {
            const int n1 = off + 6 - 32;                                      // bits that spill into the next word
            o[k] |= rev >> n1;
            o[k + 1] |= (rev & ((1u << n1) - 1u)) << (32 - n1);
        }
        A = ~(A ^ ((uint32_t)__builtin_amdgcn_readlane((int)wd[grp >> 2], (int)A) >> (1 + 8 * (grp & 3))));
    }
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < 3; ++k) out32[3 * b96 + k] = __builtin_bswap32(o[k] ^ prbs32[3 * b96 + k]);
    }
    return A;
}

// Decode the words [w_lo, w_hi) (multiples of 4) backwards from position A at the end of word w_hi - 1.  Words from
// w_ring on are in the wave's LDS ring, older ones (spilled because no merge was found in time) in global scratch.
// (SPILL = false, the first pass: nothing is ever spilled, every word is in the ring)
template <bool SPILL>
__device__ __forceinline__ void trace_words(const uint32_t *ring, const uint32_t *dec, int w_ring, int w_lo, int w_hi, uint32_t A,
                                            uint32_t *out32, const uint32_t *__restrict__ prbs32, int lane)
{
    for (int w = w_hi - 4; w >= w_lo; w -= 4) {
        uint32_t wd[4];
        if (!SPILL || w >= w_ring) {
            // w is a multiple of four and so is the ring's length: the four rows are consecutive — one address, immediate offsets
            const uint32_t *row = ring + (w & (VIT_RING - 1)) * 64 + lane;
#pragma unroll
            for (int k = 0; k < 4; ++k) wd[k] = row[64 * k];
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k) wd[k] = dec[(w + k) * 64 + lane];
        }
        A = walk96(wd, A, out32, prbs32, w >> 2, lane);
    }
}

// Have the 64 survivors at the end of word w_hi - 1 merged by the start of word B?  Every lane walks its own
// survivor back through the ring (a gather per six steps); if all arrive at one position, every path through any
// later trellis state passes through it: the words before B can be decoded now, exactly.
__device__ __forceinline__ bool survivors_merged(const uint32_t *ring, int w_hi, int B, int coordA, uint32_t &O)
{
    // The walk runs on LDS byte addresses: a = row + 4 P.  The rows are 256 bytes and 256-byte aligned, so the position sits
    // in bits 2..7 of a, and "P = ~(P ^ tags)" is one xor of those bits with the complemented tags (which the decision word
    // holds at bits 1 + 8 g, i.e. times four at bits 8 g - 1): a shift and one three-operand logic instruction per six steps.
    const uint32_t ring_a = (uint32_t)(uintptr_t)ring;           // low 32 bits of a shared pointer = the LDS address
    uint32_t a = ring_a + (uint32_t)((w_hi - 1) & (VIT_RING - 1)) * 256u + ((uint32_t)coordA << 2);
    for (int w = w_hi - 1; w >= B; --w) {
        uint32_t x;
        // a ^= ~(word(a) >> (8 g - 1)) & 0xFC for g = 3, 2, 1, 0 (v_bitop3 with the table of s0 ^ (~s1 & s2)); written out: the
        // compiler keeps two copies of the address and spends four instructions per group
        asm volatile("ds_read_b32 %[x], %[a]\n\ts_waitcnt lgkmcnt(0)\n\tv_lshrrev_b32 %[x], 23, %[x]\n\tv_bitop3_b32 %[a], %[a], %[x], %[m] bitop3:0xd2\n\t"
                     "ds_read_b32 %[x], %[a]\n\ts_waitcnt lgkmcnt(0)\n\tv_lshrrev_b32 %[x], 15, %[x]\n\tv_bitop3_b32 %[a], %[a], %[x], %[m] bitop3:0xd2\n\t"
                     "ds_read_b32 %[x], %[a]\n\ts_waitcnt lgkmcnt(0)\n\tv_lshrrev_b32 %[x], 7, %[x]\n\tv_bitop3_b32 %[a], %[a], %[x], %[m] bitop3:0xd2\n\t"
                     "ds_read_b32 %[x], %[a]\n\ts_waitcnt lgkmcnt(0)\n\tv_add_u32 %[x], %[x], %[x]\n\tv_bitop3_b32 %[a], %[a], %[x], %[m] bitop3:0xd2\n\t"
                     "v_lshl_add_u32 %[a], %[d], 8, %[a]"                                 // to the row of the word before (in the statement: outside it the compiler adds into a new register and copies back)
                     : [a] "+v"(a), [x] "=&v"(x) : [m] "s"(0xFC), [d] "s"(((w - 1) & (VIT_RING - 1)) - (w & (VIT_RING - 1))) : "memory");
    }
    uint32_t P = a >> 2;
    P &= 63u;
    O = (uint32_t)__builtin_amdgcn_readfirstlane((int)P);
#ifdef DABX_PROBE_FORCE_MERGE
    return (__builtin_amdgcn_ballot_w64(P != O) | 1) != 0;          // timing probes whose decisions are wrong: always "merged"
#else
    return __builtin_amdgcn_ballot_w64(P != O) == 0;
#endif
}

// Decode one terminated codeword with the calling wave.  Every DAB codeword has 192 k + 6 steps: n_in is a multiple
// of 192 bits (24 ms x 8 kbit/s; the FIC's 768), followed by the six tail steps.
//   ring:   the wave's VIT_RING x 64 decision words in LDS: word w (the 24 steps of chunk w) of the lane with basis
//           coordinates A at [(w mod VIT_RING) * 64 + A]; byte i of the word is the field of the chunk's i-th group of six
//           steps: its tags at bits 1 + 8 i .. 6 + 8 i (bits 8 i and 7 + 8 i are not part of it)
//   dec:    the codeword's block of global scratch, same layout without the modulus: touched only when the
//           survivors of a stretch of 192 steps have not merged within another 192 (erased or tied input)
//   prbs32: energy dispersal, bit 31-j of word h = PRBS bit 32 h + j
//   out:    n_in/8 bytes.  nsteps (= n_in + 6), n_in and all pointers are wave-uniform.
//   soft:   the codeword's first soft bit (wave-uniform); info: its gather map (dabx_spec.hpp: step_gather), linear or residue-major
//   xs:     the wave's soft-value staging ring in LDS: the A rows of 192 steps (row of step t at dword t mod 192)
//   SPILL:  false = the first pass (k_viterbi): a codeword whose survivors do not merge gives up (returns false, dec is not
//           touched) and is decoded again by k_viterbi_requeue, which has a block of scratch per wave (SPILL = true)
template <bool SPILL>
__device__ __forceinline__ bool viterbi_wave(const int8_t *soft, const uint32_t *__restrict__ info, int nsteps, int n_in,
                             const uint32_t *__restrict__ prbs32, uint32_t *dec, uint8_t *out, int *xs, uint32_t *ring)
{
    const int lane = threadIdx.x & 63;
    int sk[6];
#pragma unroll
    for (int ph = 0; ph < 6; ++ph) {
        int st = 0;
#pragma unroll
        for (int i = 0; i < 6; ++i) st |= lane_coord(lane, (i + ph) % 6) << i;
        int u = st & 1, o = conv_out0(st), kg = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            int neg = ((o >> (3 - j)) & 1) ^ u;
            kg |= (neg ? 0xC0 : 0x40) << (8 * j);          // -+64
        }
        sk[ph] = kg;                                       // (the step's tag 1 << ph is the C operand of the MFMA)
    }
    const int coordA = lane ^ (((lane >> 2) & 1) * 3);          // coordinates of this lane in the basis XV
    const int lane_x32 = (lane ^ 32) << 2;                       // ds_bpermute address of the xor-32 partner
    int pm = (lane == 0 ? 0 : PM_INIT * 128) + 63;              // metric x 128 + the field's start value
    const int nblk = nsteps / VIT_BLK;                           // full blocks of two chunks (even); the tail follows
    uint32_t *out32 = reinterpret_cast<uint32_t *>(out);
    int w_dec = 0, w_ring = 0;                                   // first word not yet decoded / first word still in the ring
    // Soft-bit pipeline.  All 64 lanes gather: round q fetches the steps 64 q .. 64 q + 63 (one per lane) into rows 64 (q mod 3) .. + 63
    // of the staging ring of 192 A rows.  Round q is made entirely inside the loop iteration before the first one that reads its
    // rows, iteration floor(4 q / 3) - 1 — three rounds per four iterations of 48 steps, none in the iterations 2, 6, 10, …; the rows
    // it replaces (round q - 3) were last read an iteration or more earlier.  Inside that iteration the step's offsets and mask are
    // loaded first (five contiguous loads across the wave), its four bytes between the two chunks, and the A row is put together
    // behind the second chunk: every load has a chunk of 24 steps to hide behind and no register of the gather is live over the
    // decoding or the loop edge.  (Carried over the edge — bytes in flight during the decoding, a full iteration of cover — the
    // compiler copies all five registers at the edge, and with the round's three parts in three conditional regions it keeps two
    // register sets and copies between them at every merge: both measured slower.  So the iteration with a round and the one
    // without are written out as two bodies.)  Every DAB codeword is 192 k + 6 steps.
    // No register of its own for the lane's place in a round: lane l takes the step (l ^ 32) of the round's 64, so that 4 (l ^ 32) — the ds_bpermute address the exchange of phase 5 keeps anyway — is
    // the lane's byte offset into the arrays and into the staging ring; what moves from round to round is scalar.  No clamping: the
    // arrays carry VIT_INFO_PAD empty entries behind their end.
    const uint32_t lane4 = (uint32_t)lane_x32;
    const uint32_t arr_bytes = 4u * (uint32_t)(nsteps + VIT_INFO_PAD);
    // A row of this lane: row r = the step of group r of a 24-step chunk (LDS byte address: the low 32 bits of a shared
    // pointer); the four addresses of a read (rows 0..3: six dwords apart) fall into four different LDS banks
    const uint32_t xs_a = (uint32_t)(uintptr_t)xs;
    const uint32_t va0 = xs_a + 24u * (lane & 3);
    const uint32_t ring_lane = (uint32_t)(uintptr_t)ring + ((uint32_t)coordA << 2);      // LDS byte address of the lane's word in row 0 of the wave's decision ring
    const char *infob = reinterpret_cast<const char *>(info);
    const uint8_t *base = reinterpret_cast<const uint8_t *>(soft);
    typedef int __attribute__((address_space(3))) *lds_int;
    Gather ga;
    gather_load(ga, infob, lane4, arr_bytes);                    // round 0 (the only loads of a codeword that nothing hides)
    gather_bytes(ga, base);
    *reinterpret_cast<lds_int>(xs_a + lane4) = gather_finish(ga);
    uint32_t A = 0;
    bool gave_up = false;
    for (int blk = 0; blk < nblk; ++blk) {
        const int it = blk & 3;                                  // iteration inside the super-block of 192 steps (wave-uniform)
        const uint32_t va = va0 + 192u * (uint32_t)it;           // the iteration's first chunk: row 48 it
        // (word 2 blk is even: the iteration's second word is the next row of the ring, never across its end)
        const uint32_t wa0 = ring_lane + (uint32_t)(((2 * blk) & (VIT_RING - 1)) * 256);
#ifndef DABX_PROBE_NOGATHER
        if (it != 2) {
            // round q = 3 (blk / 4) + (it + 1 or, in the last iteration of a super-block, 3)
            const uint32_t q = 3u * (uint32_t)(blk >> 2) + (it == 3 ? 3u : (uint32_t)it + 1u);
            const uint32_t slot = it == 3 ? 0u : (uint32_t)it + 1u;               // q mod 3
            gather_load(ga, infob, lane4 + 256u * q, arr_bytes);
            acs24(pm, sk, va, lane_x32, wa0);
            gather_bytes(ga, base);
            acs24<true>(pm, sk, va, lane_x32, wa0);
            *reinterpret_cast<lds_int>(xs_a + 256u * slot + lane4) = gather_finish(ga);
        } else
#endif
        {
            acs24(pm, sk, va, lane_x32, wa0);
            acs24<true>(pm, sk, va, lane_x32, wa0);
        }
        // ---- decode what can be decoded: the ring holds the words [w_ring, w_hi)
        const int w_hi = 2 * blk + 2, pend = w_hi - w_ring;
#ifdef DABX_PROBE_NOTRACE
        if (false) {
#else
        if (pend >= VIT_UNIT + 2) {                              // 48, 96, 144, 192 steps after the unit's end (pend is even)
#endif
            const int B = w_ring + VIT_UNIT;
            uint32_t O;
            if (survivors_merged(ring, w_hi, B, coordA, O)) {
                if (!SPILL) {
                    // first pass: exactly the unit's VIT_UNIT words, all in the ring (w_dec == w_ring): two walks of 96 steps, no loop
                    uint32_t P = O;
#pragma unroll
                    for (int u = VIT_UNIT / 4 - 1; u >= 0; --u) {
                        const int w = w_ring + 4 * u;
                        const uint32_t *row = ring + (w & (VIT_RING - 1)) * 64 + lane;
                        uint32_t wd[4];
#pragma unroll
                        for (int k = 0; k < 4; ++k) wd[k] = row[64 * k];
                        P = walk96(wd, P, out32, prbs32, w >> 2, lane);
                    }
                } else
                    trace_words<SPILL>(ring, dec, w_ring, w_dec, B, O, out32, prbs32, lane);
                w_dec = w_ring = B;
            } else if (pend == VIT_RING) {                       // no merge and the ring is full
                if (!SPILL) { gave_up = true; blk = nblk; }      // first pass: the codeword is decoded again by k_viterbi_requeue (the loop ends by its own condition: a second way out of it costs every iteration)
                else {                                           // the unit goes to global scratch
#pragma unroll
                    for (int k = 0; k < VIT_UNIT; ++k) dec[(w_ring + k) * 64 + lane] = ring[((w_ring + k) & (VIT_RING - 1)) * 64 + lane];
                    w_ring = B;
                }
            }
        }
    }
    if (!SPILL && gave_up) return false;
    {   // the six tail steps (rows 0..5 of the staging ring: nblk is a multiple of four): no output, from state 0 (lane 0)
        uint32_t bits = 0;
        acs6(pm, sk, va0, lane_x32, bits);
        A = ~((uint32_t)__builtin_amdgcn_readlane((int)bits, 0) >> 26);
    }
#ifndef DABX_PROBE_NOTRACE
    trace_words<SPILL>(ring, dec, w_ring, w_dec, 2 * nblk, A, out32, prbs32, lane);
#endif
    return true;
}

}  // namespace

// One wave per codeword (work item): FIC codeword c (0..3) of (stream, frame) when sub < 0, else MSC
// sub-channel `sub` of CIF c.  Decisions live in a 4 KB ring of LDS per wave (eight waves per SIMD stay resident)
// and are decoded as soon as the survivors have merged.  A codeword whose survivors do not merge within 192 steps (erased or
// tied input: silence, a punctured-out stretch) is put on the requeue list; k_viterbi_requeue decodes those again with a
// block of global scratch per wave, so no scratch has to be held for the 155 000 codewords of a step that never need it.
template <bool SPILL>
__device__ __forceinline__ bool viterbi_item(const DevCtx &C, const DevWork &w, uint32_t *dec, int *xs, uint32_t *ring)
{
    const DevState &st = C.state[w.stream];
    if (st.acq_fail) return true;
    // one call for both kinds of codeword (one copy of the decoder in the kernel): everything that differs is an argument
    const int8_t *soft;
    const uint32_t *info;
    uint8_t *out;
    int nsteps, n_in;
    if (w.sub < 0) {
        soft = C.fic_soft + scalar_u64(((size_t)w.stream * C.max_frames + w.frame) * FICBITS + w.c * 2304);
        out = C.fib + (((size_t)w.stream * C.max_frames + w.frame) * 12 + 3 * w.c) * 32;
        info = C.stepinfo + C.fic_info_off; nsteps = 774; n_in = 768;
    } else {
        const DevSub &sc = C.sub[(size_t)w.stream * 64 + w.sub];
        const int64_t r = st.cif + 4 * (int64_t)w.frame + w.c - 15;
        if (r < 0) return true;                      // time de-interleaver still filling (k_finish flags it)
        // the row of logical frame r (k_demod files every residue class where it belongs); sub-channels start on 64-bit boundaries,
        // so start_bit >> 4 is the sub-channel's place inside each of the row's 16 residue segments
        soft = C.ti + scalar_u64(((size_t)w.stream * C.ti_slots + (size_t)(r & (C.ti_slots - 1))) * CIFBITS + (sc.start_bit >> 4));
        out = C.msc + (((size_t)w.stream * C.max_frames + w.frame) * 4 + w.c) * (size_t)C.msc_stride + sc.out_off;
        info = C.stepinfo + sc.info_off; nsteps = sc.nsteps; n_in = sc.n_in;
    }
    return viterbi_wave<SPILL>(soft, info, nsteps, n_in, C.prbs, dec, out, xs, ring);
}

// (amdgpu_waves_per_eu: the kernel is written for eight waves per SIMD, 64 registers; telling the compiler so is worth half a percent)
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_viterbi(DevCtx C, const DevWork *__restrict__ work, int n_work)
{
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);      // provably wave-uniform
    const int wi = blockIdx.x * 4 + wave;
    if (wi >= n_work) return;
    __shared__ __attribute__((aligned(16))) int xs_all[4][VIT_XS_WORDS];
    __shared__ __attribute__((aligned(256))) uint32_t ring_all[4][VIT_RING * 64];     // survivors_merged() relies on the alignment
    if (!viterbi_item<false>(C, work[wi], nullptr, xs_all[wave], ring_all[wave]) && (threadIdx.x & 63) == 0)
        C.requeue[1 + atomicAdd(C.requeue, 1u)] = (uint32_t)wi;              // [0] = count, then the items
}

// Shader clock under load (timing mode only): ONE wave, launched on a stream of its own beside k_viterbi, brackets the time until
// the host-queued stop flag appears with the shader-cycle counter (s_memtime) and the constant 100 MHz counter (s_memrealtime);
// the ratio is the clock the chip held meanwhile (bench.py: valu_issue_frac at the measured clock).  Not inside k_viterbi: that
// kernel sits at exactly 64 vector registers (8 waves per SIMD), and every way of putting the two time stamps into it cost a 65th
// (measured: 1.59 -> 1.68 ms).  `max_ticks` bounds the wait whatever happens to the flag.
__global__ void k_clock_monitor(uint64_t *out, const uint32_t *stop, uint64_t max_ticks)
{
    if (threadIdx.x != 0) return;
    const uint64_t c0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
    uint64_t c1 = c0, r1 = r0;
    for (;;) {
        __builtin_amdgcn_s_sleep(32);
        c1 = __builtin_readcyclecounter(); r1 = __builtin_amdgcn_s_memrealtime();
        if (__hip_atomic_load(stop, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0 || r1 - r0 > max_ticks) break;
    }
    out[0] = c1 - c0;
    out[1] = r1 - r0;
}

// The codewords the first pass gave up on (normally none: the launch ends after reading the count).  VIT_RQ_BLOCKS workgroups
// walk the list; every wave owns scratch_words_per_wave words of C.dec_scratch (the longest codeword's decision words).
constexpr int VIT_RQ_BLOCKS = 64;
__global__ __launch_bounds__(256) void k_viterbi_requeue(DevCtx C, const DevWork *__restrict__ work, uint32_t scratch_words_per_wave)
{
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    __shared__ __attribute__((aligned(16))) int xs_all[4][VIT_XS_WORDS];
    __shared__ __attribute__((aligned(256))) uint32_t ring_all[4][VIT_RING * 64];
    const uint32_t n = C.requeue[0];
    const uint32_t me = blockIdx.x * 4 + wave;
    uint32_t *dec = C.dec_scratch + (size_t)me * scratch_words_per_wave;
    for (uint32_t i = me; i < n; i += VIT_RQ_BLOCKS * 4)
        (void)viterbi_item<true>(C, work[C.requeue[1 + i]], dec, xs_all[wave], ring_all[wave]);
}

// stage-level: n_cw linear codewords of one profile (unit tests, BASELINE config 2 helper)
__global__ __launch_bounds__(256) void k_viterbi_linear(const int8_t *soft, int n_coded, const uint32_t *info, int nsteps,
                                                        int n_in, const uint32_t *prbs, uint32_t *scratch, uint8_t *out, int n_cw)
{
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wi = blockIdx.x * 4 + wave;
    if (wi >= n_cw) return;
    __shared__ __attribute__((aligned(16))) int xs_all[4][VIT_XS_WORDS];
    __shared__ __attribute__((aligned(256))) uint32_t ring_all[4][VIT_RING * 64];     // survivors_merged() relies on the alignment
    (void)viterbi_wave<true>(soft + (size_t)wi * n_coded, info, nsteps, n_in, prbs, scratch + (size_t)wi * ((nsteps / 24 + 1) * 64), out + (size_t)wi * (n_in / 8), xs_all[wave], ring_all[wave]);
}

// stage-level FFT: one workgroup per vector, natural order in and out
__global__ __launch_bounds__(256) void k_fft(DevTables T, const float2 *in, float2 *out)
{
    __shared__ __attribute__((aligned(16))) float2 buf[FFT_LDS];
    const int t = threadIdx.x;
    Twiddles tw;
    load_twiddles(tw, T.W, t);
    cf v[8];
    for (int j = 0; j < 8; ++j) { float2 x = in[(size_t)blockIdx.x * TU + t + 256 * j]; v[j] = {x.x, x.y}; }
    fft2048(v, buf, t, tw);
    for (int e = 0; e < 8; ++e) out[(size_t)blockIdx.x * TU + T.bin_of_pos[8 * t + e]] = make_float2(v[e].x, v[e].y);
}

// ----------------------------------------------------------------------- finish
// FIB CRC (one thread per FIB) and the per-stream tracking state for the next step.
__global__ __launch_bounds__(256) void k_finish(DevCtx C, int n_frames)
{
    const int s = blockIdx.x, t = threadIdx.x;
    DevState &st = C.state[s];
    // the frames' synchronisation records for the tracker below, fetched by one thread each while the others check FIBs (read one
    // after the other by the thread that runs the tracker they were a chain of memory round trips)
    __shared__ int64_t sh_t0[64];
    __shared__ int32_t sh_flags[64], sh_inc[64];
    if (t < n_frames) {
        const DevSync &r = C.sync[(size_t)s * C.max_frames + t];
        sh_t0[t] = r.t_sym0; sh_flags[t] = r.flags; sh_inc[t] = r.inc;
    }
    for (int k = t; k < n_frames * 12; k += 256) {
        const uint8_t *fb = C.fib + ((size_t)s * C.max_frames * 12 + k) * 32;
        const uint4 w0 = reinterpret_cast<const uint4 *>(fb)[0], w1 = reinterpret_cast<const uint4 *>(fb)[1];      // the FIB in two loads
        const uint32_t w[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
        uint32_t crc = 0xFFFF;
#pragma unroll
        for (int i = 0; i < 30; ++i) {
            crc ^= ((w[i >> 2] >> (8 * (i & 3))) & 0xFFu) << 8;
#pragma unroll
            for (int b = 0; b < 8; ++b) crc = (crc & 0x8000) ? ((crc << 1) ^ 0x1021) & 0xFFFF : (crc << 1) & 0xFFFF;
        }
        crc = ~crc & 0xFFFF;
        C.fib_ok[(size_t)s * C.max_frames * 12 + k] = (!st.acq_fail && crc == ((((w[7] >> 16) & 0xFFu) << 8) | (w[7] >> 24))) ? 1 : 0;
    }
    for (int k = t; k < n_frames * 4; k += 256)
        C.msc_valid[(size_t)s * C.max_frames * 4 + k] = (!st.acq_fail && st.cif + k - 15 >= 0) ? 1 : 0;
    if (s == 0 && t == 0) {                   // the list of k_viterbi / k_viterbi_requeue has been served
        C.requeue[1 + C.requeue_cap] += C.requeue[0];      // running total (diagnostics: dabx_get_requeue_total)
        C.requeue[0] = 0;
    }
    __syncthreads();
    if (t == 0) {
        if (st.acq_fail) {                    // no null symbol found: skip ahead, stay unlocked
            st.pos += (int64_t)n_frames * TF;
            st.acq_fail = 0;
            return;
        }
        const int wide = !st.locked;
        int nbad = st.bad;
        // sampling-clock tracker (first order, gain 1/4): timing error of the good frames against their prediction
        const int32_t slope0 = wide ? 0 : st.slope;
        int64_t e_first = 0, e_last = 0;
        int f_first = -1, f_last = -1;
        for (int f = 0; f < n_frames; ++f) {
            const int32_t flags = sh_flags[f];
            nbad = (flags & 1) ? 0 : nbad + 1;
            if (flags & 1) {
                const int64_t e = sh_t0[f] + BACKOFF - TG - TNULL - (st.pos + (int64_t)f * TF + (((int64_t)f * slope0) >> 16));
                if (f_first < 0) { f_first = f; e_first = e; }
                f_last = f; e_last = e;
            }
        }
        // tracking: the error of frame f has built up over f + 1 frames since the last measured frame start;
        // acquisition: the drift between two good frames of the step
        int32_t sl = slope0;
        if (!wide && f_last >= 0) sl += (int32_t)(((e_last * 65536) / (f_last + 1)) / 4);
        else if (wide && f_last > f_first) sl = (int32_t)(((e_last - e_first) * 65536) / (f_last - f_first));
        sl = min(max(sl, -SLOPE_MAX), SLOPE_MAX);
        st.slope = sl;
        st.pos = sh_t0[n_frames - 1] + BACKOFF - TG - TNULL + TF + (sl >> 16);
        st.inc = sh_inc[n_frames - 1];
        st.cif += 4 * (int64_t)n_frames;
        st.bad = nbad;
        st.locked = wide ? (nbad == 0) : (nbad < 4);
    }
}

// explicit instantiations used by the host side
template __global__ void k_null_search<0>(DevCtx);
template __global__ void k_null_search<1>(DevCtx);
template __global__ void k_sync<0>(DevCtx, int);
template __global__ void k_sync<1>(DevCtx, int);
template __global__ void k_demod<0, false>(DevCtx, int);
template __global__ void k_demod<1, false>(DevCtx, int);
template __global__ void k_demod<0, true>(DevCtx, int);
template __global__ void k_demod<1, true>(DevCtx, int);
