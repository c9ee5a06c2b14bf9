// dabx_resample.hip — sample-rate conversion in front of the ring (gfx950).
//
// Replaces the reference's host-side converters that SDR devices run before the dabsdr input FIFO
// (reference: src/input/inputdevicesrc.h:78-150; selection src/input/inputdevicesrc.cpp:33-47):
//   k_resample_ds2     4096 kHz -> 2048 kHz, 43-tap half-band FIR          (inputdevicesrc.cpp:109-200)
//   k_farrow_segments  any rate -> 2048 kHz, transposed Farrow, 6 x 4      (inputdevicesrc.cpp:233-316)
//   k_farrow_outputs
// writing s16 IQ straight into the stream's ring.  Same float operations in the same order as the cited lines
// (one binary32 operation per step, no contraction), so the results equal the CPU checker (oracle/dab_src.c)
// bit for bit.  The reference's signal-level detector (a serial attack/release recursion used for device gain
// control) is not part of the decode path; k_level runs it when the caller asks for it (dabx_enable_level).
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rs {

constexpr int DS2_TAPS = 43, DS2_HIST = DS2_TAPS - 1;
__constant__ float ds2_coef[12] = {          // inputdevicesrc.h:105-109
    0.000223158782894952853123604619156594708f,  -0.00070774549637065342286290636764078954f,  0.001735782601167994458266075064045708132f,
    -0.003619832275410410967614316390950079949f, 0.006788741778432844271862212082169207861f,  -0.01183550169261274320753329902800032869f,
    0.019680477383812611941182879604639310855f,  -0.032073581325677551212560700832909788005f, 0.053382280107447499517547839786857366562f,
    -0.099631117404426483563639749263529665768f, 0.316099577146216947909351802081800997257f,  0.5f};
constexpr int FW_M = 4, FW_N = 6;
__constant__ float fw_coef[FW_N][FW_M] = {   // inputdevicesrc.h:142-149
    {0.001667349914006070960362f, 0.032712194697834547085780f, -0.146457831613232558609639f, 0.004040531324696360060411f},
    {-0.103347648141097675500433f, -0.244367915078825215235980f, 0.233146907266815583970043f, 0.243745693669456003904727f},
    {0.123959393981824803065983f, 0.873574620095563081356715f, 0.586104518066954516264389f, -0.711183949124104208827646f},
    {0.873450879200179830519346f, 0.039931348783534291457809f, -1.514110581690161660972649f, 0.711183949124100878158572f},
    {0.114518381640217964401174f, -0.923204505507555395205088f, 0.952958408884428287421997f, -0.243745693669455892882425f},
    {-0.104194288733202022889657f, 0.243880907754789599817258f, -0.134525637544989168370435f, -0.004040531324696002707375f},
};

struct State {                       // per stream, device memory
    float2 ds2_hist[DS2_HIST];       // the 42 input samples before the next one, oldest first
    float2 fw_x[FW_M];               // Farrow: integrators of the segment in progress
    float2 fw_a[FW_N - 1][FW_N];     // Farrow: polynomial outputs A_n of the last five finished segments, oldest first
    float level;                     // signal level (k_level), when the caller asked for it
    float pad;
};

template <int FMT>                   // 1: s16 pairs, 2: float pairs
__device__ __forceinline__ float2 in_sample(const void *in, int64_t k)
{
    if (FMT == 1) { const short2 v = reinterpret_cast<const short2 *>(in)[k]; return make_float2((float)v.x, (float)v.y); }
    return reinterpret_cast<const float2 *>(in)[k];
}

__device__ __forceinline__ short2 to_s16(float2 y, float gain)
{
    float a = rintf(y.x * gain), b = rintf(y.y * gain);
    a = __builtin_amdgcn_fmed3f(a, -32768.0f, 32767.0f); b = __builtin_amdgcn_fmed3f(b, -32768.0f, 32767.0f);
    return make_short2((short)a, (short)b);
}

// sample `pos` (absolute) of a stream: into the ring and, for the head of the ring, into its mirror behind the end
constexpr int RING_MIRROR = 4096;            // = DABX_RING_MIRROR (include/dabx.h)
__device__ __forceinline__ void ring_store(short2 *ring, int64_t ring_len, int64_t pos, short2 v)
{
    const int64_t w = pos % ring_len;
    ring[w] = v;
    if (w < RING_MIRROR) ring[ring_len + w] = v;
}

// Largest |I|, |Q| among the samples a launch writes, folded into *peak (a per-stream word the host latches per decode
// step: dabx_get_input_peak).  The legacy adapter's gain hysteresis needs it (dabsdr_shim.cpp).  Every thread of the wave
// must call it (inactive threads with active = false).
__device__ __forceinline__ void note_peak(uint32_t *peak, short2 v, bool active)
{
    int a = active ? max(abs((int)v.x), abs((int)v.y)) : 0;
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) a = max(a, __shfl_xor(a, d, 64));
    if (peak && (threadIdx.x & 63) == 0 && a) atomicMax(peak, (uint32_t)a);
}

// one output per thread: y[n] = sum_c coef[c] (x[2n-42+2c] + x[2n-2c]) + 0.5 x[2n-21]
template <int FMT>
__global__ __launch_bounds__(256) void k_resample_ds2(const void *in, int n_out, const State *st, short2 *ring, int64_t ring_len, int64_t wr,
                                                      float gain, uint32_t *peak)
{
    __shared__ float2 x[2 * 256 + DS2_HIST];
    const int t = threadIdx.x, n0 = blockIdx.x * 256;
    for (int k = t; k < 2 * 256 + DS2_HIST; k += 256) {          // x[k] = input sample 2 n0 - 42 + k
        const int64_t idx = 2 * (int64_t)n0 - DS2_HIST + k;
        float2 v = make_float2(0.0f, 0.0f);
        if (idx < 0) v = st->ds2_hist[DS2_HIST + idx];
        else if (idx < 2 * (int64_t)n_out) v = in_sample<FMT>(in, idx);
        x[k] = v;
    }
    __syncthreads();
    const int n = n0 + t;
    if (n >= n_out) { note_peak(peak, make_short2(0, 0), false); return; }
    float accI = 0.0f, accQ = 0.0f;
#pragma unroll
    for (int c = 0; c < 11; ++c) {
        const float2 o = x[2 * t + 2 * c], w = x[2 * t + DS2_HIST - 2 * c];
        float u = o.x + w.x;
        u = u * ds2_coef[c];
        accI = accI + u;
        u = o.y + w.y;
        u = u * ds2_coef[c];
        accQ = accQ + u;
    }
    {
        const float2 m = x[2 * t + 21];
        float u = m.x * ds2_coef[11];
        accI = accI + u;
        u = m.y * ds2_coef[11];
        accQ = accQ + u;
    }
    const short2 o16 = to_s16(make_float2(accI, accQ), gain);
    ring_store(ring, ring_len, wr + n, o16);
    note_peak(peak, o16, true);
}

// after the block: the history for the next call = the last 42 input samples (n_in even, >= 0)
template <int FMT>
__global__ void k_ds2_tail(const void *in, int64_t n_in, State *st)
{
    const int t = threadIdx.x;
    if (t >= DS2_HIST) return;
    const int64_t idx = n_in - DS2_HIST + t;
    const float2 v = idx >= 0 ? in_sample<FMT>(in, idx) : st->ds2_hist[DS2_HIST + idx];     // idx < 0: the block was shorter than the history
    __syncthreads();
    st->ds2_hist[t] = v;
}

// pass-through (2048 kHz in): float or s16 -> the ring's s16
template <int FMT>
__global__ __launch_bounds__(256) void k_resample_copy(const void *in, int n, short2 *ring, int64_t ring_len, int64_t wr, float gain, uint32_t *peak)
{
    const int k = blockIdx.x * 256 + threadIdx.x;
    short2 v = make_short2(0, 0);
    if (k < n) {
        v = to_s16(in_sample<FMT>(in, k), gain);
        ring_store(ring, ring_len, wr + k, v);
    }
    note_peak(peak, v, k < n);
}

// Transposed Farrow, pass 1: one thread per segment (the input samples integrated between two dumps).
//   seg[j] .. seg[j+1]-1: the segment's input samples; mu[k]: the fractional interval sample k is integrated with —
//   the data-independent schedule the host runs ahead (the reference's mu recursion is serial in float).
// Segment 0 continues the integrators carried over from the previous call.  A[j][n] = sum_m x[m] coef[n][m].
template <int FMT>
__global__ __launch_bounds__(256) void k_farrow_segments(const void *in, const int32_t *seg, const float *mu, int n_seg, const State *st,
                                                         float2 *A /*[n_seg][6]*/)
{
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= n_seg) return;
    float2 x[FW_M];
#pragma unroll
    for (int m = 0; m < FW_M; ++m) x[m] = j == 0 ? st->fw_x[m] : make_float2(0.0f, 0.0f);
    for (int k = seg[j]; k < seg[j + 1]; ++k) {
        float2 v = in_sample<FMT>(in, k);
        const float u = mu[k];
        x[0].x = x[0].x + v.x; x[0].y = x[0].y + v.y;
#pragma unroll
        for (int m = 1; m < FW_M; ++m) {
            v.x = v.x * u; v.y = v.y * u;
            x[m].x = x[m].x + v.x; x[m].y = x[m].y + v.y;
        }
    }
#pragma unroll
    for (int n = 0; n < FW_N; ++n) {
        float aI = 0.0f, aQ = 0.0f;
#pragma unroll
        for (int m = 0; m < FW_M; ++m) {
            float p = x[m].x * fw_coef[n][m];
            aI = aI + p;
            p = x[m].y * fw_coef[n][m];
            aQ = aQ + p;
        }
        A[(size_t)j * FW_N + n] = make_float2(aI, aQ);
    }
}

// pass 2: output j (dumped when segment j closes) = R (((((A5(j-5) + A4(j-4)) + A3(j-3)) + A2(j-2)) + A1(j-1)) + A0(j)):
// the order in which the reference's delay line y[] picks the contributions up.  n_done segments are complete (the last
// segment of a call may still be open).
__global__ __launch_bounds__(256) void k_farrow_outputs(const float2 *A, int n_done, const State *st, short2 *ring, int64_t ring_len, int64_t wr,
                                                        float R, float gain, uint32_t *peak)
{
    const int j = blockIdx.x * 256 + threadIdx.x;
    short2 o16 = make_short2(0, 0);
    if (j < n_done) {
        float yI = 0.0f, yQ = 0.0f;
#pragma unroll
        for (int i = FW_N - 1; i >= 0; --i) {
            const int s = j - i;                          // segment whose polynomial i lands in this output
            const float2 a = s >= 0 ? A[(size_t)s * FW_N + i] : st->fw_a[FW_N - 1 + s][i];
            yI = yI + a.x; yQ = yQ + a.y;
        }
        o16 = to_s16(make_float2(R * yI, R * yQ), gain);
        ring_store(ring, ring_len, wr + j, o16);
    }
    note_peak(peak, o16, j < n_done);
}

// carry-over for the next call: A of the last five finished segments and the integrators of the open one
__global__ void k_farrow_tail(const float2 *A, int n_done, State *st, const float2 *x_open)
{
    const int t = threadIdx.x;                            // 64 threads
    float2 v = make_float2(0.0f, 0.0f);
    const int r = t / FW_N, n = t % FW_N;                 // row r of the new fw_a: segment n_done - 5 + r
    if (t < (FW_N - 1) * FW_N) {
        const int s = n_done - (FW_N - 1) + r;
        v = s >= 0 ? A[(size_t)s * FW_N + n] : st->fw_a[FW_N - 1 + s][n];
    }
    float2 xo = make_float2(0.0f, 0.0f);
    if (t < FW_M) xo = x_open[t];
    __syncthreads();
    if (t < (FW_N - 1) * FW_N) st->fw_a[r][n] = v;
    if (t < FW_M) st->fw_x[t] = xo;
}

// integrators of the open (last, unfinished) segment of a call, continued from the carried state when it is segment 0
template <int FMT>
__global__ void k_farrow_open(const void *in, const int32_t *seg, const float *mu, int j, int64_t n_in, const State *st, float2 *x_open)
{
    if (threadIdx.x != 0) return;
    float2 x[FW_M];
#pragma unroll
    for (int m = 0; m < FW_M; ++m) x[m] = j == 0 ? st->fw_x[m] : make_float2(0.0f, 0.0f);
    for (int64_t k = seg[j]; k < n_in; ++k) {
        float2 v = in_sample<FMT>(in, k);
        const float u = mu[k];
        x[0].x = x[0].x + v.x; x[0].y = x[0].y + v.y;
#pragma unroll
        for (int m = 1; m < FW_M; ++m) {
            v.x = v.x * u; v.y = v.y * u;
            x[m].x = x[m].x + v.x; x[m].y = x[m].y + v.y;
        }
    }
#pragma unroll
    for (int m = 0; m < FW_M; ++m) x_open[m] = x[m];
}

// The converters' signal level output (device gain control of the reference's SDR inputs: inputdevicesrc.h:60-75): a rectifier with
// fast attack and slow release on |x|^2 of the INPUT samples — every sample (Farrow inputdevicesrc.cpp:282-292, pass-through :330-341) or
// every second one (half-band :154-173: the odd samples) —
//     c = |x|^2 > level ? c_attack : c_release;   level = (c |x|^2 + level) - c level
// in binary32, each operation rounded: a serial recursion whose every step depends on the one before, so ONE wave runs it: its
// lanes square 64 samples at a time, then the recursion takes them in order (v_readlane).  Off the decode path and slow by
// nature (a few ms per frame of input), hence only when asked for: dabx_enable_level.
template <int FMT>
__global__ void k_level(const void *in, int64_t n, int first, int stride, float catt, float crel, State *st)
{
    const int lane = threadIdx.x;                         // 64 threads
    float level = st->level;
    for (int64_t base = first; base < n; base += (int64_t)64 * stride) {
        const int64_t k = base + (int64_t)lane * stride;
        float a = 0.0f;
        if (k < n) {
            const float2 v = in_sample<FMT>(in, k);
            const float q2 = v.y * v.y;
            a = v.x * v.x;
            a = a + q2;
        }
        const int64_t left = (n - base + stride - 1) / stride;
        const int cnt = left < 64 ? (int)left : 64;
        for (int i = 0; i < cnt; ++i) {
            const float x = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(a), i));
            const float c = x > level ? catt : crel;
            float p = c * x;
            const float b = c * level;
            p = p + level;
            level = p - b;
        }
    }
    if (lane == 0) st->level = level;
}

// ---- Transposed Farrow, one launch, for input rates up to 4096 kHz (R = 2048 kHz / rate >= 0.5).
// There the reference's serial float recursion `mu -= R; if (mu < 0) { dump; mu += 1 }` (inputdevicesrc.cpp:241-245) is EXACT:
// mu and R are multiples of 2^-24 below 1, so is every difference and every sum the recursion forms, and a float holds them
// all.  With M = mu 2^24 and Ri = R 2^24 it is integer arithmetic modulo 2^24:
//     mu after sample k      = ((M0 - (k + 1) Ri) mod 2^24) 2^-24
//     dump number j (j >= 1) falls on sample floor((M0 + (j - 1) 2^24) / Ri)         (tests/test_resampler.py checks both)
// so every thread finds its own segment and the schedule needs no host loop and no upload.  One thread per output: it
// integrates its segment and evaluates the six polynomials (pass 1 of the two-kernel form) into LDS, then adds the six
// contributions of its output in the reference's order (pass 2).  The five segments before the block are recomputed by the
// block (or come from the carried state).
struct Sched { uint32_t M0, Ri; };
__device__ __forceinline__ int64_t farrow_seg(const Sched &sc, int64_t j)          // first sample of segment j
{
    if (j <= 0) return 0;
    const uint64_t num = (uint64_t)sc.M0 + ((uint64_t)(j - 1) << 24);
    int64_t k = (int64_t)((double)num / (double)sc.Ri);                            // off by one at most: corrected exactly
    while ((uint64_t)(k + 1) * sc.Ri <= num) ++k;
    while ((uint64_t)k * sc.Ri > num) --k;
    return k;
}
__device__ __forceinline__ float farrow_mu(const Sched &sc, int64_t k)             // mu sample k is integrated with
{
    const uint64_t v = (uint64_t)sc.M0 - (uint64_t)(k + 1) * sc.Ri;
    return (float)(uint32_t)(v & 0xFFFFFFu) * 5.9604644775390625e-08f;             // 2^-24: exact
}

// integrators of segment j over samples [k0, k1), continued from the carried state when j = 0; then A_n = sum_m x[m] coef[n][m]
template <int FMT>
__device__ __forceinline__ void farrow_segment(const void *in, const Sched &sc, int64_t j, int64_t k0, int64_t k1, const State *st, float2 x[FW_M])
{
#pragma unroll
    for (int m = 0; m < FW_M; ++m) x[m] = j == 0 ? st->fw_x[m] : make_float2(0.0f, 0.0f);
    for (int64_t k = k0; k < k1; ++k) {
        float2 v = in_sample<FMT>(in, k);
        const float u = farrow_mu(sc, k);
        x[0].x = x[0].x + v.x; x[0].y = x[0].y + v.y;
#pragma unroll
        for (int m = 1; m < FW_M; ++m) {
            v.x = v.x * u; v.y = v.y * u;
            x[m].x = x[m].x + v.x; x[m].y = x[m].y + v.y;
        }
    }
}
__device__ __forceinline__ void farrow_poly(const float2 x[FW_M], float2 A[FW_N])
{
#pragma unroll
    for (int n = 0; n < FW_N; ++n) {
        float aI = 0.0f, aQ = 0.0f;
#pragma unroll
        for (int m = 0; m < FW_M; ++m) {
            float p = x[m].x * fw_coef[n][m];
            aI = aI + p;
            p = x[m].y * fw_coef[n][m];
            aQ = aQ + p;
        }
        A[n] = make_float2(aI, aQ);
    }
}

template <int FMT>
__global__ __launch_bounds__(256) void k_farrow_fused(const void *in, Sched sc, int n_done, const State *st, short2 *ring, int64_t ring_len, int64_t wr,
                                                      float R, float gain, uint32_t *peak)
{
    __shared__ float2 A[256 + FW_N - 1][FW_N];
    const int t = threadIdx.x;
    const int64_t j0 = (int64_t)blockIdx.x * 256;
    for (int row = t; row < 256 + FW_N - 1; row += 256) {
        const int64_t j = j0 - (FW_N - 1) + row;
        float2 a[FW_N];
        if (j < 0) {
#pragma unroll
            for (int n = 0; n < FW_N; ++n) a[n] = st->fw_a[FW_N - 1 + j][n];
        } else if (j < n_done) {
            float2 x[FW_M];
            farrow_segment<FMT>(in, sc, j, farrow_seg(sc, j), farrow_seg(sc, j + 1), st, x);
            farrow_poly(x, a);
        } else {
#pragma unroll
            for (int n = 0; n < FW_N; ++n) a[n] = make_float2(0.0f, 0.0f);
        }
#pragma unroll
        for (int n = 0; n < FW_N; ++n) A[row][n] = a[n];
    }
    __syncthreads();
    const int64_t j = j0 + t;
    short2 o16 = make_short2(0, 0);
    if (j < n_done) {
        float yI = 0.0f, yQ = 0.0f;
#pragma unroll
        for (int i = FW_N - 1; i >= 0; --i) {                     // y = ((((A5(j-5) + A4(j-4)) + A3(j-3)) + A2(j-2)) + A1(j-1)) + A0(j)
            const float2 a = A[t + (FW_N - 1) - i][i];
            yI = yI + a.x; yQ = yQ + a.y;
        }
        o16 = to_s16(make_float2(R * yI, R * yQ), gain);
        ring_store(ring, ring_len, wr + j, o16);
    }
    note_peak(peak, o16, j < n_done);
}

// carry-over of the fused form: the polynomial outputs of the last five finished segments and the integrators of the open one,
// recomputed from the input (one small workgroup, after k_farrow_fused)
template <int FMT>
__global__ void k_farrow_finish(const void *in, Sched sc, int n_done, int64_t n_in, State *st)
{
    const int t = threadIdx.x;                            // 64 threads
    float2 a[FW_N], xo[FW_M];
    if (t < FW_N - 1) {                                   // row t of the new fw_a: segment n_done - 5 + t
        const int64_t j = (int64_t)n_done - (FW_N - 1) + t;
        if (j < 0) {
#pragma unroll
            for (int n = 0; n < FW_N; ++n) a[n] = st->fw_a[FW_N - 1 + j][n];
        } else {
            float2 x[FW_M];
            farrow_segment<FMT>(in, sc, j, farrow_seg(sc, j), farrow_seg(sc, j + 1), st, x);
            farrow_poly(x, a);
        }
    }
    if (t == FW_N - 1) farrow_segment<FMT>(in, sc, n_done, farrow_seg(sc, n_done), n_in, st, xo);
    __syncthreads();
    if (t < FW_N - 1) {
#pragma unroll
        for (int n = 0; n < FW_N; ++n) st->fw_a[t][n] = a[n];
    }
    if (t == FW_N - 1) {
#pragma unroll
        for (int m = 0; m < FW_M; ++m) st->fw_x[m] = xo[m];
    }
}

}  // namespace rs
