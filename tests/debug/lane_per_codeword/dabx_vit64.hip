// dabx_vit64.hip — lane-per-codeword Viterbi for groups of 64 codewords with one protection
// profile (included by dabx_api.hip after dabx_kernels.hip).
//
// k_viterbi (one wave per codeword, one state per lane) pays ~8 VALU issues per trellis step for
// 64 states.  When 64 codewords share a profile — the usual case: the same sub-channel in many
// CIFs/streams, and all FIC codewords — it is cheaper to give every LANE its own codeword and
// keep the 64 path metrics of that codeword in the lane's registers: a step is then 32 in-place
// butterflies of 10 plain VALU ops (4 add/sub, 2 max, 2 sub + 2 alignbit for the decisions), no
// cross-lane traffic, no LDS, and 64 trellis steps per wave-step.  The price is memory: soft
// values must arrive transposed ([step][lane], k_xgather) and the 64 decision bits per step go to
// an HBM scratch ([step][lane] uint2) that the traceback kernel reads back.
// Same arithmetic as the oracle: int32 correlation metrics, start state 0, same tie rule.
#include "dabx_vit64_steps.inc"

struct DevGroup {          // 64 codewords, one per lane
    uint32_t nsteps, n_in, info_off;
    uint32_t first_cw;     // index of lane 0's codeword in the DevWork array of grouped codewords
    uint64_t x_off;        // dwords into xbuf:   x[(x_off + t*64 + lane)]
    uint64_t d_off;        // uint2  into decbuf: d[(d_off + t*64 + lane)]
};

namespace {

// source and destination of one codeword; returns false when there is nothing to decode
__device__ __forceinline__ bool cw_setup(const DevCtx &C, const DevWork &w, VitSrc &src, uint8_t *&out)
{
    const DevState &st = C.state[w.stream];
    if (w.sub < 0) {
        src = {C.fic_soft + ((size_t)w.stream * C.max_frames + w.frame) * FICBITS + w.c * 2304, 0, -1};
        out = C.fib + (((size_t)w.stream * C.max_frames + w.frame) * 12 + 3 * w.c) * 32;
        return !st.acq_fail;
    }
    const DevSub &sc = C.sub[(size_t)w.stream * 64 + w.sub];
    const int64_t r = st.cif + 4 * (int64_t)w.frame + w.c - 15;
    src = {C.ti + (size_t)w.stream * C.ti_slots * CIFBITS + (sc.start_bit >> 4), r, C.ti_slots - 1};
    out = C.msc + (((size_t)w.stream * C.max_frames + w.frame) * 4 + w.c) * (size_t)C.msc_stride + sc.out_off;
    return r >= 0 && !st.acq_fail;
}

}  // namespace

// ---- soft values of 64 steps x 64 codewords, depunctured and time de-interleaved, transposed
// through LDS to [step][lane].  One workgroup per (group, block of 64 steps).
__global__ __launch_bounds__(256) void k_xgather(DevCtx C, const DevGroup *__restrict__ groups, const DevWork *__restrict__ cws,
                                                 const uint32_t *__restrict__ blk_group, const uint32_t *__restrict__ blk_index,
                                                 int *__restrict__ xbuf)
{
    __shared__ int tile[64][65];
    const DevGroup g = groups[blk_group[blockIdx.x]];
    const int blk = blk_index[blockIdx.x], t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int tau = blk * 64 + lane;
    const uint32_t w = step_word(C.stepinfo + g.info_off, tau, g.nsteps);
    // the loads of all 16 codewords are requested before any is consumed; punctured positions are
    // predicated off (exec-masked loads), nothing is written to LDS until every byte has arrived
    const uint32_t off = w >> 4;
    const uint32_t o1 = off + ((w >> 3) & 1), o2 = o1 + ((w >> 2) & 1), o3 = o2 + ((w >> 1) & 1);
    int xs[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        VitSrc src; uint8_t *out;
        const bool ok = cw_setup(C, cws[g.first_cw + wave * 16 + k], src, out);
        int b0 = 0, b1 = 0, b2 = 0, b3 = 0;
        if (ok && (w & 8)) b0 = (uint8_t)soft_at(src, off);
        if (ok && (w & 4)) b1 = (uint8_t)soft_at(src, o1);
        if (ok && (w & 2)) b2 = (uint8_t)soft_at(src, o2);
        if (ok && (w & 1)) b3 = (uint8_t)soft_at(src, o3);
        xs[k] = b0 | (b1 << 8) | (b2 << 16) | (b3 << 24);
    }
#pragma unroll
    for (int k = 0; k < 16; ++k) tile[wave * 16 + k][lane] = xs[k];
    __syncthreads();
    int *dst = xbuf + g.x_off + (size_t)blk * 64 * 64;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int idx = i * 256 + t, s = idx >> 6, cwl = idx & 63;
        if (blk * 64 + s < (int)g.nsteps) dst[idx] = tile[cwl][s];
    }
}

// ---- forward pass: one wave per group, lane = codeword
#define VIT64_BFP(a, b, w)                                                                       \
    {                                                                                            \
        const int E = R[a], O = R[b];                                                            \
        const int A0 = E + W[w], B0 = O - W[w], A1 = O + W[w], B1 = E - W[w];                    \
        VIT64_PUSH(A0 - B0, A1 - B1)                                                             \
        R[a] = max(A0, B0); R[b] = max(A1, B1);                                                  \
    }
#define VIT64_BFM(a, b, w)                                                                       \
    {                                                                                            \
        const int E = R[a], O = R[b];                                                            \
        const int A0 = E - W[w], B0 = O + W[w], A1 = O - W[w], B1 = E + W[w];                    \
        VIT64_PUSH(A0 - B0, A1 - B1)                                                             \
        R[a] = max(A0, B0); R[b] = max(A1, B1);                                                  \
    }
// decisions are shifted into four accumulators per word in rotation (independent chains of 8
// instead of one chain of 32); `q` counts butterflies, 31 - q is the state index j
#define VIT64_PUSH(d0, d1)                                                                       \
    {                                                                                            \
        aL[q >> 3] = __builtin_amdgcn_alignbit(aL[q >> 3], (uint32_t)(d0), 31);                  \
        aH[q >> 3] = __builtin_amdgcn_alignbit(aH[q >> 3], (uint32_t)(d1), 31);                  \
        ++q;                                                                                     \
    }

// the eight branch metrics with +x0: W[c] = x0 +- x1 +- x2 +- x3, bit (2-q) of c set = minus x_{q+1}
#define VIT64_METRICS(xw)                                                                        \
    {                                                                                            \
        const int x0 = (int)(int8_t)(xw), x1 = (int)(int8_t)((xw) >> 8), x2 = (int)(int8_t)((xw) >> 16), x3 = (xw) >> 24; \
        const int p01 = x0 + x1, m01 = x0 - x1, p23 = x2 + x3, m23 = x2 - x3;                    \
        W[0] = p01 + p23; W[1] = p01 + m23; W[2] = p01 - m23; W[3] = p01 - p23;                  \
        W[4] = m01 + p23; W[5] = m01 + m23; W[6] = m01 - m23; W[7] = m01 - p23;                  \
    }

__global__ __launch_bounds__(64) void k_vit64_fwd(const DevGroup *__restrict__ groups, const int *__restrict__ xbuf,
                                                  uint2 *__restrict__ decbuf)
{
    const DevGroup g = groups[blockIdx.x];
    const int lane = threadIdx.x;
    const int *x = xbuf + g.x_off + lane;
    uint2 *d = decbuf + g.d_off + lane;
    const int T = (int)g.nsteps;
    int R[64];
#pragma unroll
    for (int i = 0; i < 64; ++i) R[i] = PM_INIT;
    R[0] = 0;
    int W[8];
    // Soft values are prefetched 12 steps ahead through a register ring.  Loads and their waits are
    // inline asm with a COUNTED vmcnt: every steady-state step issues one load and one decision store,
    // so the load consumed now is followed by 11 loads + 12 stores = 23 younger operations.  (Left to
    // the compiler the waits degrade to vmcnt(2..9), i.e. every step waits for an HBM store.)
    int xq[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) {
        const int *p = x + (size_t)min(k, T - 1) * 64;
        asm volatile("global_load_dword %0, %1, off" : "=v"(xq[k]) : "v"(p) : "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    for (int t0 = 0; t0 < T; t0 += 12) {
#define VIT64_ONE(K, PH)                                                                         \
        if (t0 + K < T) {                                                                        \
            uint32_t aL[4] = {0, 0, 0, 0}, aH[4] = {0, 0, 0, 0};                                 \
            int q = 0;                                                                           \
            if (t0) asm volatile("s_waitcnt vmcnt(23)" : "+v"(xq[K]));                           \
            VIT64_METRICS(xq[K])                                                                 \
            {                                                                                    \
                const int *p = x + (size_t)min(t0 + K + 12, T - 1) * 64;                         \
                asm volatile("global_load_dword %0, %1, off" : "=v"(xq[K]) : "v"(p) : "memory"); \
            }                                                                                    \
            VIT64_STEP_##PH                                                                      \
            /* butterflies ran j = 31..0: aL[0] holds states 31..24 (state 24 in bit 0) ... aL[3] states 7..0 */ \
            const uint32_t wlo = (aL[0] << 24) | (aL[1] << 16) | (aL[2] << 8) | aL[3];           \
            const uint32_t whi = (aH[0] << 24) | (aH[1] << 16) | (aH[2] << 8) | aH[3];           \
            d[(size_t)(t0 + K) * 64] = make_uint2(wlo, whi);                                     \
        }
        VIT64_ONE(0, 0) VIT64_ONE(1, 1) VIT64_ONE(2, 2) VIT64_ONE(3, 3) VIT64_ONE(4, 4) VIT64_ONE(5, 5)
        VIT64_ONE(6, 0) VIT64_ONE(7, 1) VIT64_ONE(8, 2) VIT64_ONE(9, 3) VIT64_ONE(10, 4) VIT64_ONE(11, 5)
#undef VIT64_ONE
    }
}

// ---- traceback: one wave per group, every lane walks its own codeword back from state 0.
// The decision words of 32 steps are fetched ahead of the 32 dependent bit-picks (the loads do
// not depend on the path), double buffered.
namespace {
__device__ __forceinline__ void tb_step(uint32_t &s, uint32_t &o, const uint2 w)
{
    const uint32_t u = s >> 5;
    const uint32_t word = (s & 32) ? w.y : w.x;
    const uint32_t dec = (word >> (s & 31)) & 1u;
    o = __builtin_amdgcn_alignbit(u, o, 1);                      // (o >> 1) | (u << 31): step 32h+j ends at bit 31-j
    s = (((s << 1) & 63u) | u) ^ dec;
}
}  // namespace

__global__ __launch_bounds__(64) void k_vit64_tb(DevCtx C, const DevGroup *__restrict__ groups, const DevWork *__restrict__ cws,
                                                 const uint2 *__restrict__ decbuf)
{
    const DevGroup g = groups[blockIdx.x];
    const int lane = threadIdx.x;
    VitSrc src; uint8_t *out;
    const bool ok = cw_setup(C, cws[g.first_cw + lane], src, out);
    const uint2 *d = decbuf + g.d_off + lane;
    const int nwords = (int)g.n_in >> 5;                         // nsteps = 32 nwords + 6
    uint32_t s = 0, o = 0;
    uint2 buf[32];
#pragma unroll
    for (int j = 0; j < 32; ++j) buf[j] = d[(size_t)((nwords - 1) * 32 + j) * 64];     // top full word, issued first
    {   // the six tail steps
        uint2 tail[6];
#pragma unroll
        for (int j = 0; j < 6; ++j) tail[j] = d[(size_t)(nwords * 32 + j) * 64];
#pragma unroll
        for (int j = 5; j >= 0; --j) tb_step(s, o, tail[j]);
    }
    for (int h = nwords - 1; h >= 0; --h) {
        uint2 cur[32];
#pragma unroll
        for (int j = 0; j < 32; ++j) cur[j] = buf[j];
        if (h > 0) {
#pragma unroll
            for (int j = 0; j < 32; ++j) buf[j] = d[(size_t)((h - 1) * 32 + j) * 64];
        }
        o = 0;
#pragma unroll
        for (int j = 31; j >= 0; --j) tb_step(s, o, cur[j]);
        if (ok) reinterpret_cast<uint32_t *>(out)[h] = __builtin_bswap32(o ^ C.prbs[h]);
    }
}
