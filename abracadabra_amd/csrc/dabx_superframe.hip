// dabx_superframe.hip — DAB+ audio super frames on the GPU (ETSI TS 102 563 §5, §6), the stage after the
// Viterbi decoder for sub-channels that carry HE-AAC: fire code synchronisation, RS(120,110) correction
// and access-unit CRCs for every flagged sub-channel of every stream in one launch.
//
// Replaces the super frame layer inside the reference's closed dabsdr library (it exports Karn's
// init_rs_char / decode_rs_char; output contract: dabsdrAudioCBFunc_t, dabsdr.h:47-78,
// consumer src/audiodecoder.cpp:183-208).  Oracle: oracle/dab_plus.c (dab_sf_push), same records bit for bit.
//
// One workgroup per (stream, sub-channel).  The logical frames of a step (the sub-channel's slice of every
// valid CIF record k_viterbi wrote) are appended to the <= 4 frames carried over from the step before;
// a window of five frames slides over that sequence exactly as a serial receiver would: decoded -> advance
// by five, rejected -> advance by one.  While the stream is synchronised up to SF_BATCH windows (positions
// i, i+5, ...) are worked on together: 10 s syndromes per window as independent look-up sums, one thread per
// code word for the (rare) Berlekamp-Massey / Chien / Forney correction, the access-unit CRCs in 32-byte chunks.
// The windows up to the first one that fails its fire code are accepted; a failing window is then taken up
// again on its own, so the outcome is exactly the serial receiver's.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "dabx_dev.h"

namespace {

constexpr int CRC_CHUNK = 32;
constexpr int SF_BATCH = 4;         // windows per pass while synchronised
constexpr int SF_WIN = 2880 + 16;   // LDS bytes per window
constexpr int SF_THREADS = 128;      // two waves: at most 96 chunks / 10 s syndromes are in flight at once, and more workgroups fit a CU

struct Gf {                          // GF(2^8), p(x) = x^8 + x^4 + x^3 + x^2 + 1, tables in LDS
    const uint8_t *exp;              // [512]
    const uint8_t *log;              // [256]
    __device__ uint8_t mul(uint8_t a, uint8_t b) const { return (a && b) ? exp[log[a] + log[b]] : 0; }
    __device__ uint8_t div(uint8_t a, uint8_t b) const { return a ? exp[log[a] + 255 - log[b]] : 0; }
    __device__ uint8_t pow(int e) const { return exp[((e % 255) + 255) % 255]; }
};

// Scratch of the (rare) error correction of one code word, in LDS rather than registers: kept in VGPRs these
// arrays would cap how many workgroups fit a CU for a path that almost never runs.
struct RsWork { uint8_t C[11], B[11], Tp[11], Om[10], fix[5], pos[5]; };

// Correct one code word whose syndromes S[0..9] are not all zero.  cw(k) = sf[j + k s], k = 0 the highest
// power.  Returns the number of corrected bytes, -1 when uncorrectable (nothing written then).
__device__ __noinline__ int rs_correct(const Gf &G, uint8_t *sf, int j, int s, const uint8_t *S, RsWork &w)
{
    constexpr int N = 120, T2 = 10;
    uint8_t b = 1;
    for (int i = 0; i <= T2; ++i) { w.C[i] = i == 0; w.B[i] = i == 0; }
    int L = 0, m = 1;
    for (int n = 0; n < T2; ++n) {                       // Berlekamp-Massey
        uint8_t d = S[n];
        for (int i = 1; i <= L; ++i) d ^= G.mul(w.C[i], S[n - i]);
        if (!d) { ++m; continue; }
        for (int i = 0; i <= T2; ++i) w.Tp[i] = w.C[i];
        const uint8_t coef = G.div(d, b);
        for (int i = 0; i + m <= T2; ++i) w.C[i + m] ^= G.mul(coef, w.B[i]);
        if (2 * L <= n) {
            L = n + 1 - L;
            for (int i = 0; i <= T2; ++i) w.B[i] = w.Tp[i];
            b = d; m = 1;
        } else ++m;
    }
    if (L > T2 / 2) return -1;
    int nerr = 0;
    for (int k = 0; k < N; ++k) {                        // Chien search over the 120 positions of the shortened code
        const int p = N - 1 - k;
        uint8_t v = 0;
        for (int i = 0; i <= L; ++i) v ^= G.mul(w.C[i], G.pow(-p * i));
        if (!v) { if (nerr == T2 / 2) return -1; w.pos[nerr++] = static_cast<uint8_t>(k); }
    }
    if (nerr != L) return -1;
    for (int i = 0; i < T2; ++i) {
        uint8_t o = 0;
        for (int q = 0; q <= L && q <= i; ++q) o ^= G.mul(S[i - q], w.C[q]);
        w.Om[i] = o;
    }
    for (int e = 0; e < nerr; ++e) {                     // Forney, first consecutive root alpha^0
        const int p = N - 1 - w.pos[e];
        const uint8_t Xinv = G.pow(-p);
        uint8_t num = 0, den = 0;
        for (int i = T2 - 1; i >= 0; --i) num = static_cast<uint8_t>(G.mul(num, Xinv) ^ w.Om[i]);
        for (int i = 1; i <= L; i += 2) den ^= G.mul(w.C[i], G.pow(-p * (i - 1)));
        if (!den) return -1;
        w.fix[e] = G.mul(G.div(num, den), G.pow(p));
    }
    for (int e = 0; e < nerr; ++e)
        if (w.pos[e] < 110) sf[j + w.pos[e] * s] ^= w.fix[e];   // only the data part is kept (parity bytes are dropped anyway)
    return nerr;
}

}  // namespace

// max_rec: record slots per sub-channel and step
__global__ __launch_bounds__(SF_THREADS, 4) void k_superframe(DevCtx C, const DevSfSub *__restrict__ subs, DevSfState *state, DevSfRec *recs,
                                                    uint8_t *data, const uint8_t *__restrict__ gf_tab, int n_frames, int max_rec)
{
    __shared__ uint8_t t_exp[512], t_log[256];
    __shared__ uint8_t t_mul[9 * 256];                   // v * alpha^r, r = 1..9
    __shared__ uint16_t t_crc[256], t_fire[256], t_shift[16];
    __shared__ uint16_t ch_crc[SF_BATCH * 96];           // CRC of every 32-byte chunk of the access units
    __shared__ int16_t ch_first[SF_BATCH * 8 + 1];       // first chunk of access unit a of window w at [8 w + a]; [8 nb] = total
    __shared__ __attribute__((aligned(16))) uint8_t sfa[SF_BATCH * SF_WIN];
    __shared__ uint8_t synd[SF_BATCH * 240];
    __shared__ RsWork rs_work[24];
    __shared__ int res[SF_BATCH * 24];
    __shared__ int sh_ok[SF_BATCH], sh_num[SF_BATCH], sh_acc;
    __shared__ uint16_t sh_start[SF_BATCH * 8];
    __shared__ int sh_auflag[SF_BATCH * 8];

    const int t = threadIdx.x;
    const DevSfSub sb = subs[blockIdx.x];
    DevSfState &stt = state[blockIdx.x];
    const DevState &st = C.state[sb.stream];
    const int s = sb.s, fb = sb.frame_bytes, sfb = 5 * fb;

    for (int i = t; i < 512; i += SF_THREADS) t_exp[i] = gf_tab[i];
    for (int v = t; v < 256; v += SF_THREADS) {
        t_log[v] = gf_tab[512 + v];
        unsigned c = static_cast<unsigned>(v) << 8, f = c;
        for (int b = 0; b < 8; ++b) {
            c = (c & 0x8000) ? ((c << 1) ^ 0x1021) & 0xFFFF : (c << 1) & 0xFFFF;
            f = (f & 0x8000) ? ((f << 1) ^ 0x782F) & 0xFFFF : (f << 1) & 0xFFFF;
        }
        t_crc[v] = static_cast<uint16_t>(c);
        t_fire[v] = static_cast<uint16_t>(f);
    }
    __syncthreads();
    for (int idx = t; idx < 9 * 256; idx += SF_THREADS) {
        const int r = (idx >> 8) + 1, v = idx & 255;
        t_mul[idx] = v ? t_exp[t_log[v] + r] : 0;
    }
    if (t < 16) {                                        // x^t * x^(8*32) mod the CRC polynomial: moves a CRC past a 32-byte chunk
        unsigned c = 1u << t;
        for (int k = 0; k < CRC_CHUNK; ++k) c = ((c << 8) ^ t_crc[(c >> 8) & 0xFF]) & 0xFFFF;
        t_shift[t] = static_cast<uint16_t>(c);
    }
    __syncthreads();
    const Gf G = {t_exp, t_log};

    // the valid CIFs of this step are a suffix (the time de-interleaver fills at the start of a stream)
    int first_valid = 4 * n_frames;
    if (!st.acq_fail) {
        const int64_t need = 15 - st.cif;
        first_valid = need <= 0 ? 0 : (need >= 4 * n_frames ? 4 * n_frames : static_cast<int>(need));
    }
    const int carry = stt.carry, n_new = 4 * n_frames - first_valid, total = carry + n_new;
    const uint32_t base = stt.frames_seen - static_cast<uint32_t>(carry);
    const uint8_t *msc = C.msc + (size_t)sb.stream * C.max_frames * 4 * C.msc_stride + sb.msc_off;
    auto frame_ptr = [&](int q) -> const uint8_t * {     // logical frame q of the sequence
        return q < carry ? stt.buf + q * fb : msc + (size_t)(first_valid + q - carry) * C.msc_stride;
    };

    int i = 0, out = 0, synced = stt.synced;
    uint32_t n_sf = 0, n_auok = 0, n_aubad = 0, n_corr = 0, n_fail = 0, n_loss = 0;      // meaningful in thread 0
    const int wpf = fb >> 2, wpw = 5 * wpf;              // 32-bit words per frame / per window (frames are multiples of 24 bytes)
    while (i + 5 <= total) {
        const int nb = synced ? min(SF_BATCH, (total - i) / 5) : 1;
        // ---- windows i, i+5, ... into LDS, six loads in flight per thread
        for (int b0 = 0; b0 < nb * wpw; b0 += 6 * SF_THREADS) {
            uint32_t wv[6];
#pragma unroll
            for (int u = 0; u < 6; ++u) {
                const int idx = b0 + t + SF_THREADS * u;
                if (idx < nb * wpw) {
                    const int f = idx / wpf, o = idx - f * wpf;          // f counts frames from i on: windows are contiguous
                    wv[u] = reinterpret_cast<const uint32_t *>(frame_ptr(i + f))[o];
                }
            }
#pragma unroll
            for (int u = 0; u < 6; ++u) {
                const int idx = b0 + t + SF_THREADS * u;
                if (idx < nb * wpw) {
                    const int w = idx / wpw;
                    reinterpret_cast<uint32_t *>(sfa + w * SF_WIN)[idx - w * wpw] = wv[u];
                }
            }
        }
        __syncthreads();
        // ---- RS(120,110) syndromes S_r = sum_k c_k alpha^(r (119 - k)), r = 0..9, of code word j of window w.  Four lanes
        // share a code word, 30 bytes each: a lane runs the ten Horner chains of its bytes side by side (one look-up in the
        // "times alpha^r" table and one xor per term: ten independent chains hide the look-up latency; S_0 is a plain xor),
        // moves its partial results to their place in the code word (times alpha^(30 r (3 - segment))) and the four are
        // xor-ed across the lanes.  (Before: one log and one antilog look-up per term, 20 per byte instead of 10.)
        for (int q = t; q < nb * s * 4; q += SF_THREADS) {
            const int cw = q >> 2, seg = q & 3, w = cw / s, j = cw - w * s;
            const uint8_t *sf = sfa + w * SF_WIN + j + 30 * seg * s;
            unsigned a[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
            for (int k = 0; k < 30; ++k) {
                const unsigned cb = sf[k * s];
                a[0] ^= cb;
#pragma unroll
                for (int r = 1; r < 10; ++r) a[r] = t_mul[(r - 1) * 256 + a[r]] ^ cb;
            }
#pragma unroll
            for (int r = 1; r < 10; ++r) a[r] = a[r] ? t_exp[t_log[a[r]] + (30 * r * (3 - seg)) % 255] : 0u;
#pragma unroll
            for (int r = 0; r < 10; ++r) {
                a[r] ^= (unsigned)__builtin_amdgcn_update_dpp(0, (int)a[r], 0xB1, 0xf, 0xf, false);     // quad_perm [1,0,3,2]
                a[r] ^= (unsigned)__builtin_amdgcn_update_dpp(0, (int)a[r], 0x4E, 0xf, 0xf, false);     // quad_perm [2,3,0,1]
            }
            if (seg == 0) {
#pragma unroll
                for (int r = 0; r < 10; ++r) synd[w * 240 + 10 * j + r] = static_cast<uint8_t>(a[r]);
            }
        }
        __syncthreads();
        if (t < s)
            for (int w = 0; w < nb; ++w) {
                const uint8_t *S = synd + w * 240 + 10 * t;
                bool clean = true;
                for (int r = 0; r < 10; ++r) clean = clean && S[r] == 0;
                res[w * 24 + t] = clean ? 0 : rs_correct(G, sfa + w * SF_WIN, t, s, S, rs_work[t]);
            }
        __syncthreads();
        if (t < nb) {                                     // one window per thread: fire code over bytes 2..10, header
            const uint8_t *sf = sfa + t * SF_WIN;
            unsigned c = 0;
            for (int k = 2; k < 11; ++k) c = ((c << 8) ^ t_fire[((c >> 8) ^ sf[k]) & 0xFF]) & 0xFFFF;
            const bool ok = c == ((static_cast<unsigned>(sf[0]) << 8) | sf[1]) && !(sf[0] == 0 && sf[1] == 0 && sf[2] == 0);
            sh_ok[t] = ok;
            const int dac = (sf[2] >> 6) & 1, sbr = (sf[2] >> 5) & 1;
            const int num = dac ? (sbr ? 3 : 6) : (sbr ? 2 : 4);
            sh_num[t] = num;
            uint16_t *st8 = sh_start + 8 * t;
            st8[0] = static_cast<uint16_t>(dac ? (sbr ? 6 : 11) : (sbr ? 5 : 8));
            for (int a = 1; a < num; ++a) {               // 12-bit big-endian fields packed from byte 3
                const int bit = 24 + 12 * (a - 1), byte = bit >> 3;
                st8[a] = static_cast<uint16_t>((bit & 4) ? (((sf[byte] & 0x0F) << 8) | sf[byte + 1]) : ((sf[byte] << 4) | (sf[byte + 1] >> 4)));
            }
            st8[num] = static_cast<uint16_t>(110 * s);
            for (int a = num + 1; a < 8; ++a) st8[a] = 0;
        }
        __syncthreads();
        if (t == 0) {                                     // accepted: the windows before the first fire code failure
            int acc = 0;
            while (acc < nb && sh_ok[acc]) ++acc;
            sh_acc = acc;
            int nch = 0;                                  // chunks of the access units with sane bounds
            for (int w = 0; w < acc; ++w) {
                const uint16_t *st8 = sh_start + 8 * w;
                for (int a = 0; a < 8; ++a) {
                    ch_first[8 * w + a] = static_cast<int16_t>(nch);
                    if (a >= sh_num[w]) continue;
                    const int a0 = st8[a], a1 = st8[a + 1], len = a1 - a0;
                    if (!(a0 < st8[0] || len < 3 || a1 > 110 * s)) nch += (len - 2 + CRC_CHUNK - 1) / CRC_CHUNK;
                }
            }
            ch_first[8 * acc] = static_cast<int16_t>(nch);
        }
        __syncthreads();
        const int acc = sh_acc;
        if (acc == 0) {                                   // not a super frame boundary (or damaged): slide by one frame
            if (t == 0 && synced) ++n_loss;
            synced = 0;
            i += 1;
            __syncthreads();
            continue;
        }
        // CRC-16-CCITT of the access units in 32-byte chunks, one chunk per item: the FIRST chunk of a unit is the
        // short one and starts from 0xFFFF, the others are whole and start from 0, so joining them is
        // crc = shift32(crc) ^ crc(chunk) with shift32 = a 16-entry table (the CRC is linear over GF(2))
        for (int q = t; q < ch_first[8 * acc]; q += SF_THREADS) {
            int wa = 0;
            while (q >= ch_first[wa + 1]) ++wa;           // (window, access unit) of this chunk; empty entries have equal bounds
            const int w = wa >> 3, a = wa & 7;
            const uint8_t *sf = sfa + w * SF_WIN;
            const int a0 = sh_start[8 * w + a], n = sh_start[8 * w + a + 1] - a0 - 2, nch = ch_first[wa + 1] - ch_first[wa], qq = q - ch_first[wa];
            const int first_len = n - (nch - 1) * CRC_CHUNK;
            const int beg = qq == 0 ? 0 : first_len + (qq - 1) * CRC_CHUNK, cnt = qq == 0 ? first_len : CRC_CHUNK;
            unsigned c = qq == 0 ? 0xFFFFu : 0u;
            for (int k = 0; k < cnt; ++k) c = ((c << 8) ^ t_crc[((c >> 8) ^ sf[a0 + beg + k]) & 0xFF]) & 0xFFFF;
            ch_crc[q] = static_cast<uint16_t>(c);
        }
        __syncthreads();
        if (t < 8 * acc) {                                // one access unit per thread: bounds, then join its chunks
            const int w = t >> 3, a = t & 7;
            int flag = 0;
            if (a < sh_num[w]) {
                const uint8_t *sf = sfa + w * SF_WIN;
                const uint16_t *st8 = sh_start + 8 * w;
                const int a0 = st8[a], a1 = st8[a + 1], len = a1 - a0;
                if (!(a0 < st8[0] || len < 3 || a1 > 110 * s)) {
                    unsigned c = ch_crc[ch_first[t]];
                    for (int q = ch_first[t] + 1; q < ch_first[t + 1]; ++q) {
                        unsigned m = 0;
                        for (int b = 0; b < 16; ++b) m ^= ((c >> b) & 1u) ? t_shift[b] : 0u;
                        c = m ^ ch_crc[q];
                    }
                    c = ~c & 0xFFFF;
                    flag = 1 | ((c == ((static_cast<unsigned>(sf[a0 + len - 2]) << 8) | sf[a0 + len - 1])) ? 2 : 0);
                }
            }
            sh_auflag[t] = flag;
        }
        __syncthreads();
        for (int w = 0; w < acc; ++w) {
            if (out + w >= max_rec) break;
            uint8_t *dst = data + sb.data_off + (size_t)(out + w) * 110 * s;
            const uint8_t *sf = sfa + w * SF_WIN;
            for (int b = t; b < 110 * s; b += SF_THREADS) dst[b] = sf[b];
        }
        if (t < acc) {                                    // one record per accepted window
            const int w = t;
            const uint8_t *sf = sfa + w * SF_WIN;
            DevSfRec rec = {};
            rec.first_frame = base + static_cast<uint32_t>(i + 5 * w);
            rec.header = sf[2] & 0x7F;
            rec.num_aus = static_cast<uint8_t>(sh_num[w]);
            int corrected = 0, failed = 0;
            for (int j = 0; j < s; ++j) { if (res[w * 24 + j] < 0) ++failed; else corrected += res[w * 24 + j]; }
            rec.rs_corrected = static_cast<uint16_t>(corrected);
            rec.rs_failed = static_cast<uint16_t>(failed);
            for (int a = 0; a < 8; ++a) rec.au_start[a] = sh_start[8 * w + a];
            for (int a = 0; a < sh_num[w]; ++a) {
                if (sh_auflag[8 * w + a] & 1) rec.au_valid |= static_cast<uint8_t>(1 << a);
                if (sh_auflag[8 * w + a] & 2) rec.au_ok |= static_cast<uint8_t>(1 << a);
            }
            if (out + w < max_rec) recs[sb.rec_off + out + w] = rec;
        }
        if (t == 0)
            for (int w = 0; w < acc; ++w) {
                ++n_sf;
                for (int j = 0; j < s; ++j) { if (res[w * 24 + j] < 0) ++n_fail; else n_corr += static_cast<uint32_t>(res[w * 24 + j]); }
                for (int a = 0; a < sh_num[w]; ++a) { if (sh_auflag[8 * w + a] & 2) ++n_auok; else ++n_aubad; }
            }
        out += acc;
        synced = 1;
        i += 5 * acc;
        __syncthreads();
    }
    // ---- carry the unconsumed frames (< 5) over to the next step, staged through LDS: the source may be the carry buffer itself
    const int left = total - i;
    for (int f = 0; f < left; ++f) {
        const uint8_t *src = frame_ptr(i + f);
        for (int b = t; b < fb; b += SF_THREADS) sfa[f * fb + b] = src[b];
    }
    __syncthreads();
    for (int b = t; b < left * fb; b += SF_THREADS) stt.buf[b] = sfa[b];
    if (t == 0) {
        stt.carry = left;
        stt.synced = synced;
        stt.frames_seen += static_cast<uint32_t>(n_new);
        stt.n_out = out < max_rec ? out : max_rec;
        stt.stats[0] += n_sf; stt.stats[1] += n_auok; stt.stats[2] += n_aubad;
        stt.stats[3] += n_corr; stt.stats[4] += n_fail; stt.stats[5] += n_loss;
    }
    (void)sfb;
}
