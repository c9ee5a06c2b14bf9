// superframe.hpp — DAB+ audio super frame decoding (ETSI TS 102 563 §5, §6):
// fire code synchronisation, RS(120,110) error correction, access unit extraction and CRC.
//
// Consumer of the sub-channel bytes the GPU decodes; produces what the reference's dabsdr
// library hands to dabsdrAudioCBFunc_t (reference: lib/linux_x86_64/dabsdr.h:47-78;
// consumer src/radiocontrol.cpp:2542, src/audiodecoder.cpp:183-208).  The reference's own
// implementation is inside the closed binary (it exports Karn's decode_rs_char);
// this one is written from the standard.
#pragma once
#include <cstdint>
#include <cstring>
#include <functional>
#include <vector>

namespace dabplus {

// ---- GF(2^8), primitive polynomial x^8+x^4+x^3+x^2+1 (0x11D)
struct GF256 {
    uint8_t exp[512], log[256];
    GF256()
    {
        unsigned x = 1;
        for (int i = 0; i < 255; ++i) {
            exp[i] = static_cast<uint8_t>(x);
            log[x] = static_cast<uint8_t>(i);
            x <<= 1;
            if (x & 0x100) x ^= 0x11D;
        }
        for (int i = 255; i < 512; ++i) exp[i] = exp[i - 255];
        log[0] = 0;
    }
    uint8_t mul(uint8_t a, uint8_t b) const { return (a && b) ? exp[log[a] + log[b]] : 0; }
    uint8_t div(uint8_t a, uint8_t b) const { return a ? exp[log[a] + 255 - log[b]] : 0; }
    uint8_t inv(uint8_t a) const { return exp[255 - log[a]]; }
    uint8_t pow_alpha(int e) const { return exp[((e % 255) + 255) % 255]; }
};

inline const GF256 &gf()
{
    static const GF256 g;
    return g;
}

// RS(120,110): shortened (255,245), generator roots alpha^0 .. alpha^9.
// cw[0..109] data, cw[110..119] parity.  Returns number of corrected bytes, -1 if uncorrectable.
inline int rs_decode_120_110(uint8_t *cw)
{
    const GF256 &G = gf();
    constexpr int N = 120, T2 = 10;
    uint8_t S[T2];
    bool clean = true;
    for (int i = 0; i < T2; ++i) {                         // S_i = c(alpha^i), cw[0] is the highest power
        uint8_t s = 0;
        const uint8_t a = G.pow_alpha(i);
        for (int k = 0; k < N; ++k) s = static_cast<uint8_t>(G.mul(s, a) ^ cw[k]);
        S[i] = s;
        clean = clean && s == 0;
    }
    if (clean) return 0;
    // Berlekamp-Massey
    uint8_t C[T2 + 1] = {1}, B[T2 + 1] = {1};
    int L = 0, m = 1;
    uint8_t b = 1;
    for (int n = 0; n < T2; ++n) {
        uint8_t d = S[n];
        for (int i = 1; i <= L; ++i) d ^= G.mul(C[i], S[n - i]);
        if (d == 0) { ++m; continue; }
        uint8_t Tp[T2 + 1];
        std::memcpy(Tp, C, sizeof Tp);
        const uint8_t coef = G.div(d, b);
        for (int i = 0; i + m <= T2; ++i) C[i + m] ^= G.mul(coef, B[i]);
        if (2 * L <= n) { L = n + 1 - L; std::memcpy(B, Tp, sizeof B); b = d; m = 1; }
        else ++m;
    }
    if (L > T2 / 2) return -1;
    // Chien search over the 120 positions of the shortened code: position k <-> power N-1-k
    int pos[T2 / 2], nerr = 0;
    for (int k = 0; k < N; ++k) {
        const int p = N - 1 - k;                           // locator root is alpha^(-p)
        uint8_t v = 0;
        for (int i = 0; i <= L; ++i) v ^= G.mul(C[i], G.pow_alpha(-p * i));
        if (v == 0) { if (nerr == T2 / 2) return -1; pos[nerr++] = k; }
    }
    if (nerr != L) return -1;
    // Forney: Omega = S(x) C(x) mod x^T2; e = Omega(X^-1) / C'(X^-1) * X^(1-fcr), fcr = 0 -> times X
    uint8_t Om[T2] = {0};
    for (int i = 0; i < T2; ++i)
        for (int j = 0; j <= L && j <= i; ++j) Om[i] ^= G.mul(S[i - j], C[j]);
    for (int e = 0; e < nerr; ++e) {
        const int p = N - 1 - pos[e];
        const uint8_t Xinv = G.pow_alpha(-p);
        uint8_t num = 0, den = 0;
        for (int i = T2 - 1; i >= 0; --i) num = static_cast<uint8_t>(G.mul(num, Xinv) ^ Om[i]);
        for (int i = 1; i <= L; i += 2) den ^= G.mul(C[i], G.pow_alpha(-p * (i - 1)));
        if (den == 0) return -1;
        cw[pos[e]] ^= G.mul(G.div(num, den), G.pow_alpha(p));
    }
    return nerr;
}

inline uint16_t crc16_ccitt(const uint8_t *d, int n)      // FIB / AU CRC: init FFFF, inverted
{
    unsigned c = 0xFFFF;
    for (int i = 0; i < n; ++i) {
        c ^= static_cast<unsigned>(d[i]) << 8;
        for (int b = 0; b < 8; ++b) c = (c & 0x8000) ? ((c << 1) ^ 0x1021) & 0xFFFF : (c << 1) & 0xFFFF;
    }
    return static_cast<uint16_t>(~c & 0xFFFF);
}

inline uint16_t firecode(const uint8_t *d, int n)          // poly 0x782F, init 0
{
    unsigned c = 0;
    for (int i = 0; i < n; ++i) {
        c ^= static_cast<unsigned>(d[i]) << 8;
        for (int b = 0; b < 8; ++b) c = (c & 0x8000) ? ((c << 1) ^ 0x782F) & 0xFFFF : (c << 1) & 0xFFFF;
    }
    return static_cast<uint16_t>(c);
}

struct AccessUnit {
    uint8_t header;          // dabsdrAudioFrameHeader_t.raw: surr[2:0] ps[3] aac_channel_mode[4] sbr[5] dac_rate[6] conceal[7]
    const uint8_t *data;     // AU without its CRC
    uint16_t len;
};

struct Stats {
    uint32_t superframes = 0, au_ok = 0, au_crc_err = 0, rs_corrected = 0, rs_uncorrectable = 0, sync_loss = 0;
};

// Feed one logical frame (24 ms, `bytes` = 3 * kbps) at a time; AUs come out through `sink`.
class Decoder {
public:
    explicit Decoder(int kbps = 0) { configure(kbps); }
    void configure(int kbps)
    {
        s_ = kbps / 8;
        frame_ = 3 * kbps;
        buf_.assign(static_cast<size_t>(5) * frame_, 0);
        filled_ = 0; synced_ = false;
    }
    Stats stats;

    void push(const uint8_t *frame, const std::function<void(const AccessUnit &)> &sink)
    {
        if (s_ <= 0) return;
        if (filled_ == 5) {                                 // slide by one logical frame
            std::memmove(buf_.data(), buf_.data() + frame_, static_cast<size_t>(4) * frame_);
            filled_ = 4;
        }
        std::memcpy(buf_.data() + static_cast<size_t>(filled_) * frame_, frame, frame_);
        if (++filled_ < 5) return;
        if (!try_decode(sink)) {                            // not a super frame boundary (or damaged): keep sliding
            if (synced_) { ++stats.sync_loss; synced_ = false; }
            return;
        }
        synced_ = true;
        filled_ = 0;                                        // consumed: next super frame starts fresh
    }

private:
    int s_ = 0, frame_ = 0, filled_ = 0;
    bool synced_ = false;
    std::vector<uint8_t> buf_, work_;

    bool try_decode(const std::function<void(const AccessUnit &)> &sink)
    {
        work_ = buf_;
        uint8_t *sf = work_.data();
        const int s = s_;
        // RS: code word j = bytes j, j+s, j+2s, ... (110 data + 10 parity)
        int corrected = 0, failed = 0;
        uint8_t cw[120];
        for (int j = 0; j < s; ++j) {
            for (int k = 0; k < 120; ++k) cw[k] = sf[j + k * s];
            const int r = rs_decode_120_110(cw);
            if (r < 0) { ++failed; continue; }
            if (r > 0) { corrected += r; for (int k = 0; k < 110; ++k) sf[j + k * s] = cw[k]; }
        }
        if (firecode(sf + 2, 9) != ((sf[0] << 8) | sf[1]) || (sf[0] == 0 && sf[1] == 0 && sf[2] == 0)) return false;
        ++stats.superframes;
        stats.rs_corrected += corrected;
        stats.rs_uncorrectable += failed;
        const int dac_rate = (sf[2] >> 6) & 1, sbr = (sf[2] >> 5) & 1, ch = (sf[2] >> 4) & 1, ps = (sf[2] >> 3) & 1, surr = sf[2] & 7;
        const int num_aus = dac_rate ? (sbr ? 3 : 6) : (sbr ? 2 : 4);
        int start[7];
        start[0] = dac_rate ? (sbr ? 6 : 11) : (sbr ? 5 : 8);
        for (int i = 1; i < num_aus; ++i) {                 // 12-bit big-endian fields packed from byte 3
            const int bit = 24 + 12 * (i - 1), byte = bit >> 3;
            start[i] = (bit & 4) ? (((sf[byte] & 0x0F) << 8) | sf[byte + 1]) : ((sf[byte] << 4) | (sf[byte + 1] >> 4));
        }
        start[num_aus] = 110 * s;
        const uint8_t hdr = static_cast<uint8_t>(surr | (ps << 3) | (ch << 4) | (sbr << 5) | (dac_rate << 6));
        for (int i = 0; i < num_aus; ++i) {
            const int len = start[i + 1] - start[i];
            if (start[i] < start[0] || len < 3 || start[i + 1] > 110 * s) { ++stats.au_crc_err; continue; }
            const uint8_t *au = sf + start[i];
            const bool ok = crc16_ccitt(au, len - 2) == ((au[len - 2] << 8) | au[len - 1]);
            if (ok) ++stats.au_ok; else ++stats.au_crc_err;
            AccessUnit u = {static_cast<uint8_t>(hdr | (ok ? 0 : 0x80)), au, static_cast<uint16_t>(len - 2)};
            sink(u);
        }
        return true;
    }
};

}  // namespace dabplus
