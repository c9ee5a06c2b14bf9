// packet.hpp — packet mode of the MSC (ETSI EN 300 401 §5.3.2, §5.3.3): packets of 24/48/72/96 bytes with
// a 3-byte header and a CRC, carried back to back in the logical frames of a data sub-channel; the useful
// data of the packets of one address between a "first" and a "last" packet form one MSC data group.
//
// Host-side consumer of the sub-channel bytes the GPU decodes; produces what the reference's dabsdr library
// hands to dabsdrDataGroupCBFunc_t for packet-mode service components (dabsdr.h:89-96; consumer
// src/radiocontrol.cpp:2519, src/data/mscdatagroup.cpp:31-58 re-checks the data group CRC).  Written from
// the standard.  The optional outer code of FIG 0/14 (§5.3.5: RS(204,188) over FEC frames of 94 + 9 units of 24 bytes)
// is applied when the sub-channel announces it: once the nine FEC packets (address 1022, counter 0..8) have been seen
// in a row the decoder follows the frame structure by position, corrects the 12 rows of every Application Data Table
// (up to 8 byte errors each) and only then cuts it into packets; before that, and for sub-channels without FEC,
// damaged packets are dropped by their CRC.
#pragma once
#include <cstdint>
#include <cstring>
#include <functional>
#include <map>
#include <vector>

namespace packet {

inline uint16_t crc16(const uint8_t *d, int n)
{
    unsigned c = 0xFFFF;
    for (int i = 0; i < n; ++i) {
        c ^= static_cast<unsigned>(d[i]) << 8;
        for (int b = 0; b < 8; ++b) c = (c & 0x8000) ? ((c << 1) ^ 0x1021) & 0xFFFF : (c << 1) & 0xFFFF;
    }
    return static_cast<uint16_t>(~c & 0xFFFF);
}

// RS(255,239) over GF(2^8) / x^8+x^4+x^3+x^2+1, generator roots 2^0 .. 2^15 (EN 300 401 §5.3.5.1), used shortened
// to (204,188).  The DAB+ super frame's RS(120,110) of the same field runs on the GPU (dabx_superframe.hip); the
// packet-mode FEC frames are a few hundred bytes per second and are decoded where the packets are parsed.
namespace rs {

struct Field {
    uint8_t exp[512], log[256];
    Field()
    {
        unsigned x = 1;
        for (int i = 0; i < 255; ++i) { exp[i] = static_cast<uint8_t>(x); log[x] = static_cast<uint8_t>(i); x <<= 1; if (x & 0x100) x ^= 0x11D; }
        for (int i = 255; i < 512; ++i) exp[i] = exp[i - 255];
        log[0] = 0;
    }
    uint8_t mul(uint8_t a, uint8_t b) const { return a && b ? exp[log[a] + log[b]] : 0; }
    uint8_t div(uint8_t a, uint8_t b) const { return a ? exp[log[a] + 255 - log[b]] : 0; }     // b != 0
    uint8_t pow_a(int e) const { e %= 255; if (e < 0) e += 255; return exp[e]; }
};
inline const Field &field() { static const Field f; return f; }

constexpr int NROOTS = 16;

inline bool syndromes(const uint8_t *cw, int n, uint8_t *S)
{
    const Field &F = field();
    bool any = false;
    for (int i = 0; i < NROOTS; ++i) {
        uint8_t s = 0;
        for (int j = 0; j < n; ++j) s = static_cast<uint8_t>(F.mul(s, F.exp[i]) ^ cw[j]);       // Horner, cw[0] = highest power
        S[i] = s;
        any |= s != 0;
    }
    return any;
}

// corrects an n-byte shortened codeword (n - 16 data bytes first, 16 parity bytes last) in place;
// returns the number of corrected bytes, -1 when it holds more than 8 errors
inline int decode(uint8_t *cw, int n)
{
    const Field &F = field();
    uint8_t S[NROOTS];
    if (!syndromes(cw, n, S)) return 0;
    // Berlekamp-Massey
    uint8_t lam[NROOTS + 1] = {1}, prev[NROOTS + 1] = {1}, tmp[NROOTS + 1];
    int L = 0, m = 1;
    uint8_t b = 1;
    for (int r = 0; r < NROOTS; ++r) {
        uint8_t d = S[r];
        for (int i = 1; i <= L; ++i) d ^= F.mul(lam[i], S[r - i]);
        if (!d) { ++m; continue; }
        std::memcpy(tmp, lam, sizeof tmp);
        const uint8_t q = F.div(d, b);
        for (int i = 0; i + m <= NROOTS; ++i) lam[i + m] ^= F.mul(q, prev[i]);
        if (2 * L <= r) { L = r + 1 - L; std::memcpy(prev, tmp, sizeof prev); b = d; m = 1; }
        else ++m;
    }
    if (L > NROOTS / 2) return -1;
    // error evaluator: S(x) * lambda(x) mod x^16
    uint8_t om[NROOTS] = {0};
    for (int i = 0; i < NROOTS; ++i)
        for (int j = 0; j <= i && j <= L; ++j) om[i] ^= F.mul(lam[j], S[i - j]);
    // Chien search over the n positions of the shortened word, Forney's values (first root 2^0: e = X * omega(1/X) / lambda'(1/X))
    int found = 0, pos[NROOTS / 2];
    uint8_t val[NROOTS / 2];
    for (int p = 0; p < n && found <= L; ++p) {
        const int inv = (255 - p) % 255;                               // log of X^-1, X = 2^p
        uint8_t v = 0;
        for (int i = 0; i <= L; ++i) v ^= F.mul(lam[i], F.pow_a(inv * i));
        if (v) continue;
        if (found == L) return -1;
        uint8_t num = 0, den = 0;
        for (int i = 0; i < NROOTS; ++i) num ^= F.mul(om[i], F.pow_a(inv * i));
        for (int i = 1; i <= L; i += 2) den ^= F.mul(lam[i], F.pow_a(inv * (i - 1)));
        if (!den) return -1;
        pos[found] = p;
        val[found] = F.mul(F.exp[p], F.div(num, den));
        ++found;
    }
    if (found != L) return -1;
    for (int k = 0; k < L; ++k) cw[n - 1 - pos[k]] ^= val[k];
    if (syndromes(cw, n, S)) {                                         // cannot happen for <= 8 errors; a miscorrection is undone
        for (int k = 0; k < L; ++k) cw[n - 1 - pos[k]] ^= val[k];
        return -1;
    }
    return L;
}

}  // namespace rs

struct Stats { uint32_t packets = 0, crc_err = 0, groups = 0, dropped = 0, fec_frames = 0, fec_corrected = 0, fec_failed_rows = 0; };

class Decoder {
public:
    int address = -1;                                                    // packet address to follow, -1 = all
    std::function<void(int addr, const uint8_t *, int)> on_data_group;   // one complete MSC data group
    Stats stats;

    void reset() { asm_.clear(); unlock(); }

    // FIG 0/14: the sub-channel carries FEC frames (scheme 1)
    void set_fec(bool on) { if (on != fec_) { fec_ = on; unlock(); } }
    bool fec_locked() const { return locked_; }

    // one logical frame of the sub-channel (3 * kbps bytes)
    void feed_frame(const uint8_t *f, int len)
    {
        if (!fec_) { parse(f, len); return; }
        // a FEC frame is 94 units of 24 bytes of packets + 9 FEC packets of one unit each; logical frames and FEC frames
        // are not aligned with each other, both are whole units
        int start = 0;                                                   // first byte of this frame not yet handed on
        for (int pos = 0; pos + UNIT <= len; pos += UNIT) {
            const uint8_t *u = f + pos;
            if (!locked_) {
                const bool is_fec = (u[0] >> 6) == 0 && (((u[0] & 3) << 8) | u[1]) == FEC_ADDR;
                const int counter = (u[0] >> 2) & 15;
                run_ = is_fec && counter == run_ ? run_ + 1 : (is_fec && counter == 0 ? 1 : 0);
                if (run_ == FEC_UNITS) {                                 // nine in a row: the next unit opens a table
                    parse(f + start, pos + UNIT - start);                // what came before goes through uncorrected
                    start = pos + UNIT;
                    locked_ = true; unit_ = 0; misses_ = 0; run_ = 0;
                }
                continue;
            }
            if (unit_ < APP_UNITS) std::memcpy(table_ + unit_ * UNIT, u, UNIT);
            else {
                const int k = unit_ - APP_UNITS;
                const bool ok = (u[0] >> 6) == 0 && (((u[0] & 3) << 8) | u[1]) == FEC_ADDR && ((u[0] >> 2) & 15) == k;
                if (!ok) ++misses_;
                std::memcpy(parity_ + k * 22, u + 2, 22);
            }
            start = pos + UNIT;
            if (++unit_ == APP_UNITS + FEC_UNITS) {
                if (misses_ > FEC_UNITS / 2) {                           // the structure is gone: hand the table on as it is
                    parse(table_, APP_UNITS * UNIT);
                    unit_ = 0;
                    unlock();
                } else {
                    correct_table();
                    parse(table_, APP_UNITS * UNIT);
                    unit_ = 0; misses_ = 0;
                }
            }
        }
        if (!locked_ && start < len) parse(f + start, len - start);
    }

private:
    static constexpr int UNIT = 24, APP_UNITS = 94, FEC_UNITS = 9, FEC_ADDR = 1022, ROWS = 12, COLS = 188;
    struct Assembly { std::vector<uint8_t> data; int next_ci = -1; bool open = false; };
    std::map<int, Assembly> asm_;
    bool fec_ = false, locked_ = false;
    int run_ = 0, unit_ = 0, misses_ = 0;
    uint8_t table_[APP_UNITS * UNIT];                                    // Application Data Table, filled column by column
    uint8_t parity_[FEC_UNITS * 22];                                     // RS Data Table (192 bytes + 6 bytes of padding)

    void unlock()
    {
        if (locked_ && unit_ > 0) parse(table_, (unit_ < APP_UNITS ? unit_ : APP_UNITS) * UNIT);
        locked_ = false; run_ = 0; unit_ = 0; misses_ = 0;
    }

    // row r of the tables: bytes r, r + 12, r + 24, ... (both tables are written and read column by column)
    void correct_table()
    {
        ++stats.fec_frames;
        uint8_t cw[COLS + rs::NROOTS];
        for (int r = 0; r < ROWS; ++r) {
            for (int c = 0; c < COLS; ++c) cw[c] = table_[c * ROWS + r];
            for (int c = 0; c < rs::NROOTS; ++c) cw[COLS + c] = parity_[c * ROWS + r];
            const int n = rs::decode(cw, COLS + rs::NROOTS);
            if (n < 0) { ++stats.fec_failed_rows; continue; }
            if (n > 0) {
                stats.fec_corrected += static_cast<uint32_t>(n);
                for (int c = 0; c < COLS; ++c) table_[c * ROWS + r] = cw[c];
            }
        }
    }

    void parse(const uint8_t *f, int len)
    {
        int pos = 0;
        while (pos + 24 <= len) {
            const int plen = 24 * ((f[pos] >> 6) + 1);
            if (pos + plen > len) break;
            packet_in(f + pos, plen);
            pos += plen;
        }
    }

    void packet_in(const uint8_t *p, int plen)
    {
        const int addr = ((p[0] & 3) << 8) | p[1];
        if (addr == 0 || (addr == FEC_ADDR && plen == UNIT)) return;     // padding packet / FEC packet (no CRC of its own)
        ++stats.packets;
        if (crc16(p, plen - 2) != ((p[plen - 2] << 8) | p[plen - 1])) { ++stats.crc_err; drop(addr); return; }
        if (address >= 0 && addr != address) return;
        const int ci = (p[0] >> 4) & 3, fl = (p[0] >> 2) & 3, useful = p[2] & 0x7F;
        if ((p[2] & 0x80) || useful > plen - 5) return;                 // command packets carry no data group bytes
        Assembly &a = asm_[addr];
        const bool first = fl & 2, last = fl & 1;
        if (first) { a.data.clear(); a.open = true; }
        else if (!a.open || ci != a.next_ci) { if (a.open) ++stats.dropped; a.open = false; return; }
        a.next_ci = (ci + 1) & 3;
        a.data.insert(a.data.end(), p + 3, p + 3 + useful);
        if (last) {
            ++stats.groups;
            if (on_data_group && !a.data.empty()) on_data_group(addr, a.data.data(), static_cast<int>(a.data.size()));
            a.open = false;
        }
    }

    void drop(int addr)
    {
        auto it = asm_.find(addr);
        if (it != asm_.end() && it->second.open) { it->second.open = false; ++stats.dropped; }
    }
};

}  // namespace packet
