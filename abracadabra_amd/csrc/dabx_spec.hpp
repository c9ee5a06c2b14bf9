// dabx_spec.hpp — host-side Mode-I tables of the product library.
//
// ETSI EN 300 401 constants generated at context creation and uploaded to the
// GPU.  The reference keeps all of this inside its closed binary
// (reference: lib/linux_x86_64/libdabsdr.so.4.0.1; public surface dabsdr.h:397-429),
// so the rules are taken from the standard.
#pragma once
#include <array>
#include <cmath>
#include <cstdint>
#include <vector>

namespace dabx {

constexpr int kTF = 196608, kTNull = 2656, kTS = 2552, kTU = 2048, kTG = 504;
constexpr int kNSym = 76, kCarriers = 1536, kSymBits = 3072;
constexpr int kFicBits = 9216, kFicCwBits = 2304, kFicCwIn = 768;
constexpr int kCifBits = 55296, kCifSyms = 18, kCuBits = 64, kNumCu = 864;
constexpr int kBackoff = 24;       // FFT window starts this far inside the guard interval
constexpr int kCfoRange = 16;      // integer carrier-offset search, +-kHz
constexpr int kSoftExp = 17;       // soft-bit scale exponent
constexpr int kPmInit = -1000000;  // path metric of the non-zero start states
constexpr float kLockThr = 48.0f;

// §14.6.1 frequency interleaver: QPSK symbol n -> carrier k
inline std::array<int16_t, kCarriers> carrier_of_symbol()
{
    std::array<int16_t, kCarriers> k{};
    int n = 0;
    unsigned v = 0;
    for (int i = 0; i < 2048 && n < kCarriers; ++i, v = (13u * v + 511u) % 2048u)
        if (v >= 256 && v <= 1792 && v != 1024) k[n++] = static_cast<int16_t>(static_cast<int>(v) - 1024);
    return k;
}

// §14.3.2 phase reference symbol as quadrant numbers per FFT bin (-1 = unused)
inline std::array<int8_t, kTU> prs_quadrants()
{
    static const char *h[4] = {"0200001120002211", "0323013021232330", "0002021322022013", "0121033223212132"};
    // index i and offset n per block of 32 carriers, negative half then positive half
    static const char *iseq = "012301230123012301230123" "032103210321032103210321";
    static const char *nseq = "120132232123123322211312" "311122102233021333303011";
    std::array<int8_t, kTU> q;
    q.fill(-1);
    for (int blk = 0; blk < 48; ++blk) {
        const int k0 = blk < 24 ? -768 + 32 * blk : 1 + 32 * (blk - 24);
        const int i = iseq[blk] - '0', n = nseq[blk] - '0';
        for (int j = 0; j < 32; ++j) q[(k0 + j) & 2047] = static_cast<int8_t>(((h[i][j & 15] - '0') + n) & 3);
    }
    return q;
}

// §11.1.2: number of kept bits in group g (of 4 mother bits) of puncturing vector PI
inline int punct_group_ones(int pi, int g)
{
    const int order = ((g & 1) << 2) | (g & 2) | ((g >> 2) & 1);   // 3-bit reversal: 0,4,2,6,1,5,3,7
    int ones = 1;
    for (int round = 0; round < 3; ++round) ones += (pi - 8 * round > order) ? 1 : 0;
    return ones;
}

struct Profile {
    int nseg = 0;
    int L[4] = {0, 0, 0, 0}, PI[4] = {0, 0, 0, 0};
    int n_in = 0, n_coded = 0, n_cu = 0;
    bool operator==(const Profile &o) const
    {
        if (nseg != o.nseg) return false;
        for (int i = 0; i < nseg; ++i)
            if (L[i] != o.L[i] || PI[i] != o.PI[i]) return false;
        return true;
    }
    void finish()
    {
        int blocks = 0, coded = 12;
        for (int i = 0; i < nseg; ++i) { blocks += L[i]; coded += L[i] * 4 * (8 + PI[i]); }
        n_in = 32 * blocks;
        n_coded = coded;
    }
    int steps() const { return n_in + 6; }
};

inline Profile fic_profile()
{
    Profile p;
    p.nseg = 2; p.L[0] = 21; p.PI[0] = 16; p.L[1] = 3; p.PI[1] = 15;
    p.finish();
    return p;
}

// §11.3.2 equal error protection
inline bool eep_profile(int option, int level, int kbps, Profile &p)
{
    p = Profile();
    p.nseg = 2;
    if (level < 1 || level > 4 || kbps <= 0) return false;
    if (option == 0) {
        if (kbps % 8) return false;
        const int n = kbps / 8;
        if (level == 1) { p.L[0] = 6 * n - 3; p.L[1] = 3; p.PI[0] = 24; p.PI[1] = 23; p.n_cu = 12 * n; }
        else if (level == 2 && n == 1) { p.L[0] = 5; p.L[1] = 1; p.PI[0] = 13; p.PI[1] = 12; p.n_cu = 8; }
        else if (level == 2) { p.L[0] = 2 * n - 3; p.L[1] = 4 * n + 3; p.PI[0] = 14; p.PI[1] = 13; p.n_cu = 8 * n; }
        else if (level == 3) { p.L[0] = 6 * n - 3; p.L[1] = 3; p.PI[0] = 8; p.PI[1] = 7; p.n_cu = 6 * n; }
        else { p.L[0] = 4 * n - 3; p.L[1] = 2 * n + 3; p.PI[0] = 3; p.PI[1] = 2; p.n_cu = 4 * n; }
    } else if (option == 1) {
        if (kbps % 32) return false;
        const int n = kbps / 32;
        const int pi1[5] = {0, 10, 6, 4, 2}, cu[5] = {0, 27, 21, 18, 15};
        p.L[0] = 24 * n - 3; p.L[1] = 3; p.PI[0] = pi1[level]; p.PI[1] = pi1[level] - 1; p.n_cu = cu[level] * n;
    } else return false;
    p.finish();
    return p.n_cu <= kNumCu && p.n_coded == p.n_cu * kCuBits;
}

// depuncturing map: per trellis step, (offset of first kept bit << 4) | keep mask (bit 3 = x0)
inline std::vector<uint32_t> step_info(const Profile &p)
{
    std::vector<uint32_t> info;
    info.reserve(p.steps());
    uint32_t off = 0;
    auto emit = [&](int ones) {
        info.push_back((off << 4) | (0xFu & ~(0xFu >> ones)));
        off += ones;
    };
    for (int s = 0; s < p.nseg; ++s)
        for (int blk = 0; blk < p.L[s]; ++blk)
            for (int g = 0; g < 32; ++g) emit(punct_group_ones(p.PI[s], g & 7));
    for (int g = 0; g < 6; ++g) emit(2);
    return info;
}

// §10 energy dispersal sequence packed for the Viterbi epilogue: bit 31-j of word h = PRBS bit 32 h + j
inline std::vector<uint32_t> prbs_words(int nbits)
{
    std::vector<uint32_t> w((nbits + 31) / 32, 0);
    unsigned reg = 0x1FF;
    for (int i = 0; i < nbits; ++i) {
        const unsigned b = ((reg >> 8) ^ (reg >> 4)) & 1u;
        reg = ((reg << 1) | b) & 0x1FF;
        w[i >> 5] |= static_cast<uint32_t>(b) << (31 - (i & 31));
    }
    return w;
}

// FFT output placement of the radix 8-8-8-4 decimation-in-frequency kernel
inline int bin_of_pos(int p) { return (p >> 8) + 8 * ((p >> 5) & 7) + 64 * ((p >> 2) & 7) + 512 * (p & 3); }

}  // namespace dabx
