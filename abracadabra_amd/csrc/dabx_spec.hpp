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

// §11.3.1 tables 31-33 unequal error protection, by the 6-bit table index of FIG 0/1 short form:
// {bit rate, protection level, L1..L4, PI1..PI4, padding bits}
inline const int16_t (&uep_rows())[64][11]
{
    static const int16_t t[64][11] = {
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
    return t;
}

inline bool uep_profile(int index, Profile &p, int *kbps = nullptr)
{
    p = Profile();
    if (index < 0 || index > 63) return false;
    const int16_t *r = uep_rows()[index];
    p.nseg = r[5] ? 4 : 3;
    for (int s = 0; s < p.nseg; ++s) { p.L[s] = r[2 + s]; p.PI[s] = r[6 + s]; }
    p.finish();
    p.n_coded += r[10];                     // padding bits after the tail
    p.n_cu = p.n_coded / kCuBits;
    if (kbps) *kbps = r[0];
    return p.n_coded % kCuBits == 0;
}

// option 0/1: EEP set A/B (level 1..4, kbps); option 2: UEP, level = table index
inline bool any_profile(int option, int level, int kbps, Profile &p)
{
    return option == 2 ? uep_profile(level, p) : eep_profile(option, level, kbps, p);
}

// Gather map of a codeword: for every trellis step the byte offsets of its four soft bits from the codeword's base and the byte mask
// of the ones it keeps — five arrays of steps() + kStepInfoPad words, one behind the other (offsets of bit 0, 1, 2, 3, masks), so that the
// 64 lanes of a wave read each of them with one contiguous load.  Every puncturing vector of EN 300 401 table 29 keeps a PREFIX of the
// four mother-code bits of a step (1000, 1100, 1110 or 1111), so the kept soft bits are the next `ones` bits of the stream; the
// offsets of the punctured ones point at the step's first bit (a valid address: the kernel loads all four and masks).
//   linear:  coded bit i at byte i (FIC rows, the stage-level entry point)
//   !linear: the MSC row of a logical frame, residue-major (dabx_dev.h): bit i at (i & 15) * kTiSeg + (i >> 4) from the
//            sub-channel's place in residue class 0
// The mask is 0xFE per kept byte (the kernel doubles the values inside their bytes: the bit that comes in from below is cleared).
// Entries [steps() ..] of every array: what k_viterbi's fetch reads for the steps past the end (it runs up to two rounds of 64 steps
// ahead and does not clamp its index): offset 0, mask 0.
constexpr int kStepInfoPad = 256;
constexpr int kStepArrays = 5;
constexpr int kTiSeg = kCifBits / 16;
inline std::vector<uint32_t> step_gather(const Profile &p, bool linear)
{
    const size_t n = static_cast<size_t>(p.steps() + kStepInfoPad);
    std::vector<uint32_t> info(n * kStepArrays, 0u);
    uint32_t bit = 0;
    size_t step = 0;
    auto place = [&](uint32_t i) { return linear ? i : (i & 15u) * static_cast<uint32_t>(kTiSeg) + (i >> 4); };
    auto emit = [&](int ones) {
        for (int j = 0; j < 4; ++j) info[static_cast<size_t>(j) * n + step] = place(bit + static_cast<uint32_t>(j < ones ? j : 0));
        info[4 * n + step] = 0xFEFEFEFEu >> (8 * (4 - ones));
        bit += static_cast<uint32_t>(ones);
        ++step;
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
