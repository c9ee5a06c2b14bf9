// tii.hpp — Transmitter Identification Information from the null symbol's power spectrum
// (ETSI EN 300 401 §14.8, Mode I).  Produces what the reference reports through
// dabsdrNtfTii_t (reference: lib/linux_x86_64/dabsdr.h:372-384; consumer src/tii/tiibackend.cpp:70,
// which plots the 384-value folded spectrum, src/radiocontrol.h:278-283).  The reference's detector
// is inside the closed binary; this one follows the standard: transmitter (p, c) lights carrier
// pairs k, k+1 with k = base + 2c + 48b for the four bases -768, -384, 1, 385 and the four b
// whose bit is set in pattern p (the 70 words of 8 bits with four ones, in increasing order).
#pragma once
#include <algorithm>
#include <cstdint>
#include <vector>

namespace tii {

struct Id { uint8_t main, sub; float level; };

inline int pattern_word(int p)          // a_0 is the most significant bit
{
    int n = 0;
    for (int w = 0; w < 256; ++w)
        if (__builtin_popcount(w) == 4 && n++ == p) return w;
    return -1;
}
inline int pattern_index(int word)
{
    int n = 0;
    for (int w = 0; w < 256; ++w)
        if (__builtin_popcount(w) == 4) { if (w == word) return n; ++n; }
    return -1;
}
inline int carrier_base(int block) { static const int base[4] = {-768, -384, 1, 385}; return base[block]; }

// power: 2048 bins, natural FFT order.  folded: 384 values = the four blocks summed.
inline void fold(const float *power, float folded[384])
{
    for (int j = 0; j < 384; ++j) {
        float s = 0.0f;
        for (int blk = 0; blk < 4; ++blk) s += power[(carrier_base(blk) + j) & 2047];
        folded[j] = s;
    }
}

// threshold: a comb is reported when its four strongest pair energies all exceed `factor` times the
// median pair energy (factor 4 = "default", 8 = "conservative")
inline std::vector<Id> detect(const float *power, float factor = 4.0f)
{
    float folded[384];
    fold(power, folded);
    float pair[24][8];
    std::vector<float> all;
    all.reserve(192);
    for (int c = 0; c < 24; ++c)
        for (int b = 0; b < 8; ++b) {
            pair[c][b] = folded[2 * c + 48 * b] + folded[2 * c + 48 * b + 1];
            all.push_back(pair[c][b]);
        }
    std::nth_element(all.begin(), all.begin() + 96, all.end());
    const float floor = all[96];
    std::vector<Id> out;
    if (!(floor >= 0.0f)) return out;
    for (int c = 0; c < 24; ++c) {
        int order[8] = {0, 1, 2, 3, 4, 5, 6, 7};
        std::sort(order, order + 8, [&](int a, int b) { return pair[c][a] > pair[c][b]; });
        if (!(pair[c][order[3]] > factor * floor) || !(pair[c][order[3]] > 2.0f * pair[c][order[4]])) continue;
        int word = 0;
        float lvl = 0.0f;
        for (int i = 0; i < 4; ++i) { word |= 0x80 >> order[i]; lvl += pair[c][order[i]]; }
        const int p = pattern_index(word);
        if (p >= 0) out.push_back({static_cast<uint8_t>(p), static_cast<uint8_t>(c), lvl * 0.25f});
    }
    std::sort(out.begin(), out.end(), [](const Id &a, const Id &b) { return a.level > b.level; });
    if (out.size() > 24) out.resize(24);
    return out;
}

}  // namespace tii
