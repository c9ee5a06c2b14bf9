// Sanitizer harness for the host-side parsers (FIG database, PAD, packet mode incl. its FEC frames, raw-file probe, TII detector):
//   g++ -std=c++17 -O1 -g -fsanitize=address,undefined -fno-sanitize-recover=all -I../../abracadabra_amd/csrc -o fuzz_host fuzz_host.cpp && ./fuzz_host
// Feeds random and mutated-valid input; any out-of-bounds access or UB aborts.  CPU only (no HIP involved).
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>

#include "fig_db.hpp"
#include "packet.hpp"
#include "pad.hpp"
#include "rawfile.hpp"
#include "tii.hpp"

static uint16_t crc16(const uint8_t *d, int n)
{
    unsigned c = 0xFFFF;
    for (int i = 0; i < n; ++i) {
        c ^= static_cast<unsigned>(d[i]) << 8;
        for (int b = 0; b < 8; ++b) c = (c & 0x8000) ? ((c << 1) ^ 0x1021) & 0xFFFF : (c << 1) & 0xFFFF;
    }
    return static_cast<uint16_t>(~c & 0xFFFF);
}

int main()
{
    std::mt19937 rng(12345);
    auto rnd = [&](int n) { return static_cast<int>(rng() % static_cast<unsigned>(n)); };
    long fibs = 0, pads = 0, pkts = 0;
    // ---- FIG database: random FIBs (the CRC is checked before parse_fib in the product; here every FIB is parsed)
    for (int round = 0; round < 300; ++round) {
        figdb::Database db;
        for (int k = 0; k < 200; ++k) {
            uint8_t fib[32];
            for (auto &b : fib) b = static_cast<uint8_t>(rng());
            if (rnd(2)) {                                  // plausible headers: type 0/1, random extension and length
                int pos = 0;
                while (pos < 30) {
                    const int len = 1 + rnd(29);
                    fib[pos] = static_cast<uint8_t>((rnd(2) << 5) | (len & 0x1F));
                    if (pos + 1 < 30) fib[pos + 1] = static_cast<uint8_t>((rnd(2) << 7) | (rnd(2) << 5) | rnd(26));
                    pos += 1 + len;
                }
            }
            db.parse_fib(fib);
            ++fibs;
        }
    }
    // ---- PAD: random access units and random PAD fields, DAB+ and MPEG Layer II framing
    for (int round = 0; round < 2000; ++round) {
        pad::Decoder dec;
        long sink = 0;
        dec.on_dynamic_label = [&](const uint8_t *d, int n) { for (int i = 0; i < n; ++i) sink += d[i]; };
        dec.on_data_group = [&](int, const uint8_t *d, int n) { for (int i = 0; i < n; ++i) sink += d[i]; };
        for (int k = 0; k < 50; ++k) {
            std::vector<uint8_t> au(static_cast<size_t>(1 + rnd(700)));
            for (auto &b : au) b = static_cast<uint8_t>(rng());
            if (rnd(2)) { au[0] = 0x80 | (au[0] & 0x1F); if (au.size() > 1) au[1] = static_cast<uint8_t>(rnd(256)); }
            if (rnd(3) == 0 && au.size() >= 4) { au[0] = 0xFF; au[1] = 0xF0 | (au[1] & 0x0F); }
            dec.feed_dabplus_au(au.data(), static_cast<int>(au.size()));
            dec.feed_mp2_frame(au.data(), static_cast<int>(au.size()));
            if (au.size() >= 2) dec.feed_pad(au.data(), static_cast<int>(au.size() > 200 ? 200 : au.size()));
            ++pads;
        }
    }
    // ---- packet mode: random frames, and valid packets with random headers
    for (int round = 0; round < 2000; ++round) {
        packet::Decoder dec;
        dec.address = rnd(2) ? -1 : rnd(1024);
        long sink = 0;
        dec.on_data_group = [&](int, const uint8_t *d, int n) { for (int i = 0; i < n; ++i) sink += d[i]; };
        for (int k = 0; k < 20; ++k) {
            const int fb = 24 * (1 + rnd(24));
            std::vector<uint8_t> f(static_cast<size_t>(fb));
            for (auto &b : f) b = static_cast<uint8_t>(rng());
            if (rnd(2))
                for (int pos = 0; pos + 24 <= fb;) {       // give the packets valid CRCs so that the assembly logic runs
                    const int plen = 24 * ((f[static_cast<size_t>(pos)] >> 6) + 1);
                    if (pos + plen > fb) break;
                    const uint16_t c = crc16(f.data() + pos, plen - 2);
                    f[static_cast<size_t>(pos + plen - 2)] = static_cast<uint8_t>(c >> 8);
                    f[static_cast<size_t>(pos + plen - 1)] = static_cast<uint8_t>(c);
                    pos += plen;
                }
            dec.feed_frame(f.data(), fb);
            ++pkts;
        }
    }
    // ---- packet mode with FEC frames: the decoder locks on runs of FEC packets (address 1022, counters 0..8), fills tables by
    // position and runs RS(204,188) over garbage rows; the switch is flipped at random, the structure broken at random
    for (int round = 0; round < 300; ++round) {
        packet::Decoder dec;
        dec.address = rnd(2) ? -1 : rnd(1024);
        dec.set_fec(rnd(4) != 0);
        long sink = 0;
        dec.on_data_group = [&](int, const uint8_t *d, int n) { for (int i = 0; i < n; ++i) sink += d[i]; };
        int unit = rnd(103);
        for (int k = 0; k < 60; ++k) {
            const int fb = 24 * (1 + rnd(24));
            std::vector<uint8_t> f(static_cast<size_t>(fb));
            for (auto &b : f) b = static_cast<uint8_t>(rnd(8) ? 0 : rng());             // mostly zeros: few RS errors, sometimes too many
            for (int pos = 0; pos + 24 <= fb; pos += 24, unit = (unit + 1) % 103)
                if (unit >= 94 && rnd(20)) {                                             // FEC packets where the frame structure puts them
                    f[static_cast<size_t>(pos)] = static_cast<uint8_t>(((unit - 94) << 2) | 3);
                    f[static_cast<size_t>(pos + 1)] = 0xFE;
                }
            if (rnd(50) == 0) unit = rnd(103);                                           // the structure slips
            if (rnd(40) == 0) dec.set_fec(rnd(2));
            dec.feed_frame(f.data(), fb);
            ++pkts;
        }
        std::vector<uint8_t> cw(204);
        for (int t = 0; t < 20; ++t) {                                                   // the RS decoder on its own: 0..12 errors in a zero word
            std::fill(cw.begin(), cw.end(), 0);
            for (int e = rnd(13); e > 0; --e) cw[static_cast<size_t>(rnd(204))] = static_cast<uint8_t>(rng());
            sink += packet::rs::decode(cw.data(), 204);
        }
    }
    // ---- raw-file probe and TII detector
    for (int round = 0; round < 20000; ++round) {
        std::vector<uint8_t> head(static_cast<size_t>(rnd(4096)));
        for (auto &b : head) b = static_cast<uint8_t>(rnd(3) ? 32 + rnd(95) : rng());
        if (head.size() > 40 && rnd(2)) std::memcpy(head.data(), "<?xml version=\"1.0\"?><SDR><Datablocks", 37);
        (void)rawfile::probe(head.data(), static_cast<int>(head.size()));
    }
    for (int round = 0; round < 2000; ++round) {
        float p[2048];
        for (auto &x : p) x = static_cast<float>(rng() % 1000) * (rnd(50) == 0 ? 1e30f : 1.0f);
        if (rnd(10) == 0) p[rnd(2048)] = 0.0f / 1.0f;
        (void)tii::detect(p, 4.0f);
    }
    std::printf("fuzz ok: %ld FIBs, %ld PAD fields, %ld packet frames\n", fibs, pads, pkts);
    return 0;
}
