// packet.hpp — packet mode of the MSC (ETSI EN 300 401 §5.3.2, §5.3.3): packets of 24/48/72/96 bytes with
// a 3-byte header and a CRC, carried back to back in the logical frames of a data sub-channel; the useful
// data of the packets of one address between a "first" and a "last" packet form one MSC data group.
//
// Host-side consumer of the sub-channel bytes the GPU decodes; produces what the reference's dabsdr library
// hands to dabsdrDataGroupCBFunc_t for packet-mode service components (dabsdr.h:89-96; consumer
// src/radiocontrol.cpp:2519, src/data/mscdatagroup.cpp:31-58 re-checks the data group CRC).  Written from
// the standard.  The optional outer code of FIG 0/14 (RS(204,188) over FEC frames) is not applied: its
// parity packets (address 1022) are skipped and damaged packets are dropped by their CRC.
#pragma once
#include <cstdint>
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

struct Stats { uint32_t packets = 0, crc_err = 0, groups = 0, dropped = 0; };

class Decoder {
public:
    int address = -1;                                                    // packet address to follow, -1 = all
    std::function<void(int addr, const uint8_t *, int)> on_data_group;   // one complete MSC data group
    Stats stats;

    void reset() { asm_.clear(); }

    // one logical frame of the sub-channel (3 * kbps bytes)
    void feed_frame(const uint8_t *f, int len)
    {
        int pos = 0;
        while (pos + 24 <= len) {
            const int plen = 24 * ((f[pos] >> 6) + 1);
            if (pos + plen > len) break;
            packet_in(f + pos, plen);
            pos += plen;
        }
    }

private:
    struct Assembly { std::vector<uint8_t> data; int next_ci = -1; bool open = false; };
    std::map<int, Assembly> asm_;

    void packet_in(const uint8_t *p, int plen)
    {
        const int addr = ((p[0] & 3) << 8) | p[1];
        if (addr == 0) return;                                           // padding packet
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
