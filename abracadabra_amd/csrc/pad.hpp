// pad.hpp — programme associated data of a DAB+ audio service (ETSI TS 102 563 §5.4: PAD travels in a
// data_stream_element at the start of every access unit; ETSI EN 300 401 §7.4: F-PAD / X-PAD, §7.4.5 dynamic
// label, §7.4.5.2 X-PAD data groups for MOT).
//
// Host-side consumer of the access units k_superframe delivers; produces what the reference's dabsdr library
// hands to dabsdrDynamicLabelCBFunc_t (dabsdr.h:81-86: one dynamic-label segment = 2 prefix bytes + character
// field, CRC already verified — consumer src/data/dldecoder.cpp:82-190) and to dabsdrDataGroupCBFunc_t
// (dabsdr.h:89-96: one complete MSC data group incl. its CRC, which the host re-checks — consumer
// src/data/mscdatagroup.cpp:31-58).  Written from the standards; the reference's own parser is in its binary.
#pragma once
#include <cstdint>
#include <functional>
#include <vector>

namespace pad {

inline uint16_t crc16(const uint8_t *d, int n)             // CRC-16-CCITT, initial word all ones, result inverted
{
    unsigned c = 0xFFFF;
    for (int i = 0; i < n; ++i) {
        c ^= static_cast<unsigned>(d[i]) << 8;
        for (int b = 0; b < 8; ++b) c = (c & 0x8000) ? ((c << 1) ^ 0x1021) & 0xFFFF : (c << 1) & 0xFFFF;
    }
    return static_cast<uint16_t>(~c & 0xFFFF);
}

struct Stats { uint32_t pads = 0, dl_ok = 0, dl_crc_err = 0, dg_ok = 0, dg_crc_err = 0; };

class Decoder {
public:
    std::function<void(const uint8_t *, int)> on_dynamic_label;              // prefix (2) + character / command field
    std::function<void(int app_type, const uint8_t *, int)> on_data_group;   // X-PAD application type (12 = MOT), data group
    Stats stats;

    void reset() { dl_.clear(); dg_.clear(); dg_len_ = -1; last_app_ = -1; dg_app_ = -1; }

    // one DAB+ access unit: the PAD is the payload of a leading data_stream_element (id_syn_ele 4)
    void feed_dabplus_au(const uint8_t *au, int len)
    {
        if (len < 2 || (au[0] >> 5) != 4) return;            // 3 bits id, 4 bits instance tag, 1 bit byte-align flag
        int count = au[1], pos = 2;
        if (count == 255) {
            if (len < 3) return;
            count += au[2];
            pos = 3;
        }
        if (count < 2 || pos + count > len) return;
        feed_pad(au + pos, count);
    }

    // One MPEG-1/2 Layer II frame of a DAB audio sub-channel (EN 300 401 §7.3.2.4, §B): the frame ends with
    // X-PAD | ScF-CRC (2 bytes below 56 kbit/s per channel, else 4) | F-PAD (2 bytes).  24 kHz (LSF) frames span two
    // logical frames; the caller passes whole audio frames.
    void feed_mp2_frame(const uint8_t *f, int len)
    {
        if (len < 8 || f[0] != 0xFF || (f[1] & 0xF0) != 0xF0) return;       // 12-bit sync word
        const bool lsf = !((f[1] >> 3) & 1);                                 // ID bit: 1 = 48 kHz (MPEG-1), 0 = 24 kHz (MPEG-2 LSF)
        static const int kBr1[16] = {0, 32, 48, 56, 64, 80, 96, 112, 128, 160, 192, 224, 256, 320, 384, 0};
        static const int kBr2[16] = {0, 8, 16, 24, 32, 40, 48, 56, 64, 80, 96, 112, 128, 144, 160, 0};
        const int kbps = (lsf ? kBr2 : kBr1)[f[2] >> 4];
        const bool mono = (f[3] >> 6) == 3;
        if (!kbps) return;
        const int scf = (kbps / (mono ? 1 : 2) >= 56) ? 4 : 2;
        const int room = len - 2 - scf - 4;                                  // bytes that can hold X-PAD (after the 4-byte header)
        if (room <= 0) return;
        const int nx = room < 196 ? room : 196;                              // 4 CIs + 4 x 48 bytes is the largest X-PAD
        uint8_t tmp[196 + 2];
        for (int i = 0; i < nx; ++i) tmp[i] = f[len - 2 - scf - nx + i];
        tmp[nx] = f[len - 2]; tmp[nx + 1] = f[len - 1];
        feed_pad(tmp, nx + 2, false);
    }

    // pad: the whole PAD field, its last two bytes are the F-PAD; the X-PAD before them is read backwards
    // exact: n is the true PAD length (DAB+); otherwise the X-PAD may be shorter than n - 2 (MPEG Layer II frames)
    void feed_pad(const uint8_t *pad, int n, bool exact = true)
    {
        if (n < 2) return;
        ++stats.pads;
        const uint8_t fpad0 = pad[n - 2], fpad1 = pad[n - 1];
        if ((fpad0 >> 6) != 0) return;                        // F-PAD type 00 carries the X-PAD indicator
        const int xind = (fpad0 >> 4) & 3;
        const bool ci_flag = (fpad1 >> 1) & 1;
        const int xlen = xind == 1 ? 4 : (xind == 2 ? n - 2 : 0);
        if (xlen <= 0 || xlen > n - 2) return;
        std::vector<uint8_t> x(static_cast<size_t>(xlen));
        for (int i = 0; i < xlen; ++i) x[static_cast<size_t>(i)] = pad[n - 3 - i];
        static const int kLen[8] = {4, 6, 8, 12, 16, 24, 32, 48};
        if (xind == 1) {                                      // short X-PAD: one contents indicator + 3 bytes, or 4 bytes of continuation
            if (ci_flag) subfield(x[0] & 0x1F, x.data() + 1, 3, true);
            else if (last_app_ >= 0) subfield(last_app_, x.data(), 4, false);
            return;
        }
        if (!ci_flag) {                                       // variable size without indicators: continuation of the last application
            if (last_app_ >= 0 && exact) subfield(last_app_, x.data(), xlen, false);
            return;
        }
        int apps[4], lens[4], nci = 0, p = 0;
        while (nci < 4 && p < xlen) {
            const uint8_t ci = x[static_cast<size_t>(p++)];
            const int app = ci & 0x1F;
            if (app == 0) break;                              // end marker
            if (app == 31) { ++p; continue; }                 // extension byte: application types >= 32 are not used by DL / MOT
            apps[nci] = app; lens[nci] = kLen[ci >> 5]; ++nci;
        }
        for (int k = 0; k < nci; ++k) {
            if (p + lens[k] > xlen) break;
            subfield(apps[k], x.data() + p, lens[k], true);
            p += lens[k];
        }
    }

private:
    std::vector<uint8_t> dl_, dg_;
    int dg_len_ = -1, last_app_ = -1, dg_app_ = -1;

    static int dl_total(const std::vector<uint8_t> &b)       // bytes of the dynamic label data group incl. CRC, -1 = unknown yet
    {
        if (b.size() < 2) return -1;
        if (!(b[0] & 0x10)) return 2 + (b[0] & 0x0F) + 1 + 2;                     // label segment: prefix, 1..16 characters, CRC
        return (b[0] & 0x0F) == 2 ? 2 + (b[1] & 0x0F) + 1 + 2 : 4;                // DL Plus command field / other commands: prefix + CRC
    }

    void subfield(int app, const uint8_t *d, int n, bool with_ci)
    {
        (void)with_ci;
        last_app_ = app;
        switch (app) {
        case 1:                                               // data group length indicator: 14-bit length + CRC
            if (n >= 4 && crc16(d, 2) == ((d[2] << 8) | d[3])) dg_len_ = ((d[0] & 0x3F) << 8) | d[1];
            else dg_len_ = -1;
            last_app_ = -1;
            break;
        case 2:                                               // dynamic label: start of a data group
            dl_.assign(d, d + n);
            dl_done();
            last_app_ = 3;
            break;
        case 3:
            if (dl_.empty()) break;
            dl_.insert(dl_.end(), d, d + n);
            dl_done();
            break;
        case 12: case 14: case 16: case 18: case 20: case 22:  // MOT and other data-group applications: start
            dg_.clear();
            dg_app_ = app;
            if (dg_len_ > 0) { dg_.assign(d, d + n); dg_done(); }
            last_app_ = app + 1;
            break;
        case 13: case 15: case 17: case 19: case 21: case 23:  // ... continuation
            if (dg_.empty() || dg_app_ != app - 1) break;
            dg_.insert(dg_.end(), d, d + n);
            dg_done();
            break;
        default: break;
        }
    }

    void dl_done()
    {
        const int total = dl_total(dl_);
        if (total < 0 || static_cast<int>(dl_.size()) < total) return;
        if (crc16(dl_.data(), total - 2) == ((dl_[static_cast<size_t>(total) - 2] << 8) | dl_[static_cast<size_t>(total) - 1])) {
            ++stats.dl_ok;
            if (on_dynamic_label) on_dynamic_label(dl_.data(), total - 2);
        } else ++stats.dl_crc_err;
        dl_.clear();
    }

    void dg_done()
    {
        if (dg_len_ <= 0 || static_cast<int>(dg_.size()) < dg_len_) return;
        const bool has_crc = dg_[0] & 0x40;                  // CRC flag of the MSC data group header (EN 300 401 §5.3.3.1)
        if (!has_crc || (dg_len_ >= 4 && crc16(dg_.data(), dg_len_ - 2) == ((dg_[static_cast<size_t>(dg_len_) - 2] << 8) | dg_[static_cast<size_t>(dg_len_) - 1]))) {
            ++stats.dg_ok;
            if (on_data_group) on_data_group(dg_app_, dg_.data(), dg_len_);
        } else ++stats.dg_crc_err;
        dg_.clear();
        dg_len_ = -1;
    }
};

}  // namespace pad
