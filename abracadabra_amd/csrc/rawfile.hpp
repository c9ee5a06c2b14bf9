// rawfile.hpp — the reference's raw IQ file formats: plain `.raw` (u8 or s16 IQ at 2.048 Msps)
// and `.uff` with a zero-padded 2048-byte XML header as written by InputDeviceRecorder
// (reference: src/input/inputdevicerecorder.cpp:195-263) and sniffed by RawFileInput::openDevice
// (src/input/rawfileinput.cpp:90-134, parse :345-549).  Only the fields the decode path needs
// are read: sample container, sample rate, centre frequency, data offset and channel count.
#pragma once
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <string>

namespace rawfile {

struct Info {
    int has_header = 0;
    int fmt = -1;              // 0 = u8, 1 = s16, -1 = unknown (caller's setting applies)
    int64_t data_offset = 0;   // first sample byte
    int64_t channel_count = 0; // <Datablock Count=...>: I and Q values, 0 if absent
    int32_t samplerate = 0;    // Hz
    int32_t frequency_khz = 0;
};

inline std::string attr(const std::string &xml, const char *tag, const char *name)
{
    size_t p = xml.find(std::string("<") + tag + " ");        // "<Datablock " must not match "<Datablocks>"
    if (p == std::string::npos) return "";
    const size_t e = xml.find('>', p);
    const std::string key = std::string(name) + "=\"";
    const size_t a = xml.find(key, p);
    if (a == std::string::npos || a > e) return "";
    const size_t q = xml.find('"', a + key.size());
    return q == std::string::npos ? "" : xml.substr(a + key.size(), q - a - key.size());
}

// head: the first bytes of the file (up to 2048 are looked at)
inline Info probe(const uint8_t *head, int n)
{
    Info info;
    constexpr int kPad = 2048;
    int len = 0;
    while (len < n && len < kPad && head[len] != 0) ++len;          // header text ends at the zero padding
    if (len == 0 || len >= kPad || len >= n) return info;            // no terminator inside the padding: plain .raw
    const std::string xml(reinterpret_cast<const char *>(head), static_cast<size_t>(len));
    if (xml.find("<SDR") == std::string::npos) return info;
    const std::string container = attr(xml, "Channels", "Container");
    const std::string bits = attr(xml, "Channels", "Bits");
    if (container == "uint8" || (container.empty() && bits == "8")) info.fmt = 0;
    else if (container == "int16" || (container.empty() && bits == "16")) info.fmt = 1;
    else return info;                                                 // unsupported container: treat as headerless
    if (attr(xml, "Channels", "Ordering") == "MSB") return info;      // big-endian samples are not supported
    info.samplerate = std::atoi(attr(xml, "Samplerate", "Value").c_str());
    info.frequency_khz = std::atoi(attr(xml, "Frequency", "Value").c_str());
    info.channel_count = std::atoll(attr(xml, "Datablock", "Count").c_str());
    const std::string off = attr(xml, "Datablock", "Offset");
    info.data_offset = off.empty() ? kPad : std::atoll(off.c_str());
    info.has_header = 1;
    return info;
}

}  // namespace rawfile
