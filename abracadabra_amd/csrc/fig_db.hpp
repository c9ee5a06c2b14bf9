// fig_db.hpp — Fast Information Group database (ETSI EN 300 401 §5.2, §6, §8).
//
// Host-side consumer of the FIBs the GPU decodes: collects the multiplex
// configuration that the reference's dabsdr library reports through
// dabsdrNtfEnsemble_t, dabsdrServiceListItem_t and dabsdrServiceCompListItem_t
// (reference: lib/linux_x86_64/dabsdr.h:176-243, :301-318; consumer
// src/radiocontrol.cpp:1381-1570).  The reference's parser is closed source;
// this one follows the standard.  Supported: FIG 0/0, 0/1, 0/2, 0/3, 0/5, 0/8, 0/9, 0/10,
// 0/13, 0/14, 0/17, 0/18, 0/19, 1/0, 1/1, 1/4, 1/5.  Unknown FIGs are skipped by their length field.
#pragma once
#include <cstdint>
#include <cstring>
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "dabx_spec.hpp"

namespace figdb {

struct SubChannel {
    int id = -1, start = -1, size = 0;
    bool long_form = false;
    int option = 0, level = 0;        // long form: EEP option 0/1, protection level 1..4
    int uep_index = -1;               // short form: table index
    int kbps = 0;
};

struct UserApp {
    int type = 0;                     // 11-bit user application type (TS 101 756 table 16)
    std::vector<uint8_t> data;        // user application data field (X-PAD: CA/AppTy byte, DG/DSCTy byte, ...)
};

struct Component {
    int tmid = 0;
    int ascty_dscty = 0;
    int subch = -1;                   // stream modes
    int scid = -1;                    // packet mode
    bool primary = false, ca = false;
    int scids = -1;                   // from FIG 0/8, else position
    bool scids_known = false;
    std::string label;
    uint16_t label_flag = 0;
    std::vector<UserApp> apps;        // FIG 0/13
};

struct PacketComponent {              // FIG 0/3
    int scid = -1, subch = -1, dscty = 0, packet_address = -1;
    bool dg_flag = false, ca_org = false;
};

struct AnnouncementSwitch {           // FIG 0/19
    int cluster = 0, subch = 0;
    uint16_t flags = 0;
    bool new_flag = false;
};

struct Service {
    uint32_t sid = 0;
    bool data = false;                // P/D
    int caid = 0;
    std::vector<Component> comp;
    std::string label;
    uint16_t label_flag = 0;
    int pty = -1;
    bool pty_dynamic = false;
    uint16_t asu = 0;                 // FIG 0/18 announcement support flags
    std::vector<uint8_t> clusters;
};

struct Ensemble {
    int eid = -1, ecc = 0, lto = 0, int_table = 0, alarm = 0;
    int cif_count = -1;
    std::string label;
    uint16_t label_flag = 0;
    uint32_t mjd = 0; int hours = 0, minutes = 0, seconds = 0, ms = 0; bool utc_valid = false;
    uint32_t date_hours_minutes = 0;  // first 32 bits of FIG 0/10 as sent: the host decodes MJD at bits 30-14,
                                      // hours at 10-6, minutes at 5-0 (reference: src/dabtables.cpp:124-134)
};

class Database {
public:
    Ensemble ens;
    std::map<int, SubChannel> subch;
    std::map<uint32_t, Service> services;
    std::map<int, int> language;                      // FIG 0/5 short form: SubChId -> language code
    std::map<int, int> language_scid;                 // FIG 0/5 long form: SCId -> language code
    std::map<int, PacketComponent> packet;            // FIG 0/3 by SCId
    std::map<int, int> fec_scheme;                    // FIG 0/14: SubChId -> FEC scheme
    std::map<int, AnnouncementSwitch> switching;      // FIG 0/19 by cluster id
    std::vector<uint32_t> pty_changed;                // SIds whose PTy changed since the owner last looked
    bool switching_changed = false;
    std::vector<std::pair<uint32_t, int>> apps_changed;   // (SId, SCIdS) whose FIG 0/13 user applications changed
    // FIG 0/0 announcing another ensemble than the one in this database (two FIBs in a row, so that one damaged FIB with
    // a matching CRC cannot do it): the owner resets (reference: DABSDR_RESET_NEW_EID, dabsdr.h; radiocontrol.cpp:118-127)
    bool eid_changed = false;
    int eid_candidate = -1, eid_candidate_count = 0;
    int fibs_seen = 0;
    // multiplex reconfiguration (EN 300 401 §6.5): FIGs of the multiplex configuration sent with C/N = 1 describe the
    // NEXT configuration; FIG 0/0 announces the change and the CIF count at which it takes effect
    std::shared_ptr<Database> next;
    bool change_pending = false, reconfigured = false;
    int occurrence = -1;

    void clear() { *this = Database(); }

    // one FIB of 32 bytes whose CRC has already been verified
    void parse_fib(const uint8_t *fib)
    {
        ++fibs_seen;
        int pos = 0;
        while (pos < 30) {
            const uint8_t head = fib[pos];
            if (head == 0xFF) break;
            const int type = head >> 5, len = head & 0x1F;
            if (len == 0 || pos + 1 + len > 30) break;
            const uint8_t *d = fib + pos + 1;
            if (type == 0) fig0(d, len);
            else if (type == 1) fig1(d, len);
            pos += 1 + len;
        }
    }

    bool ensemble_ready() const { return ens.eid >= 0 && !ens.label.empty(); }

    const Service *find_service(uint32_t sid) const
    {
        auto it = services.find(sid);
        return it == services.end() ? nullptr : &it->second;
    }

    // bit rate of an EEP sub-channel from its size (§6.2.1 table 7/8)
    static int eep_kbps(int option, int level, int size)
    {
        static const int a_div[5] = {0, 12, 8, 6, 4}, b_div[5] = {0, 27, 21, 18, 15};
        if (level < 1 || level > 4) return 0;
        return option == 0 ? 8 * size / a_div[level] : (option == 1 ? 32 * size / b_div[level] : 0);
    }

private:
    void fig0(const uint8_t *d, int len)
    {
        const bool pd = (d[0] >> 5) & 1, cn = d[0] >> 7;
        const int ext = d[0] & 0x1F;
        const uint8_t *p = d + 1;
        int n = len - 1;
        if (cn && (ext == 1 || ext == 2 || ext == 3 || ext == 8 || ext == 14)) {   // part of the next multiplex configuration
            if (!next) next = std::make_shared<Database>();
            uint8_t tmp[32];
            std::memcpy(tmp, d, static_cast<size_t>(len));
            tmp[0] &= 0x7F;
            next->fig0(tmp, len);
            return;
        }
        switch (ext) {
        case 0:
            if (n >= 4) {
                const int eid_now = (p[0] << 8) | p[1];
                if (ens.eid >= 0 && eid_now != ens.eid) {
                    eid_candidate_count = eid_now == eid_candidate ? eid_candidate_count + 1 : 1;
                    eid_candidate = eid_now;
                    if (eid_candidate_count >= 2) eid_changed = true;
                    break;                            // nothing of the other ensemble goes into this database
                }
                eid_candidate_count = 0;
                ens.eid = eid_now;
                ens.alarm = (p[2] >> 5) & 1;
                ens.cif_count = (p[2] & 0x1F) * 250 + p[3];
                const bool change = (p[2] >> 6) != 0;
                if (change && n >= 5) occurrence = p[4];
                // the new configuration is valid from the CIF whose count (lower part) equals the occurrence value;
                // a receiver that missed that FIB applies it when the announcement is withdrawn
                if (next && ((change && p[3] == occurrence) || (!change && change_pending))) apply_next();
                change_pending = change;
            }
            break;
        case 1:
            while (n >= 3) {
                SubChannel s;
                s.id = p[0] >> 2;
                s.start = ((p[0] & 3) << 8) | p[1];
                s.long_form = (p[2] >> 7) & 1;
                if (s.long_form) {
                    if (n < 4) return;
                    s.option = (p[2] >> 4) & 7;
                    s.level = ((p[2] >> 2) & 3) + 1;
                    s.size = ((p[2] & 3) << 8) | p[3];
                    s.kbps = eep_kbps(s.option, s.level, s.size);
                    p += 4; n -= 4;
                } else {
                    s.uep_index = p[2] & 0x3F;
                    dabx::Profile prof;
                    if (dabx::uep_profile(s.uep_index, prof, &s.kbps)) { s.size = prof.n_cu; s.level = dabx::uep_rows()[s.uep_index][1]; }
                    p += 3; n -= 3;
                }
                subch[s.id] = s;
            }
            break;
        case 2:
            while (n >= (pd ? 5 : 3)) {
                uint32_t sid;
                if (pd) { sid = (uint32_t(p[0]) << 24) | (p[1] << 16) | (p[2] << 8) | p[3]; p += 4; n -= 4; }
                else { sid = (p[0] << 8) | p[1]; p += 2; n -= 2; }
                const int caid = (p[0] >> 4) & 7, ncomp = p[0] & 0xF;
                ++p; --n;
                if (n < 2 * ncomp) return;
                Service &sv = services[sid];
                sv.sid = sid; sv.data = pd; sv.caid = caid;
                std::vector<Component> comps;
                for (int i = 0; i < ncomp; ++i, p += 2, n -= 2) {
                    Component c;
                    c.tmid = p[0] >> 6;
                    if (c.tmid == 3) {
                        c.scid = ((p[0] & 0x3F) << 6) | (p[1] >> 2);
                    } else {
                        c.ascty_dscty = p[0] & 0x3F;
                        c.subch = p[1] >> 2;
                    }
                    c.primary = (p[1] >> 1) & 1;
                    c.ca = p[1] & 1;
                    c.scids = i;
                    comps.push_back(c);
                }
                // keep labels of components already known
                for (size_t i = 0; i < comps.size() && i < sv.comp.size(); ++i) {
                    comps[i].label = sv.comp[i].label;
                    comps[i].label_flag = sv.comp[i].label_flag;
                }
                sv.comp = comps;
            }
            break;
        case 9:
            if (n >= 3) {
                const int v = p[0] & 0x1F;
                ens.lto = (p[0] & 0x20) ? -v : v;
                ens.ecc = p[1];
                ens.int_table = p[2];
            }
            break;
        case 10:
            if (n >= 4) {
                ens.date_hours_minutes = (uint32_t(p[0]) << 24) | (uint32_t(p[1]) << 16) | (uint32_t(p[2]) << 8) | p[3];
                ens.mjd = ((uint32_t(p[0]) & 0x7F) << 10) | (uint32_t(p[1]) << 2) | (p[2] >> 6);
                const bool utc_long = (p[2] >> 3) & 1;
                ens.hours = ((p[2] & 7) << 2) | (p[3] >> 6);
                ens.minutes = p[3] & 0x3F;
                ens.seconds = ens.ms = 0;
                if (utc_long && n >= 6) { ens.seconds = p[4] >> 2; ens.ms = ((p[4] & 3) << 8) | p[5]; }
                ens.utc_valid = true;
            }
            break;
        case 3:                       // service component in packet mode
            while (n >= 5) {
                PacketComponent pc;
                pc.scid = (p[0] << 4) | (p[1] >> 4);
                pc.ca_org = p[1] & 1;
                pc.dg_flag = !((p[2] >> 7) & 1);      // the bit says "data groups are NOT used"
                pc.dscty = p[2] & 0x3F;
                pc.subch = p[3] >> 2;
                pc.packet_address = ((p[3] & 3) << 8) | p[4];
                const int step = pc.ca_org ? 7 : 5;
                if (n < step) return;
                packet[pc.scid] = pc;
                p += step; n -= step;
            }
            break;
        case 5:                       // service component language
            while (n >= 2) {
                if (p[0] & 0x80) {                    // long form: SCId
                    if (n < 3) return;
                    language_scid[((p[0] & 0x0F) << 8) | p[1]] = p[2];
                    p += 3; n -= 3;
                } else {
                    language[p[0] & 0x3F] = p[1];
                    p += 2; n -= 2;
                }
            }
            break;
        case 8:                       // service component global definition: SCIdS <-> SubChId / SCId
            while (n >= (pd ? 6 : 4)) {
                uint32_t sid;
                if (pd) { sid = (uint32_t(p[0]) << 24) | (p[1] << 16) | (p[2] << 8) | p[3]; p += 4; n -= 4; }
                else { sid = (p[0] << 8) | p[1]; p += 2; n -= 2; }
                const bool ext = p[0] >> 7;
                const int scids = p[0] & 0x0F;
                const bool ls = p[1] >> 7;
                int subch = -1, scid = -1, used = 2;
                if (ls) { if (n < 3) return; scid = ((p[1] & 0x0F) << 8) | p[2]; used = 3; }
                else subch = p[1] & 0x3F;
                if (ext) ++used;                      // Rfa byte
                if (n < used) return;
                auto it = services.find(sid);
                if (it != services.end())
                    for (auto &c : it->second.comp)
                        if ((subch >= 0 && c.tmid != 3 && c.subch == subch) || (scid >= 0 && c.tmid == 3 && c.scid == scid)) {
                            c.scids = scids; c.scids_known = true;
                        }
                p += used; n -= used;
            }
            break;
        case 13:                      // user application information
            while (n >= (pd ? 5 : 3)) {
                uint32_t sid;
                if (pd) { sid = (uint32_t(p[0]) << 24) | (p[1] << 16) | (p[2] << 8) | p[3]; p += 4; n -= 4; }
                else { sid = (p[0] << 8) | p[1]; p += 2; n -= 2; }
                const int scids = p[0] >> 4, napps = p[0] & 0x0F;
                ++p; --n;
                std::vector<UserApp> apps;
                for (int i = 0; i < napps; ++i) {
                    if (n < 2) return;
                    UserApp a;
                    a.type = (p[0] << 3) | (p[1] >> 5);
                    const int dl = p[1] & 0x1F;
                    if (n < 2 + dl) return;
                    a.data.assign(p + 2, p + 2 + dl);
                    apps.push_back(a);
                    p += 2 + dl; n -= 2 + dl;
                }
                auto it = services.find(sid);
                if (it != services.end())
                    for (auto &c : it->second.comp)
                        if (c.scids == scids) {
                            bool same = c.apps.size() == apps.size();
                            for (size_t i = 0; same && i < apps.size(); ++i) same = c.apps[i].type == apps[i].type && c.apps[i].data == apps[i].data;
                            if (!same) { c.apps = apps; apps_changed.emplace_back(sid, scids); }
                        }
            }
            break;
        case 14:                      // FEC sub-channel organisation (packet mode)
            while (n >= 1) { fec_scheme[p[0] >> 2] = p[0] & 3; ++p; --n; }
            break;
        case 17:
            while (n >= 4) {          // editions before V2.1.1 may carry a language (L) and a complementary code (CC)
                const uint32_t sid = (p[0] << 8) | p[1];
                const int l = (p[2] >> 5) & 1, cc = (p[2] >> 4) & 1, used = 4 + l + cc;
                if (n < used) return;
                auto it = services.find(sid);
                if (it != services.end()) {
                    const int pty = p[3 + l] & 0x1F;
                    const bool dyn = p[2] >> 7;
                    if (it->second.pty != pty || it->second.pty_dynamic != dyn) pty_changed.push_back(sid);
                    it->second.pty = pty; it->second.pty_dynamic = dyn;
                }
                p += used; n -= used;
            }
            break;
        case 18:                      // announcement support
            while (n >= 5) {
                const uint32_t sid = (p[0] << 8) | p[1];
                const uint16_t asu = uint16_t((p[2] << 8) | p[3]);
                const int ncl = p[4] & 0x1F;
                if (n < 5 + ncl) return;
                auto it = services.find(sid);
                if (it != services.end()) { it->second.asu = asu; it->second.clusters.assign(p + 5, p + 5 + ncl); }
                p += 5 + ncl; n -= 5 + ncl;
            }
            break;
        case 19:                      // announcement switching
            while (n >= 4) {
                AnnouncementSwitch a;
                a.cluster = p[0];
                a.flags = uint16_t((p[1] << 8) | p[2]);
                a.new_flag = p[3] >> 7;
                const bool region = (p[3] >> 6) & 1;
                a.subch = p[3] & 0x3F;
                const int used = region ? 5 : 4;
                if (n < used) return;
                auto it = switching.find(a.cluster);
                if (it == switching.end() || it->second.flags != a.flags || it->second.subch != a.subch) switching_changed = true;
                switching[a.cluster] = a;
                p += used; n -= used;
            }
            break;
        default: break;
        }
    }

    void apply_next()
    {
        subch = next->subch;
        packet = next->packet;
        fec_scheme = next->fec_scheme;
        std::map<uint32_t, Service> now;
        for (auto &kv : next->services) {                 // organisation from the new configuration, service information (labels,
            Service sv = kv.second;                       // programme type, announcements, user applications) carried over
            auto old = services.find(kv.first);
            if (old != services.end()) {
                sv.label = old->second.label; sv.label_flag = old->second.label_flag;
                sv.pty = old->second.pty; sv.pty_dynamic = old->second.pty_dynamic;
                sv.asu = old->second.asu; sv.clusters = old->second.clusters;
                for (auto &c : sv.comp)
                    for (const auto &oc : old->second.comp)
                        if (oc.tmid == c.tmid && oc.subch == c.subch && oc.scid == c.scid) {
                            c.label = oc.label; c.label_flag = oc.label_flag; c.apps = oc.apps;
                            if (!c.scids_known) { c.scids = oc.scids; c.scids_known = oc.scids_known; }
                        }
            }
            now[kv.first] = sv;
        }
        services = now;
        next.reset();
        reconfigured = true;
    }

    static std::string label16(const uint8_t *c)
    {
        std::string s(reinterpret_cast<const char *>(c), 16);
        return s;
    }

    void fig1(const uint8_t *d, int len)
    {
        const int ext = d[0] & 7;
        const uint8_t *p = d + 1;
        const int n = len - 1;
        if (ext == 0 && n >= 20) {
            const int eid = (p[0] << 8) | p[1];
            if (ens.eid < 0 || ens.eid == eid) { ens.label = label16(p + 2); ens.label_flag = uint16_t((p[18] << 8) | p[19]); }
        } else if (ext == 1 && n >= 20) {
            const uint32_t sid = (p[0] << 8) | p[1];
            Service &sv = services[sid];
            sv.sid = sid;
            sv.label = label16(p + 2);
            sv.label_flag = uint16_t((p[18] << 8) | p[19]);
        } else if (ext == 5 && n >= 22) {
            const uint32_t sid = (uint32_t(p[0]) << 24) | (p[1] << 16) | (p[2] << 8) | p[3];
            Service &sv = services[sid];
            sv.sid = sid; sv.data = true;
            sv.label = label16(p + 4);
            sv.label_flag = uint16_t((p[20] << 8) | p[21]);
        } else if (ext == 4 && n >= 21) {
            const bool pd = (p[0] >> 7) & 1;
            const int scids = p[0] & 0xF;
            if (pd) return;
            const uint32_t sid = (p[1] << 8) | p[2];
            auto it = services.find(sid);
            if (it == services.end()) return;
            for (auto &c : it->second.comp)
                if (c.scids == scids) { c.label = label16(p + 3); c.label_flag = uint16_t((p[19] << 8) | p[20]); }
        }
    }
};

}  // namespace figdb
