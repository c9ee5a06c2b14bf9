// dabsdr_shim.cpp — the reference's 24-function dabsdr C API over a one-stream
// dabx context (include/dabsdr_amd.h cites the declaration each function replaces).
//
// Threading mirrors the reference: dabsdr() spawns one
// worker thread named "dabsdr"; every callback fires on that thread; requests
// are queued from any thread and answered by notifications.  The worker pulls
// whole transmission frames through the registered input callback
// (reference contract: src/input/inputdevice.cpp:70-108 — blocking, fills the
// buffer completely, zeros when flushed) straight into page-locked memory; the
// GPU turns the floats into the ring's s16 (dabx_push_resampled_from), so the
// reference's raw-file input drops in unchanged and the host never touches a
// sample.  When the callback returns faster than real time (a file read as fast
// as it goes, SURVEY.md §8b "un-paced") up to kMaxBatch frames are decoded per
// step, and the step runs while the callback fills the next buffer.
#include "../../include/dabsdr_amd.h"
#include "../../include/dabx.h"
#include "fig_db.hpp"
#include "packet.hpp"
#include "pad.hpp"
#include "tii.hpp"

#include <pthread.h>

#include <atomic>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstdio>
#include <cstring>
#include <deque>
#include <set>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

namespace {

enum class Req { Tune, GetEnsemble, GetServiceList, GetServiceComponents, GetUserAppList, GetAnnouncementSupport,
                 ServiceSelection, ServiceStop, XPadAppStart, SetPeriodicNotify, SetTII, SignalSpectrum, InjectFibs, Exit };

struct Request {
    Req kind;
    uint32_t a = 0;      // frequency / SId
    int32_t b = 0;       // SCIdS / period / enable
    int32_t c = 0;       // decoder id / cfg
    std::vector<uint8_t> blob;   // InjectFibs (test hook): FIBs of 32 bytes
};

constexpr int kPullChunk = 16384;        // complex samples per input-callback call (uint16_t length)
constexpr int kMaxBatch = 8;             // frames per decode step at most (input not paced)
constexpr int kRingFrames = 32;          // device ring: the step in flight, the one being filled, one frame of history, slack

// what one selected sub-channel contributed to a decode step, identified by the selection's serial number: a selection
// made or dropped while the step was in flight must not be served from bytes that were laid out for another list
struct SubSnap {
    uint64_t serial = 0;
    int kbps = 0;
    size_t off = 0;                       // byte offset inside a CIF record
    bool dabplus = false;
    std::vector<dabx_superframe_t> recs;  // DAB+: the super frames the step completed ...
    std::vector<uint8_t> data;            // ... and their RS-corrected bytes
    uint32_t frames_seen = 0;             // valid logical frames of the sub-channel so far, the step's included
};

// everything the worker needs from one decode step, fetched before the next step is submitted
struct StepResults {
    int n_frames = 0;
    std::vector<uint8_t> fib, ok;
    std::vector<dabx_sync_rec_t> rec;
    dabx_stream_state_t st = {};
    bool have_null = false, have_spec = false;
    std::vector<float> spectrum, null_power;      // of the step's last frame
    int shift = 0;                                // the samples were scaled by 2^shift
    int32_t peak = 0;                             // their largest |I|, |Q| after scaling
    size_t stride = 0;
    bool have_msc = false;
    std::vector<uint8_t> msc, valid;
    std::vector<SubSnap> subs;
};

}  // namespace

// one selected service component (dabsdrRequest_ServiceSelection): what it needs from the GPU context and its
// host-side decoders.  The reference runs a primary audio, a secondary audio and data decoders side by side
// (dabsdrDecoderId_t, dabsdr.h:22-28).
struct Selection {
    uint32_t sid = 0;
    int scids = -1;
    dabsdrDecoderId_t id = DABSDR_ID_AUDIO_PRIMARY;
    bool packet = false;                  // TMId 3: packet mode data
    int ascty = 0;                        // audio: 63 = DAB+, 0 = MPEG Layer II; packet: DSCTy
    int kbps = 0, scid = 0;
    dabx_subch_t sub = {};                // the sub-channel to decode
    int subch_id = -1;
    std::vector<figdb::UserApp> apps;     // FIG 0/13: maps X-PAD application types to user application types
    pad::Decoder pad;                     // X-PAD -> dynamic label / data groups
    packet::Decoder pkt;                  // packet mode -> data groups
    std::set<int> xpad_on;                // X-PAD application types the host started (dabsdrRequest_XPadAppStart)
    std::vector<uint8_t> mp2_half;        // first half of a 24 kHz Layer II frame
    uint64_t serial = 0;                  // unique per selection (StepResults refer to it)
};

struct dabsdr_s {
    dabx_ctx *ctx = nullptr;
    std::thread worker;
    std::mutex mu;
    std::condition_variable cv;
    std::deque<Request> queue;
    std::atomic<bool> exit_req{false};
    bool started = false;

    dabsdrInputFunc_t input = nullptr, dummy = nullptr;
    dabsdrAudioCBFunc_t audio_cb = nullptr; void *audio_ctx = nullptr;
    dabsdrDynamicLabelCBFunc_t dl_cb = nullptr; void *dl_ctx = nullptr;
    dabsdrDataGroupCBFunc_t dg_cb = nullptr; void *dg_ctx = nullptr;
    dabsdrSpectrumCBFunc_t spec_cb = nullptr; void *spec_ctx = nullptr;
    dabsdrNotificationCBFunc_t ntf_cb = nullptr; void *ntf_ctx = nullptr;

    // worker-thread state
    uint32_t frequency = 0;
    int gain_shift = 0;                  // float -> s16 scaling: x * 2^gain_shift
    bool gain_set = false;
    bool locked_hint = false;            // the last step left the receiver locked: whole frames are collected; while it is not, every chunk is
                                         // handed on at once, so that the lock comes as early as the samples allow (2.1 frames, not 3)
    float *fbuf[2] = {nullptr, nullptr}; // page-locked: kMaxBatch frames of float IQ each, filled by the input callback, read by the GPU
    bool buf_busy[2] = {false, false};   // a copy out of the buffer may still be in flight
    int cur = 0, fill = 0;               // buffer being filled / complex samples in it
    bool inflight = false;               // a decode step has been submitted and not waited for
    int inflight_frames = 0, inflight_shift = 0;
    std::vector<SubSnap> inflight_subs;
    std::deque<StepResults> done;        // fetched, not yet turned into callbacks
    double pull_s = 0.096;               // running estimate of the time the input callback takes per frame
    int batch = 1;                       // frames per step: what arrives within 48 ms, at most kMaxBatch
    bool pipelined = false;              // the step runs while the next buffer is filled (input faster than ~10 x real time)
    uint64_t next_serial = 1;
    int64_t pushed = 0;                  // samples handed to the context since it was created
    std::deque<std::pair<int64_t, int>> gain_epochs;   // (first sample, gain shift): where the gain changed, for the spectra's units
    int msc_stride = 0;                  // bytes per CIF of the running selections, as the GPU context has them
    std::atomic<bool> worker_done{false};
    dabsdrSyncLevel_t sync_level = DABSDR_SYNC_LEVEL_NO_SYNC;
    int period_log2 = -1;                // periodic notification every 2^n frames, <0 = off
    int period_frames = 0;
    uint32_t fib_err_acc = 0;
    figdb::Database db;
    std::vector<std::unique_ptr<Selection>> sel;   // running decoders: primary / secondary audio, data components
    uint32_t per_sf = 0, per_au_ok = 0, per_au_bad = 0, per_rs_corr = 0, per_rs_fail = 0;   // primary audio since the last periodic notification
    int per_kbps = 0;
    uint32_t audio_bytes_acc = 0;
    std::vector<figdb::UserApp> app_snapshot;
    bool spectrum_on = false, tii_on = false;
    int tii_mode = DABSDR_TII_MODE_DEFAULT;
    std::vector<float> spectrum, null_power;
    float tii_folded[384] = {0};
    std::vector<figdb::Service> list_snapshot;       // for the list getters (valid during a callback)
    std::vector<figdb::Component> comp_snapshot;
    uint32_t comp_sid = 0;
};

namespace {

void notify(dabsdr_s *h, dabsdrNotificationId_t nid, dabsdrNotificationStatus_t st, const void *data, uint16_t len)
{
    if (!h->ntf_cb) return;
    dabsdrNotificationCBData_t d;
    d.nid = nid; d.status = st; d.len = len; d.pData = data;
    h->ntf_cb(&d, h->ntf_ctx);
}

void fill_label(dabsdrLabel_t &l, const std::string &s, uint16_t flag)
{
    std::memset(&l, 0, sizeof l);
    std::memcpy(l.str, s.data(), s.size() < 16 ? s.size() : 16);
    l.charField = flag;
    l.charset = 0;
}

int protection_enum(const figdb::SubChannel &sc)
{
    // numbering of the reference's DabProtectionLevel (src/dabtables.h:35-55):
    // UEP 1..5 = 1..5, EEP 1-A..4-A = 6..9, EEP 1-B..4-B = 10..13
    if (!sc.long_form) return sc.level;             // UEP 1..5
    return (sc.option == 0 ? 5 : 9) + sc.level;
}

int get_service_item(dabsdrHandle_t h, uint8_t idx, dabsdrServiceListItem_t *out)
{
    if (!h || !out || idx >= h->list_snapshot.size()) return -1;
    const figdb::Service &s = h->list_snapshot[idx];
    std::memset(out, 0, sizeof *out);
    out->sid = s.sid;
    fill_label(out->label, s.label, s.label_flag);
    out->pty.s = out->pty.d = s.pty < 0 ? 255 : static_cast<uint8_t>(s.pty);
    out->CAId = static_cast<uint8_t>(s.caid);
    return 0;
}

int get_component_item(dabsdrHandle_t h, uint8_t idx, dabsdrServiceCompListItem_t *out)
{
    if (!h || !out || idx >= h->comp_snapshot.size()) return -1;
    const figdb::Component &c = h->comp_snapshot[idx];
    std::memset(out, 0, sizeof *out);
    out->SCIdS = static_cast<uint8_t>(c.scids);
    out->SubChAddr = -1;
    // P/S bit of FIG 0/2 as the reference reports it: masked, not shifted (the reference's library
    // returned ps = 2 for a primary component; the host only tests non-zero, radiocontrol.cpp:1522-1523)
    out->ps = c.primary ? 2 : 0;
    out->CAflag = c.ca;
    out->TMId = static_cast<uint8_t>(c.tmid);
    out->numUserApps = static_cast<uint8_t>(c.apps.size());
    fill_label(out->label, c.label, c.label_flag);
    int subch = c.subch;
    if (c.tmid == 3) {                                  // packet mode: FIG 0/3 links the SCId to a sub-channel
        auto pk = h->db.packet.find(c.scid);
        if (pk != h->db.packet.end()) subch = pk->second.subch;
        auto lg = h->db.language_scid.find(c.scid);
        if (lg != h->db.language_scid.end()) out->lang = static_cast<uint8_t>(lg->second);
    } else {
        auto lg = h->db.language.find(c.subch);
        if (lg != h->db.language.end()) out->lang = static_cast<uint8_t>(lg->second);
    }
    auto it = h->db.subch.find(subch);
    if (c.tmid != 3 && it != h->db.subch.end()) {
        const figdb::SubChannel &sc = it->second;
        out->SubChId = static_cast<uint8_t>(sc.id);
        out->SubChAddr = static_cast<int16_t>(sc.start);
        out->SubChSize = static_cast<uint16_t>(sc.size);
        out->protectionLevel = static_cast<uint8_t>(protection_enum(sc));
        if (!sc.long_form) out->uepIdx = static_cast<uint8_t>(sc.uep_index);
        if (c.tmid == 0) { out->streamAudio.ASCTy = static_cast<uint8_t>(c.ascty_dscty); out->streamAudio.bitRate = static_cast<uint16_t>(sc.kbps); }
        else { out->streamData.DSCTy = static_cast<uint8_t>(c.ascty_dscty); out->streamData.bitRate = static_cast<uint16_t>(sc.kbps); }
    } else if (c.tmid == 3) {
        out->packetData.SCId = static_cast<uint16_t>(c.scid);
        out->packetData.packetAddress = -1;
        auto pk = h->db.packet.find(c.scid);
        if (pk != h->db.packet.end()) {
            out->packetData.DSCTy = static_cast<uint8_t>(pk->second.dscty);
            out->packetData.DGflag = pk->second.dg_flag ? 1 : 0;
            out->packetData.packetAddress = static_cast<int16_t>(pk->second.packet_address);
            if (it != h->db.subch.end()) {
                const figdb::SubChannel &sc = it->second;
                out->SubChId = static_cast<uint8_t>(sc.id);
                out->SubChAddr = static_cast<int16_t>(sc.start);
                out->SubChSize = static_cast<uint16_t>(sc.size);
                out->protectionLevel = static_cast<uint8_t>(protection_enum(sc));
                auto fe = h->db.fec_scheme.find(sc.id);
                out->fecScheme = fe != h->db.fec_scheme.end() ? static_cast<uint8_t>(fe->second) : 0;
            }
        }
    }
    return 0;
}

void reset_receiver(dabsdr_s *h, dabsdrNtfResetFlags_t flag)
{
    h->db.clear();
    h->sync_level = DABSDR_SYNC_LEVEL_NO_SYNC;
    h->fib_err_acc = 0; h->period_frames = 0;
    h->sel.clear();
    notify(h, DABSDR_NID_RESET, DABSDR_NSTAT_SUCCESS, &flag, 0);
}

void drain(dabsdr_s *h);

// (re)program the GPU context with the sub-channels of all running selections, in the order of h->sel
bool apply_selections(dabsdr_s *h)
{
    if (!h->ctx) return false;
    drain(h);                                              // the context takes a new layout only between steps
    std::vector<dabx_subch_t> subs;
    uint64_t dabplus = 0;
    for (size_t k = 0; k < h->sel.size(); ++k) {
        subs.push_back(h->sel[k]->sub);
        if (!h->sel[k]->packet && h->sel[k]->ascty == 63) dabplus |= 1ull << k;
    }
    const int stride = dabx_set_subchannels(h->ctx, 0, static_cast<int>(subs.size()), subs.empty() ? nullptr : subs.data());
    if (stride < 0) return false;
    h->msc_stride = stride;
    if (dabplus && dabx_set_dabplus(h->ctx, 0, dabplus) != DABX_OK) return false;
    return true;                                           // running decoders (and their super frame state on the GPU) carry on
}

// The same, insisting: a selection the GPU context refuses (a sub-channel a corrupt or hostile FIC describes beyond the
// 864 CU, an unknown protection profile) is dropped and reported stopped, until the context and h->sel agree again.
void apply_selections_or_drop(dabsdr_s *h)
{
    while (!apply_selections(h)) {
        if (h->sel.empty() || !h->ctx) { h->msc_stride = 0; return; }
        // find one selection the context refuses on its own; failing that, give up the newest
        size_t victim = h->sel.size() - 1;
        for (size_t k = 0; k < h->sel.size(); ++k)
            if (dabx_set_subchannels(h->ctx, 0, 1, &h->sel[k]->sub) < 0) { victim = k; break; }
        dabsdrNtfServiceStop_t stop = {h->sel[victim]->sid, static_cast<uint8_t>(h->sel[victim]->scids), h->sel[victim]->id};
        h->sel.erase(h->sel.begin() + static_cast<long>(victim));
        notify(h, DABSDR_NID_SERVICE_STOP, DABSDR_NSTAT_SERVICE_NOT_SUPPORTED, &stop, sizeof stop);
    }
}

void wire_callbacks(dabsdr_s *h, Selection *sp)
{
    sp->pad.on_dynamic_label = [h, sp](const uint8_t *d, int n) {
        if (!h->dl_cb) return;
        dabsdrDynamicLabelCBData_t cb = {sp->id, static_cast<uint16_t>(n), d};
        h->dl_cb(&cb, h->dl_ctx);
    };
    sp->pad.on_data_group = [h, sp](int xpad_app, const uint8_t *d, int n) {
        // X-PAD applications run only while the host has them started (radiocontrol.cpp:601-618 starts / stops the slide show)
        if (!h->dg_cb || !sp->xpad_on.count(xpad_app)) return;
        uint16_t type = xpad_app == 12 ? 0x002 : 0;                   // default: MOT slide show; else what FIG 0/13 announces
        for (const auto &a : sp->apps)
            if (!a.data.empty() && (a.data[0] & 0x1F) == xpad_app) type = static_cast<uint16_t>(a.type);
        dabsdrDataGroupCBData_t cb = {sp->id, 0, type, static_cast<uint16_t>(n), d};
        h->dg_cb(&cb, h->dg_ctx);
    };
    sp->pkt.on_data_group = [h, sp](int, const uint8_t *d, int n) {
        if (!h->dg_cb) return;
        const uint16_t type = sp->apps.empty() ? 0 : static_cast<uint16_t>(sp->apps[0].type);
        dabsdrDataGroupCBData_t cb = {sp->id, static_cast<uint16_t>(sp->scid), type, static_cast<uint16_t>(n), d};
        h->dg_cb(&cb, h->dg_ctx);
    };
}

bool after_fibs(dabsdr_s *h);

void handle_request(dabsdr_s *h, const Request &r)
{
    switch (r.kind) {
    case Req::Tune: {
        if (r.a == 0) {
            reset_receiver(h, DABSDR_RESET_INIT);
            h->frequency = 0;
            uint32_t f = 0;
            notify(h, DABSDR_NID_TUNE, DABSDR_NSTAT_SUCCESS, &f, 0);
        } else {
            h->frequency = r.a;
            h->gain_set = false; h->locked_hint = false;
            h->fill = 0; h->inflight = false; h->done.clear();
            h->pushed = 0; h->gain_epochs.clear();
            h->buf_busy[0] = h->buf_busy[1] = false;       // dabx_destroy drains the copy stream
            h->msc_stride = 0;
            // a fresh context state: drop everything buffered so far
            if (h->ctx) {
                dabx_config_t cfg = {1, DABX_FMT_S16, static_cast<int64_t>(kRingFrames) * DABX_FRAME_SAMPLES, kMaxBatch, 0};
                dabx_destroy(h->ctx);
                h->ctx = nullptr;
                if (dabx_create(&cfg, &h->ctx) != DABX_OK) h->ctx = nullptr;
                else dabx_enable_spectrum(h->ctx, (h->spectrum_on ? 1 : 0) | 2);
            }
            uint32_t f = r.a;
            notify(h, DABSDR_NID_TUNE, h->ctx ? DABSDR_NSTAT_SUCCESS : DABSDR_NSTAT_GENERIC_ERROR, &f, 0);
            reset_receiver(h, DABSDR_RESET_INIT);
        }
        break;
    }
    case Req::GetEnsemble: {
        dabsdrNtfEnsemble_t e;
        std::memset(&e, 0, sizeof e);
        e.frequency = h->frequency;
        const bool ok = h->db.ens.eid >= 0;
        e.ueid = ok ? (static_cast<uint32_t>(h->db.ens.ecc) << 16) | static_cast<uint32_t>(h->db.ens.eid) : 0;
        e.LTO = static_cast<int8_t>(h->db.ens.lto);
        e.intTable = static_cast<uint8_t>(h->db.ens.int_table);
        e.alarm = static_cast<uint8_t>(h->db.ens.alarm);
        fill_label(e.label, h->db.ens.label, h->db.ens.label_flag);
        notify(h, DABSDR_NID_ENSEMBLE_INFO, ok ? DABSDR_NSTAT_SUCCESS : DABSDR_NSTAT_GENERIC_ERROR, &e, sizeof e);
        break;
    }
    case Req::GetServiceList: {
        h->list_snapshot.clear();
        for (const auto &kv : h->db.services)
            if (!kv.second.comp.empty()) h->list_snapshot.push_back(kv.second);
        dabsdrNtfServiceList_t l;
        l.numServices = static_cast<uint8_t>(h->list_snapshot.size() > 255 ? 255 : h->list_snapshot.size());
        l.getServiceListItem = get_service_item;
        notify(h, DABSDR_NID_SERVICE_LIST, DABSDR_NSTAT_SUCCESS, &l, sizeof l);
        break;
    }
    case Req::GetServiceComponents: {
        dabsdrNtfServiceComponentList_t l;
        std::memset(&l, 0, sizeof l);
        l.SId = r.a;
        l.getServiceComponentListItem = get_component_item;
        const figdb::Service *s = h->db.find_service(r.a);
        h->comp_snapshot.clear();
        if (s) h->comp_snapshot = s->comp;
        h->comp_sid = r.a;
        l.numServiceComponents = static_cast<uint8_t>(h->comp_snapshot.size());
        notify(h, DABSDR_NID_SERVICE_COMPONENT_LIST, s ? DABSDR_NSTAT_SUCCESS : DABSDR_NSTAT_SERVICE_NOT_FOUND, &l, sizeof l);
        break;
    }
    case Req::GetUserAppList: {
        dabsdrNtfUserAppList_t l;
        std::memset(&l, 0, sizeof l);
        l.SId = r.a; l.SCIdS = static_cast<uint8_t>(r.b);
        h->app_snapshot.clear();
        const figdb::Service *s = h->db.find_service(r.a);
        if (s)
            for (const auto &c : s->comp)
                if (c.scids == static_cast<int>(r.b)) h->app_snapshot = c.apps;
        l.numUserApps = static_cast<uint8_t>(h->app_snapshot.size());
        l.getUserAppListItem = [](dabsdrHandle_t hh, uint8_t idx, dabsdrUserAppListItem_t *out) {
            if (!hh || !out || idx >= hh->app_snapshot.size()) return -1;
            const figdb::UserApp &a = hh->app_snapshot[idx];
            std::memset(out, 0, sizeof *out);
            out->type = static_cast<uint16_t>(a.type);
            out->dataLen = static_cast<uint8_t>(std::min<size_t>(a.data.size(), sizeof out->data));
            std::memcpy(out->data, a.data.data(), out->dataLen);
            return 0;
        };
        notify(h, DABSDR_NID_USER_APP_LIST, s ? DABSDR_NSTAT_SUCCESS : DABSDR_NSTAT_SERVICE_NOT_FOUND, &l, sizeof l);
        break;
    }
    case Req::GetAnnouncementSupport: {
        dabsdrNtfAnnouncementSupport_t a;
        std::memset(&a, 0, sizeof a);
        a.SId = r.a;
        if (const figdb::Service *s = h->db.find_service(r.a)) {
            a.ASu = s->asu;
            a.numClusterIds = static_cast<uint8_t>(std::min<size_t>(s->clusters.size(), sizeof a.clusterIds));
            std::memcpy(a.clusterIds, s->clusters.data(), a.numClusterIds);
        }
        notify(h, DABSDR_NID_ANNOUNCEMENT_SUPPORT, DABSDR_NSTAT_SUCCESS, &a, sizeof a);
        break;
    }
    case Req::ServiceSelection: {
        dabsdrNtfServiceSelection_t s = {r.a, static_cast<uint8_t>(r.b), static_cast<dabsdrDecoderId_t>(r.c)};
        dabsdrNotificationStatus_t st = DABSDR_NSTAT_SERVICE_NOT_FOUND;
        const figdb::Service *sv = h->db.find_service(r.a);
        if (sv && h->ctx) {
            for (const auto &c : sv->comp) {
                if (c.scids != static_cast<int>(r.b)) continue;
                int subch = c.subch;
                const figdb::PacketComponent *pk = nullptr;
                if (c.tmid == 3) {                        // packet mode: FIG 0/3 names the sub-channel and the packet address
                    auto pi = h->db.packet.find(c.scid);
                    if (pi == h->db.packet.end()) { st = DABSDR_NSTAT_SERVICE_NOT_READY; break; }
                    pk = &pi->second;
                    subch = pk->subch;
                }
                auto it = h->db.subch.find(subch);
                if (it == h->db.subch.end()) { st = DABSDR_NSTAT_SERVICE_NOT_READY; break; }
                auto sp = std::make_unique<Selection>();
                sp->sid = r.a; sp->scids = static_cast<int>(r.b); sp->id = static_cast<dabsdrDecoderId_t>(r.c);
                sp->serial = h->next_serial++;
                sp->packet = pk != nullptr;
                sp->ascty = pk ? pk->dscty : c.ascty_dscty;
                sp->kbps = it->second.kbps;
                sp->scid = c.scid;
                sp->subch_id = subch;
                sp->apps = c.apps;
                sp->sub = {it->second.start, it->second.option, it->second.level, it->second.kbps};
                if (!it->second.long_form) sp->sub = {it->second.start, 2, it->second.uep_index, 0};   // UEP short form
                if (pk) sp->pkt.address = pk->packet_address;
                // an audio decoder id holds one component; a data component replaces an earlier selection of itself
                std::vector<std::unique_ptr<Selection>> keep, displaced;
                for (auto &o : h->sel) {
                    const bool same = sp->id == DABSDR_ID_DATA ? (o->id == DABSDR_ID_DATA && o->sid == sp->sid && o->scids == sp->scids) : o->id == sp->id;
                    (same ? displaced : keep).push_back(std::move(o));
                }
                h->sel = std::move(keep);
                wire_callbacks(h, sp.get());
                const bool primary = sp->id == DABSDR_ID_AUDIO_PRIMARY;
                h->sel.push_back(std::move(sp));
                if (apply_selections(h)) {
                    st = DABSDR_NSTAT_SUCCESS;
                    if (primary) h->per_sf = h->per_au_ok = h->per_au_bad = h->per_rs_corr = h->per_rs_fail = 0;
                } else {                                   // the decoder that was running on this id carries on
                    h->sel.pop_back();
                    for (auto &o : displaced) h->sel.push_back(std::move(o));
                    apply_selections_or_drop(h);
                    st = DABSDR_NSTAT_SERVICE_NOT_SUPPORTED;
                }
                break;
            }
        }
        notify(h, DABSDR_NID_SERVICE_SELECTION, st, &s, sizeof s);
        break;
    }
    case Req::ServiceStop: {
        dabsdrNtfServiceStop_t s = {r.a, static_cast<uint8_t>(r.b), static_cast<dabsdrDecoderId_t>(r.c)};
        std::vector<std::unique_ptr<Selection>> keep;
        for (auto &o : h->sel) {
            const bool hit = static_cast<dabsdrDecoderId_t>(r.c) == DABSDR_ID_DATA ? (o->id == DABSDR_ID_DATA && o->sid == r.a && o->scids == static_cast<int>(r.b))
                                                                                  : o->id == static_cast<dabsdrDecoderId_t>(r.c);
            if (!hit) keep.push_back(std::move(o));
        }
        h->sel = std::move(keep);
        apply_selections_or_drop(h);
        notify(h, DABSDR_NID_SERVICE_STOP, DABSDR_NSTAT_SUCCESS, &s, sizeof s);
        break;
    }
    case Req::XPadAppStart: {
        // start / stop the delivery of one X-PAD application type (e.g. 12: MOT slide show) of the audio decoder r.c
        dabsdrNtfXpadAppStartStop_t x = {static_cast<uint8_t>(r.a), static_cast<int8_t>(r.b)};
        dabsdrNotificationStatus_t st = DABSDR_NSTAT_SERVICE_NOT_FOUND;
        for (auto &sp : h->sel)
            if (sp->id == static_cast<dabsdrDecoderId_t>(r.c) && !sp->packet) {
                if (r.b) sp->xpad_on.insert(static_cast<int>(r.a)); else sp->xpad_on.erase(static_cast<int>(r.a));
                st = DABSDR_NSTAT_SUCCESS;
            }
        notify(h, DABSDR_NID_XPAD_APP_START_STOP, st, &x, sizeof x);
        break;
    }
    case Req::SetPeriodicNotify:
        h->period_log2 = r.b;
        h->period_frames = 0; h->fib_err_acc = 0;
        notify(h, DABSDR_NID_PERIODIC, DABSDR_NSTAT_SUCCESS, nullptr, 0);      // acknowledgement, no payload
        break;
    case Req::SignalSpectrum:
        h->spectrum_on = r.a != 0;
        drain(h);
        if (h->ctx) dabx_enable_spectrum(h->ctx, (h->spectrum_on ? 1 : 0) | 2);
        break;
    case Req::SetTII:
        h->tii_on = r.a != 0;
        h->tii_mode = r.b;
        drain(h);
        if (h->ctx) dabx_enable_spectrum(h->ctx, (h->spectrum_on ? 1 : 0) | 2);
        break;
    case Req::InjectFibs:                                   // test hook: FIBs as if the FIC had delivered them
        for (size_t o = 0; o + 32 <= r.blob.size(); o += 32) h->db.parse_fib(r.blob.data() + o);
        after_fibs(h);
        break;
    case Req::Exit:
        break;
    }
}

// float IQ from the host -> s16 IQ with a power-of-two gain, on the GPU (dabx_push_resampled_from: rint(x * 2^shift), clamped).
// Raw-file input arrives as exact integers (reference: src/input/rawfileinput.cpp:657,692): as long as those fit int16 the
// gain is 1 and the conversion is lossless.  Anything else (SDR floats in +-1, scaled recordings) gets a gain that puts the
// peak of the FIRST WHOLE FRAME THAT CARRIES A SIGNAL between 2^12 and 2^13 — headroom of 12 dB upwards, 12 bits downwards —
// and a slow hysteresis afterwards: a step whose samples peak above 30000 halves the gain, one that stays below 256 doubles
// it.  The receiver normalises every OFDM symbol on its own, so a gain step costs at most the symbol it falls into.
// Returns false for a frame without a usable level (all zeros: the host's FIFO hands out zeros while it is flushed,
// inputdevice.cpp:80-85; or non-finite values): the gain stays open and the next frame decides.
bool choose_shift(const float *in, size_t n_values, int &shift)
{
    float mx = 0.0f;
    bool integral = true, nan = false;
    for (size_t i = 0; i < n_values; ++i) {
        mx = std::fmax(mx, std::fabs(in[i]));
        nan = nan || in[i] != in[i];
        integral = integral && in[i] == std::nearbyint(in[i]);
    }
    if (nan || !(mx > 0.0f) || !std::isfinite(mx)) return false;
    shift = 0;
    if (integral && mx <= 32767.0f && mx >= 16.0f) return true;        // integer samples that fit: lossless
    while (mx * std::ldexp(1.0f, shift) >= 8192.0f) --shift;
    while (mx * std::ldexp(1.0f, shift) < 4096.0f && shift < 40) ++shift;
    return true;
}

// Dynamic range control of MPEG Layer II audio (EN 300 401 §7.4.1): the F-PAD — the last two bytes of an audio frame — of
// type 00 with byte L indicator 0001 carries six bits of DRC data in byte L, a gain in steps of 0.25 dB for the FOLLOWING
// frame.  The reference hands it to the host as header.mp2DRC (dabsdr.h:47-60), which applies it to the next decoded frame
// (src/audiodecoder.cpp:285-294, 326).  `second_half`: this logical frame completes a 24 kHz (LSF) audio frame.
uint8_t mp2_drc(bool second_half, const uint8_t *frame, int len)
{
    if (len < 4) return 0;
    if (!second_half) {
        if (frame[0] != 0xFF || (frame[1] & 0xF0) != 0xF0) return 0;   // not the start of an audio frame
        if (!((frame[1] >> 3) & 1)) return 0;                          // LSF: the F-PAD comes with the second logical frame
    }
    const uint8_t f0 = frame[len - 2], f1 = frame[len - 1];
    return ((f0 >> 6) == 0 && (f0 & 0x0F) == 1) ? static_cast<uint8_t>(f1 >> 2) : 0;
}

// MPEG Layer II audio frames: 48 kHz frames are one logical frame long, 24 kHz (LSF) frames two; the second half
// of an LSF frame does not start with a sync word
void feed_mp2_pad(Selection *sp, const uint8_t *frame, int len)
{
    const bool sync = len >= 4 && frame[0] == 0xFF && (frame[1] & 0xF0) == 0xF0;
    if (!sp->mp2_half.empty()) {                                      // second logical frame of an LSF audio frame
        sp->mp2_half.insert(sp->mp2_half.end(), frame, frame + len);
        sp->pad.feed_mp2_frame(sp->mp2_half.data(), static_cast<int>(sp->mp2_half.size()));
        sp->mp2_half.clear();
        return;
    }
    if (!sync) return;
    if (!((frame[1] >> 3) & 1)) sp->mp2_half.assign(frame, frame + len);  // LSF: wait for the other half
    else sp->pad.feed_mp2_frame(frame, len);
}

// what the FIG database learnt from the FIBs just parsed: multiplex reconfiguration, another ensemble, changed user
// applications, programme type, announcement switching.  Returns false when the receiver was reset.
bool after_fibs(dabsdr_s *h)
{
    if (h->db.reconfigured) {                             // multiplex reconfiguration took effect (EN 300 401 §6.5)
        h->db.reconfigured = false;
        notify(h, DABSDR_NID_RECONFIGURATION, DABSDR_NSTAT_SUCCESS, nullptr, 0);
        // running selections follow their component into the new configuration; one whose component is gone stops
        std::vector<std::unique_ptr<Selection>> keep;
        for (auto &sp : h->sel) {
            bool found = false;
            if (const figdb::Service *sv = h->db.find_service(sp->sid))
                for (const auto &c : sv->comp) {
                    if (c.scids != sp->scids) continue;
                    int subch = c.subch;
                    if (c.tmid == 3) {
                        auto pi = h->db.packet.find(c.scid);
                        if (pi == h->db.packet.end()) break;
                        subch = pi->second.subch;
                        sp->pkt.address = pi->second.packet_address;
                    }
                    auto it = h->db.subch.find(subch);
                    if (it == h->db.subch.end()) break;
                    sp->kbps = it->second.kbps;
                    sp->subch_id = subch;
                    sp->sub = {it->second.start, it->second.option, it->second.level, it->second.kbps};
                    if (!it->second.long_form) sp->sub = {it->second.start, 2, it->second.uep_index, 0};
                    found = true;
                    break;
                }
            if (found) keep.push_back(std::move(sp));
            else {
                dabsdrNtfServiceStop_t stop = {sp->sid, static_cast<uint8_t>(sp->scids), sp->id};
                notify(h, DABSDR_NID_SERVICE_STOP, DABSDR_NSTAT_SUCCESS, &stop, sizeof stop);
            }
        }
        h->sel = std::move(keep);
        apply_selections_or_drop(h);
    }
    if (h->db.eid_changed) {                              // another ensemble on this frequency (two recordings in one file,
        h->sel.clear();                                   // a transmitter that changed its multiplex): everything known is stale
        apply_selections_or_drop(h);
        reset_receiver(h, DABSDR_RESET_NEW_EID);          // the host restarts on it (radiocontrol.cpp:118-127)
        return false;
    }
    for (const auto &ch : h->db.apps_changed)             // FIG 0/13 changed for a running component -> the host asks for the list again
        for (const auto &sp : h->sel)                     // (radiocontrol.cpp:214-222, 2323-2333)
            if (sp->sid == ch.first && sp->scids == ch.second) {
                if (const figdb::Service *sv = h->db.find_service(ch.first))
                    for (const auto &c : sv->comp)
                        if (c.scids == ch.second) sp->apps = c.apps;
                dabsdrNtfUserAppUpdate_t u = {ch.first, static_cast<uint8_t>(ch.second)};
                notify(h, DABSDR_NID_USER_APP_UPDATE, DABSDR_NSTAT_SUCCESS, &u, sizeof u);
                break;
            }
    h->db.apps_changed.clear();
    for (uint32_t sid : h->db.pty_changed)                // FIG 0/17 -> DABSDR_NID_PTY (dabsdr.h:353-357)
        if (const figdb::Service *sv = h->db.find_service(sid)) {
            dabsdrNtfPTy_t p = {sid, static_cast<uint8_t>(sv->pty), static_cast<uint8_t>(sv->pty)};
            notify(h, DABSDR_NID_PTY, DABSDR_NSTAT_SUCCESS, &p, sizeof p);
        }
    h->db.pty_changed.clear();
    if (h->db.switching_changed) {                        // FIG 0/19 -> DABSDR_NID_ANNOUNCEMENT_SWITCHING (dabsdr.h:341-351)
        h->db.switching_changed = false;
        dabsdrNtfAnnouncementSwitching_t a;
        std::memset(&a, 0, sizeof a);
        int k = 0;
        for (const auto &kv : h->db.switching) {
            if (k >= 8) break;
            a.asw[k++] = {static_cast<uint8_t>(kv.second.cluster), static_cast<uint8_t>(kv.second.subch), kv.second.flags};
        }
        notify(h, DABSDR_NID_ANNOUNCEMENT_SWITCHING, DABSDR_NSTAT_SUCCESS, &a, sizeof a);
    }
    return true;
}

// Fetch the results of the n_frames-frame step that has just been waited for (the context is idle) into h->done.
void collect(dabsdr_s *h, int n_frames, int shift, std::vector<SubSnap> subs)
{
    dabx_ctx *c = h->ctx;
    if (!c || n_frames < 1) return;
    StepResults r;
    r.n_frames = n_frames;
    r.shift = shift;
    r.fib.resize(static_cast<size_t>(n_frames) * 384);
    r.ok.resize(static_cast<size_t>(n_frames) * 12);
    r.rec.resize(static_cast<size_t>(n_frames));
    if (dabx_get_fib(c, 0, r.fib.data(), r.ok.data()) || dabx_get_state(c, 0, &r.st) || dabx_get_sync(c, 0, r.rec.data())) return;
    (void)dabx_get_input_peak(c, 0, &r.peak);
    r.null_power.resize(static_cast<size_t>(n_frames) * 2048);
    r.have_null = dabx_get_null_spectra(c, 0, r.null_power.data()) == DABX_OK;
    if (h->spectrum_on && h->spec_cb) {
        r.spectrum.resize(2048);
        r.have_spec = dabx_get_spectrum(c, 0, r.spectrum.data()) == DABX_OK;
    }
    r.subs = std::move(subs);
    if (!r.subs.empty()) {
        for (const auto &sb : r.subs) r.stride += static_cast<size_t>(3 * sb.kbps);
        // dabx_get_msc copies 4 x the stride the CONTEXT has per frame; it must be the one the records are sliced by
        if (r.stride == static_cast<size_t>(h->msc_stride)) {
            r.msc.resize(static_cast<size_t>(n_frames) * 4 * r.stride);
            r.valid.assign(static_cast<size_t>(n_frames) * 4, 0);
            r.have_msc = dabx_get_msc(c, 0, r.msc.data(), r.valid.data()) == DABX_OK;
        }
        const int max_rec = (4 + 4 * kMaxBatch) / 5;
        for (size_t k = 0; k < r.subs.size(); ++k) {
            SubSnap &sb = r.subs[k];
            if (!sb.dabplus) continue;
            const size_t s8 = static_cast<size_t>(sb.kbps / 8);
            sb.recs.resize(max_rec);
            sb.data.resize(static_cast<size_t>(max_rec) * 110 * s8);
            const int n = dabx_get_superframes(c, 0, static_cast<int>(k), sb.recs.data(), sb.data.data(), max_rec);
            sb.recs.resize(n > 0 ? static_cast<size_t>(n) : 0);
            uint32_t pos[2] = {0, 0};
            if (n > 0 && dabx_get_superframe_pos(c, 0, static_cast<int>(k), pos) == DABX_OK) sb.frames_seen = pos[0];
            else sb.recs.clear();
        }
    }
    h->done.push_back(std::move(r));
}

// wait for the step in flight, if any, and fetch its results
void drain(dabsdr_s *h)
{
    if (!h->inflight) return;
    h->inflight = false;
    if (!h->ctx) return;
    const bool ok = dabx_wait(h->ctx) == DABX_OK;
    h->buf_busy[0] = h->buf_busy[1] = false;               // the step waited for every copy queued before it
    if (ok) collect(h, h->inflight_frames, h->inflight_shift, std::move(h->inflight_subs));
    h->inflight_subs.clear();
}

// the gain (power of two) the sample with absolute index `pos` was written to the ring with
int gain_at(const dabsdr_s *h, int64_t pos)
{
    int sh = h->gain_epochs.empty() ? h->gain_shift : h->gain_epochs.front().second;
    for (const auto &e : h->gain_epochs)
        if (e.first <= pos) sh = e.second;
    return sh;
}

Selection *find_selection(dabsdr_s *h, uint64_t serial)
{
    for (auto &sp : h->sel)
        if (sp->serial == serial) return sp.get();
    return nullptr;
}

// frame f of a fetched step -> notifications and callbacks, in the order the reference's single-frame loop produces them
void deliver_frame(dabsdr_s *h, StepResults &r, int f, bool locked)
{
    const uint8_t *fib = r.fib.data() + static_cast<size_t>(f) * 384, *ok = r.ok.data() + static_cast<size_t>(f) * 12;
    const dabx_sync_rec_t &rec = r.rec[static_cast<size_t>(f)];
    const bool last = f == r.n_frames - 1;
    int good = 0;
    for (int i = 0; i < 12; ++i)
        if (ok[i]) { ++good; h->db.parse_fib(fib + 32 * i); }
    const dabsdrSyncLevel_t lvl = !locked ? DABSDR_SYNC_LEVEL_NO_SYNC : (good ? DABSDR_SYNC_LEVEL_FIC : DABSDR_SYNC_LEVEL_ON_NULL);
    // SNR: (signal+noise energy of the PRS window - noise energy of the null symbol) / noise energy.
    // The null symbol may carry TII carriers (32 of 1536), so its noise level is taken from the median
    // in-band bin of its spectrum (median of an exponential distribution = mean * ln 2) rather than
    // from its total energy; for a TII-free null the two agree.
    double noise = static_cast<double>(rec.e_null);
    float *null_power = r.null_power.data() + static_cast<size_t>(f) * 2048;
    if (r.have_null && locked) {
        float band[1536];
        std::memcpy(band, null_power + 1, 768 * sizeof(float));
        std::memcpy(band + 768, null_power + 2048 - 768, 768 * sizeof(float));
        std::nth_element(band, band + 768, band + 1536);
        noise = static_cast<double>(band[768]) / 0.6931471805599453;
    }
    int16_t snr10 = 0;
    const double sig = static_cast<double>(rec.e_sig);
    if (noise > 0 && sig > noise) snr10 = static_cast<int16_t>(std::min(600L, std::lround(100.0 * std::log10((sig - noise) / noise))));
    else if (sig > 0 && noise <= 0) snr10 = 600;
    if (!after_fibs(h)) return;
    if (lvl != h->sync_level) {
        h->sync_level = lvl;
        dabsdrNtfSyncStatus_t s = {lvl, snr10};
        notify(h, DABSDR_NID_SYNC_STATUS, DABSDR_NSTAT_SUCCESS, &s, sizeof s);
    }
    h->fib_err_acc += static_cast<uint32_t>(12 - good);
    // spectra are powers of the HOST's samples (the reference: un-normalised |FFT|^2 of what the input callback delivered,
    // signalbackend.cpp:203 subtracts the FFT gain only): the adapter's power-of-two gain is taken out again
    // (the gain in force where the window was read: gain_at)
    if (last && r.have_spec && h->spectrum_on && h->spec_cb) {
        const int sh = gain_at(h, rec.t_sym0);
        const float inv_g2 = std::ldexp(1.0f, -2 * sh);
        if (sh) for (float &v : r.spectrum) v *= inv_g2;
        h->spec_cb(r.spectrum.data(), DABSDR_SPECT_SIGNAL, h->spec_ctx);
    }
    {
        if ((h->tii_on || h->spectrum_on) && locked && r.have_null) {
            const int sh = gain_at(h, rec.t_sym0 - 2400);
            const float inv_g2 = std::ldexp(1.0f, -2 * sh);
            if (sh) for (int i = 0; i < 2048; ++i) null_power[i] *= inv_g2;
            if (h->spectrum_on && h->spec_cb) h->spec_cb(null_power, DABSDR_SPECT_NULL, h->spec_ctx);
            if (h->tii_on) {
                const auto ids = tii::detect(null_power, h->tii_mode == DABSDR_TII_MODE_CONSERVATIVE ? 8.0f : 4.0f);
                tii::fold(null_power, h->tii_folded);
                dabsdrNtfTii_t n;
                std::memset(&n, 0, sizeof n);
                n.numIds = static_cast<uint8_t>(ids.size());
                for (size_t i = 0; i < ids.size(); ++i) n.id[i] = {ids[i].main, ids[i].sub, ids[i].level};
                // the header declares float[192] but the host passes 384 floats (radiocontrol.h:280)
                n.getSpectrumTii = [](dabsdrHandle_t hh, float *b) { std::memcpy(b, hh->tii_folded, sizeof hh->tii_folded); return 0; };
                notify(h, DABSDR_NID_TII, DABSDR_NSTAT_SUCCESS, &n, sizeof n);
            }
        }
    }
    // running selections.  Audio (dabsdrAudioCBFunc_t, dabsdr.h:47-78): DAB+ access units come from k_superframe's records
    // (a damaged unit keeps its place with the conceal bit set, as audiodecoder.cpp:183-208 expects), MPEG Layer II
    // sub-channels are handed over one logical frame at a time; their PAD feeds the dynamic label / data group
    // callbacks.  Packet-mode data components: packets -> MSC data groups -> dabsdrDataGroupCBFunc_t.
    // A super frame belongs to the frame whose CIF completed it: the step's last CIF has index frames_seen - 1.
    for (SubSnap &sb : r.subs) {
        Selection *sp = find_selection(h, sb.serial);
        if (!sp) continue;                                   // stopped or replaced since the step was submitted
        const size_t fb = static_cast<size_t>(3 * sb.kbps);
        if (sb.dabplus) {
            const size_t s8 = static_cast<size_t>(sb.kbps / 8);
            for (size_t q = 0; q < sb.recs.size(); ++q) {
                const dabx_superframe_t &sr = sb.recs[q];
                const int64_t cif = static_cast<int64_t>(sr.first_frame) + 4 - (static_cast<int64_t>(sb.frames_seen) - 4 * r.n_frames);
                const int64_t fr = cif < 0 ? 0 : cif / 4;
                if (fr != f) continue;
                const uint8_t *base = sb.data.data() + q * 110 * s8;
                const bool primary = sp->id == DABSDR_ID_AUDIO_PRIMARY;
                if (primary) { ++h->per_sf; h->per_rs_corr += sr.rs_corrected; h->per_rs_fail += sr.rs_failed; h->per_kbps = sb.kbps; }
                for (int a = 0; a < sr.num_aus; ++a) {
                    const bool au_ok = (sr.au_ok >> a) & 1;
                    if (primary) { if (au_ok) ++h->per_au_ok; else ++h->per_au_bad; }
                    if (!((sr.au_valid >> a) & 1)) continue;
                    dabsdrAudioCBData_t d;
                    d.id = sp->id; d.ASCTy = 63;
                    d.header.raw = static_cast<uint8_t>(sr.header | (au_ok ? 0 : 0x80));
                    d.auLen = static_cast<uint16_t>(sr.au_start[a + 1] - sr.au_start[a] - 2);
                    d.pAuData = base + sr.au_start[a];
                    if (primary) h->audio_bytes_acc += d.auLen;
                    if (h->audio_cb) h->audio_cb(&d, h->audio_ctx);
                    if (au_ok) sp->pad.feed_dabplus_au(d.pAuData, d.auLen);
                }
            }
        } else if (r.have_msc) {
            for (int c = 0; c < 4; ++c) {
                if (!r.valid[static_cast<size_t>(4 * f + c)]) continue;
                const uint8_t *frame = r.msc.data() + static_cast<size_t>(4 * f + c) * r.stride + sb.off;
                if (sp->packet) {
                    auto fe = h->db.fec_scheme.find(sp->subch_id);          // FIG 0/14 may arrive after the selection
                    sp->pkt.set_fec(fe != h->db.fec_scheme.end() && fe->second == 1);
                    sp->pkt.feed_frame(frame, static_cast<int>(fb));
                    continue;
                }
                dabsdrAudioCBData_t d;
                d.id = sp->id; d.ASCTy = static_cast<uint8_t>(sp->ascty); d.header.raw = 0;
                if (sp->ascty == 0) d.header.mp2DRC = mp2_drc(!sp->mp2_half.empty(), frame, static_cast<int>(fb));
                d.auLen = static_cast<uint16_t>(fb); d.pAuData = frame;
                if (sp->id == DABSDR_ID_AUDIO_PRIMARY) h->audio_bytes_acc += d.auLen;
                if (h->audio_cb) h->audio_cb(&d, h->audio_ctx);
                if (sp->ascty == 0) feed_mp2_pad(sp, frame, static_cast<int>(fb));   // MPEG Layer II: PAD at the end of the audio frame
            }
        }
    }
    if (h->period_log2 >= 0 && ++h->period_frames >= (1 << h->period_log2)) {
        dabsdrNtfPeriodic_t p;
        std::memset(&p, 0, sizeof p);
        p.syncLevel = lvl;
        p.snr10 = snr10;
        // inc is 2^-32 turn per sample at 2.048 MHz; the host wants Hz * 10
        p.freqOffset = static_cast<int32_t>(std::llround(static_cast<double>(rec.inc) * 2048000.0 * 10.0 / 4294967296.0));
        if (h->db.ens.utc_valid) {
            p.dateHoursMinutes = h->db.ens.date_hours_minutes & 0x7FFFFFFFu;
            p.secMsec = static_cast<uint16_t>((h->db.ens.seconds << 10) | h->db.ens.ms);
        }
        p.fibErrorCntr = static_cast<uint16_t>(h->fib_err_acc);
        p.mscCrcOkCntr = static_cast<uint8_t>(h->per_au_ok);
        p.mscCrcErrorCntr = static_cast<uint8_t>(h->per_au_bad);
        p.rsUncorrectableCntr = static_cast<uint16_t>(h->per_rs_fail);
        p.rsBitErrors = static_cast<uint16_t>(h->per_rs_corr);
        p.rsBytes = static_cast<uint16_t>(h->per_sf * 120u * static_cast<unsigned>(h->per_kbps / 8));
        p.audioServiceBytes = static_cast<uint16_t>(h->audio_bytes_acc);
        h->per_sf = h->per_au_ok = h->per_au_bad = h->per_rs_corr = h->per_rs_fail = 0;
        h->audio_bytes_acc = 0;
        notify(h, DABSDR_NID_PERIODIC, DABSDR_NSTAT_SUCCESS, &p, sizeof p);
        h->period_frames = 0; h->fib_err_acc = 0;
    }
}

// turn every fetched step into callbacks.  A callback chain may fetch further steps (a reconfiguration drains the
// context: apply_selections), which land at the end of h->done and are served by the same loop.
void deliver(dabsdr_s *h)
{
    while (!h->done.empty()) {
        StepResults r = std::move(h->done.front());
        h->done.pop_front();
        // The sync level the host sees is the frame's own: a frame whose phase reference symbol was not found reports
        // NO_SYNC (the reference: "brief SYNC 0" at the wrap of a looped file) although the receiver's
        // flywheel keeps its timing for up to four such frames (k_finish) before it searches from scratch.
        for (int f = 0; f < r.n_frames; ++f) deliver_frame(h, r, f, (r.rec[static_cast<size_t>(f)].flags & 1) != 0);
        // whole frames are batched only while the receiver tracks without doubt (a missing PRS makes the next step look for the
        // null symbol again, which needs a frame more in the ring: chunk by chunk gets there soonest)
        h->locked_hint = r.st.locked != 0 && r.st.bad == 0;
        // gain hysteresis on the level the GPU saw (int16 units after the gain)
        if (r.peak > 30000) --h->gain_shift;
        else if (r.peak > 0 && r.peak < 256 && h->gain_shift < 40) ++h->gain_shift;
    }
}

std::vector<SubSnap> snapshot_selections(const dabsdr_s *h)
{
    std::vector<SubSnap> v;
    size_t off = 0;
    for (const auto &sp : h->sel) {
        SubSnap sb;
        sb.serial = sp->serial; sb.kbps = sp->kbps; sb.off = off;
        sb.dabplus = !sp->packet && sp->ascty == 63;
        off += static_cast<size_t>(3 * sp->kbps);
        v.push_back(std::move(sb));
    }
    return v;
}

// hand n complex float samples at `src` (page-locked) to the GPU and decode what is complete.
//   async: the copy is queued and the step left in flight (the caller fills the other buffer meanwhile);
//   otherwise everything is waited for and delivered before returning.
void submit(dabsdr_s *h, const float *src, int64_t n, bool async)
{
    if (!h->ctx) return;
    drain(h);                                                            // at most one step in flight
    const float gain = std::ldexp(1.0f, h->gain_shift);
    if (h->gain_epochs.empty() || h->gain_epochs.back().second != h->gain_shift) {
        h->gain_epochs.emplace_back(h->pushed, h->gain_shift);
        if (h->gain_epochs.size() > 16) h->gain_epochs.pop_front();
    }
    const int64_t rc = dabx_push_resampled_from(h->ctx, 0, src, n, DABX_FMT_F32, 2048000.0, gain, async ? DABX_SRC_PINNED : DABX_SRC_HOST);
    if (rc < 0) { if (!async) deliver(h); return; }                      // (overrun: cannot happen with a ring of kRingFrames)
    h->pushed += rc;
    for (;;) {
        const int avail = h->ctx ? dabx_frames_available(h->ctx) : 0;
        if (avail < 1 || h->exit_req.load()) break;
        const int nf = std::min(avail, kMaxBatch);
        if (dabx_process_async(h->ctx, nf) != DABX_OK) break;
        h->inflight = true;
        h->inflight_frames = nf;
        h->inflight_shift = h->gain_shift;
        h->inflight_subs = snapshot_selections(h);
        if (async) break;                                                // one step; what is left joins the next buffer
        drain(h);
        deliver(h);
    }
    if (!async) deliver(h);
}

void worker_loop(dabsdr_s *h);
void worker_main(dabsdr_s *h)
{
    pthread_setname_np(pthread_self(), "dabsdr");
    // dabsdrDeinit may cancel this thread when it sits in the host's input callback for good; nowhere else (never inside a
    // HIP call or a notification): cancellation is enabled around that call only
    int old;
    pthread_setcancelstate(PTHREAD_CANCEL_DISABLE, &old);
    struct Done { dabsdr_s *h; ~Done() { h->worker_done.store(true); } } done{h};     // also when the thread is cancelled
    worker_loop(h);
}

void pull(dabsdr_s *h, float *dst, int n)
{
    int old;
    const auto t0 = std::chrono::steady_clock::now();
    pthread_setcancelstate(PTHREAD_CANCEL_ENABLE, &old);
    h->input(dst, static_cast<uint16_t>(n));
    pthread_setcancelstate(PTHREAD_CANCEL_DISABLE, &old);
    // how fast the host delivers: a frame's worth of this call's time, smoothed (1/8 per chunk)
    const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() * (static_cast<double>(DABX_FRAME_SAMPLES) / n);
    h->pull_s += (dt - h->pull_s) * 0.125;
}

void worker_loop(dabsdr_s *h)
{
    // requests are served between input calls (every 8 ms of signal), like the reference, whose
    // reads never exceed one OFDM symbol: a host that paces or gates its input still gets answers
    auto serve = [h](bool may_block) -> bool {
        std::deque<Request> todo;
        {
            std::unique_lock<std::mutex> lk(h->mu);
            if (may_block && h->queue.empty()) h->cv.wait(lk, [&] { return !h->queue.empty() || h->exit_req.load(); });
            todo.swap(h->queue);
        }
        if (!todo.empty()) { drain(h); deliver(h); }                     // answers come after what was decoded before the question
        for (const Request &r : todo) {
            if (r.kind == Req::Exit) return false;
            handle_request(h, r);
        }
        return true;
    };
    const size_t buf_floats = 2 * static_cast<size_t>(kMaxBatch) * DABX_FRAME_SAMPLES;
    while (!h->exit_req.load()) {
        if (!serve(h->frequency == 0)) return;
        if (h->frequency == 0 || !h->ctx || !h->input) continue;
        for (float *&b : h->fbuf)
            if (!b && !(b = static_cast<float *>(dabx_alloc_pinned(buf_floats * sizeof(float))))) return;
        float *buf = h->fbuf[h->cur];
        if (h->fill == 0 && h->buf_busy[h->cur]) { (void)dabx_flush_copies(h->ctx); h->buf_busy[0] = h->buf_busy[1] = false; }
        pull(h, buf + 2 * static_cast<size_t>(h->fill), kPullChunk);
        h->fill += kPullChunk;
        if (!h->gain_set) {                                              // the first frame with a signal after a tune fixes the gain
            if (h->fill < DABX_FRAME_SAMPLES) continue;
            h->gain_set = choose_shift(buf, 2 * static_cast<size_t>(h->fill), h->gain_shift);
            if (h->gain_set) submit(h, buf, h->fill, false);
            h->fill = 0;                                                 // (a silent frame is dropped: nothing to decode in it)
            continue;
        }
        if (!h->locked_hint) {                                           // acquiring: every chunk at once, steps in between
            submit(h, buf, h->fill, false);
            h->fill = 0;
            continue;
        }
        // locked: whole frames, as many per step as arrive within 48 ms (one when the host paces its input at real time)
        h->batch = std::max(1, std::min(kMaxBatch, static_cast<int>(0.048 / std::max(h->pull_s, 1e-6))));
        h->pipelined = h->pull_s < 0.0096;
        if (h->fill < h->batch * DABX_FRAME_SAMPLES && h->fill + kPullChunk <= kMaxBatch * DABX_FRAME_SAMPLES) continue;
        if (h->pipelined) {
            // the step of the buffer filled before this one has had a whole fill to finish: fetch it, start this one,
            // then turn the fetched one into callbacks while the GPU works
            submit(h, buf, h->fill, true);
            h->buf_busy[h->cur] = true;
            h->cur ^= 1;
            deliver(h);
        } else {
            submit(h, buf, h->fill, false);
        }
        h->fill = 0;
    }
}

void post(dabsdrHandle_t h, Request r)
{
    if (!h) return;
    {
        std::lock_guard<std::mutex> lk(h->mu);
        h->queue.push_back(r);
    }
    h->cv.notify_all();
}

}  // namespace

extern "C" {

uint8_t dabsdrInit(dabsdrHandle_t *handle)
{
    if (!handle) return EXIT_FAILURE;
    dabsdr_s *h = new (std::nothrow) dabsdr_s;
    if (!h) return EXIT_FAILURE;
    dabx_config_t cfg = {1, DABX_FMT_S16, static_cast<int64_t>(kRingFrames) * DABX_FRAME_SAMPLES, kMaxBatch, 0};
    if (dabx_create(&cfg, &h->ctx) != DABX_OK) {     // no GPU: fail loudly, there is no CPU path
        delete h;
        *handle = nullptr;
        return EXIT_FAILURE;
    }
    dabx_enable_spectrum(h->ctx, 2);                 // null-symbol spectrum: noise estimate and TII
    *handle = h;
    return EXIT_SUCCESS;
}

void dabsdrGetVersion(dabsdrVersion_t *v)
{
    if (!v) return;
    v->major = 4; v->minor = 0; v->patch = 1;       // API level of the reference library this replaces
    v->flags = 0x80;                                // bit 7: GPU implementation
}

void dabsdr(dabsdrHandle_t h)
{
    if (!h || h->started) return;
    h->started = true;
    h->worker = std::thread(worker_main, h);
}

void dabsdrDeinit(dabsdrHandle_t *handle)
{
    if (!handle || !*handle) return;
    dabsdr_s *h = *handle;
    h->exit_req.store(true);
    h->cv.notify_all();
    if (h->worker.joinable()) {
        // The worker may sit inside the host's input callback, which blocks until samples arrive (reference:
        // inputdevice.cpp:70-85 waits on a condition variable).  The reference library's thread is cancelled by Deinit
        // (radiocontrol.cpp:76 relies on it), so: a grace period, then pthread_cancel — the wait is a cancellation
        // point.  All GPU teardown happens here, on the calling thread, after the join.
        for (int i = 0; i < 100 && !h->worker_done.load(); ++i) std::this_thread::sleep_for(std::chrono::milliseconds(10));
        if (!h->worker_done.load()) pthread_cancel(h->worker.native_handle());
        h->worker.join();
    }
    if (h->ctx) dabx_destroy(h->ctx);
    for (float *b : h->fbuf)
        if (b) dabx_free_pinned(b);
    delete h;
    *handle = nullptr;
}

void dabsdrRegisterInputFcn(dabsdrHandle_t h, dabsdrInputFunc_t f) { if (h) h->input = f; }
void dabsdrRegisterDummyInputFcn(dabsdrHandle_t h, dabsdrInputFunc_t f) { if (h) h->dummy = f; }   // never needed: whole frames are consumed
void dabsdrRegisterAudioCb(dabsdrHandle_t h, dabsdrAudioCBFunc_t f, void *c) { if (h) { h->audio_cb = f; h->audio_ctx = c; } }
void dabsdrRegisterDynamicLabelCb(dabsdrHandle_t h, dabsdrDynamicLabelCBFunc_t f, void *c) { if (h) { h->dl_cb = f; h->dl_ctx = c; } }
void dabsdrRegisterDataGroupCb(dabsdrHandle_t h, dabsdrDataGroupCBFunc_t f, void *c) { if (h) { h->dg_cb = f; h->dg_ctx = c; } }
void dabsdrRegisterSignalSpectrumCb(dabsdrHandle_t h, dabsdrSpectrumCBFunc_t f, void *c) { if (h) { h->spec_cb = f; h->spec_ctx = c; } }
void dabsdrRegisterNotificationCb(dabsdrHandle_t h, dabsdrNotificationCBFunc_t f, void *c) { if (h) { h->ntf_cb = f; h->ntf_ctx = c; } }

void dabsdrRequest_Tune(dabsdrHandle_t h, uint32_t f) { post(h, {Req::Tune, f, 0, 0}); }
void dabsdrRequest_GetEnsemble(dabsdrHandle_t h) { post(h, {Req::GetEnsemble, 0, 0, 0}); }
void dabsdrRequest_GetServiceList(dabsdrHandle_t h) { post(h, {Req::GetServiceList, 0, 0, 0}); }
void dabsdrRequest_GetServiceComponents(dabsdrHandle_t h, uint32_t sid) { post(h, {Req::GetServiceComponents, sid, 0, 0}); }
void dabsdrRequest_GetUserAppList(dabsdrHandle_t h, uint32_t sid, uint8_t scids) { post(h, {Req::GetUserAppList, sid, scids, 0}); }
void dabsdrRequest_GetAnnouncementSupport(dabsdrHandle_t h, uint32_t sid) { post(h, {Req::GetAnnouncementSupport, sid, 0, 0}); }
void dabsdrRequest_ServiceSelection(dabsdrHandle_t h, uint32_t sid, uint8_t scids, dabsdrDecoderId_t id) { post(h, {Req::ServiceSelection, sid, scids, id}); }
void dabsdrRequest_ServiceStop(dabsdrHandle_t h, uint32_t sid, uint8_t scids, dabsdrDecoderId_t id) { post(h, {Req::ServiceStop, sid, scids, id}); }
void dabsdrRequest_XPadAppStart(dabsdrHandle_t h, uint8_t app, int8_t start, dabsdrDecoderId_t id) { post(h, {Req::XPadAppStart, app, start, id}); }
void dabsdrRequest_SetPeriodicNotify(dabsdrHandle_t h, uint8_t period, uint32_t cfg) { post(h, {Req::SetPeriodicNotify, 0, period, static_cast<int32_t>(cfg)}); }
void dabsdrRequest_SetTII(dabsdrHandle_t h, uint8_t ena, dabsdrTiiMode_t mode) { post(h, {Req::SetTII, ena, mode, 0}); }
void dabsdrRequest_SignalSpectrum(dabsdrHandle_t h, uint8_t ena) { post(h, {Req::SignalSpectrum, ena, 0, 0}); }
void dabsdrRequest_Exit(dabsdrHandle_t h)
{
    if (!h) return;
    h->exit_req.store(true);
    post(h, {Req::Exit, 0, 0, 0});
}

// test hook (CPU only): TII detection on a 2048-bin null-symbol power spectrum; ids = {main, sub} pairs
DABSDR_API int dabsdr_amd_tii_detect(const float *power, float factor, uint8_t *ids, int max_ids)
{
    const auto v = tii::detect(power, factor);
    int n = 0;
    for (const auto &id : v) {
        if (n >= max_ids) break;
        ids[2 * n] = id.main; ids[2 * n + 1] = id.sub; ++n;
    }
    return n;
}

// test hook (CPU only): feed DAB+ access units (len lo, len hi, bytes; concatenated) to the PAD decoder; out receives the
// dynamic-label segments and data groups as records {kind 'L' | 'G', app type, len lo, len hi, bytes}
DABSDR_API int dabsdr_amd_pad_decode(const uint8_t *aus, int n_bytes, uint8_t *out, int cap, uint32_t *stats)
{
    pad::Decoder dec;
    int used = 0;
    bool overflow = false;
    auto emit = [&](char kind, int app, const uint8_t *d, int n) {
        if (used + 4 + n > cap) { overflow = true; return; }
        out[used] = static_cast<uint8_t>(kind); out[used + 1] = static_cast<uint8_t>(app);
        out[used + 2] = static_cast<uint8_t>(n & 0xFF); out[used + 3] = static_cast<uint8_t>(n >> 8);
        std::memcpy(out + used + 4, d, static_cast<size_t>(n));
        used += 4 + n;
    };
    dec.on_dynamic_label = [&](const uint8_t *d, int n) { emit('L', 2, d, n); };
    dec.on_data_group = [&](int app, const uint8_t *d, int n) { emit('G', app, d, n); };
    for (int pos = 0; pos + 2 <= n_bytes;) {
        const int len = aus[pos] | (aus[pos + 1] << 8);
        if (pos + 2 + len > n_bytes) break;
        dec.feed_dabplus_au(aus + pos + 2, len);
        pos += 2 + len;
    }
    if (stats) { stats[0] = dec.stats.pads; stats[1] = dec.stats.dl_ok; stats[2] = dec.stats.dl_crc_err; stats[3] = dec.stats.dg_ok; stats[4] = dec.stats.dg_crc_err; }
    return overflow ? -1 : used;
}

// test hook (CPU only): the same for MPEG Layer II audio frames (records: len lo, len hi, frame bytes)
DABSDR_API int dabsdr_amd_pad_decode_mp2(const uint8_t *frames, int n_bytes, uint8_t *out, int cap, uint32_t *stats)
{
    pad::Decoder dec;
    int used = 0;
    bool overflow = false;
    dec.on_dynamic_label = [&](const uint8_t *d, int n) {
        if (used + 4 + n > cap) { overflow = true; return; }
        out[used] = 'L'; out[used + 1] = 2; out[used + 2] = static_cast<uint8_t>(n & 0xFF); out[used + 3] = static_cast<uint8_t>(n >> 8);
        std::memcpy(out + used + 4, d, static_cast<size_t>(n));
        used += 4 + n;
    };
    for (int pos = 0; pos + 2 <= n_bytes;) {
        const int len = frames[pos] | (frames[pos + 1] << 8);
        if (pos + 2 + len > n_bytes) break;
        dec.feed_mp2_frame(frames + pos + 2, len);
        pos += 2 + len;
    }
    if (stats) { stats[0] = dec.stats.pads; stats[1] = dec.stats.dl_ok; stats[2] = dec.stats.dl_crc_err; stats[3] = dec.stats.dg_ok; stats[4] = dec.stats.dg_crc_err; }
    return overflow ? -1 : used;
}

// test hook (CPU only): header.mp2DRC for each of n logical frames of frame_bytes (MPEG Layer II sub-channel), as after_step() sets it
DABSDR_API int dabsdr_amd_mp2_drc(const uint8_t *frames, int n_frames, int frame_bytes, uint8_t *out)
{
    Selection sel;
    for (int i = 0; i < n_frames; ++i) {
        const uint8_t *f = frames + static_cast<size_t>(i) * frame_bytes;
        out[i] = mp2_drc(!sel.mp2_half.empty(), f, frame_bytes);
        feed_mp2_pad(&sel, f, frame_bytes);
    }
    return n_frames;
}

// test hook (CPU only): logical frames of a packet-mode sub-channel (n_frames x frame_bytes) -> data groups as records
// {addr lo, addr hi, len lo, len hi, bytes}; address < 0 follows every address; stats[4] = packets, CRC errors, groups, dropped
// dabsdr_amd_packet_decode_fec: the same for a sub-channel with FEC frames (FIG 0/14 scheme 1); stats[7] adds FEC frames, corrected bytes, failed rows
static int packet_decode(const uint8_t *frames, int n_frames, int frame_bytes, int address, bool fec, uint8_t *out, int cap, uint32_t *stats, int n_stats)
{
    packet::Decoder dec;
    dec.address = address;
    dec.set_fec(fec);
    int used = 0;
    bool overflow = false;
    dec.on_data_group = [&](int addr, const uint8_t *d, int n) {
        if (used + 4 + n > cap) { overflow = true; return; }
        out[used] = static_cast<uint8_t>(addr & 0xFF); out[used + 1] = static_cast<uint8_t>(addr >> 8);
        out[used + 2] = static_cast<uint8_t>(n & 0xFF); out[used + 3] = static_cast<uint8_t>(n >> 8);
        std::memcpy(out + used + 4, d, static_cast<size_t>(n));
        used += 4 + n;
    };
    for (int i = 0; i < n_frames; ++i) dec.feed_frame(frames + static_cast<size_t>(i) * frame_bytes, frame_bytes);
    dec.set_fec(false);                                   // hands on what an unfinished table holds
    const uint32_t all[7] = {dec.stats.packets, dec.stats.crc_err, dec.stats.groups, dec.stats.dropped, dec.stats.fec_frames, dec.stats.fec_corrected, dec.stats.fec_failed_rows};
    if (stats) std::memcpy(stats, all, sizeof(uint32_t) * static_cast<size_t>(n_stats));
    return overflow ? -1 : used;
}

DABSDR_API int dabsdr_amd_packet_decode(const uint8_t *frames, int n_frames, int frame_bytes, int address, uint8_t *out, int cap, uint32_t *stats)
{
    return packet_decode(frames, n_frames, frame_bytes, address, false, out, cap, stats, 4);
}

DABSDR_API int dabsdr_amd_packet_decode_fec(const uint8_t *frames, int n_frames, int frame_bytes, int address, uint8_t *out, int cap, uint32_t *stats)
{
    return packet_decode(frames, n_frames, frame_bytes, address, true, out, cap, stats, 7);
}

// test hook (CPU only): parse FIBs and print the ensemble as text, see tests/test_figdb.py
DABSDR_API int dabsdr_amd_fig_dump(const uint8_t *fibs, int n_fibs, char *out, int cap)
{
    figdb::Database db;
    for (int i = 0; i < n_fibs; ++i) db.parse_fib(fibs + 32 * i);
    std::string s;
    char line[160];
    std::snprintf(line, sizeof line, "ensemble eid=%04X ecc=%02X lto=%d label='%s' cif=%d utc=%u %02d:%02d:%02d.%03d\n", db.ens.eid & 0xFFFF,
                  db.ens.ecc, db.ens.lto, db.ens.label.c_str(), db.ens.cif_count, db.ens.mjd, db.ens.hours, db.ens.minutes, db.ens.seconds, db.ens.ms);
    s += line;
    for (const auto &kv : db.subch) {
        std::snprintf(line, sizeof line, "subch id=%d start=%d size=%d opt=%d level=%d kbps=%d\n", kv.second.id, kv.second.start,
                      kv.second.size, kv.second.option, kv.second.level, kv.second.kbps);
        s += line;
    }
    for (const auto &kv : db.services) {
        std::snprintf(line, sizeof line, "service sid=%04X label='%s' ncomp=%zu", kv.second.sid, kv.second.label.c_str(), kv.second.comp.size());
        s += line;
        for (const auto &c : kv.second.comp) {
            std::snprintf(line, sizeof line, " [tmid=%d ty=%d subch=%d ps=%d]", c.tmid, c.ascty_dscty, c.subch, c.primary ? 1 : 0);
            s += line;
        }
        s += "\n";
        if (kv.second.pty >= 0 || kv.second.asu || !kv.second.clusters.empty()) {
            std::snprintf(line, sizeof line, "  pty=%d dyn=%d asu=%04X clusters=", kv.second.pty, kv.second.pty_dynamic ? 1 : 0, kv.second.asu);
            s += line;
            for (uint8_t cl : kv.second.clusters) { std::snprintf(line, sizeof line, "%d,", cl); s += line; }
            s += "\n";
        }
        for (const auto &c : kv.second.comp) {
            if (!c.scids_known && c.apps.empty() && c.tmid != 3) continue;
            std::snprintf(line, sizeof line, "  comp scids=%d scid=%d apps=", c.scids, c.scid);
            s += line;
            for (const auto &a : c.apps) {
                std::snprintf(line, sizeof line, "%03X:", a.type);
                s += line;
                for (uint8_t b : a.data) { std::snprintf(line, sizeof line, "%02X", b); s += line; }
                s += ",";
            }
            s += "\n";
        }
    }
    for (const auto &kv : db.language) { std::snprintf(line, sizeof line, "language subch=%d code=%d\n", kv.first, kv.second); s += line; }
    for (const auto &kv : db.language_scid) { std::snprintf(line, sizeof line, "language scid=%d code=%d\n", kv.first, kv.second); s += line; }
    for (const auto &kv : db.packet) {
        std::snprintf(line, sizeof line, "packet scid=%d subch=%d dscty=%d addr=%d dg=%d\n", kv.second.scid, kv.second.subch, kv.second.dscty,
                      kv.second.packet_address, kv.second.dg_flag ? 1 : 0);
        s += line;
    }
    for (const auto &kv : db.fec_scheme) { std::snprintf(line, sizeof line, "fec subch=%d scheme=%d\n", kv.first, kv.second); s += line; }
    for (const auto &kv : db.switching) {
        std::snprintf(line, sizeof line, "switching cluster=%d flags=%04X subch=%d new=%d\n", kv.second.cluster, kv.second.flags, kv.second.subch,
                      kv.second.new_flag ? 1 : 0);
        s += line;
    }
    std::snprintf(line, sizeof line, "reconfiguration pending=%d next=%d applied=%d\n", db.change_pending ? 1 : 0, db.next ? 1 : 0, db.reconfigured ? 1 : 0);
    s += line;
    if (static_cast<int>(s.size()) + 1 > cap) return -1;
    std::memcpy(out, s.c_str(), s.size() + 1);
    return static_cast<int>(s.size());
}

// test hook: hand FIBs (CRC already verified) to a running receiver as if its FIC had delivered them; served on the
// library's thread like any request (tests/test_gpu_legacy_api.py: reconfiguration, ensemble change, FIG 0/13 update)
DABSDR_API void dabsdr_amd_inject_fibs(dabsdrHandle_t h, const uint8_t *fibs, int n_fibs)
{
    Request r;
    r.kind = Req::InjectFibs;
    r.blob.assign(fibs, fibs + 32 * static_cast<size_t>(n_fibs));
    post(h, r);
}

// test hook (CPU only): parse FIBs, then answer GetEnsemble / GetServiceList / GetServiceComponents exactly as the
// worker does and print the notification structs the host would receive (tests/test_reference_observed.py)
namespace {
void struct_dump_cb(dabsdrNotificationCBData_t *d, void *ctx)
{
    auto *hs = static_cast<std::pair<dabsdr_s *, std::string *> *>(ctx);
    std::string &s = *hs->second;
    char line[256];
    if (d->nid == DABSDR_NID_ENSEMBLE_INFO) {
        const auto *e = static_cast<const dabsdrNtfEnsemble_t *>(d->pData);
        std::snprintf(line, sizeof line, "ENSEMBLE status=%d ueid=0x%08X LTO=%d intTable=%d label='%.16s' charField=0x%04X\n", d->status, e->ueid, e->LTO,
                      e->intTable, e->label.str, e->label.charField);
        s += line;
    } else if (d->nid == DABSDR_NID_SERVICE_LIST) {
        const auto *l = static_cast<const dabsdrNtfServiceList_t *>(d->pData);
        std::snprintf(line, sizeof line, "SERVICE_LIST n=%d\n", l->numServices);
        s += line;
        for (int i = 0; i < l->numServices; ++i) {
            dabsdrServiceListItem_t it;
            l->getServiceListItem(hs->first, static_cast<uint8_t>(i), &it);
            std::snprintf(line, sizeof line, "  SId=0x%04X label='%.16s' pty=%d/%d CAId=%d\n", it.sid, it.label.str, it.pty.s, it.pty.d, it.CAId);
            s += line;
        }
    } else if (d->nid == DABSDR_NID_SERVICE_COMPONENT_LIST) {
        const auto *l = static_cast<const dabsdrNtfServiceComponentList_t *>(d->pData);
        std::snprintf(line, sizeof line, "SC_LIST SId=0x%04X n=%d\n", l->SId, l->numServiceComponents);
        s += line;
        for (int i = 0; i < l->numServiceComponents; ++i) {
            dabsdrServiceCompListItem_t it;
            l->getServiceComponentListItem(hs->first, static_cast<uint8_t>(i), &it);
            std::snprintf(line, sizeof line, "  SCIdS=%d SubChId=%d addr=%d size=%d prot=%d ps=%d TMId=%d ASCTy=%d bitrate=%d\n", it.SCIdS, it.SubChId,
                          it.SubChAddr, it.SubChSize, it.protectionLevel, it.ps, it.TMId, it.streamAudio.ASCTy, it.streamAudio.bitRate);
            s += line;
        }
    }
}
}  // namespace

DABSDR_API int dabsdr_amd_struct_dump(const uint8_t *fibs, int n_fibs, char *out, int cap)
{
    dabsdr_s h;
    std::string s;
    std::pair<dabsdr_s *, std::string *> ctx(&h, &s);
    h.ntf_cb = struct_dump_cb; h.ntf_ctx = &ctx;
    for (int i = 0; i < n_fibs; ++i) h.db.parse_fib(fibs + 32 * i);
    handle_request(&h, {Req::GetEnsemble, 0, 0, 0});
    handle_request(&h, {Req::GetServiceList, 0, 0, 0});
    std::vector<uint32_t> sids;
    for (const auto &kv : h.db.services) sids.push_back(kv.second.sid);
    for (uint32_t sid : sids) handle_request(&h, {Req::GetServiceComponents, sid, 0, 0});
    if (static_cast<int>(s.size()) + 1 > cap) return -1;
    std::memcpy(out, s.c_str(), s.size() + 1);
    return static_cast<int>(s.size());
}

}  // extern "C"
