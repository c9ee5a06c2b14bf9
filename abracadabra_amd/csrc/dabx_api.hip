// dabx_api.hip — host side of the batch C ABI (include/dabx.h).
//
// Owns the HBM buffers (layout in dabx_dev.h), the per-stream bookkeeping that
// the reference keeps in its global sample FIFO (reference:
// src/input/inputdevice.cpp:30-131) and the launch sequence of one decode step.
#include "dabx_kernels.hip"
#include "dabx_superframe.hip"
#include "dabx_resample.hip"
#include "dabx_spec.hpp"
#include "rawfile.hpp"
#include "../../include/dabx.h"

#include <algorithm>
#include <cmath>
#include <cstddef>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <new>
#include <vector>

namespace {


#define HIPCHK(expr)                                                                                   \
    do {                                                                                               \
        hipError_t e_ = (expr);                                                                        \
        if (e_ != hipSuccess) {                                                                        \
            std::fprintf(stderr, "dabx: %s failed: %s (%s:%d)\n", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
            return e_ == hipErrorOutOfMemory ? DABX_E_NOMEM : DABX_E_NODEV;                            \
        }                                                                                              \
    } while (0)

struct DevTmp {                              // device scratch of one call: freed on every way out
    void *p = nullptr;
    ~DevTmp() { if (p) (void)hipFree(p); }
    template <class T> T *as() const { return static_cast<T *>(p); }
};

template <class T>
int dev_upload(T *&dst, const std::vector<T> &src)
{
    if (dst) (void)hipFree(dst);
    dst = nullptr;
    if (src.empty()) return DABX_OK;
    HIPCHK(hipMalloc(reinterpret_cast<void **>(&dst), src.size() * sizeof(T)));
    HIPCHK(hipMemcpy(dst, src.data(), src.size() * sizeof(T), hipMemcpyHostToDevice));
    return DABX_OK;
}

struct StreamHost {
    int64_t wr = 0;                         // samples pushed
    DevState st = {};                       // last state read back from the device
    std::vector<dabx_subch_t> sub;
    std::vector<dabx::Profile> prof;
    int msc_bytes = 0;
    uint64_t dabplus = 0;                   // bit k: sub-channel k carries DAB+ audio (dabx_set_dabplus)
    float rs_mu = 0.0f;                     // Farrow resampler: fractional interval after the last input sample
    bool level_on = false;                  // dabx_enable_level: the converters also run the reference's signal-level detector
};

}  // namespace

struct dabx_ctx {
    dabx_config_t cfg = {};
    int bps = 2;                            // bytes per complex sample
    int ti_slots = 64;
    hipStream_t stream = nullptr;
    hipEvent_t ev[6] = {};
    bool timing = false;
    float last_ms[5] = {0, 0, 0, 0, 0};
    int last_frames = 0;
    bool pending = false;

    // device buffers
    uint8_t *d_ring = nullptr;
    DevState *d_state = nullptr;
    DevSync *d_sync = nullptr;
    int8_t *d_fic = nullptr, *d_ti = nullptr;
    uint8_t *d_fib = nullptr, *d_fib_ok = nullptr, *d_msc = nullptr, *d_msc_valid = nullptr;
    DevSub *d_sub = nullptr;
    uint32_t *d_info = nullptr;
    uint32_t *d_prbs = nullptr, *d_scratch = nullptr, *d_requeue = nullptr;
    uint32_t rq_words_per_wave = 0;
    uint64_t *d_clock = nullptr;            // k_clock_monitor (timing enabled): {shader cycles, 100 MHz ticks} beside the last k_viterbi, then the stop flag
    hipStream_t mon_stream = nullptr;
    hipEvent_t mon_start = nullptr;
    size_t work_cap_rq = 0;                 // entries the requeue list holds (its running total sits behind them)
    DevWork *d_work = nullptr;
    float2 *d_W = nullptr, *d_nhi = nullptr, *d_nlo = nullptr;
    int16_t *d_bop = nullptr, *d_nob = nullptr, *d_car = nullptr;
    int8_t *d_pq = nullptr, *d_pdq = nullptr;
    float *d_spectrum = nullptr, *d_null_spectrum = nullptr;   // optional, dabx_enable_spectrum
    DevState *h_state = nullptr;            // pinned mirror
    // single-stream contexts (the legacy 24-function path: one ensemble, one frame per step) get their small per-step
    // results with the step itself, into page-locked memory: the getters then cost no HIP call (each synchronous
    // device-to-host copy of a few hundred bytes took ~22 us, eight of them per 96 ms frame)
    uint8_t *h_res = nullptr;               // [F][384] fib | [F][12] fib_ok | [F][64] sync | [2048] float spectrum | [F][2048] null spectra
    bool res_valid = false;

    std::vector<StreamHost> streams;
    std::vector<uint32_t> info_pool;        // depuncturing maps, FIC first
    std::vector<dabx::Profile> pool_prof;
    std::vector<int> pool_off;
    std::vector<char> pool_linear;
    std::vector<DevSub> h_sub;
    bool work_dirty = true;
    int work_frames = 0, n_work = 0;
    // DAB+ super frame stage (dabx_superframe.hip)
    std::vector<DevSfSub> sf_subs;
    std::vector<dabx_subch_t> sf_ident;     // what each entry decodes: a sub-channel that stays keeps its synchronisation state
    DevSfSub *d_sf_subs = nullptr; DevSfState *d_sf_state = nullptr; DevSfRec *d_sf_recs = nullptr; uint8_t *d_sf_data = nullptr, *d_gf = nullptr;
    bool sf_dirty = false;
    int sf_max_rec = 0;
    hipStream_t copy_stream = nullptr;      // DABX_SRC_PINNED pushes: overlap with the kernels of a step in flight
    hipEvent_t copy_done = nullptr;
    bool copies_queued = false;
    size_t scratch_words = 0, work_cap = 0;
    // sample-rate conversion in front of the ring (dabx_resample.hip)
    rs::State *d_rs_state = nullptr;        // [S], zeroed
    uint8_t *d_rs_in = nullptr; float *d_rs_mu = nullptr; int32_t *d_rs_seg = nullptr; float2 *d_rs_A = nullptr, *d_rs_x = nullptr;
    size_t rs_in_cap = 0, rs_mu_cap = 0, rs_seg_cap = 0, rs_A_cap = 0;
    struct RsSched { float *mu = nullptr; int32_t *seg = nullptr; size_t cap_mu = 0, cap_seg = 0; hipEvent_t ev = nullptr; } rs_sched[2];
    int rs_slot = 0, rs_last_async = -1;
    // largest |I|, |Q| the converters wrote since the step before last was submitted: [2][S] on the device (the pushes between two
    // steps use slot step_count & 1), latched into h_peak [S] by the step that follows them
    uint32_t *d_peak = nullptr, *h_peak = nullptr;
    uint64_t step_count = 0;
    std::mutex mu;

    // bytes per stream in d_ring: the ring itself plus a mirror of its first DABX_RING_MIRROR samples behind its end, so
    // that a 2048-sample FFT window never has to wrap (kernels read it with a scalar base + immediate offsets)
    size_t stride() const { return (static_cast<size_t>(cfg.ring_samples) + DABX_RING_MIRROR) * bps; }
    uint8_t *ring_of(int s) const { return d_ring + static_cast<size_t>(s) * stride(); }

    DevCtx dev() const
    {
        DevCtx c = {};
        c.tab = {d_W, d_nhi, d_nlo, d_bop, d_nob, d_pq, d_pdq, d_car};
        c.state = d_state; c.sync = d_sync; c.ring = d_ring; c.fic_soft = d_fic; c.ti = d_ti;
        c.fib = d_fib; c.fib_ok = d_fib_ok; c.msc = d_msc; c.msc_valid = d_msc_valid;
        c.sub = d_sub; c.stepinfo = d_info; c.prbs = d_prbs; c.dec_scratch = d_scratch; c.requeue = d_requeue; c.spectrum = d_spectrum; c.null_spectrum = d_null_spectrum;
        c.ring_len = cfg.ring_samples; c.ring_bytes = stride();
        c.n_streams = cfg.n_streams; c.max_frames = cfg.max_frames; c.ti_slots = ti_slots;
        c.msc_stride = DABX_MSC_STRIDE; c.fic_info_off = 0; c.requeue_cap = static_cast<int32_t>(work_cap_rq);
        return c;
    }

    // gather maps (dabx_spec.hpp: step_gather) of the profiles in use, each once; offset in words.  FIC codewords are read from
    // linear rows, sub-channels from the residue-major rows of the time de-interleaver.
    int pool_lookup(const dabx::Profile &p, bool linear = false)
    {
        for (size_t i = 0; i < pool_prof.size(); ++i)
            if (pool_prof[i] == p && pool_linear[i] == linear) return pool_off[i];
        const int off = static_cast<int>(info_pool.size());
        const auto info = dabx::step_gather(p, linear);
        info_pool.insert(info_pool.end(), info.begin(), info.end());
        pool_prof.push_back(p);
        pool_linear.push_back(linear);
        pool_off.push_back(off);
        return off;
    }
};

namespace {

int upload_tables(dabx_ctx *c)
{
    std::vector<float2> W(2048), hi(2048), lo(2048);
    for (int k = 0; k < 2048; ++k) {
        const double a = 2.0 * M_PI * k / 2048.0, b = 2.0 * M_PI * k / 4194304.0;
        W[k] = make_float2(static_cast<float>(std::cos(a)), static_cast<float>(-std::sin(a)));
        hi[k] = make_float2(static_cast<float>(std::cos(a)), static_cast<float>(std::sin(a)));
        lo[k] = make_float2(static_cast<float>(std::cos(b)), static_cast<float>(std::sin(b)));
    }
    const auto car = dabx::carrier_of_symbol();
    const auto pq = dabx::prs_quadrants();
    std::vector<int16_t> bop(2048), nob(2048, -1), cfo;
    std::vector<int8_t> vq(pq.begin(), pq.end()), vdq(2048, -1);
    for (int p = 0; p < 2048; ++p) bop[p] = static_cast<int16_t>(dabx::bin_of_pos(p));
    for (int n = 0; n < dabx::kCarriers; ++n) nob[car[n] & 2047] = static_cast<int16_t>(n);
    for (int k = -767; k <= 768; ++k) {
        if (k == 0 || k == 1) continue;
        cfo.push_back(static_cast<int16_t>(k));
        vdq[k & 2047] = static_cast<int8_t>((pq[k & 2047] - pq[(k - 1) & 2047]) & 3);
    }
    int rc;
    if ((rc = dev_upload(c->d_W, W)) || (rc = dev_upload(c->d_nhi, hi)) || (rc = dev_upload(c->d_nlo, lo)) ||
        (rc = dev_upload(c->d_bop, bop)) || (rc = dev_upload(c->d_nob, nob)) || (rc = dev_upload(c->d_car, cfo)) ||
        (rc = dev_upload(c->d_pq, vq)) || (rc = dev_upload(c->d_pdq, vdq)))
        return rc;
    return dev_upload(c->d_prbs, dabx::prbs_words(dabx::kCifBits));
}

// (re)build the Viterbi work of a step of n_frames frames: one item per codeword (4 FIC codewords per
// frame, one per sub-channel and CIF), longest first, each with its own block of decision words.
int build_work(dabx_ctx *c, int n_frames)
{
    std::vector<DevWork> all;
    for (int s = 0; s < c->cfg.n_streams; ++s)
        for (int f = 0; f < n_frames; ++f) {
            for (int cw = 0; cw < 4; ++cw) all.push_back({s, static_cast<int16_t>(f), static_cast<int8_t>(cw), -1, 774u});
            const auto &sh = c->streams[s];
            for (int cif = 0; cif < 4; ++cif)
                for (size_t k = 0; k < sh.prof.size(); ++k)
                    all.push_back({s, static_cast<int16_t>(f), static_cast<int8_t>(cif), static_cast<int8_t>(k),
                                   static_cast<uint32_t>(sh.prof[k].steps())});
        }
    std::stable_sort(all.begin(), all.end(), [](const DevWork &a, const DevWork &b) { return a.nsteps > b.nsteps; });
    // scratch for the codewords whose survivors do not merge (k_viterbi_requeue): one block per WAVE of that kernel, sized for
    // the longest codeword — 64 words per 24 steps.  (Until round 3 every codeword had its own block: 2.5 GB for the benchmark's
    // 155 648 codewords per step, none of it ever touched.)
    const size_t words_per_wave = all.empty() ? 0 : (static_cast<size_t>(all.front().nsteps) / 24 + 1) * 64;
    const size_t blocks = words_per_wave / 64 * VIT_RQ_BLOCKS * 4;
    c->n_work = static_cast<int>(all.size());
    if (all.size() > c->work_cap) {                      // grows only: the one-frame legacy path alternates between sizes
        if (c->d_work) (void)hipFree(c->d_work);
        c->d_work = nullptr; c->work_cap = 0;
        HIPCHK(hipMalloc(reinterpret_cast<void **>(&c->d_work), all.size() * sizeof(DevWork)));
        uint32_t total = 0;                              // the running total moves with the list
        if (c->d_requeue) HIPCHK(hipMemcpy(&total, c->d_requeue + 1 + c->work_cap_rq, sizeof total, hipMemcpyDeviceToHost));
        if (c->d_requeue) (void)hipFree(c->d_requeue);
        c->d_requeue = nullptr;
        HIPCHK(hipMalloc(reinterpret_cast<void **>(&c->d_requeue), (all.size() + 2) * sizeof(uint32_t)));
        HIPCHK(hipMemset(c->d_requeue, 0, sizeof(uint32_t)));
        HIPCHK(hipMemcpy(c->d_requeue + 1 + all.size(), &total, sizeof total, hipMemcpyHostToDevice));
        c->work_cap_rq = all.size();
        c->work_cap = all.size();
    }
    c->rq_words_per_wave = static_cast<uint32_t>(words_per_wave);

    if (!all.empty()) HIPCHK(hipMemcpy(c->d_work, all.data(), all.size() * sizeof(DevWork), hipMemcpyHostToDevice));
    if (blocks * 64 > c->scratch_words) {
        if (c->d_scratch) (void)hipFree(c->d_scratch);
        c->d_scratch = nullptr;
        c->scratch_words = 0;
        if (hipMalloc(reinterpret_cast<void **>(&c->d_scratch), blocks * 64 * sizeof(uint32_t)) != hipSuccess) return DABX_E_NOMEM;
        c->scratch_words = blocks * 64;
    }
    c->work_frames = n_frames;
    c->work_dirty = false;
    return DABX_OK;
}

// (re)build the list of DAB+ sub-channels.  A sub-channel that was already being decoded (same stream, same
// position, protection and rate) keeps its state, so that changing the set next to it does not interrupt it.
int build_superframe_work(dabx_ctx *c)
{
    const std::vector<DevSfSub> old_subs = c->sf_subs;
    const std::vector<dabx_subch_t> old_ident = c->sf_ident;
    DevSfState *old_state = c->d_sf_state;
    DevTmp old_guard;                                                 // the previous state array goes on every way out
    old_guard.p = old_state;
    c->sf_subs.clear();
    c->sf_ident.clear();
    c->sf_max_rec = (4 + 4 * c->cfg.max_frames) / 5;
    uint32_t rec = 0, data = 0;
    for (int s = 0; s < c->cfg.n_streams; ++s) {
        const auto &sh = c->streams[s];
        for (size_t k = 0; k < sh.sub.size() && k < 64; ++k) {
            if (!((sh.dabplus >> k) & 1)) continue;
            const int kbps = sh.prof[k].n_in / 24;                    // n_in = 24 ms x kbps
            if (kbps < 8 || kbps > 192 || kbps % 8) return DABX_E_PROFILE;
            DevSfSub d = {s, static_cast<int32_t>(k), kbps / 8, 3 * kbps, c->h_sub[static_cast<size_t>(s) * 64 + k].out_off, rec, data, 0u};
            rec += static_cast<uint32_t>(c->sf_max_rec);
            data += static_cast<uint32_t>(c->sf_max_rec) * 110u * static_cast<uint32_t>(kbps / 8);
            c->sf_subs.push_back(d);
            c->sf_ident.push_back(sh.sub[k]);
        }
    }
    for (void *p : {static_cast<void *>(c->d_sf_subs), static_cast<void *>(c->d_sf_recs), static_cast<void *>(c->d_sf_data)})
        if (p) (void)hipFree(p);
    c->d_sf_subs = nullptr; c->d_sf_state = nullptr; c->d_sf_recs = nullptr; c->d_sf_data = nullptr;
    c->sf_dirty = false;
    if (c->sf_subs.empty()) return DABX_OK;
    if (!c->d_gf) {                                                   // GF(2^8) tables: exp[512], log[256]
        std::vector<uint8_t> tab(768, 0);
        unsigned x = 1;
        for (int i = 0; i < 255; ++i) { tab[i] = static_cast<uint8_t>(x); tab[512 + x] = static_cast<uint8_t>(i); x <<= 1; if (x & 0x100) x ^= 0x11D; }
        for (int i = 255; i < 512; ++i) tab[i] = tab[i - 255];
        int rc = dev_upload(c->d_gf, tab);
        if (rc) return rc;
    }
    const size_t n = c->sf_subs.size();
    int rc = dev_upload(c->d_sf_subs, c->sf_subs);
    if (rc) return rc;
    HIPCHK(hipMalloc(reinterpret_cast<void **>(&c->d_sf_state), n * sizeof(DevSfState)));
    HIPCHK(hipMemset(c->d_sf_state, 0, n * sizeof(DevSfState)));
    for (size_t i = 0; i < n && old_state; ++i)
        for (size_t j = 0; j < old_subs.size(); ++j) {
            const dabx_subch_t &a = c->sf_ident[i], &b = old_ident[j];
            if (old_subs[j].stream == c->sf_subs[i].stream && a.start_cu == b.start_cu && a.option == b.option && a.level == b.level && a.kbps == b.kbps) {
                HIPCHK(hipMemcpy(c->d_sf_state + i, old_state + j, sizeof(DevSfState), hipMemcpyDeviceToDevice));
                break;
            }
        }
    HIPCHK(hipMalloc(reinterpret_cast<void **>(&c->d_sf_recs), static_cast<size_t>(rec) * sizeof(DevSfRec)));
    HIPCHK(hipMalloc(reinterpret_cast<void **>(&c->d_sf_data), std::max<size_t>(data, 16)));
    return DABX_OK;
}

int64_t samples_needed(const StreamHost &s, int n_frames)
{
    // (one frame more while searching for the null symbol: not locked, or locked with the last frame's PRS missing)
    return s.st.pos + static_cast<int64_t>(n_frames + ((s.st.locked && s.st.bad == 0) ? 0 : 1)) * dabx::kTF + 4096;
}

bool valid_stream(const dabx_ctx *c, int s) { return c && s >= 0 && s < c->cfg.n_streams; }

}  // namespace

namespace {
template <class T>
int grow(T *&p, size_t &cap, size_t need)
{
    if (need <= cap) return DABX_OK;
    if (p) (void)hipFree(p);
    p = nullptr; cap = 0;
    const size_t n = need + need / 2 + 1024;
    HIPCHK(hipMalloc(reinterpret_cast<void **>(&p), n * sizeof(T)));
    cap = n;
    return DABX_OK;
}
}  // namespace

extern "C" {

const char *dabx_strerror(int code)
{
    switch (code) {
    case DABX_OK: return "ok";
    case DABX_E_ARG: return "bad argument";
    case DABX_E_NODEV: return "HIP device/runtime error";
    case DABX_E_NOMEM: return "out of device memory";
    case DABX_E_UNDERRUN: return "not enough samples buffered";
    case DABX_E_OVERRUN: return "ring overrun";
    case DABX_E_PROFILE: return "unsupported protection profile";
    default: return "unknown error";
    }
}

static int create_body(dabx_ctx *c, const dabx_config_t *cfg);

int dabx_create(const dabx_config_t *cfg, dabx_ctx **out)
{
    if (!cfg || !out || cfg->n_streams < 1 || cfg->max_frames < 1 || cfg->max_frames > 60 ||
        (cfg->fmt != DABX_FMT_U8 && cfg->fmt != DABX_FMT_S16) ||
        cfg->ring_samples < static_cast<int64_t>(cfg->max_frames + 2) * dabx::kTF ||
        cfg->ring_samples > (1LL << 30))                    // the kernels index one stream's ring with 32 bits (5 000 s of signal)
        return DABX_E_ARG;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || cfg->device < 0 || cfg->device >= ndev) {
        std::fprintf(stderr, "dabx: no HIP device %d (found %d) — this library has no CPU path\n", cfg->device, ndev);
        return DABX_E_NODEV;
    }
    HIPCHK(hipSetDevice(cfg->device));
    dabx_ctx *c = new (std::nothrow) dabx_ctx;
    if (!c) return DABX_E_NOMEM;
    const int rc = create_body(c, cfg);
    if (rc) { dabx_destroy(c); return rc; }          // whatever had been allocated goes with it
    *out = c;
    return DABX_OK;
}

static int create_body(dabx_ctx *c, const dabx_config_t *cfg)
{
    c->cfg = *cfg;
    c->bps = cfg->fmt == DABX_FMT_U8 ? 2 : 4;
    c->ti_slots = 16;
    while (c->ti_slots < 15 + 4 * cfg->max_frames) c->ti_slots *= 2;
    c->streams.resize(cfg->n_streams);
    const size_t S = cfg->n_streams, F = cfg->max_frames;
#define ALLOC(ptr, bytes)                                                         \
    do {                                                                          \
        HIPCHK(hipMalloc(reinterpret_cast<void **>(&(ptr)), (bytes)));            \
        HIPCHK(hipMemset((ptr), 0, (bytes)));                                     \
    } while (0)
    HIPCHK(hipStreamCreate(&c->stream));
    HIPCHK(hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking));
    HIPCHK(hipEventCreateWithFlags(&c->copy_done, hipEventDisableTiming));
    for (auto &e : c->ev) HIPCHK(hipEventCreate(&e));
    ALLOC(c->d_ring, S * c->stride());
    if (cfg->fmt == DABX_FMT_U8) HIPCHK(hipMemset(c->d_ring, 128, S * c->stride()));
    ALLOC(c->d_state, S * sizeof(DevState));
    ALLOC(c->d_sync, S * F * sizeof(DevSync));
    ALLOC(c->d_fic, S * F * DABX_FIC_SOFT_BITS + 64);
    ALLOC(c->d_ti, S * static_cast<size_t>(c->ti_slots) * DABX_CIF_SOFT_BITS + 64);   // + slack for the 4-byte soft-bit fetches

    ALLOC(c->d_fib, S * F * 12 * 32);
    ALLOC(c->d_fib_ok, S * F * 12);
    ALLOC(c->d_msc, S * F * 4 * DABX_MSC_STRIDE);
    ALLOC(c->d_msc_valid, S * F * 4);
    ALLOC(c->d_sub, S * 64 * sizeof(DevSub));
#undef ALLOC
    HIPCHK(hipHostMalloc(reinterpret_cast<void **>(&c->h_state), S * sizeof(DevState)));
    if (S == 1) HIPCHK(hipHostMalloc(reinterpret_cast<void **>(&c->h_res), F * (384 + 12 + sizeof(DevSync)) + (1 + F) * 2048 * sizeof(float)));
    std::memset(c->h_state, 0, S * sizeof(DevState));
    c->h_sub.assign(S * 64, DevSub{});
    int rc = upload_tables(c);
    if (rc) return rc;
    c->pool_lookup(dabx::fic_profile(), true);           // offset 0
    return dev_upload(c->d_info, c->info_pool);
}

void dabx_destroy(dabx_ctx *c)
{
    if (!c) return;
    (void)hipSetDevice(c->cfg.device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->copy_stream) { (void)hipStreamSynchronize(c->copy_stream); (void)hipStreamDestroy(c->copy_stream); }
    if (c->copy_done) (void)hipEventDestroy(c->copy_done);
    if (c->mon_stream) { (void)hipStreamSynchronize(c->mon_stream); (void)hipStreamDestroy(c->mon_stream); }
    if (c->mon_start) (void)hipEventDestroy(c->mon_start);
    if (c->d_clock) (void)hipFree(c->d_clock);
    void *bufs[] = {c->d_ring, c->d_state, c->d_sync, c->d_fic, c->d_ti, c->d_fib, c->d_fib_ok, c->d_msc, c->d_msc_valid,
                    c->d_sub, c->d_info, c->d_prbs, c->d_scratch, c->d_requeue, c->d_work, c->d_sf_subs, c->d_sf_state, c->d_sf_recs, c->d_sf_data, c->d_gf, c->d_spectrum, c->d_null_spectrum, c->d_W, c->d_nhi, c->d_nlo, c->d_bop,
                    c->d_nob, c->d_car, c->d_pq, c->d_pdq, c->d_rs_state, c->d_rs_in, c->d_rs_mu, c->d_rs_seg, c->d_rs_A, c->d_rs_x};
    for (void *b : bufs)
        if (b) (void)hipFree(b);
    if (c->h_state) (void)hipHostFree(c->h_state);
    if (c->h_res) (void)hipHostFree(c->h_res);
    if (c->h_peak) (void)hipHostFree(c->h_peak);
    if (c->d_peak) (void)hipFree(c->d_peak);
    for (auto &sl : c->rs_sched) {
        if (sl.mu) (void)hipHostFree(sl.mu);
        if (sl.seg) (void)hipHostFree(sl.seg);
        if (sl.ev) (void)hipEventDestroy(sl.ev);
    }
    for (auto &e : c->ev)
        if (e) (void)hipEventDestroy(e);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

int dabx_set_subchannels(dabx_ctx *c, int s, int n, const dabx_subch_t *sub)
{
    if (!valid_stream(c, s) || n < 0 || n > DABX_MAX_SUBCH || (n && !sub)) return DABX_E_ARG;
    std::lock_guard<std::mutex> lk(c->mu);
    (void)hipSetDevice(c->cfg.device);              /* a host may drive several contexts (GPUs) from one thread */
    if (c->pending) return DABX_E_ARG;
    std::vector<dabx::Profile> prof(n);
    std::vector<DevSub> ds(64, DevSub{});
    int out_off = 0;
    for (int i = 0; i < n; ++i) {
        if (!dabx::any_profile(sub[i].option, sub[i].level, sub[i].kbps, prof[i])) return DABX_E_PROFILE;
        if (sub[i].start_cu < 0 || sub[i].start_cu + prof[i].n_cu > dabx::kNumCu) return DABX_E_ARG;
        ds[i] = {sub[i].start_cu * dabx::kCuBits, prof[i].steps(), prof[i].n_in, c->pool_lookup(prof[i]), out_off};
        out_off += prof[i].n_in / 8;
    }
    if (out_off > DABX_MSC_STRIDE) return DABX_E_ARG;
    auto &sh = c->streams[s];
    sh.sub.assign(sub, sub + n);
    if (sh.dabplus) { sh.dabplus = 0; c->sf_dirty = true; }          // a new layout: the caller flags its DAB+ sub-channels again
    sh.prof = prof;
    sh.msc_bytes = out_off;
    std::copy(ds.begin(), ds.end(), c->h_sub.begin() + static_cast<size_t>(s) * 64);
    HIPCHK(hipMemcpy(c->d_sub + static_cast<size_t>(s) * 64, ds.data(), 64 * sizeof(DevSub), hipMemcpyHostToDevice));
    int rc = dev_upload(c->d_info, c->info_pool);
    if (rc) return rc;
    c->work_dirty = true;
    return out_off;
}

int dabx_push(dabx_ctx *c, int s, const void *src, int64_t n, int kind)
{
    if (!valid_stream(c, s) || n < 0 || (n && !src) || kind < DABX_SRC_HOST || kind > DABX_SRC_PINNED) return DABX_E_ARG;
    std::lock_guard<std::mutex> lk(c->mu);
    (void)hipSetDevice(c->cfg.device);              /* a host may drive several contexts (GPUs) from one thread */
    auto &sh = c->streams[s];
    const int64_t len = c->cfg.ring_samples;
    // samples older than (pos - one frame) are no longer needed by any kernel; while a step is in flight
    // sh.st.pos is still the position before it, so the region its kernels read is protected as well
    if (sh.wr + n - std::max<int64_t>(0, sh.st.pos - dabx::kTF) > len) return DABX_E_OVERRUN;
    uint8_t *ring = c->ring_of(s);
    const hipMemcpyKind mk = kind == DABX_SRC_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
    hipStream_t q = kind == DABX_SRC_PINNED ? c->copy_stream : c->stream;
    int64_t done = 0;
    while (done < n) {
        const int64_t w = (sh.wr + done) % len, chunk = std::min(n - done, len - w);
        const uint8_t *from = static_cast<const uint8_t *>(src) + done * c->bps;
        HIPCHK(hipMemcpyAsync(ring + w * c->bps, from, static_cast<size_t>(chunk) * c->bps, mk, q));
        if (w < DABX_RING_MIRROR)                    // the head of the ring is mirrored behind its end
            HIPCHK(hipMemcpyAsync(ring + (len + w) * c->bps, from, static_cast<size_t>(std::min<int64_t>(chunk, DABX_RING_MIRROR - w)) * c->bps, mk, q));
        done += chunk;
    }
    if (kind == DABX_SRC_HOST) HIPCHK(hipStreamSynchronize(c->stream));   // the caller may reuse its buffer
    if (kind == DABX_SRC_PINNED) c->copies_queued = true;                 // the next step waits for the copy stream
    sh.wr += n;
    return DABX_OK;
}

int dabx_push_all(dabx_ctx *c, const void *src, size_t stride, int64_t n, int kind)
{
    if (!c || n < 0 || (n && !src) || kind < DABX_SRC_HOST || kind > DABX_SRC_PINNED) return DABX_E_ARG;
    const int S = c->cfg.n_streams;
    const int64_t len = c->cfg.ring_samples;
    bool lockstep = true;
    {
        std::lock_guard<std::mutex> lk(c->mu);
    (void)hipSetDevice(c->cfg.device);              /* a host may drive several contexts (GPUs) from one thread */
        for (int s = 0; s < S; ++s) {
            const auto &sh = c->streams[s];
            lockstep = lockstep && sh.wr == c->streams[0].wr;
            if (sh.wr + n - std::max<int64_t>(0, sh.st.pos - dabx::kTF) > len) return DABX_E_OVERRUN;
        }
        if (lockstep && n) {
            const hipMemcpyKind mk = kind == DABX_SRC_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
            hipStream_t q = kind == DABX_SRC_PINNED ? c->copy_stream : c->stream;
            int64_t done = 0;
            while (done < n) {                       // at most two pieces: the ring wraps at the same place in every stream
                const int64_t w = (c->streams[0].wr + done) % len, chunk = std::min(n - done, len - w);
                const uint8_t *from = static_cast<const uint8_t *>(src) + done * c->bps;
                HIPCHK(hipMemcpy2DAsync(c->d_ring + w * c->bps, c->stride(), from, stride, static_cast<size_t>(chunk) * c->bps, S, mk, q));
                if (w < DABX_RING_MIRROR)
                    HIPCHK(hipMemcpy2DAsync(c->d_ring + (len + w) * c->bps, c->stride(), from, stride,
                                            static_cast<size_t>(std::min<int64_t>(chunk, DABX_RING_MIRROR - w)) * c->bps, S, mk, q));
                done += chunk;
            }
            if (kind == DABX_SRC_HOST) HIPCHK(hipStreamSynchronize(c->stream));
            if (kind == DABX_SRC_PINNED) c->copies_queued = true;
            for (auto &sh : c->streams) sh.wr += n;
            return DABX_OK;
        }
    }
    for (int s = 0; s < S; ++s) {
        const int rc = dabx_push(c, s, static_cast<const uint8_t *>(src) + s * stride, n, kind);
        if (rc) return rc;
    }
    return DABX_OK;
}

// Host side of the Farrow schedule for an asynchronous push: two page-locked slots (mu per input sample, the first sample of
// every segment), each guarded by an event recorded behind the copies that read it.
static int rs_sched_slot(dabx_ctx *c, size_t n_mu, size_t n_seg, float *&mu, int32_t *&seg, int &slot)
{
    slot = c->rs_slot ^= 1;
    auto &sl = c->rs_sched[slot];
    if (sl.ev) HIPCHK(hipEventSynchronize(sl.ev));                      // the copies queued from this slot two pushes ago
    else HIPCHK(hipEventCreateWithFlags(&sl.ev, hipEventDisableTiming));
    if (n_mu > sl.cap_mu) {
        if (sl.mu) (void)hipHostFree(sl.mu);
        sl.mu = nullptr; sl.cap_mu = 0;
        HIPCHK(hipHostMalloc(reinterpret_cast<void **>(&sl.mu), (n_mu + n_mu / 2 + 1024) * sizeof(float)));
        sl.cap_mu = n_mu + n_mu / 2 + 1024;
    }
    if (n_seg > sl.cap_seg) {
        if (sl.seg) (void)hipHostFree(sl.seg);
        sl.seg = nullptr; sl.cap_seg = 0;
        HIPCHK(hipHostMalloc(reinterpret_cast<void **>(&sl.seg), (n_seg + n_seg / 2 + 1024) * sizeof(int32_t)));
        sl.cap_seg = n_seg + n_seg / 2 + 1024;
    }
    mu = sl.mu; seg = sl.seg;
    return DABX_OK;
}

int64_t dabx_push_resampled_from(dabx_ctx *c, int s, const void *src, int64_t n, int src_fmt, double in_rate_hz, float gain, int kind)
{
    if (!valid_stream(c, s) || n < 0 || (n && !src) || (src_fmt != DABX_FMT_S16 && src_fmt != DABX_FMT_F32) || !(in_rate_hz >= 2048000.0) ||
        in_rate_hz > 32768000.0 || n > (1 << 28) || c->cfg.fmt != DABX_FMT_S16 || kind < DABX_SRC_HOST || kind > DABX_SRC_PINNED)
        return DABX_E_ARG;
    const bool ds2 = in_rate_hz == 4096000.0, copy = in_rate_hz == 2048000.0;
    if (ds2 && (n & 1)) return DABX_E_ARG;
    std::lock_guard<std::mutex> lk(c->mu);
    (void)hipSetDevice(c->cfg.device);
    auto &sh = c->streams[s];
    const int64_t len = c->cfg.ring_samples;
    const size_t bpc = src_fmt == DABX_FMT_S16 ? 4 : 8;                  // bytes per complex input sample
    // DABX_SRC_PINNED: staging copy and converter kernels go to the copy stream and overlap with a decode step in flight;
    // the carried filter state lives in device memory and is handed from launch to launch in stream order, so a change of
    // the source kind first drains the stream the previous push used
    const bool async = kind == DABX_SRC_PINNED;
    hipStream_t q = async ? c->copy_stream : c->stream;
    if (c->rs_last_async != static_cast<int>(async)) {
        if (c->rs_last_async >= 0) HIPCHK(hipStreamSynchronize(c->rs_last_async ? c->copy_stream : c->stream));
        c->rs_last_async = static_cast<int>(async);
    }
    if (!c->d_rs_state) {
        HIPCHK(hipMalloc(reinterpret_cast<void **>(&c->d_rs_state), c->cfg.n_streams * sizeof(rs::State)));
        HIPCHK(hipMemset(c->d_rs_state, 0, c->cfg.n_streams * sizeof(rs::State)));
        HIPCHK(hipMalloc(reinterpret_cast<void **>(&c->d_rs_x), rs::FW_M * sizeof(float2)));
        HIPCHK(hipMalloc(reinterpret_cast<void **>(&c->d_peak), 2 * static_cast<size_t>(c->cfg.n_streams) * sizeof(uint32_t)));
        HIPCHK(hipMemset(c->d_peak, 0, 2 * static_cast<size_t>(c->cfg.n_streams) * sizeof(uint32_t)));
    }
    if (n == 0) return 0;
    int rc;
    if (static_cast<size_t>(n) * bpc > c->rs_in_cap) {               // growing frees the old buffer: nothing may still read it
        HIPCHK(hipStreamSynchronize(q));
        if ((rc = grow(c->d_rs_in, c->rs_in_cap, static_cast<size_t>(n) * bpc))) return rc;
    }
    const void *in = c->d_rs_in;
    if (kind == DABX_SRC_DEVICE) in = src;                               // already on this GPU: converted in place
    else HIPCHK(hipMemcpyAsync(c->d_rs_in, src, static_cast<size_t>(n) * bpc, hipMemcpyHostToDevice, q));
    rs::State *st = c->d_rs_state + s;
    short2 *ring = reinterpret_cast<short2 *>(c->ring_of(s));
    uint32_t *peak = c->d_peak + static_cast<size_t>(c->step_count & 1) * c->cfg.n_streams + s;
    int64_t n_out = 0;
    // the reference picks the converter by rate (inputdevicesrc.cpp:33-47)
    if (copy || ds2) {
        n_out = copy ? n : n / 2;
        if (sh.wr + n_out - std::max<int64_t>(0, sh.st.pos - dabx::kTF) > len) return DABX_E_OVERRUN;
        const unsigned blocks = static_cast<unsigned>((n_out + 255) / 256);
        if (copy) {
            if (src_fmt == DABX_FMT_S16) hipLaunchKernelGGL(rs::k_resample_copy<1>, dim3(blocks), dim3(256), 0, q, in, static_cast<int>(n_out), ring, len, sh.wr, gain, peak);
            else hipLaunchKernelGGL(rs::k_resample_copy<2>, dim3(blocks), dim3(256), 0, q, in, static_cast<int>(n_out), ring, len, sh.wr, gain, peak);
        } else if (src_fmt == DABX_FMT_S16) {
            hipLaunchKernelGGL(rs::k_resample_ds2<1>, dim3(blocks), dim3(256), 0, q, in, static_cast<int>(n_out), st, ring, len, sh.wr, gain, peak);
            hipLaunchKernelGGL(rs::k_ds2_tail<1>, dim3(1), dim3(64), 0, q, in, n, st);
        } else {
            hipLaunchKernelGGL(rs::k_resample_ds2<2>, dim3(blocks), dim3(256), 0, q, in, static_cast<int>(n_out), st, ring, len, sh.wr, gain, peak);
            hipLaunchKernelGGL(rs::k_ds2_tail<2>, dim3(1), dim3(64), 0, q, in, n, st);
        }
    } else {
        // the schedule of the transposed Farrow structure does not depend on the data: mu <- mu - R, a dump (one output)
        // whenever it turns negative (inputdevicesrc.cpp:241-245).  The recursion is serial in float (every step rounds), so
        // the host runs it ahead into page-locked memory; the kernels do the arithmetic on the samples.
        const float R = static_cast<float>(2048e3 / static_cast<double>(static_cast<float>(in_rate_hz)));
        if (R >= 0.5f) {
            // up to 4096 kHz the recursion is exact integer arithmetic modulo 2^24 (dabx_resample.hip: k_farrow_fused): the schedule is
            // closed form, the threads work it out themselves, nothing is computed or uploaded by the host
            const uint32_t Ri = static_cast<uint32_t>(R * 16777216.0f), M0 = static_cast<uint32_t>(sh.rs_mu * 16777216.0f);
            const uint64_t drop = static_cast<uint64_t>(n) * Ri;                  // how far mu travels over the block
            const int64_t n_done = drop <= M0 ? 0 : static_cast<int64_t>((drop - M0 + 16777215u) >> 24);
            n_out = n_done;
            if (sh.wr + n_out - std::max<int64_t>(0, sh.st.pos - dabx::kTF) > len) return DABX_E_OVERRUN;
            const rs::Sched sc = {M0, Ri};
            const unsigned blocks = static_cast<unsigned>((n_done + 255) / 256);
            if (src_fmt == DABX_FMT_S16) {
                if (n_done) hipLaunchKernelGGL(rs::k_farrow_fused<1>, dim3(blocks), dim3(256), 0, q, in, sc, static_cast<int>(n_done), st, ring, len, sh.wr, R, gain, peak);
                hipLaunchKernelGGL(rs::k_farrow_finish<1>, dim3(1), dim3(64), 0, q, in, sc, static_cast<int>(n_done), n, st);
            } else {
                if (n_done) hipLaunchKernelGGL(rs::k_farrow_fused<2>, dim3(blocks), dim3(256), 0, q, in, sc, static_cast<int>(n_done), st, ring, len, sh.wr, R, gain, peak);
                hipLaunchKernelGGL(rs::k_farrow_finish<2>, dim3(1), dim3(64), 0, q, in, sc, static_cast<int>(n_done), n, st);
            }
            sh.rs_mu = static_cast<float>(static_cast<uint32_t>((static_cast<uint64_t>(M0) - drop) & 0xFFFFFFu)) * 5.9604644775390625e-08f;
        } else {
        float *mu = nullptr;
        int32_t *seg = nullptr;
        int slot = 0;
        if ((rc = rs_sched_slot(c, static_cast<size_t>(n), static_cast<size_t>(n) + 2, mu, seg, slot))) return rc;
        float m = sh.rs_mu;
        size_t n_seg = 1;
        seg[0] = 0;
        for (int64_t k = 0; k < n; ++k) {
            m = m - R;
            if (m < 0) { m = m + 1.0f; seg[n_seg++] = static_cast<int32_t>(k); }
            mu[static_cast<size_t>(k)] = m;
        }
        const int n_done = static_cast<int>(n_seg) - 1;
        n_out = n_done;
        if (sh.wr + n_out - std::max<int64_t>(0, sh.st.pos - dabx::kTF) > len) return DABX_E_OVERRUN;
        const size_t need_A = static_cast<size_t>(n_done + 1) * rs::FW_N;
        if (static_cast<size_t>(n) > c->rs_mu_cap || n_seg > c->rs_seg_cap || need_A > c->rs_A_cap) {
            HIPCHK(hipStreamSynchronize(q));
            if ((rc = grow(c->d_rs_mu, c->rs_mu_cap, static_cast<size_t>(n))) || (rc = grow(c->d_rs_seg, c->rs_seg_cap, n_seg)) ||
                (rc = grow(c->d_rs_A, c->rs_A_cap, need_A)))
                return rc;
        }
        HIPCHK(hipMemcpyAsync(c->d_rs_mu, mu, static_cast<size_t>(n) * sizeof(float), hipMemcpyHostToDevice, q));
        HIPCHK(hipMemcpyAsync(c->d_rs_seg, seg, n_seg * sizeof(int32_t), hipMemcpyHostToDevice, q));
        HIPCHK(hipEventRecord(c->rs_sched[slot].ev, q));
        const unsigned blocks = static_cast<unsigned>((n_done + 255) / 256);
        if (src_fmt == DABX_FMT_S16) {
            if (n_done) hipLaunchKernelGGL(rs::k_farrow_segments<1>, dim3(blocks), dim3(256), 0, q, in, c->d_rs_seg, c->d_rs_mu, n_done, st, c->d_rs_A);
            hipLaunchKernelGGL(rs::k_farrow_open<1>, dim3(1), dim3(64), 0, q, in, c->d_rs_seg, c->d_rs_mu, n_done, n, st, c->d_rs_x);
        } else {
            if (n_done) hipLaunchKernelGGL(rs::k_farrow_segments<2>, dim3(blocks), dim3(256), 0, q, in, c->d_rs_seg, c->d_rs_mu, n_done, st, c->d_rs_A);
            hipLaunchKernelGGL(rs::k_farrow_open<2>, dim3(1), dim3(64), 0, q, in, c->d_rs_seg, c->d_rs_mu, n_done, n, st, c->d_rs_x);
        }
        if (n_done) hipLaunchKernelGGL(rs::k_farrow_outputs, dim3(blocks), dim3(256), 0, q, c->d_rs_A, n_done, st, ring, len, sh.wr, R, gain, peak);
        hipLaunchKernelGGL(rs::k_farrow_tail, dim3(1), dim3(64), 0, q, c->d_rs_A, n_done, st, c->d_rs_x);
        sh.rs_mu = m;
        }
    }
    if (sh.level_on) {
        // the reference's level detector over the INPUT samples (inputdevicesrc.cpp:167-173 / 282-292 / 330-341); constants as its
        // constructors form them (:84-85, :206-207, :316-317)
        const double base_rate = (copy || ds2) ? 2048e3 : static_cast<double>(static_cast<float>(in_rate_hz));
        const float catt = static_cast<float>(1 - std::exp(-1 / (5e-5 * base_rate))), crel = static_cast<float>(1 - std::exp(-1 / (5e-2 * base_rate)));
        if (src_fmt == DABX_FMT_S16) hipLaunchKernelGGL(rs::k_level<1>, dim3(1), dim3(64), 0, q, in, n, ds2 ? 1 : 0, ds2 ? 2 : 1, catt, crel, st);
        else hipLaunchKernelGGL(rs::k_level<2>, dim3(1), dim3(64), 0, q, in, n, ds2 ? 1 : 0, ds2 ? 2 : 1, catt, crel, st);
    }
    HIPCHK(hipGetLastError());
    if (async) c->copies_queued = true;                  // the next step waits for the copy stream
    else HIPCHK(hipStreamSynchronize(q));                // the caller may reuse its buffer
    sh.wr += n_out;
    return n_out;
}

int64_t dabx_push_resampled(dabx_ctx *c, int s, const void *src, int64_t n, int src_fmt, double in_rate_hz, float gain)
{
    return dabx_push_resampled_from(c, s, src, n, src_fmt, in_rate_hz, gain, DABX_SRC_HOST);
}

int dabx_enable_level(dabx_ctx *c, int s, int on)
{
    if (!valid_stream(c, s)) return DABX_E_ARG;
    std::lock_guard<std::mutex> lk(c->mu);
    (void)hipSetDevice(c->cfg.device);
    auto &sh = c->streams[s];
    if (c->d_rs_state && (on != 0) != sh.level_on) {             // switching it starts from level 0, like resetSignalLevel()
        HIPCHK(hipStreamSynchronize(c->copy_stream));
        HIPCHK(hipStreamSynchronize(c->stream));
        HIPCHK(hipMemset(reinterpret_cast<char *>(c->d_rs_state + s) + offsetof(rs::State, level), 0, sizeof(float)));
    }
    sh.level_on = on != 0;
    return DABX_OK;
}

int dabx_get_level(dabx_ctx *c, int s, float *level)
{
    if (!valid_stream(c, s) || !level) return DABX_E_ARG;
    std::lock_guard<std::mutex> lk(c->mu);
    (void)hipSetDevice(c->cfg.device);
    *level = 0.0f;
    if (!c->d_rs_state) return DABX_OK;
    HIPCHK(hipStreamSynchronize(c->copy_stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    HIPCHK(hipMemcpy(level, reinterpret_cast<const char *>(c->d_rs_state + s) + offsetof(rs::State, level), sizeof(float), hipMemcpyDeviceToHost));
    return DABX_OK;
}

int dabx_get_input_peak(dabx_ctx *c, int s, int32_t *peak)
{
    if (!valid_stream(c, s) || !peak) return DABX_E_ARG;
    std::lock_guard<std::mutex> lk(c->mu);
    if (c->pending) return DABX_E_ARG;
    *peak = c->h_peak ? static_cast<int32_t>(c->h_peak[s]) : 0;
    return DABX_OK;
}

int dabx_read_ring(dabx_ctx *c, int s, int64_t from, int64_t n, void *dst)
{
    if (!valid_stream(c, s) || from < 0 || n < 0 || (n && !dst) || n > c->cfg.ring_samples) return DABX_E_ARG;
    std::lock_guard<std::mutex> lk(c->mu);
    (void)hipSetDevice(c->cfg.device);
    const int64_t len = c->cfg.ring_samples;
    const uint8_t *ring = c->ring_of(s);
    HIPCHK(hipStreamSynchronize(c->stream));
    int64_t done = 0;
    while (done < n) {
        const int64_t w = (from + done) % len, chunk = std::min(n - done, len - w);
        HIPCHK(hipMemcpy(static_cast<uint8_t *>(dst) + done * c->bps, ring + w * c->bps, static_cast<size_t>(chunk) * c->bps, hipMemcpyDeviceToHost));
        done += chunk;
    }
    return DABX_OK;
}

int dabx_flush_copies(dabx_ctx *c)
{
    if (!c) return DABX_E_ARG;
    std::lock_guard<std::mutex> lk(c->mu);
    (void)hipSetDevice(c->cfg.device);
    HIPCHK(hipStreamSynchronize(c->copy_stream));
    return DABX_OK;
}

void *dabx_alloc_pinned(size_t bytes)
{
    void *p = nullptr;
    return hipHostMalloc(&p, bytes, hipHostMallocDefault) == hipSuccess ? p : nullptr;
}

void dabx_free_pinned(void *p)
{
    if (p) (void)hipHostFree(p);
}

void *dabx_ring_ptr(dabx_ctx *c, int s)
{
    if (!valid_stream(c, s)) return nullptr;
    return c->ring_of(s);
}

int dabx_set_write_pos(dabx_ctx *c, int s, int64_t wr)
{
    if (!valid_stream(c, s) || wr < 0) return DABX_E_ARG;
    std::lock_guard<std::mutex> lk(c->mu);
    (void)hipSetDevice(c->cfg.device);              /* a host may drive several contexts (GPUs) from one thread */
    c->streams[s].wr = wr;
    return DABX_OK;
}

int dabx_frames_available(dabx_ctx *c)
{
    if (!c) return DABX_E_ARG;
    std::lock_guard<std::mutex> lk(c->mu);
    (void)hipSetDevice(c->cfg.device);              /* a host may drive several contexts (GPUs) from one thread */
    int best = c->cfg.max_frames;
    for (const auto &sh : c->streams) {
        int n = 0;
        while (n < c->cfg.max_frames && samples_needed(sh, n + 1) <= sh.wr) ++n;
        best = std::min(best, n);
    }
    return best;
}

int dabx_process_async(dabx_ctx *c, int n_frames)
{
    if (!c || n_frames < 1 || n_frames > c->cfg.max_frames) return DABX_E_ARG;
    std::lock_guard<std::mutex> lk(c->mu);
    (void)hipSetDevice(c->cfg.device);              /* a host may drive several contexts (GPUs) from one thread */
    if (c->pending) return DABX_E_ARG;
    for (const auto &sh : c->streams) {
        if (samples_needed(sh, n_frames) > sh.wr) return DABX_E_UNDERRUN;
        if (sh.wr < (1LL << 62) && sh.wr - std::max<int64_t>(0, sh.st.pos - dabx::kTF) > c->cfg.ring_samples) return DABX_E_OVERRUN;
    }
    if (c->work_dirty || c->work_frames != n_frames) {
        int rc = build_work(c, n_frames);
        if (rc) return rc;
    }
    if (c->sf_dirty) {
        int rc = build_superframe_work(c);
        if (rc) return rc;
    }
    const DevCtx d = c->dev();
    const int S = c->cfg.n_streams;
    hipStream_t q = c->stream;
    if (c->copies_queued) {                 // samples pushed from pinned memory must have landed
        HIPCHK(hipEventRecord(c->copy_done, c->copy_stream));
        HIPCHK(hipStreamWaitEvent(q, c->copy_done, 0));
        c->copies_queued = false;
    }
    if (c->d_peak) {                        // the converters' peak over the pushes since the previous step was submitted
        uint32_t *slot = c->d_peak + static_cast<size_t>(c->step_count & 1) * S;
        if (!c->h_peak) HIPCHK(hipHostMalloc(reinterpret_cast<void **>(&c->h_peak), static_cast<size_t>(S) * sizeof(uint32_t)));
        HIPCHK(hipMemcpyAsync(c->h_peak, slot, static_cast<size_t>(S) * sizeof(uint32_t), hipMemcpyDeviceToHost, q));
        HIPCHK(hipMemsetAsync(slot, 0, static_cast<size_t>(S) * sizeof(uint32_t), q));   // next written by the pushes after the NEXT step's submission
    }
    ++c->step_count;
    const bool u8 = c->cfg.fmt == DABX_FMT_U8;
    if (c->timing) HIPCHK(hipEventRecord(c->ev[0], q));
    if (u8) hipLaunchKernelGGL(k_null_search<0>, dim3(S), dim3(256), 0, q, d);
    else hipLaunchKernelGGL(k_null_search<1>, dim3(S), dim3(256), 0, q, d);
    if (u8) hipLaunchKernelGGL(k_sync<0>, dim3(S * n_frames), dim3(256), 0, q, d, n_frames);
    else hipLaunchKernelGGL(k_sync<1>, dim3(S * n_frames), dim3(256), 0, q, d, n_frames);
    if (c->timing) HIPCHK(hipEventRecord(c->ev[1], q));
    // the variant for streams whose sampling clock is off is launched only if the state the last step left says one exists
    // (the host mirror is what the kernels will read: only k_finish and the calls that write both change it)
    bool any_sco = false;
    for (const auto &sh : c->streams) any_sco |= sh.st.locked && (sh.st.slope >= SCO_MIN || sh.st.slope <= -SCO_MIN);
    if (u8) {
        hipLaunchKernelGGL((k_demod<0, false>), dim3(S * n_frames * DEMOD_GROUPS), dim3(256), 0, q, d, n_frames);
        if (any_sco) hipLaunchKernelGGL((k_demod<0, true>), dim3(S * n_frames * DEMOD_GROUPS), dim3(256), 0, q, d, n_frames);
    } else {
        hipLaunchKernelGGL((k_demod<1, false>), dim3(S * n_frames * DEMOD_GROUPS), dim3(256), 0, q, d, n_frames);
        if (any_sco) hipLaunchKernelGGL((k_demod<1, true>), dim3(S * n_frames * DEMOD_GROUPS), dim3(256), 0, q, d, n_frames);
    }
    if (c->timing) HIPCHK(hipEventRecord(c->ev[2], q));
    if (c->n_work) {
        if (c->timing) {                    // the clock monitor runs beside k_viterbi on a stream of its own, until the flag set behind it
            if (!c->d_clock) {
                HIPCHK(hipMalloc(reinterpret_cast<void **>(&c->d_clock), 3 * sizeof(uint64_t)));
                HIPCHK(hipStreamCreateWithFlags(&c->mon_stream, hipStreamNonBlocking));
                HIPCHK(hipEventCreateWithFlags(&c->mon_start, hipEventDisableTiming));
            }
            HIPCHK(hipMemsetAsync(c->d_clock, 0, 3 * sizeof(uint64_t), q));
            HIPCHK(hipEventRecord(c->mon_start, q));
            HIPCHK(hipStreamWaitEvent(c->mon_stream, c->mon_start, 0));
            hipLaunchKernelGGL(k_clock_monitor, dim3(1), dim3(64), 0, c->mon_stream, c->d_clock, reinterpret_cast<const uint32_t *>(c->d_clock + 2),
                               static_cast<uint64_t>(5000000));      // at most 50 ms
        }
        hipLaunchKernelGGL(k_viterbi, dim3((c->n_work + 3) / 4), dim3(256), 0, q, d, c->d_work, c->n_work);
        if (c->timing) HIPCHK(hipMemsetAsync(c->d_clock + 2, 1, sizeof(uint32_t), q));
        hipLaunchKernelGGL(k_viterbi_requeue, dim3(VIT_RQ_BLOCKS), dim3(256), 0, q, d, c->d_work, c->rq_words_per_wave);
    }
    if (c->timing) HIPCHK(hipEventRecord(c->ev[5], q));
    if (!c->sf_subs.empty())
        hipLaunchKernelGGL(k_superframe, dim3(static_cast<unsigned>(c->sf_subs.size())), dim3(SF_THREADS), 0, q, d, c->d_sf_subs, c->d_sf_state,
                           c->d_sf_recs, c->d_sf_data, c->d_gf, n_frames, c->sf_max_rec);
    if (c->timing) HIPCHK(hipEventRecord(c->ev[3], q));
    hipLaunchKernelGGL(k_finish, dim3(S), dim3(256), 0, q, d, n_frames);
    if (c->timing) HIPCHK(hipEventRecord(c->ev[4], q));
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(c->h_state, c->d_state, static_cast<size_t>(S) * sizeof(DevState), hipMemcpyDeviceToHost, q));
    c->res_valid = false;
    if (c->h_res) {
        const size_t F = c->cfg.max_frames, n = n_frames;
        uint8_t *p = c->h_res;
        HIPCHK(hipMemcpyAsync(p, c->d_fib, n * 384, hipMemcpyDeviceToHost, q));
        HIPCHK(hipMemcpyAsync(p + F * 384, c->d_fib_ok, n * 12, hipMemcpyDeviceToHost, q));
        HIPCHK(hipMemcpyAsync(p + F * 396, c->d_sync, n * sizeof(DevSync), hipMemcpyDeviceToHost, q));
        float *sp = reinterpret_cast<float *>(p + F * (396 + sizeof(DevSync)));
        if (c->d_spectrum) HIPCHK(hipMemcpyAsync(sp, c->d_spectrum, 2048 * sizeof(float), hipMemcpyDeviceToHost, q));
        if (c->d_null_spectrum) HIPCHK(hipMemcpyAsync(sp + 2048, c->d_null_spectrum, n * 2048 * sizeof(float), hipMemcpyDeviceToHost, q));
        c->res_valid = true;
    }
    c->pending = true;
    c->last_frames = n_frames;
    return DABX_OK;
}

int dabx_wait(dabx_ctx *c)
{
    if (!c) return DABX_E_ARG;
    std::lock_guard<std::mutex> lk(c->mu);
    (void)hipSetDevice(c->cfg.device);              /* a host may drive several contexts (GPUs) from one thread */
    if (!c->pending) return DABX_OK;
    c->pending = false;                              // also when the synchronisation fails: the context stays usable ...
    const bool mirror = c->res_valid;
    c->res_valid = false;                            // ... but the result mirror holds nothing then: the getters go to the device (and fail there)
    HIPCHK(hipStreamSynchronize(c->stream));
    c->res_valid = mirror;
    for (int s = 0; s < c->cfg.n_streams; ++s) c->streams[s].st = c->h_state[s];
    if (c->timing) {
        HIPCHK(hipEventElapsedTime(&c->last_ms[0], c->ev[0], c->ev[1]));
        HIPCHK(hipEventElapsedTime(&c->last_ms[1], c->ev[1], c->ev[2]));
        HIPCHK(hipEventElapsedTime(&c->last_ms[2], c->ev[2], c->ev[5]));      // k_viterbi alone
        HIPCHK(hipEventElapsedTime(&c->last_ms[3], c->ev[5], c->ev[4]));      // k_superframe (if any) + k_finish
        HIPCHK(hipEventElapsedTime(&c->last_ms[4], c->ev[0], c->ev[4]));
    }
    return DABX_OK;
}

int dabx_process(dabx_ctx *c, int n_frames)
{
    int rc = dabx_process_async(c, n_frames);
    return rc ? rc : dabx_wait(c);
}

#define GETTER_PROLOGUE                                            \
    if (!valid_stream(c, s)) return DABX_E_ARG;                    \
    std::lock_guard<std::mutex> lk(c->mu);                         \
    (void)hipSetDevice(c->cfg.device);                             \
    if (c->pending) return DABX_E_ARG;                             \
    const size_t F = c->cfg.max_frames, n = c->last_frames;        \
    (void)F; (void)n;

int dabx_get_fib(dabx_ctx *c, int s, uint8_t *fib, uint8_t *ok)
{
    GETTER_PROLOGUE
    if (c->res_valid) {
        if (fib) std::memcpy(fib, c->h_res, n * 384);
        if (ok) std::memcpy(ok, c->h_res + F * 384, n * 12);
        return DABX_OK;
    }
    if (fib) HIPCHK(hipMemcpy(fib, c->d_fib + s * F * 384, n * 384, hipMemcpyDeviceToHost));
    if (ok) HIPCHK(hipMemcpy(ok, c->d_fib_ok + s * F * 12, n * 12, hipMemcpyDeviceToHost));
    return DABX_OK;
}

int dabx_get_msc(dabx_ctx *c, int s, uint8_t *msc, uint8_t *valid)
{
    GETTER_PROLOGUE
    const size_t nb = c->streams[s].msc_bytes;
    if (msc && nb)
        HIPCHK(hipMemcpy2D(msc, nb, c->d_msc + s * F * 4 * DABX_MSC_STRIDE, DABX_MSC_STRIDE, nb, n * 4, hipMemcpyDeviceToHost));
    if (valid) HIPCHK(hipMemcpy(valid, c->d_msc_valid + s * F * 4, n * 4, hipMemcpyDeviceToHost));
    return DABX_OK;
}

int dabx_get_sync(dabx_ctx *c, int s, dabx_sync_rec_t *rec)
{
    GETTER_PROLOGUE
    static_assert(sizeof(dabx_sync_rec_t) == sizeof(DevSync), "sync record layout");
    if (!rec) return DABX_E_ARG;
    if (c->res_valid) { std::memcpy(rec, c->h_res + F * 396, n * sizeof(DevSync)); return DABX_OK; }
    HIPCHK(hipMemcpy(rec, c->d_sync + s * F, n * sizeof(DevSync), hipMemcpyDeviceToHost));
    return DABX_OK;
}

int dabx_get_state(dabx_ctx *c, int s, dabx_stream_state_t *st)
{
    GETTER_PROLOGUE
    if (!st) return DABX_E_ARG;
    const auto &sh = c->streams[s];
    *st = {sh.st.pos, sh.st.inc, sh.st.locked, sh.st.cif, sh.st.bad, sh.st.slope, sh.wr};
    return DABX_OK;
}

int dabx_get_fic_soft(dabx_ctx *c, int s, int8_t *fic)
{
    GETTER_PROLOGUE
    if (!fic) return DABX_E_ARG;
    HIPCHK(hipMemcpy(fic, c->d_fic + s * F * DABX_FIC_SOFT_BITS, n * DABX_FIC_SOFT_BITS, hipMemcpyDeviceToHost));
    return DABX_OK;
}

int dabx_get_msc_soft(dabx_ctx *c, int s, int8_t *msc)
{
    GETTER_PROLOGUE
    if (!msc) return DABX_E_ARG;
    // k_demod files residue class q of CIF c in the row of the logical frame it belongs to, c - bitrev4(q) (dabx_dev.h); the CIFs of
    // the last step are cif_end - 4n .. cif_end - 1.  Hand them out CIF by CIF in natural bit order.
    const int64_t cif_end = c->streams[s].st.cif;
    constexpr int seg = DABX_CIF_SOFT_BITS / 16;
    std::vector<int8_t> ringbuf(static_cast<size_t>(c->ti_slots) * DABX_CIF_SOFT_BITS);
    HIPCHK(hipMemcpy(ringbuf.data(), c->d_ti + static_cast<size_t>(s) * c->ti_slots * DABX_CIF_SOFT_BITS, ringbuf.size(), hipMemcpyDeviceToHost));
    static const int delay[16] = {0, 8, 4, 12, 2, 10, 6, 14, 1, 9, 5, 13, 3, 11, 7, 15};
    for (size_t k = 0; k < n * 4; ++k) {
        const int64_t cif = cif_end - static_cast<int64_t>(n * 4) + static_cast<int64_t>(k);
        for (int q = 0; q < 16; ++q) {
            const size_t row = static_cast<size_t>((cif - delay[q]) & (c->ti_slots - 1));
            const int8_t *src = ringbuf.data() + row * DABX_CIF_SOFT_BITS + static_cast<size_t>(q) * seg;
            for (int m = 0; m < seg; ++m) msc[k * DABX_CIF_SOFT_BITS + static_cast<size_t>(16 * m + q)] = src[m];
        }
    }
    return DABX_OK;
}

int dabx_get_fib_counts(dabx_ctx *c, int64_t *ok, int64_t *bad)
{
    if (!c) return DABX_E_ARG;
    std::lock_guard<std::mutex> lk(c->mu);
    (void)hipSetDevice(c->cfg.device);              /* a host may drive several contexts (GPUs) from one thread */
    if (c->pending) return DABX_E_ARG;
    const size_t S = c->cfg.n_streams, F = c->cfg.max_frames, n = c->last_frames;
    std::vector<uint8_t> h(S * F * 12);
    HIPCHK(hipMemcpy(h.data(), c->d_fib_ok, h.size(), hipMemcpyDeviceToHost));
    int64_t good = 0;
    for (size_t s = 0; s < S; ++s)
        for (size_t k = 0; k < n * 12; ++k) good += h[s * F * 12 + k];
    if (ok) *ok = good;
    if (bad) *bad = static_cast<int64_t>(S * n * 12) - good;
    return DABX_OK;
}

int dabx_fft2048(dabx_ctx *c, const float *in, float *out, int n_vec)
{
    if (!c || !in || !out || n_vec < 1) return DABX_E_ARG;
    std::lock_guard<std::mutex> lk(c->mu);
    (void)hipSetDevice(c->cfg.device);              /* a host may drive several contexts (GPUs) from one thread */
    DevTmp t_in, t_out;
    const size_t bytes = static_cast<size_t>(n_vec) * 2048 * sizeof(float2);
    HIPCHK(hipMalloc(&t_in.p, bytes));
    HIPCHK(hipMalloc(&t_out.p, bytes));
    HIPCHK(hipMemcpy(t_in.p, in, bytes, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_fft, dim3(n_vec), dim3(256), 0, c->stream, c->dev().tab, t_in.as<float2>(), t_out.as<float2>());
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(c->stream));
    HIPCHK(hipMemcpy(out, t_out.p, bytes, hipMemcpyDeviceToHost));
    return DABX_OK;
}

int dabx_viterbi(dabx_ctx *c, int kind, int option, int level, int kbps, const int8_t *soft, int n_cw, uint8_t *out)
{
    if (!c || !soft || !out || n_cw < 1) return DABX_E_ARG;
    std::lock_guard<std::mutex> lk(c->mu);
    (void)hipSetDevice(c->cfg.device);              /* a host may drive several contexts (GPUs) from one thread */
    dabx::Profile p = dabx::fic_profile();
    if (kind != 0 && !dabx::any_profile(option, level, kbps, p)) return DABX_E_PROFILE;
    const auto info = dabx::step_gather(p, true);
    const int nsteps = p.steps();
    const size_t words = static_cast<size_t>((nsteps / 24 + 1) * 64);
    if (nsteps % 192 != 6) return DABX_E_PROFILE;            // every DAB codeword: 192 k + 6 trellis steps (24 ms x 8 kbit/s units; the FIC's 768)
    // the soft-bit contract of the decoder (twice the sum of two soft bits must fit a byte): refuse what breaks it
    for (size_t i = 0, n = static_cast<size_t>(n_cw) * p.n_coded; i < n; ++i)
        if (soft[i] > 31 || soft[i] < -31) return DABX_E_ARG;
    DevTmp t_soft, t_info, t_scr, t_out;
    HIPCHK(hipMalloc(&t_soft.p, static_cast<size_t>(n_cw) * p.n_coded + 64));
    HIPCHK(hipMalloc(&t_info.p, info.size() * 4));
    HIPCHK(hipMalloc(&t_scr.p, static_cast<size_t>(n_cw) * words * 4));
    HIPCHK(hipMalloc(&t_out.p, static_cast<size_t>(n_cw) * (p.n_in / 8)));
    HIPCHK(hipMemcpy(t_soft.p, soft, static_cast<size_t>(n_cw) * p.n_coded, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(t_info.p, info.data(), info.size() * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_viterbi_linear, dim3((n_cw + 3) / 4), dim3(256), 0, c->stream, t_soft.as<int8_t>(), p.n_coded, t_info.as<uint32_t>(), nsteps,
                       p.n_in, c->d_prbs, t_scr.as<uint32_t>(), t_out.as<uint8_t>(), n_cw);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(c->stream));
    HIPCHK(hipMemcpy(out, t_out.p, static_cast<size_t>(n_cw) * (p.n_in / 8), hipMemcpyDeviceToHost));
    return p.n_in / 8;
}

int dabx_enable_spectrum(dabx_ctx *c, int mask)
{
    if (!c) return DABX_E_ARG;
    std::lock_guard<std::mutex> lk(c->mu);
    (void)hipSetDevice(c->cfg.device);              /* a host may drive several contexts (GPUs) from one thread */
    if (c->pending) return DABX_E_ARG;
    c->res_valid = false;                            // the result mirror of the last step does not hold what is enabled now
    float **bufs[2] = {&c->d_spectrum, &c->d_null_spectrum};
    for (int k = 0; k < 2; ++k) {
        const bool want = (mask >> k) & 1;
        // the PRS spectrum of a step's last frame; the null-symbol spectrum of every frame
        const size_t bytes = static_cast<size_t>(c->cfg.n_streams) * (k ? c->cfg.max_frames : 1) * 2048 * sizeof(float);
        if (want && !*bufs[k]) {
            HIPCHK(hipMalloc(reinterpret_cast<void **>(bufs[k]), bytes));
            HIPCHK(hipMemset(*bufs[k], 0, bytes));
        } else if (!want && *bufs[k]) {
            (void)hipFree(*bufs[k]);
            *bufs[k] = nullptr;
        }
    }
    return DABX_OK;
}

int dabx_get_spectrum(dabx_ctx *c, int s, float *power)
{
    GETTER_PROLOGUE
    if (!power || !c->d_spectrum) return DABX_E_ARG;
    if (c->res_valid) { std::memcpy(power, c->h_res + F * (396 + sizeof(DevSync)), 2048 * sizeof(float)); return DABX_OK; }
    HIPCHK(hipMemcpy(power, c->d_spectrum + static_cast<size_t>(s) * 2048, 2048 * sizeof(float), hipMemcpyDeviceToHost));
    return DABX_OK;
}

int dabx_get_null_spectrum(dabx_ctx *c, int s, float *power)
{
    GETTER_PROLOGUE
    if (!power || !c->d_null_spectrum || n < 1) return DABX_E_ARG;
    if (c->res_valid) { std::memcpy(power, c->h_res + F * (396 + sizeof(DevSync)) + n * 2048 * sizeof(float), 2048 * sizeof(float)); return DABX_OK; }
    HIPCHK(hipMemcpy(power, c->d_null_spectrum + (static_cast<size_t>(s) * F + n - 1) * 2048, 2048 * sizeof(float), hipMemcpyDeviceToHost));
    return DABX_OK;
}

int dabx_get_null_spectra(dabx_ctx *c, int s, float *power)
{
    GETTER_PROLOGUE
    if (!power || !c->d_null_spectrum || n < 1) return DABX_E_ARG;
    if (c->res_valid) { std::memcpy(power, c->h_res + F * (396 + sizeof(DevSync)) + 2048 * sizeof(float), n * 2048 * sizeof(float)); return DABX_OK; }
    HIPCHK(hipMemcpy(power, c->d_null_spectrum + static_cast<size_t>(s) * F * 2048, n * 2048 * sizeof(float), hipMemcpyDeviceToHost));
    return DABX_OK;
}

int dabx_set_dabplus(dabx_ctx *c, int s, uint64_t mask)
{
    if (!valid_stream(c, s)) return DABX_E_ARG;
    std::lock_guard<std::mutex> lk(c->mu);
    (void)hipSetDevice(c->cfg.device);              /* a host may drive several contexts (GPUs) from one thread */
    if (c->pending) return DABX_E_ARG;
    auto &sh = c->streams[s];
    if (sh.sub.size() < 64 && (mask >> sh.sub.size())) return DABX_E_ARG;
    for (size_t k = 0; k < sh.sub.size(); ++k)
        if (((mask >> k) & 1) && (sh.prof[k].n_in % 192 || sh.prof[k].n_in > 24 * 192)) return DABX_E_PROFILE;
    sh.dabplus = mask;
    c->sf_dirty = true;
    return DABX_OK;
}

static int find_sf(const dabx_ctx *c, int s, int sub)
{
    for (size_t i = 0; i < c->sf_subs.size(); ++i)
        if (c->sf_subs[i].stream == s && c->sf_subs[i].sub == sub) return static_cast<int>(i);
    return -1;
}

int dabx_get_superframes(dabx_ctx *c, int s, int sub, dabx_superframe_t *recs, uint8_t *data, int max)
{
    GETTER_PROLOGUE
    if (!recs || !data || max < 0 || c->sf_dirty) return DABX_E_ARG;
    const int i = find_sf(c, s, sub);
    if (i < 0) return DABX_E_ARG;
    DevSfState st;
    HIPCHK(hipMemcpy(&st, c->d_sf_state + i, sizeof(int32_t) * 4, hipMemcpyDeviceToHost));
    const int cnt = std::min(max, st.n_out);
    if (cnt > 0) {
        static_assert(sizeof(dabx_superframe_t) == sizeof(DevSfRec), "record layouts must match");
        HIPCHK(hipMemcpy(recs, c->d_sf_recs + c->sf_subs[i].rec_off, static_cast<size_t>(cnt) * sizeof(DevSfRec), hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(data, c->d_sf_data + c->sf_subs[i].data_off, static_cast<size_t>(cnt) * 110 * c->sf_subs[i].s, hipMemcpyDeviceToHost));
    }
    return cnt;
}

int dabx_get_superframe_stats(dabx_ctx *c, int s, int sub, uint32_t stats[6])
{
    GETTER_PROLOGUE
    if (!stats || c->sf_dirty) return DABX_E_ARG;
    const int i = find_sf(c, s, sub);
    if (i < 0) return DABX_E_ARG;
    HIPCHK(hipMemcpy(stats, reinterpret_cast<const uint8_t *>(c->d_sf_state + i) + offsetof(DevSfState, stats), 6 * sizeof(uint32_t), hipMemcpyDeviceToHost));
    return DABX_OK;
}

int dabx_get_superframe_pos(dabx_ctx *c, int s, int sub, uint32_t pos[2])
{
    GETTER_PROLOGUE
    if (!pos || c->sf_dirty) return DABX_E_ARG;
    const int i = find_sf(c, s, sub);
    if (i < 0) return DABX_E_ARG;
    DevSfState st;
    HIPCHK(hipMemcpy(&st, c->d_sf_state + i, sizeof(int32_t) * 4, hipMemcpyDeviceToHost));
    pos[0] = st.frames_seen;
    pos[1] = static_cast<uint32_t>(st.carry);
    return DABX_OK;
}

int dabx_rawfile_probe(const uint8_t *head, int n_bytes, dabx_rawfile_info_t *info)
{
    if (!head || n_bytes < 0 || !info) return DABX_E_ARG;
    const rawfile::Info r = rawfile::probe(head, n_bytes);
    *info = {r.has_header, r.fmt, r.data_offset, r.channel_count, r.samplerate, r.frequency_khz};
    return DABX_OK;
}

int dabx_get_requeue_total(dabx_ctx *c, uint64_t *total)
{
    if (!c || !total) return DABX_E_ARG;
    std::lock_guard<std::mutex> lk(c->mu);
    (void)hipSetDevice(c->cfg.device);
    if (c->pending) return DABX_E_ARG;
    uint32_t t = 0;
    if (c->d_requeue) HIPCHK(hipMemcpy(&t, c->d_requeue + 1 + c->work_cap_rq, sizeof t, hipMemcpyDeviceToHost));
    *total = t;
    return DABX_OK;
}

int dabx_enable_timing(dabx_ctx *c, int on)
{
    if (!c) return DABX_E_ARG;
    std::lock_guard<std::mutex> lk(c->mu);
    if (c->pending) return DABX_E_ARG;
    c->timing = on != 0;
    return DABX_OK;
}

int dabx_last_shader_clock(dabx_ctx *c, double *ghz)
{
    if (!c || !ghz) return DABX_E_ARG;
    std::lock_guard<std::mutex> lk(c->mu);
    (void)hipSetDevice(c->cfg.device);
    if (c->pending || !c->timing || !c->d_clock) return DABX_E_ARG;
    HIPCHK(hipStreamSynchronize(c->mon_stream));
    uint64_t h[2] = {0, 0};
    HIPCHK(hipMemcpy(h, c->d_clock, sizeof h, hipMemcpyDeviceToHost));
    // s_memrealtime counts at 100 MHz; a monitor that was scheduled only after the flag had been set measured nothing
    *ghz = h[1] > 1000 ? static_cast<double>(h[0]) / static_cast<double>(h[1]) * 0.1 : 0.0;
    return DABX_OK;
}

int dabx_last_timing(dabx_ctx *c, float ms[5])
{
    if (!c || !ms) return DABX_E_ARG;
    std::memcpy(ms, c->last_ms, sizeof c->last_ms);
    return DABX_OK;
}

}  // extern "C"
