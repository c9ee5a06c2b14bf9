// dabx_dev.h — device-side data layout shared by the kernels and the host API.
//
// HBM layout per context (S = streams, F = max_frames, all arrays stream-major so
// one workgroup's traffic is contiguous):
//   ring      [S][ring_samples]            raw IQ, u8 pairs or s16 pairs (the reference's
//                                           raw-file formats, src/input/rawfileinput.cpp:640-713)
//   fic_soft  [S][F][9216]        int8      FIC soft bits, frequency de-interleaved
//   ti        [S][ti_slots][55296] int8     MSC soft bits, time DE-interleaved: one row per LOGICAL
//                                           frame (>= 15 + 4 F rows).  Inside a row bit b sits at
//                                           (b & 15) * 3456 + (b >> 4) (residue-major).  k_demod files
//                                           residue class q of CIF c in row (c - bitrev4(q)) & (ti_slots-1),
//                                           192 contiguous bytes per symbol and class, so a codeword's
//                                           bits sit at offsets that depend on its profile only
//                                           (dabx_spec.hpp: step_gather)
//   fib       [S][F][12][32]                decoded FIBs;  fib_ok [S][F][12]
//   msc       [S][F][4][msc_stride]         decoded sub-channel bytes;  msc_valid [S][F][4]
//   sync      [S][F]                        per-frame synchronisation records
#pragma once
#include <stdint.h>
#include <hip/hip_runtime.h>

struct DevState {
    int64_t pos;        // estimated start of the next frame's null symbol
    int64_t cif;        // CIFs demodulated since lock (row of logical frame r = r & (ti_slots-1); frame cif - 15 is the newest complete one)
    int32_t inc;        // carrier offset, 2^-32 turn per sample
    int32_t locked;
    int32_t bad;        // consecutive frames without PRS
    int32_t acq_fail;   // set by k_null_search for the current step only
    int32_t slope;      // tracked sampling-clock drift of the recording: samples per frame, Q16
    int32_t pad;
};

struct DevSync {        // same layout as dabx_sync_rec_t
    int64_t t_sym0;
    int32_t inc, flags, peak_idx, m_int;
    float peak, total;
    int64_t cp_re, cp_im;
    int64_t e_null, e_sig;
};

struct DevWork {        // one Viterbi codeword = one wave
    int32_t stream;
    int16_t frame;
    int8_t c;           // FIC codeword 0..3 or CIF 0..3
    int8_t sub;         // -1 = FIC
    uint32_t nsteps;
};

struct DevSub {
    int32_t start_bit;  // first soft bit of the sub-channel inside a CIF row
    int32_t nsteps, n_in;
    int32_t info_off;   // offset of the profile's gather map in stepinfo (words)
    int32_t out_off;    // byte offset inside the CIF's output record
};

// ---- DAB+ audio super frames (dabx_superframe.hip)
struct DevSfSub {           // one sub-channel that carries DAB+ audio
    int32_t stream, sub;    // sub: index into the stream's sub-channel list
    int32_t s;              // kbps / 8 = number of RS code words per super frame
    int32_t frame_bytes;    // 3 kbps: the sub-channel's bytes per logical frame (CIF)
    int32_t msc_off;        // its byte offset inside a CIF record
    uint32_t rec_off;       // first record slot of this sub-channel
    uint32_t data_off;      // byte offset of its super frame data
    uint32_t pad;
};

struct DevSfRec {           // = dabx_superframe_t (include/dabx.h) = dab_sf_rec_t (oracle/dab_plus.c)
    uint32_t first_frame;
    uint8_t header, num_aus, au_valid, au_ok;
    uint16_t au_start[8];
    uint16_t rs_corrected, rs_failed;
    uint32_t pad;
};

struct DevSfState {         // persists from step to step
    int32_t carry, synced;
    uint32_t frames_seen;
    int32_t n_out;          // records written by the last step
    uint32_t stats[6];      // super frames, AUs good, AUs bad, bytes corrected, code words uncorrectable, sync losses
    uint8_t buf[2304 + 24]; // up to four carried logical frames of 3 * 192 bytes
};

struct DevTables {
    const float2 *W;            // [2048] exp(-j 2 pi k / 2048)
    const float2 *nco_hi;       // [2048] exp(+j 2 pi k / 2^11)
    const float2 *nco_lo;       // [2048] exp(+j 2 pi k / 2^22)
    const int16_t *bin_of_pos;  // [2048] FFT output placement
    const int16_t *n_of_bin;    // [2048] frequency de-interleaver (-1 unused)
    const int8_t *prs_q;        // [2048] PRS quadrant (-1 unused)
    const int8_t *prs_dq;       // [2048] PRS quadrant difference k vs k-1
    const int16_t *cfo_car;     // [1534] carriers usable for the differential CFO search
};

struct DevCtx {
    DevTables tab;
    DevState *state;
    DevSync *sync;
    const uint8_t *ring;
    int8_t *fic_soft;
    int8_t *ti;
    uint8_t *fib, *fib_ok, *msc, *msc_valid;
    const DevSub *sub;          // [S][64]
    const uint32_t *stepinfo;   // pooled gather maps (step_gather: five arrays of nsteps + 256 words each)
    const uint32_t *prbs;       // energy dispersal, bit 31-j of word h = PRBS bit 32 h + j
    uint32_t *dec_scratch;      // k_viterbi_requeue: per wave the decision words of one codeword, 64 words per 24 trellis steps
    uint32_t *requeue;          // [0] number of codewords k_viterbi gave up on (survivors did not merge), [1..cap] their work
                                // indices, [1 + cap] running total over the steps
    float *spectrum;            // [S][2048] |FFT|^2 of the last frame's PRS window, natural bin order; may be null
    float *null_spectrum;       // [S][F][2048] same for 2048 samples in the middle of every frame's null symbol (noise level, TII); may be null
    int64_t ring_len;           // samples
    size_t ring_bytes;          // bytes per stream
    int32_t n_streams, max_frames, ti_slots, msc_stride, fic_info_off, requeue_cap;
};
