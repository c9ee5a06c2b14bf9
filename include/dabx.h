/*
 * dabx.h — batch C ABI of the MI355X DAB Mode-I PHY decode library.
 *
 * The reference's dabsdr library decodes ONE ensemble per handle and pulls
 * samples through a context-free callback (reference:
 * lib/linux_x86_64/dabsdr.h:387 `dabsdrInputFunc_t`, :397-429 the 24 entry
 * points; caller src/radiocontrol.cpp:81-93; producer src/input/inputdevice.cpp:70-131).
 * That shape cannot feed a GPU, so the hot path is exposed here as a batch
 * interface over many independent raw-IQ streams; include/dabsdr_amd.h keeps the
 * reference's 24 entry points as an adapter over stream 0 of a one-stream context.
 *
 * Every entry point takes plain pointers and sizes.  Device memory is owned by
 * the context; host pointers are copied in/out on the context's HIP stream.
 * All functions return 0 on success or a negative DABX_E_* code.
 */
#ifndef DABX_H
#define DABX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DABX_API __attribute__((visibility("default")))

/* Mode I constants visible to callers (ETSI EN 300 401 §14.2) */
#define DABX_FRAME_SAMPLES   196608     /* 96 ms at 2.048 Msps                          */
#define DABX_FIBS_PER_FRAME  12
#define DABX_FIB_BYTES       32
#define DABX_FIC_SOFT_BITS   9216       /* 3 OFDM symbols                               */
#define DABX_CIF_SOFT_BITS   55296      /* 864 capacity units x 64 bit                  */
#define DABX_CIFS_PER_FRAME  4
#define DABX_MSC_STRIDE      6912       /* bytes reserved per CIF in the MSC output     */
#define DABX_MAX_SUBCH       64

enum {
    DABX_OK = 0,
    DABX_E_ARG = -1,        /* bad argument                                          */
    DABX_E_NODEV = -2,      /* no HIP device / HIP runtime error                     */
    DABX_E_NOMEM = -3,
    DABX_E_UNDERRUN = -4,   /* a stream does not hold enough samples for the request */
    DABX_E_OVERRUN = -5,    /* push would overwrite samples not yet consumed         */
    DABX_E_PROFILE = -6,    /* protection profile not in EN 300 401 §11.3            */
};

/* sample formats of the reference's raw-file input
 * (reference: src/input/rawfileinput.cpp:640-713: u8 -> float(v-128), s16 -> float(v)) */
enum { DABX_FMT_U8 = 0, DABX_FMT_S16 = 1 };

typedef struct {
    int32_t n_streams;            /* independent ensembles decoded side by side           */
    int32_t fmt;                  /* DABX_FMT_*                                           */
    int64_t ring_samples;         /* per-stream IQ ring capacity in complex samples: (max_frames + 2) frames .. 2^30 */
    int32_t max_frames;           /* largest n_frames dabx_process will be called with    */
    int32_t device;               /* HIP device ordinal                                   */
} dabx_config_t;

/* per-frame synchronisation record (one per stream and frame of the last step) */
typedef struct {
    int64_t t_sym0;               /* absolute sample index of the PRS FFT window          */
    int32_t inc;                  /* carrier offset, 2^-32 turn per sample                */
    int32_t flags;                /* bit0: PRS found (frame ok); bit1: wide search used   */
    int32_t peak_idx;             /* impulse-response peak position                       */
    int32_t m_int;                /* integer carrier offset found by the wide search      */
    float   peak, total;          /* |h|^2 peak and sum                                   */
    int64_t cp_re, cp_im;         /* guard-interval correlation                           */
    int64_t e_null, e_sig;        /* energy of 2048 samples inside the null symbol / PRS  */
} dabx_sync_rec_t;

typedef struct {
    int64_t pos;                  /* estimated start of the next frame's null symbol      */
    int32_t inc;                  /* carrier offset estimate                              */
    int32_t locked;               /* 0 searching, 1 tracking                              */
    int64_t cif;                  /* CIFs demodulated since lock                          */
    int32_t bad;                  /* consecutive frames without PRS                       */
    int32_t slope_q16;            /* tracked sampling-clock drift of the recording: samples per frame, Q16 */
    int64_t wr;                   /* samples pushed so far (host side)                    */
} dabx_stream_state_t;

/* one sub-channel of the MSC, as FIG 0/1 announces it (ETSI EN 300 401 §6.2.1, §11.3):
 *   option 0 / 1: EEP set A / B, level 1..4, kbps = bit rate          (long form)
 *   option 2:     UEP, level = the 6-bit table index 0..63, kbps unused (short form) */
typedef struct {
    int32_t start_cu;
    int32_t option;
    int32_t level;
    int32_t kbps;
} dabx_subch_t;

typedef struct dabx_ctx dabx_ctx;

DABX_API int  dabx_create(const dabx_config_t *cfg, dabx_ctx **out);
DABX_API void dabx_destroy(dabx_ctx *ctx);
DABX_API const char *dabx_strerror(int code);

/* MSC configuration of one stream (what FIG 0/1 announces).  Returns the number
 * of decoded bytes per CIF (sub-channels concatenated in the given order). */
DABX_API int dabx_set_subchannels(dabx_ctx *ctx, int stream, int n, const dabx_subch_t *sub);

/* Append n complex samples (interleaved I,Q in the context's format) to a
 * stream's ring.  Replaces the producer side of the reference's global FIFO
 * (reference: src/input/inputdevice.cpp:30, src/input/rawfileinput.cpp:602-747).
 *   src_kind DABX_SRC_HOST    pageable host memory; returns when the buffer may be reused
 *            DABX_SRC_DEVICE  device memory of the context's GPU
 *            DABX_SRC_PINNED  host memory from dabx_alloc_pinned(): the copy is queued on the context's
 *                             copy stream and overlaps with a running dabx_process_async(); the buffer
 *                             must stay untouched until the next dabx_process / dabx_process_async
 *                             call has returned and its dabx_wait() has completed */
#define DABX_SRC_HOST   0
#define DABX_SRC_DEVICE 1
#define DABX_SRC_PINNED 2
DABX_API int dabx_push(dabx_ctx *ctx, int stream, const void *src, int64_t n, int src_kind);

/* The same for every stream in one call: stream s reads n samples at src + s * src_stride_bytes.  When all
 * streams stand at the same write position (a batch fed in lock step) this is one strided copy instead of
 * n_streams copies. */
DABX_API int dabx_push_all(dabx_ctx *ctx, const void *src, size_t src_stride_bytes, int64_t n, int src_kind);

/* Samples at another rate: the converters SDR devices run in front of the reference's FIFO (reference:
 * src/input/inputdevicesrc.h:78-150, selection src/input/inputdevicesrc.cpp:33-47), on the GPU, writing into the
 * stream's ring (the context must have been created with DABX_FMT_S16).
 *   in_rate_hz 2048000: conversion only; 4096000: 43-tap half-band decimator (n even); any other rate: transposed
 *   Farrow resampler.  src: host memory, n complex samples, interleaved I,Q as int16 (DABX_FMT_S16) or float
 *   (DABX_FMT_F32); every output sample is multiplied by `gain` and rounded to int16 (1.0 for int16 input; e.g.
 *   8192 for floats in +-1).  Filter state is kept per stream from call to call.  Returns the number of samples
 *   appended to the ring, or a negative error code.  (The reference's signal-level output: dabx_enable_level.) */
#define DABX_FMT_F32 2
DABX_API int64_t dabx_push_resampled(dabx_ctx *ctx, int stream, const void *src, int64_t n, int src_fmt, double in_rate_hz, float gain);
/* The same with the source kinds of dabx_push.  DABX_SRC_PINNED: the staging copy and the converter kernels are queued on the
 * context's copy stream and the call returns at once (the buffer must stay untouched until the next decode step has been
 * waited for, or dabx_flush_copies); DABX_SRC_DEVICE: src is device memory of the context's GPU, converted in place.
 * This is how the legacy adapter hands the float samples of the reference's input callback (dabsdr.h:387) to the GPU
 * without touching them on the host: in_rate_hz = 2048000, src_fmt = DABX_FMT_F32, gain = a power of two. */
DABX_API int64_t dabx_push_resampled_from(dabx_ctx *ctx, int stream, const void *src, int64_t n, int src_fmt, double in_rate_hz, float gain, int src_kind);
/* The converters' signal-level output, what the reference's SDR inputs steer their gain with (src/input/inputdevicesrc.h:60-75
 * signalLevel(); the detector: inputdevicesrc.cpp:167-173, :282-292, :330-341 — fast attack 50 us, slow release 50 ms on |x|^2 of the
 * input samples).  A serial binary32 recursion, run by one wave per push on the GPU (a few ms per frame of input, on the stream of
 * the push): off by default, enabled per stream; enabling or disabling starts from level 0.  dabx_get_level waits for the pushes
 * queued so far.  Bit-exact against the restatement in oracle/dab_src.c. */
DABX_API int dabx_enable_level(dabx_ctx *ctx, int stream, int on);
DABX_API int dabx_get_level(dabx_ctx *ctx, int stream, float *level);
/* Largest |I| or |Q| (int16 units, after the gain) the converters wrote into the stream's ring between the submission of the
 * step before the last one and the submission of the last one; valid after dabx_wait.  (Input level for a caller's gain control.) */
DABX_API int dabx_get_input_peak(dabx_ctx *ctx, int stream, int32_t *peak);
/* Test/diagnostic accessor: copies n complex samples starting at absolute sample index `from` out of a stream's ring. */
DABX_API int dabx_read_ring(dabx_ctx *ctx, int stream, int64_t from, int64_t n, void *dst);

/* Wait until every DABX_SRC_PINNED copy queued so far has left its source buffer (for a producer that wants to refill the
 * buffer without a decode step in between). */
DABX_API int dabx_flush_copies(dabx_ctx *ctx);

/* Page-locked host buffers for DABX_SRC_PINNED (a file reader fills them directly, as RawFileWorker
 * fills its read buffer: reference src/input/rawfileinput.cpp:640-713). */
DABX_API void *dabx_alloc_pinned(size_t bytes);
DABX_API void  dabx_free_pinned(void *p);

/* Device address of a stream's ring and a way to declare samples resident
 * without copying (zero-copy producers, and the benchmark's periodic signal).  The ring is followed by a mirror of its
 * first DABX_RING_MIRROR samples (so that no FFT window wraps): dabx_push and dabx_push_resampled maintain it; a producer
 * that writes through this pointer writes sample i < DABX_RING_MIRROR of the ring also at index ring_samples + i. */
#define DABX_RING_MIRROR 4096
DABX_API void *dabx_ring_ptr(dabx_ctx *ctx, int stream);
DABX_API int   dabx_set_write_pos(dabx_ctx *ctx, int stream, int64_t wr);

/* Decode n_frames transmission frames on every stream: sync, FFT + DQPSK demap,
 * de-interleave, Viterbi (FIC and all configured sub-channels), FIB CRC.
 * This is the hot path that replaces the dabsdr worker loop
 * (reference: SURVEY.md §3.3; lib/linux_x86_64/dabsdr.h:397 `dabsdr()`).
 * dabx_process returns when the results are complete; the _async form only
 * enqueues on the context's HIP stream and dabx_wait completes it. */
DABX_API int dabx_process(dabx_ctx *ctx, int n_frames);
DABX_API int dabx_process_async(dabx_ctx *ctx, int n_frames);
DABX_API int dabx_wait(dabx_ctx *ctx);
/* frames that can be processed right now on every stream (min over streams) */
DABX_API int dabx_frames_available(dabx_ctx *ctx);

/* Results of the last step, copied to host memory.
 *   fib     [n_frames][12][32]   decoded FIBs incl. CRC     (reference: FIB -> FIG database, dabsdr.h:158 fibErrorCntr)
 *   fib_ok  [n_frames][12]       1 where the FIB CRC matched
 *   msc     [n_frames][4][bytes_per_cif]  decoded sub-channel bytes   (reference: audio/data callbacks, dabsdr.h:71-96)
 *   valid   [n_frames][4]        0 while the 16-CIF time de-interleaver is filling */
DABX_API int dabx_get_fib(dabx_ctx *ctx, int stream, uint8_t *fib, uint8_t *fib_ok);
DABX_API int dabx_get_msc(dabx_ctx *ctx, int stream, uint8_t *msc, uint8_t *valid);
DABX_API int dabx_get_sync(dabx_ctx *ctx, int stream, dabx_sync_rec_t *rec);
DABX_API int dabx_get_state(dabx_ctx *ctx, int stream, dabx_stream_state_t *st);
/* parity taps: soft bits of the last step, fic [n_frames][9216], msc [n_frames][4][55296] */
DABX_API int dabx_get_fic_soft(dabx_ctx *ctx, int stream, int8_t *fic);
DABX_API int dabx_get_msc_soft(dabx_ctx *ctx, int stream, int8_t *msc);
/* sum over streams and frames of the last step: FIBs with good / bad CRC */
DABX_API int dabx_get_fib_counts(dabx_ctx *ctx, int64_t *ok, int64_t *bad);

/* Stage-level entry points (BASELINE config "FFT + DQPSK demap kernel only",
 * and unit parity tests).  Host pointers.
 *   dabx_fft2048: n_vec vectors of 2048 complex float (re,im interleaved), natural bin order out
 *   dabx_viterbi: n_cw punctured codewords of one profile, soft bits linear in memory, |soft| <= 31, else DABX_E_ARG (the range the
 *                 demapper produces: the decoder adds the two soft bits of the repeated generator in one byte);
 *                 kind 0 = FIC codeword (2304 soft -> 96 bytes), 1 = EEP(option, level, kbps) */
DABX_API int dabx_fft2048(dabx_ctx *ctx, const float *in, float *out, int n_vec);
DABX_API int dabx_viterbi(dabx_ctx *ctx, int kind, int option, int level, int kbps,
                          const int8_t *soft, int n_cw, uint8_t *out);

/* Spectra of the last decoded frame, 2048 bins of linear power of the un-normalised FFT in natural
 * order (bin 0 = DC, 1024.. = negative frequencies): the payload of the reference's spectrum
 * callback (dabsdr.h:365-370, :393; consumer src/signalbackend.cpp:383-429).
 *   mask bit 0: phase-reference symbol (DABSDR_SPECT_SIGNAL)
 *   mask bit 1: null symbol (DABSDR_SPECT_NULL; input of the TII detector, dabsdr.h:372-384) */
DABX_API int dabx_enable_spectrum(dabx_ctx *ctx, int mask);
DABX_API int dabx_get_spectrum(dabx_ctx *ctx, int stream, float *power);
DABX_API int dabx_get_null_spectrum(dabx_ctx *ctx, int stream, float *power);
/* the null-symbol spectrum of EVERY frame of the last step: power [n_frames][2048] (dabx_get_null_spectrum = the last one) */
DABX_API int dabx_get_null_spectra(dabx_ctx *ctx, int stream, float *power);

/* DAB+ audio super frames (ETSI TS 102 563): fire code synchronisation, RS(120,110) correction and access
 * unit CRCs on the GPU for the sub-channels flagged here (bit k of mask = k-th entry of the list given to
 * dabx_set_subchannels; what FIG 0/2 announces as ASCTy 63).  Replaces the super frame layer of the
 * reference's library; a record is what its audio callback delivers per access unit (dabsdr.h:47-78). */
typedef struct {
    uint32_t first_frame;      /* index of the super frame's first logical frame among the sub-channel's valid frames */
    uint8_t  header;           /* dabsdrAudioFrameHeader_t bits: dac_rate<<6 | sbr<<5 | aac_channel_mode<<4 | ps<<3 | mpeg_surround */
    uint8_t  num_aus;
    uint8_t  au_valid;         /* bit i: access unit i has sane bounds */
    uint8_t  au_ok;            /* bit i: ... and a good CRC            */
    uint16_t au_start[8];      /* byte offsets into the 110 * (kbps/8) data bytes; au_start[num_aus] = their count; AU i is
                                  [au_start[i], au_start[i+1] - 2), followed by its 2 CRC bytes */
    uint16_t rs_corrected;     /* bytes corrected by the RS decoder in this super frame */
    uint16_t rs_failed;        /* code words it could not correct                       */
    uint32_t reserved;
} dabx_superframe_t;           /* 32 bytes */

DABX_API int dabx_set_dabplus(dabx_ctx *ctx, int stream, uint64_t mask);
/* Super frames completed by the last step for sub-channel `sub` of `stream`: up to max records and
 * max * 110 * (kbps/8) data bytes (RS-corrected, parity dropped).  Returns their number. */
DABX_API int dabx_get_superframes(dabx_ctx *ctx, int stream, int sub, dabx_superframe_t *recs, uint8_t *data, int max);
/* running totals: super frames, AUs good, AUs bad, bytes corrected, code words uncorrectable, sync losses */
DABX_API int dabx_get_superframe_stats(dabx_ctx *ctx, int stream, int sub, uint32_t stats[6]);
/* pos[0]: valid logical frames (CIFs) of the sub-channel seen so far, the last step's included — record r of that step was
 * completed by the frame with index r.first_frame + 4, and the step's last CIF has index pos[0] - 1; pos[1]: frames carried
 * over (< 5) into the next step.  Lets a caller place the records of a multi-frame step in time. */
DABX_API int dabx_get_superframe_pos(dabx_ctx *ctx, int stream, int sub, uint32_t pos[2]);

/* Raw-file front end: the reference's RawFileInput accepts headerless `.raw` files and `.uff`
 * files whose first 2048 bytes hold a zero-padded XML description (reference:
 * src/input/rawfileinput.cpp:90-134, writer src/input/inputdevicerecorder.cpp:195-263).
 * dabx_rawfile_probe inspects the first bytes of a file and tells where the samples start and in
 * which DABX_FMT_* they are; fmt = -1 and data_offset = 0 for a headerless file. */
typedef struct {
    int32_t has_header;
    int32_t fmt;
    int64_t data_offset;
    int64_t channel_count;
    int32_t samplerate;
    int32_t frequency_khz;
} dabx_rawfile_info_t;
DABX_API int dabx_rawfile_probe(const uint8_t *head, int n_bytes, dabx_rawfile_info_t *info);

/* Diagnostics: codewords (all streams, all steps so far) whose survivor paths did not merge within 192 trellis steps, so that
 * the Viterbi kernel had to decode them a second time with their decisions spilled to device memory (erased or tied input:
 * silence, a sub-channel that is not there).  Zero on any live signal. */
DABX_API int dabx_get_requeue_total(dabx_ctx *ctx, uint64_t *total);

/* Timing of the last step, HIP events on the context's stream (ms):
 * [0] acquire+sync, [1] FFT/demap, [2] Viterbi, [3] super frames + CRC/state, [4] whole step */
DABX_API int dabx_last_timing(dabx_ctx *ctx, float ms[5]);
/* With timing enabled: the shader clock (GHz) the Viterbi kernel of the last step ran at, measured inside the kernel (shader-cycle
 * counter against the constant 100 MHz counter, over sampled codewords).  The GPU lowers its clock under this load. */
DABX_API int dabx_last_shader_clock(dabx_ctx *ctx, double *ghz);
DABX_API int dabx_enable_timing(dabx_ctx *ctx, int on);

#ifdef __cplusplus
}
#endif
#endif
