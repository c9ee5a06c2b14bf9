"""ctypes view of include/dabx.h (the batch C ABI of libdabsdr_amd.so)."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

FRAME_SAMPLES = 196608
FIC_SOFT_BITS = 9216
CIF_SOFT_BITS = 55296
MSC_STRIDE = 6912

SYNC_DTYPE = np.dtype([("t_sym0", "<i8"), ("inc", "<i4"), ("flags", "<i4"), ("peak_idx", "<i4"),
                       ("m_int", "<i4"), ("peak", "<f4"), ("total", "<f4"), ("cp_re", "<i8"), ("cp_im", "<i8"),
                       ("e_null", "<i8"), ("e_sig", "<i8")])

# every symbol include/dabx.h declares; tests check that the library exports them all
DABX_SYMBOLS = [
    "dabx_create", "dabx_destroy", "dabx_strerror", "dabx_set_subchannels", "dabx_push", "dabx_push_all", "dabx_set_dabplus", "dabx_get_superframes", "dabx_get_superframe_stats", "dabx_alloc_pinned", "dabx_free_pinned", "dabx_ring_ptr",
    "dabx_push_resampled", "dabx_push_resampled_from", "dabx_get_input_peak", "dabx_get_superframe_pos", "dabx_read_ring", "dabx_flush_copies",
    "dabx_set_write_pos", "dabx_process", "dabx_process_async", "dabx_wait", "dabx_frames_available",
    "dabx_get_fib", "dabx_get_msc", "dabx_get_sync", "dabx_get_state", "dabx_get_fic_soft", "dabx_get_msc_soft",
    "dabx_get_fib_counts", "dabx_fft2048", "dabx_viterbi", "dabx_last_timing", "dabx_enable_timing", "dabx_rawfile_probe", "dabx_enable_spectrum", "dabx_get_spectrum", "dabx_get_null_spectrum", "dabx_get_null_spectra", "dabx_get_requeue_total", "dabx_last_shader_clock", "dabx_enable_level", "dabx_get_level",
]


class DabxError(RuntimeError):
    pass


class Config(C.Structure):
    _fields_ = [("n_streams", C.c_int32), ("fmt", C.c_int32), ("ring_samples", C.c_int64),
                ("max_frames", C.c_int32), ("device", C.c_int32)]


class SubCh(C.Structure):
    _fields_ = [("start_cu", C.c_int32), ("option", C.c_int32), ("level", C.c_int32), ("kbps", C.c_int32)]


class StreamState(C.Structure):
    _fields_ = [("pos", C.c_int64), ("inc", C.c_int32), ("locked", C.c_int32), ("cif", C.c_int64),
                ("bad", C.c_int32), ("slope_q16", C.c_int32), ("wr", C.c_int64)]


class RawFileInfo(C.Structure):
    _fields_ = [("has_header", C.c_int32), ("fmt", C.c_int32), ("data_offset", C.c_int64), ("channel_count", C.c_int64),
                ("samplerate", C.c_int32), ("frequency_khz", C.c_int32)]


def rawfile_probe(head):
    """Inspect the first bytes of a .raw/.uff file (reference: rawfileinput.cpp:90-134)."""
    L = load_library()
    head = np.frombuffer(bytes(head[:4096]), dtype=np.uint8)
    info = RawFileInfo()
    L.dabx_rawfile_probe.argtypes = [C.c_void_p, C.c_int, C.POINTER(RawFileInfo)]
    _chk(L.dabx_rawfile_probe(head.ctypes.data, head.size, C.byref(info)))
    return dict(has_header=bool(info.has_header), fmt=info.fmt, data_offset=info.data_offset, channel_count=info.channel_count,
                samplerate=info.samplerate, frequency_khz=info.frequency_khz)


def library_path():
    return os.environ.get("DABX_LIBRARY", os.path.join(_HERE, "libdabsdr_amd.so"))


def build_library():
    """Compile csrc/ for gfx950 with hipcc (works without a GPU)."""
    subprocess.check_call(["make", "-s", "-C", os.path.join(_HERE, "csrc")])


def load_library():
    """Load libdabsdr_amd.so.  There is no fallback: a missing library is an error."""
    global _LIB
    if _LIB is not None:
        return _LIB
    path = library_path()
    if not os.path.exists(path):
        raise DabxError(f"{path} is missing: build it with `make -C abracadabra_amd/csrc` "
                        "(or __graft_entry__.build()); there is no CPU fallback")
    L = C.CDLL(path)
    L.dabx_strerror.restype = C.c_char_p
    L.dabx_strerror.argtypes = [C.c_int]
    L.dabx_create.argtypes = [C.POINTER(Config), C.POINTER(C.c_void_p)]
    L.dabx_destroy.argtypes = [C.c_void_p]
    L.dabx_destroy.restype = None
    L.dabx_set_subchannels.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    L.dabx_push.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int64, C.c_int]
    L.dabx_ring_ptr.restype = C.c_void_p
    L.dabx_ring_ptr.argtypes = [C.c_void_p, C.c_int]
    L.dabx_set_write_pos.argtypes = [C.c_void_p, C.c_int, C.c_int64]
    for name in ("dabx_process", "dabx_process_async"):
        getattr(L, name).argtypes = [C.c_void_p, C.c_int]
    L.dabx_wait.argtypes = [C.c_void_p]
    L.dabx_frames_available.argtypes = [C.c_void_p]
    for name in ("dabx_get_fib", "dabx_get_msc"):
        getattr(L, name).argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    for name in ("dabx_get_sync", "dabx_get_state", "dabx_get_fic_soft", "dabx_get_msc_soft"):
        getattr(L, name).argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    L.dabx_get_fib_counts.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    L.dabx_fft2048.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
    L.dabx_viterbi.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
    L.dabx_last_timing.argtypes = [C.c_void_p, C.c_void_p]
    L.dabx_enable_timing.argtypes = [C.c_void_p, C.c_int]
    _LIB = L
    return L


def _chk(rc):
    if rc < 0:
        raise DabxError(f"dabx error {rc}: {load_library().dabx_strerror(rc).decode()}")
    return rc


SF_REC_DTYPE = np.dtype([("first_frame", "<u4"), ("header", "u1"), ("num_aus", "u1"), ("au_valid", "u1"), ("au_ok", "u1"),
                         ("au_start", "<u2", (8,)), ("rs_corrected", "<u2"), ("rs_failed", "<u2"), ("reserved", "<u4")])   # dabx_superframe_t


class Context:
    """A batch decoder over n_streams independent raw-IQ streams on one GPU."""

    def __init__(self, n_streams, fmt=0, ring_frames=16, max_frames=4, device=0):
        self.L = load_library()
        self.n_streams, self.fmt, self.max_frames = n_streams, fmt, max_frames
        self.ring_samples = ring_frames * FRAME_SAMPLES
        cfg = Config(n_streams, fmt, self.ring_samples, max_frames, device)
        h = C.c_void_p()
        _chk(self.L.dabx_create(C.byref(cfg), C.byref(h)))
        self.h = h
        self.msc_bytes = [0] * n_streams
        self.last_frames = 0

    def close(self):
        if getattr(self, "h", None):
            self.L.dabx_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def set_subchannels(self, stream, subch):
        arr = (SubCh * max(len(subch), 1))()
        for i, s in enumerate(subch):
            arr[i] = SubCh(*[int(v) for v in s])
        self.msc_bytes[stream] = _chk(self.L.dabx_set_subchannels(self.h, stream, len(subch), arr))
        return self.msc_bytes[stream]

    def push(self, stream, iq):
        iq = np.ascontiguousarray(iq)
        _chk(self.L.dabx_push(self.h, stream, iq.ctypes.data, iq.size // 2, 0))

    def push_pinned(self, stream, host_ptr, n_samples):
        """host_ptr: address inside a buffer from alloc_pinned(); asynchronous (see include/dabx.h)"""
        _chk(self.L.dabx_push(self.h, stream, C.c_void_p(host_ptr), n_samples, 2))

    def set_dabplus(self, stream, mask):
        """bit k of mask: the k-th sub-channel of the stream carries DAB+ audio -> super frames are decoded on the GPU"""
        self.L.dabx_set_dabplus.argtypes = [C.c_void_p, C.c_int, C.c_uint64]
        _chk(self.L.dabx_set_dabplus(self.h, stream, mask))

    def superframes(self, stream, sub, kbps, max_rec=64):
        """records (SF_REC_DTYPE) and RS-corrected data [n, 110 * kbps/8] of the super frames the last step completed"""
        recs = np.zeros(max_rec, dtype=SF_REC_DTYPE)
        data = np.zeros((max_rec, 110 * (kbps // 8)), dtype=np.uint8)
        self.L.dabx_get_superframes.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int]
        n = _chk(self.L.dabx_get_superframes(self.h, stream, sub, recs.ctypes.data, data.ctypes.data, max_rec))
        return recs[:n].copy(), data[:n].copy()

    def superframe_stats(self, stream, sub):
        st = np.zeros(6, dtype=np.uint32)
        self.L.dabx_get_superframe_stats.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        _chk(self.L.dabx_get_superframe_stats(self.h, stream, sub, st.ctypes.data))
        return dict(zip(("superframes", "au_ok", "au_crc_err", "rs_corrected", "rs_uncorrectable", "sync_loss"), st.tolist()))

    def push_all(self, src_ptr, stride_bytes, n_samples, kind=2):
        """n_samples for every stream, stream s from src_ptr + s * stride_bytes (kind: 0 host, 1 device, 2 pinned)"""
        self.L.dabx_push_all.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int64, C.c_int]
        _chk(self.L.dabx_push_all(self.h, C.c_void_p(src_ptr), stride_bytes, n_samples, kind))

    def alloc_pinned(self, n_bytes):
        """page-locked host buffer as a numpy uint8 array (free with free_pinned)"""
        self.L.dabx_alloc_pinned.restype = C.c_void_p
        self.L.dabx_alloc_pinned.argtypes = [C.c_size_t]
        p = self.L.dabx_alloc_pinned(n_bytes)
        if not p:
            raise MemoryError("dabx_alloc_pinned")
        return np.ctypeslib.as_array((C.c_uint8 * n_bytes).from_address(p))

    def free_pinned(self, arr):
        self.L.dabx_free_pinned.argtypes = [C.c_void_p]
        self.L.dabx_free_pinned(C.c_void_p(arr.ctypes.data))

    def push_resampled(self, stream, iq, in_rate_hz, gain=1.0):
        """iq: interleaved I,Q as int16 or float32 at in_rate_hz; returns the number of 2.048 Msps samples appended"""
        iq = np.ascontiguousarray(iq)
        fmt = {np.dtype(np.int16): 1, np.dtype(np.float32): 2}[iq.dtype]
        self.L.dabx_push_resampled.restype = C.c_int64
        self.L.dabx_push_resampled.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int64, C.c_int, C.c_double, C.c_float]
        n = self.L.dabx_push_resampled(self.h, stream, iq.ctypes.data, iq.size // 2, fmt, float(in_rate_hz), float(gain))
        if n < 0:
            _chk(int(n))
        return int(n)

    def push_resampled_from(self, stream, iq, in_rate_hz, gain=1.0, kind=2):
        """as push_resampled with the source kinds of push(): kind 2 = iq lives in memory from alloc_pinned(), the copy and the
        converter kernels are queued (flush_copies() or the next waited step completes them)"""
        fmt = {np.dtype(np.int16): 1, np.dtype(np.float32): 2}[iq.dtype]
        self.L.dabx_push_resampled_from.restype = C.c_int64
        self.L.dabx_push_resampled_from.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int64, C.c_int, C.c_double, C.c_float, C.c_int]
        n = self.L.dabx_push_resampled_from(self.h, stream, iq.ctypes.data, iq.size // 2, fmt, float(in_rate_hz), float(gain), int(kind))
        if n < 0:
            _chk(int(n))
        return int(n)

    def enable_level(self, stream, on=True):
        self.L.dabx_enable_level.argtypes = [C.c_void_p, C.c_int, C.c_int]
        _chk(self.L.dabx_enable_level(self.h, stream, 1 if on else 0))

    def level(self, stream):
        v = C.c_float()
        self.L.dabx_get_level.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        _chk(self.L.dabx_get_level(self.h, stream, C.byref(v)))
        return v.value

    def flush_copies(self):
        self.L.dabx_flush_copies.argtypes = [C.c_void_p]
        _chk(self.L.dabx_flush_copies(self.h))

    def read_ring(self, stream, start, n):
        out = np.zeros(2 * n, dtype=np.int16 if self.fmt else np.uint8)
        self.L.dabx_read_ring.argtypes = [C.c_void_p, C.c_int, C.c_int64, C.c_int64, C.c_void_p]
        _chk(self.L.dabx_read_ring(self.h, stream, start, n, out.ctypes.data))
        return out

    def push_device(self, stream, dev_ptr, n_samples):
        _chk(self.L.dabx_push(self.h, stream, C.c_void_p(dev_ptr), n_samples, 1))

    def ring_ptr(self, stream):
        return self.L.dabx_ring_ptr(self.h, stream)

    def set_write_pos(self, stream, wr):
        _chk(self.L.dabx_set_write_pos(self.h, stream, wr))

    def frames_available(self):
        return _chk(self.L.dabx_frames_available(self.h))

    def process(self, n_frames):
        _chk(self.L.dabx_process(self.h, n_frames))
        self.last_frames = n_frames

    def process_async(self, n_frames):
        _chk(self.L.dabx_process_async(self.h, n_frames))
        self.last_frames = n_frames

    def wait(self):
        _chk(self.L.dabx_wait(self.h))

    def fib(self, stream):
        n = self.last_frames
        fib = np.zeros((n, 12, 32), dtype=np.uint8)
        ok = np.zeros((n, 12), dtype=np.uint8)
        _chk(self.L.dabx_get_fib(self.h, stream, fib.ctypes.data, ok.ctypes.data))
        return fib, ok

    def msc(self, stream):
        n = self.last_frames
        msc = np.zeros((n, 4, max(self.msc_bytes[stream], 1)), dtype=np.uint8)
        valid = np.zeros((n, 4), dtype=np.uint8)
        _chk(self.L.dabx_get_msc(self.h, stream, msc.ctypes.data, valid.ctypes.data))
        return msc[:, :, :self.msc_bytes[stream]], valid

    def sync(self, stream):
        rec = np.zeros(self.last_frames, dtype=SYNC_DTYPE)
        _chk(self.L.dabx_get_sync(self.h, stream, rec.ctypes.data))
        return rec

    def state(self, stream):
        st = StreamState()
        _chk(self.L.dabx_get_state(self.h, stream, C.byref(st)))
        return dict(pos=st.pos, inc=st.inc, locked=st.locked, cif=st.cif, bad=st.bad, wr=st.wr, slope=st.slope_q16)

    def fic_soft(self, stream):
        a = np.zeros((self.last_frames, FIC_SOFT_BITS), dtype=np.int8)
        _chk(self.L.dabx_get_fic_soft(self.h, stream, a.ctypes.data))
        return a

    def msc_soft(self, stream):
        a = np.zeros((self.last_frames, 4, CIF_SOFT_BITS), dtype=np.int8)
        _chk(self.L.dabx_get_msc_soft(self.h, stream, a.ctypes.data))
        return a

    def fib_counts(self):
        ok, bad = C.c_int64(), C.c_int64()
        _chk(self.L.dabx_get_fib_counts(self.h, C.byref(ok), C.byref(bad)))
        return ok.value, bad.value

    def last_shader_clock(self):
        """GHz the Viterbi kernel of the last step ran at (timing must be enabled)"""
        g = C.c_double()
        self.L.dabx_last_shader_clock.argtypes = [C.c_void_p, C.c_void_p]
        _chk(self.L.dabx_last_shader_clock(self.h, C.byref(g)))
        return g.value

    def requeue_total(self):
        """codewords decoded a second time with spilled decisions (survivors did not merge), all steps so far"""
        n = C.c_uint64()
        self.L.dabx_get_requeue_total.argtypes = [C.c_void_p, C.c_void_p]
        _chk(self.L.dabx_get_requeue_total(self.h, C.byref(n)))
        return n.value

    def fft2048(self, x):
        x = np.ascontiguousarray(x, dtype=np.complex64).reshape(-1, 2048)
        out = np.zeros_like(x)
        _chk(self.L.dabx_fft2048(self.h, x.ctypes.data, out.ctypes.data, x.shape[0]))
        return out

    def viterbi(self, soft, kind=0, option=0, level=3, kbps=64):
        soft = np.ascontiguousarray(soft, dtype=np.int8)
        n_cw = soft.shape[0]
        out = np.zeros((n_cw, 8192), dtype=np.uint8)
        flat = np.zeros(n_cw * 8192, dtype=np.uint8)
        nb = _chk(self.L.dabx_viterbi(self.h, kind, option, level, kbps, soft.ctypes.data, n_cw, flat.ctypes.data))
        return flat[:n_cw * nb].reshape(n_cw, nb).copy()

    def enable_spectrum(self, mask=1):
        """mask bit 0: PRS (signal) spectrum, bit 1: null-symbol spectrum"""
        self.L.dabx_enable_spectrum.argtypes = [C.c_void_p, C.c_int]
        _chk(self.L.dabx_enable_spectrum(self.h, int(mask)))

    def null_spectrum(self, stream):
        p = np.zeros(2048, dtype=np.float32)
        self.L.dabx_get_null_spectrum.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        _chk(self.L.dabx_get_null_spectrum(self.h, stream, p.ctypes.data))
        return p

    def spectrum(self, stream):
        p = np.zeros(2048, dtype=np.float32)
        self.L.dabx_get_spectrum.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        _chk(self.L.dabx_get_spectrum(self.h, stream, p.ctypes.data))
        return p

    def enable_timing(self, on=True):
        _chk(self.L.dabx_enable_timing(self.h, 1 if on else 0))

    def last_timing(self):
        ms = (C.c_float * 5)()
        _chk(self.L.dabx_last_timing(self.h, ms))
        return list(ms)
