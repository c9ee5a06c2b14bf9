"""ctypes binding of the CPU oracle (oracle/liboracle.so) — TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module.  The product (abracadabra_amd) never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

TF = 196608
FIC_BITS = 9216
CIF_BITS = 55296


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(path):
            build()
        L = C.CDLL(path)
        L.orx_create.restype = C.c_void_p
        L.orx_create.argtypes = [C.c_int, C.c_int64, C.c_int]
        L.orx_destroy.argtypes = [C.c_void_p]
        L.orx_set_subch.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.orx_push.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
        L.orx_get_state.argtypes = [C.c_void_p, C.c_void_p]
        L.orx_process.argtypes = [C.c_void_p, C.c_int] + [C.c_void_p] * 7
        L.orx_cordic.restype = C.c_int32
        L.orx_cordic.argtypes = [C.c_int64, C.c_int64]
        L.orx_fft.argtypes = [C.c_void_p, C.c_void_p]
        L.orx_viterbi.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.orx_decode_linear.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        L.dab_tx_generate.argtypes = [C.c_void_p] * 4
        L.dab_tx_msc_bytes_per_cif.argtypes = [C.c_void_p]
        L.dab_crc16.restype = C.c_uint16
        L.dab_crc16.argtypes = [C.c_void_p, C.c_int]
        L.dab_conv_output.argtypes = [C.c_int, C.c_int]
        L.dab_profile_eep.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p]
        L.dab_profile_any.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p]
        L.dab_superframe_build.argtypes = [C.c_int] * 6 + [C.c_void_p, C.c_void_p, C.c_void_p]
        L.dab_rs_encode_120_110.argtypes = [C.c_void_p, C.c_void_p]
        _LIB = L
    return _LIB


class TxCfg(C.Structure):
    _fields_ = [("seed", C.c_uint64), ("eid", C.c_int32), ("n_frames", C.c_int32), ("n_subch", C.c_int32),
                ("delay", C.c_int32), ("loop", C.c_int32), ("fmt", C.c_int32), ("snr_db", C.c_double),
                ("cfo_hz", C.c_double), ("rms", C.c_double), ("subch", (C.c_int32 * 4) * 64), ("payload_given", C.c_int32), ("tii_main", C.c_int32), ("tii_sub", C.c_int32), ("extra_figs", C.c_int32), ("packet_sub", C.c_int32),
                ("sco_ppm", C.c_double), ("dc_i", C.c_double), ("dc_q", C.c_double), ("echo_db", C.c_double), ("echo_phase", C.c_double),
                ("echo_delay", C.c_int32), ("eid2_from", C.c_int32)]          # = dab_tx_cfg_t (oracle/dab_tx.h)


class Profile(C.Structure):
    _fields_ = [("nseg", C.c_int), ("L", C.c_int * 4), ("PI", C.c_int * 4), ("n_in", C.c_int),
                ("n_coded", C.c_int), ("n_cu", C.c_int)]


SYNC_DTYPE = np.dtype([("t_sym0", "<i8"), ("inc", "<i4"), ("flags", "<i4"), ("peak_idx", "<i4"),
                       ("m_int", "<i4"), ("peak", "<f4"), ("total", "<f4"), ("cp_re", "<i8"), ("cp_im", "<i8"),
                       ("e_null", "<i8"), ("e_sig", "<i8")])
assert SYNC_DTYPE.itemsize == 64


def eep_profile(option, level, kbps):
    p = Profile()
    if lib().dab_profile_eep(option, level, kbps, C.byref(p)):
        raise ValueError("invalid EEP profile")
    return p


def any_profile(option, level, kbps=0):
    """option 0/1: EEP set A/B (level 1..4, kbps); option 2: UEP with level = table index 0..63"""
    p = Profile()
    L = lib()
    L.dab_profile_any.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p]
    if L.dab_profile_any(option, level, kbps, C.byref(p)):
        raise ValueError("invalid protection profile")
    return p


def subch_layout(n=18, kbps=64, option=0, level=3):
    """n equal sub-channels packed from CU 0: [[start_cu, option, level, kbps], …]"""
    size = eep_profile(option, level, kbps).n_cu
    return [[i * size, option, level, kbps] for i in range(n)]


def tx_generate(seed=1, eid=0x1000, n_frames=2, subch=(), delay=0, loop=0, fmt=0, snr_db=30.0, cfo_hz=0.0,
                rms=28.0, payload=None, tii=None, extra_figs=False, packet_sub=0, sco_ppm=0.0, dc=(0.0, 0.0), echo=None, eid2_from=0):
    """Synthetic Mode-I signal.  Returns (iq, fib[n_frames,12,32], msc[n_frames*4, bytes_per_cif]).
    payload: optional uint8 array [n_frames*4, bytes_per_cif] to transmit instead of random bytes.
    Channel impairments: sco_ppm (sampling clock offset of the recording), dc = (I, Q) offset in LSB,
    echo = (delay_samples, attenuation_db, phase_rad) second path; eid2_from: frames from this index on carry EId + 1."""
    L = lib()
    cfg = TxCfg()
    cfg.seed, cfg.eid, cfg.n_frames, cfg.n_subch = seed, eid, n_frames, len(subch)
    cfg.delay, cfg.loop, cfg.fmt = delay, loop, fmt
    cfg.snr_db, cfg.cfo_hz, cfg.rms = snr_db, cfo_hz, rms
    cfg.tii_main, cfg.tii_sub = (tii if tii is not None else (-1, 0))
    cfg.extra_figs = 1 if extra_figs else 0
    cfg.packet_sub = int(packet_sub)
    cfg.sco_ppm, cfg.dc_i, cfg.dc_q, cfg.eid2_from = float(sco_ppm), float(dc[0]), float(dc[1]), int(eid2_from)
    if echo is not None:
        cfg.echo_delay, cfg.echo_db, cfg.echo_phase = int(echo[0]), float(echo[1]), float(echo[2])
    for i, s in enumerate(subch):
        for j in range(4):
            cfg.subch[i][j] = int(s[j])
    mb = L.dab_tx_msc_bytes_per_cif(C.byref(cfg))
    if mb < 0:
        raise ValueError("bad sub-channel configuration")
    nsamp = delay + n_frames * TF
    iq = np.zeros(nsamp * 2, dtype=np.int16 if fmt else np.uint8)
    fib = np.zeros((n_frames, 12, 32), dtype=np.uint8)
    msc = np.zeros((n_frames * 4, max(mb, 1)), dtype=np.uint8)
    if payload is not None:
        msc[:, :mb] = np.asarray(payload, dtype=np.uint8).reshape(n_frames * 4, mb)
        cfg.payload_given = 1
    rc = L.dab_tx_generate(C.byref(cfg), iq.ctypes.data, fib.ctypes.data, msc.ctypes.data)
    if rc:
        raise RuntimeError(f"dab_tx_generate failed: {rc}")
    return iq, fib, msc[:, :mb]


class Stream:
    """One oracle receiver instance (mirrors one stream of the dabx batch API)."""

    def __init__(self, fmt=0, ring_len=16 * TF, subch=(), ti_slots=64):
        self.L = lib()
        self.h = self.L.orx_create(fmt, ring_len, ti_slots)
        self.fmt = fmt
        self.msc_bytes = 0
        if len(subch):
            self.set_subch(subch)

    def close(self):
        if self.h:
            self.L.orx_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def set_subch(self, subch):
        a = np.ascontiguousarray(np.array(subch, dtype=np.int32).reshape(-1, 4))
        rc = self.L.orx_set_subch(self.h, len(a), a.ctypes.data)
        if rc < 0:
            raise ValueError(f"orx_set_subch: {rc}")
        self.msc_bytes = rc

    def push(self, iq):
        iq = np.ascontiguousarray(iq)
        self.L.orx_push(self.h, iq.ctypes.data, iq.size // 2)

    def set_soft_bits(self, bits):
        """test knob: 6 (the product's contract, +-31) or 8 (+-127) bit soft decisions"""
        self.L.orx_set_soft_bits.argtypes = [C.c_void_p, C.c_int]
        self.L.orx_set_soft_bits(self.h, int(bits))

    def set_write_pos(self, wr):
        """wr = 1 << 62: resident periodic ring that never underruns (as dabx_set_write_pos)"""
        self.L.orx_set_wr.argtypes = [C.c_void_p, C.c_int64]
        self.L.orx_set_wr(self.h, wr)

    def spectrum(self):
        out = np.zeros(2048, dtype=np.float32)
        self.L.orx_get_spectrum.argtypes = [C.c_void_p, C.c_void_p]
        self.L.orx_get_spectrum(self.h, out.ctypes.data)
        return out

    def null_spectrum(self):
        out = np.zeros(2048, dtype=np.float32)
        self.L.orx_get_null_spectrum.argtypes = [C.c_void_p, C.c_void_p]
        self.L.orx_get_null_spectrum(self.h, out.ctypes.data)
        return out

    def state(self):
        st = np.zeros(7, dtype=np.int64)
        self.L.orx_get_state(self.h, st.ctypes.data)
        return dict(pos=int(st[0]), inc=int(st[1]), locked=int(st[2]), cif=int(st[3]), bad=int(st[4]), wr=int(st[5]), slope=int(st[6]))

    def process(self, n_frames, want_soft=True):
        out = dict(
            sync=np.zeros(n_frames, dtype=SYNC_DTYPE),
            fic_soft=np.zeros((n_frames, FIC_BITS), dtype=np.int8) if want_soft else None,
            msc_soft=np.zeros((n_frames, 4, CIF_BITS), dtype=np.int8) if want_soft else None,
            fib=np.zeros((n_frames, 12, 32), dtype=np.uint8),
            fib_ok=np.zeros((n_frames, 12), dtype=np.uint8),
            msc=np.zeros((n_frames, 4, max(self.msc_bytes, 1)), dtype=np.uint8),
            msc_valid=np.zeros((n_frames, 4), dtype=np.uint8),
        )
        ptr = lambda a: a.ctypes.data if a is not None else None
        rc = self.L.orx_process(self.h, n_frames, ptr(out["sync"]), ptr(out["fic_soft"]), ptr(out["msc_soft"]),
                                ptr(out["fib"]), ptr(out["fib_ok"]), ptr(out["msc"]), ptr(out["msc_valid"]))
        out["rc"] = rc
        out["msc"] = out["msc"][:, :, :self.msc_bytes]
        return out


def fft(x):
    """Oracle FFT of a complex64 vector of length 2048, natural bin order."""
    re = np.ascontiguousarray(x.real.astype(np.float32))
    im = np.ascontiguousarray(x.imag.astype(np.float32))
    lib().orx_fft(re.ctypes.data, im.ctypes.data)
    return re + 1j * im


def decode_linear(soft, kind=0, option=0, level=3, kbps=64):
    soft = np.ascontiguousarray(soft, dtype=np.int8)
    out = np.zeros(8192, dtype=np.uint8)
    n = lib().orx_decode_linear(kind, option, level, kbps, soft.ctypes.data, out.ctypes.data)
    if n < 0:
        raise ValueError("bad profile")
    return out[:n].copy()


def superframes(kbps, n_superframes, seed=0, dac_rate=1, sbr=1, ch_mode=1, ps=0, au_heads=None, equal_aus=False):
    """Random DAB+ audio super frames for a kbps sub-channel.
    au_heads: optional list of byte strings; access unit i starts with au_heads[i] (e.g. a data_stream_element
    carrying PAD), the rest of it is random.
    equal_aus: access units of equal size (64 kbit/s, 3 units: 290 / 289 / 289 bytes — the sizes the survey's probe used).
    Returns (bytes [n_superframes*5, 3*kbps] = one row per logical frame, list of AU payloads)."""
    L = lib()
    s = kbps // 8
    num_aus = (3 if sbr else 6) if dac_rate else (2 if sbr else 4)
    first = (6 if sbr else 11) if dac_rate else (5 if sbr else 8)
    room = 110 * s - first - 2 * num_aus
    rng = np.random.default_rng(seed)
    out = np.zeros((n_superframes, 120 * s), dtype=np.uint8)
    aus = []
    for f in range(n_superframes):
        cuts = np.sort(rng.choice(np.arange(8, room - 8), num_aus - 1, replace=False)) if num_aus > 1 else np.array([], dtype=int)
        if equal_aus:                                       # access units of (nearly) one size, the longer ones first
            cuts = np.cumsum([room // num_aus + (1 if i < room % num_aus else 0) for i in range(num_aus - 1)])
        lens = np.diff(np.concatenate([[0], cuts, [room]])).astype(np.int32)
        while au_heads is not None and num_aus > 1 and lens.min() < 40:         # room for the prescribed heads
            cuts = np.sort(rng.choice(np.arange(8, room - 8), num_aus - 1, replace=False))
            lens = np.diff(np.concatenate([[0], cuts, [room]])).astype(np.int32)
        data = [rng.integers(0, 256, int(n), dtype=np.uint8) for n in lens]
        if au_heads is not None:
            for d in data:
                k = len(aus) + next(i for i, x in enumerate(data) if x is d)
                if k < len(au_heads):
                    h = np.frombuffer(au_heads[k], dtype=np.uint8)
                    assert len(h) <= len(d)
                    d[:len(h)] = h
                else:
                    d[0] &= 0x1F                                               # not a data_stream_element
        ptrs = (C.c_void_p * num_aus)(*[d.ctypes.data for d in data])
        ln = (C.c_int * num_aus)(*[int(n) for n in lens])
        rc = L.dab_superframe_build(s, dac_rate, sbr, ch_mode, ps, 0, ptrs, ln, out[f].ctypes.data)
        assert rc == 0
        aus += data
    return out.reshape(n_superframes * 5, 24 * s), aus


SF_REC_DTYPE = np.dtype([("first_frame", "<u4"), ("header", "u1"), ("num_aus", "u1"), ("au_valid", "u1"), ("au_ok", "u1"),
                         ("au_start", "<u2", (8,)), ("rs_corrected", "<u2"), ("rs_failed", "<u2"), ("pad", "<u4")])
assert SF_REC_DTYPE.itemsize == 32


class SuperframeDecoder:
    """oracle/dab_plus.c streaming super frame decoder (one sub-channel)"""

    def __init__(self, kbps):
        self.L = lib()
        self.kbps, self.s = kbps, kbps // 8
        self.state = np.zeros(4096, dtype=np.uint8)            # dab_sf_state_t, opaque
        self.L.dab_sf_init.argtypes = [C.c_void_p, C.c_int]
        self.L.dab_sf_push.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int]
        self.L.dab_sf_init(self.state.ctypes.data, kbps)

    def push(self, frames):
        """frames: uint8 [n, 3*kbps]; returns (records, data [n_rec, 110*s])"""
        frames = np.ascontiguousarray(frames, dtype=np.uint8).reshape(-1, 3 * self.kbps)
        cap = len(frames) // 5 + 2
        recs = np.zeros(cap, dtype=SF_REC_DTYPE)
        data = np.zeros((cap, 110 * self.s), dtype=np.uint8)
        n = self.L.dab_sf_push(self.state.ctypes.data, frames.ctypes.data, len(frames), recs.ctypes.data, data.ctypes.data, cap)
        return recs[:n].copy(), data[:n].copy()

    def stats(self):
        v = self.state.view(np.int32)
        return dict(zip(("superframes", "au_ok", "au_crc_err", "rs_corrected", "rs_uncorrectable", "sync_loss"), v[4:10].tolist()),
                    carry=int(v[2]), frames_seen=int(v[3]), synced=int(v[10]))


def rs_decode(cw):
    cw = np.ascontiguousarray(cw, dtype=np.uint8).copy()
    L = lib()
    L.dab_rs_decode_120_110.argtypes = [C.c_void_p]
    r = L.dab_rs_decode_120_110(cw.ctypes.data)
    return r, cw


class Resampler:
    """oracle/dab_src.c: the reference's InputDeviceSRC converters (inputdevicesrc.cpp:33-47 picks by rate)"""

    def __init__(self, in_rate_hz):
        self.L = lib()
        self.rate = float(in_rate_hz)
        self.kind = "copy" if self.rate == 2048e3 else ("ds2" if self.rate == 4096e3 else "farrow")
        self.L.osrc_sizeof_ds2.restype = self.L.osrc_sizeof_farrow.restype = C.c_int
        self.state = np.zeros(max(self.L.osrc_sizeof_ds2(), self.L.osrc_sizeof_farrow()) + 16, dtype=np.uint8)
        self.L.osrc_ds2_reset.argtypes = [C.c_void_p]
        self.L.osrc_farrow_reset.argtypes = [C.c_void_p, C.c_float]
        self.L.osrc_ds2_process.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        self.L.osrc_farrow_process.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        self.L.osrc_ds2_level.restype = self.L.osrc_farrow_level.restype = C.c_float
        self.L.osrc_ds2_level.argtypes = self.L.osrc_farrow_level.argtypes = [C.c_void_p]
        if self.kind == "ds2":
            self.L.osrc_ds2_reset(self.state.ctypes.data)
        elif self.kind == "farrow":
            self.L.osrc_farrow_reset(self.state.ctypes.data, self.rate)

    def process(self, iq_f32):
        """iq_f32: interleaved I,Q float32; returns the float32 output samples (interleaved)"""
        x = np.ascontiguousarray(iq_f32, dtype=np.float32)
        n = x.size // 2
        if self.kind == "copy":
            self.L.osrc_passthrough_level.restype = C.c_float
            self.L.osrc_passthrough_level.argtypes = [C.c_float, C.c_void_p, C.c_int]
            self.copy_level = self.L.osrc_passthrough_level(getattr(self, "copy_level", 0.0), x.ctypes.data, n)
            return x.copy()
        out = np.zeros(2 * (n + 8), dtype=np.float32)
        if self.kind == "ds2":
            m = self.L.osrc_ds2_process(self.state.ctypes.data, x.ctypes.data, n, out.ctypes.data)
        else:
            m = self.L.osrc_farrow_process(self.state.ctypes.data, x.ctypes.data, n, out.ctypes.data, None, None)
        return out[:2 * m].copy()

    def level(self):
        if self.kind == "copy":
            return float(getattr(self, "copy_level", 0.0))
        return float(self.L.osrc_ds2_level(self.state.ctypes.data) if self.kind == "ds2" else self.L.osrc_farrow_level(self.state.ctypes.data))


def to_s16(y, gain=1.0):
    """the ring's sample format: rint(y * gain) clamped to int16 (one float32 multiply, round to nearest even)"""
    v = np.rint(np.asarray(y, dtype=np.float32) * np.float32(gain))
    return np.clip(v, -32768, 32767).astype(np.int16)
