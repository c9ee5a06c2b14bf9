"""DAB+ super frame decoder of the product (csrc/superframe.hpp) against the oracle's encoder:
fire-code sync at an arbitrary logical-frame offset, RS(120,110) correction, AU CRC (CPU only)."""
import ctypes as C

import numpy as np
import pytest

import abracadabra_amd as aa
from oracle import binding as ob


def _decode(frames, kbps):
    L = aa.load_library()
    L.dabsdr_amd_superframe_decode.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
    frames = np.ascontiguousarray(frames, dtype=np.uint8)
    out = np.zeros(1 << 20, dtype=np.uint8)
    stats = np.zeros(6, dtype=np.uint32)
    n = L.dabsdr_amd_superframe_decode(frames.ctypes.data, frames.shape[0], kbps, out.ctypes.data, out.size, stats.ctypes.data)
    assert n >= 0
    aus, pos = [], 0
    while pos < n:
        hdr, ln = int(out[pos]), int(out[pos + 1]) | (int(out[pos + 2]) << 8)
        aus.append((hdr, out[pos + 3:pos + 3 + ln].copy()))
        pos += 3 + ln
    return aus, dict(zip(["superframes", "au_ok", "au_crc_err", "rs_corrected", "rs_uncorrectable", "sync_loss"], stats.tolist()))


@pytest.mark.parametrize("kbps,dac,sbr,hdr", [(64, 1, 1, 0x70), (32, 0, 1, 0x30), (96, 1, 0, 0x50), (48, 0, 0, 0x10)])
def test_clean_superframes(kbps, dac, sbr, hdr):
    frames, aus = ob.superframes(kbps, 4, seed=kbps, dac_rate=dac, sbr=sbr)
    lead = np.random.default_rng(1).integers(0, 256, (3, 3 * kbps), dtype=np.uint8)      # arbitrary start offset
    got, st = _decode(np.concatenate([lead, frames]), kbps)
    assert st["superframes"] == 4 and st["au_crc_err"] == 0 and st["rs_corrected"] == 0
    assert len(got) == len(aus)
    for (h, d), ref in zip(got, aus):
        assert h == hdr and np.array_equal(d, ref)


def test_rs_corrects_five_byte_errors_per_codeword_and_flags_more():
    kbps, s = 64, 8
    frames, aus = ob.superframes(kbps, 2, seed=5)
    sf = frames.reshape(2, 120 * s).copy()
    rng = np.random.default_rng(3)
    for j in range(s):                                     # 5 errors in every code word of super frame 0
        for k in rng.choice(120, 5, replace=False):
            sf[0, j + k * s] ^= rng.integers(1, 256)
    for k in rng.choice(120, 6, replace=False):            # 6 errors in one code word of super frame 1
        sf[1, 3 + k * s] ^= 0x55
    got, st = _decode(sf.reshape(10, 3 * kbps), kbps)
    assert st["superframes"] == 2 and st["rs_corrected"] == 5 * s and st["rs_uncorrectable"] == 1
    n0 = len(aus) // 2
    for (h, d), ref in zip(got[:n0], aus[:n0]):
        assert h == 0x70 and np.array_equal(d, ref)        # fully repaired
    assert st["au_crc_err"] >= 1 and any(h & 0x80 for h, _ in got[n0:])   # damaged AUs are flagged for concealment
