"""DAB+ super frames on the CPU: the oracle's decoder (oracle/dab_plus.c: dab_sf_push — the checker of the GPU kernel
k_superframe, see tests/test_gpu_superframe.py) against the oracle's encoder: fire-code sync at an arbitrary
logical-frame offset, RS(120,110) correction, AU CRC, carry across arbitrary step boundaries."""
import numpy as np
import pytest

from oracle import binding as ob


def _decode(frames, kbps):
    """[(header byte with the conceal bit for a bad CRC, AU bytes)], statistics"""
    dec = ob.SuperframeDecoder(kbps)
    recs, data = dec.push(np.ascontiguousarray(frames, dtype=np.uint8))
    aus = []
    for r, d in zip(recs, data):
        for i in range(r["num_aus"]):
            if (r["au_valid"] >> i) & 1:
                aus.append((int(r["header"]) | (0 if (r["au_ok"] >> i) & 1 else 0x80), d[r["au_start"][i]:r["au_start"][i + 1] - 2].copy()))
    st = dec.stats()
    return aus, {k: st[k] for k in ("superframes", "au_ok", "au_crc_err", "rs_corrected", "rs_uncorrectable", "sync_loss")}


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


# ---- the oracle's own decoder (oracle/dab_plus.c: dab_sf_push), the checker of the GPU kernel k_superframe

def _aus_of(recs, data):
    out = []
    for r, d in zip(recs, data):
        for i in range(r["num_aus"]):
            if (r["au_valid"] >> i) & 1:
                out.append(((r["au_ok"] >> i) & 1, d[r["au_start"][i]:r["au_start"][i + 1] - 2].tobytes()))
    return out


@pytest.mark.parametrize("kbps,dac,sbr", [(64, 1, 1), (32, 0, 1), (96, 1, 0), (48, 0, 0), (8, 1, 1), (192, 1, 0)])
def test_oracle_decoder_round_trip_in_pieces(kbps, dac, sbr):
    frames, aus = ob.superframes(kbps, 5, seed=kbps + 1, dac_rate=dac, sbr=sbr)
    lead = np.random.default_rng(2).integers(0, 256, (2, 3 * kbps), dtype=np.uint8)
    seq = np.concatenate([lead, frames])
    dec = ob.SuperframeDecoder(kbps)
    recs, data = [], []
    for a, b in ((0, 3), (3, 4), (4, 13), (13, len(seq))):          # arbitrary step boundaries: the carry must bridge them
        r, d = dec.push(seq[a:b])
        recs += list(r); data += list(d)
    assert [r["first_frame"] for r in recs] == [2, 7, 12, 17, 22]
    assert [a.tobytes() for a in aus] == [x for ok, x in _aus_of(recs, data) if ok]
    assert dec.stats()["superframes"] == 5 and dec.stats()["au_crc_err"] == 0 and dec.stats()["carry"] == 0


def test_oracle_decoder_on_damaged_input_loses_and_regains_sync():
    kbps, s = 64, 8
    frames, aus = ob.superframes(kbps, 6, seed=9)
    sf = frames.reshape(6, 120 * s).copy()
    rng = np.random.default_rng(8)
    for j in range(s):
        for k in rng.choice(120, 1 + j % 5, replace=False):
            sf[1, j + k * s] ^= rng.integers(1, 256)
    for k in 20 + rng.choice(100, 7, replace=False):
        sf[2, 5 + k * s] ^= 0xA5                                     # uncorrectable code word
    rows = sf.reshape(30, 3 * kbps)
    seq = np.concatenate([rows[:12], np.zeros((3, 3 * kbps), np.uint8), rows[12:]])   # a gap inside super frame 2: sync loss
    got, st = _decode(seq, kbps)
    # the fire code only covers the header: the window holding the first two frames of super frame 2 followed by the gap is
    # accepted (its code words are beyond repair, its access units fail their CRCs), then synchronisation is lost and regained
    assert st["superframes"] == 6 and st["sync_loss"] == 1 and st["rs_uncorrectable"] == s and st["au_crc_err"] >= 1
    assert st["rs_corrected"] == sum(1 + j % 5 for j in range(s))
    tx = [a.tobytes() for a in aus]
    good = [d.tobytes() for h, d in got if not h & 0x80]
    assert all(g in tx for g in good) and [g for g in good if g not in tx[6:9]] == tx[:6] + tx[9:]   # only AUs of super frame 2 are lost


def test_rs_decoder_corrects_up_to_five_errors():
    rng = np.random.default_rng(4)
    msg = rng.integers(0, 256, 110, dtype=np.uint8)
    par = np.zeros(10, dtype=np.uint8)
    ob.lib().dab_rs_encode_120_110(msg.ctypes.data, par.ctypes.data)
    cw = np.concatenate([msg, par])
    for nerr in range(0, 8):
        bad = cw.copy()
        for k in rng.choice(120, nerr, replace=False):
            bad[k] ^= rng.integers(1, 256)
        r, fixed = ob.rs_decode(bad)
        if nerr <= 5:
            assert r == nerr and np.array_equal(fixed, cw)
        else:
            assert r == -1 or not np.array_equal(fixed, cw)          # beyond the bound: detected, or (rarely) miscorrected
