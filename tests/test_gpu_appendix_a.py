"""The remaining observations of SURVEY.md Appendix A (what the reference's closed library was seen to do when the survey
session drove it through its public API), encoded as GPU tests through the same 24-function ABI.  The binary is not run
here; the numbers are quoted from the survey.  Where this library deliberately differs the test says so.

  A.3  a raw file looped sample-discontinuously: ONE period with fibErrorCntr = 12, brief SYNC 0, re-lock
  A.4  one DAB+ 64 kbit/s service: mscCrcOkCntr 15..21 per 8-frame period, mscCrcErrorCntr = 0, access units hdr 0x70,
       289/290 bytes, bit-exact; first unit delivered = super frame 7 after the selection (16-CIF interleaver fill + super
       frame sync) — here the interleaver is kept warm for the whole MSC, so audio starts with the NEXT super frame
  A.5  9 dB SNR: fibErrorCntr = 0, mscCrcErrorCntr = 0 (reported snr 7.6 dB)
  A.6  snr10 reads low/compressed at the bottom end: 4.1..6.7 dB reported for 2..8 dB in, 28.5 for 30 — this library reports
       an unbiased estimate (2..8 dB read 2..8 dB); the test pins it to the truth and to the reference where they agree
  ADVICE r02: a silent first frame must not fix the gain of the float -> s16 conversion
"""
import ctypes as C
import time

import numpy as np
import pytest

from oracle import binding as ob

pytestmark = pytest.mark.gpu
TF = 196608
SID = 0x1A01
PACE = 0.0025            # 30 ms per frame of input: frame by frame, delivered at once (the reference's raw-file input is timer paced)


def _host(*a, **k):
    from legacy_host import LegacyHost
    return LegacyHost(*a, **k)


def _nid():
    from legacy_host import NID
    return NID


def _dabplus_signal(n_frames, snr, seed=7, equal_aus=True, delay=2500, cfo=300.0):
    sub = [[0, 0, 3, 64]]
    rows, aus = ob.superframes(64, n_frames * 4 // 5, seed=seed, equal_aus=equal_aus)
    payload = np.zeros((n_frames * 4, 192), dtype=np.uint8)
    payload[:len(rows)] = rows
    iq, fib_tx, _ = ob.tx_generate(seed=seed, eid=0x1234, n_frames=n_frames, subch=sub, delay=delay, snr_db=snr, cfo_hz=cfo, payload=payload)
    return iq.astype(np.float32) - 128.0, aus, fib_tx


def _select(host, sid=SID):
    NID = _nid()
    host.wait_for(lambda e: e["nid"] == NID["SYNC_STATUS"] and e.get("level") == 3)
    t0 = time.time()
    while host.gate_at is not None and host.pos // 2 < host.gate_at:        # select with the input held at the gate
        time.sleep(0.02)
        assert time.time() - t0 < 20
    while True:                                                  # the service appears in the FIG database within a few frames
        host.L.dabsdrRequest_GetServiceList(host.handle)
        time.sleep(0.05)
        with host.lock:
            lists = [e for e in host.events if e["nid"] == NID["SERVICE_LIST"]]
        if lists and any(s["sid"] == sid for s in lists[-1]["services"]):
            break
        assert time.time() - t0 < 20
    host.L.dabsdrRequest_ServiceSelection(host.handle, sid, 0, 0)
    sel = host.wait_for(lambda e: e["nid"] == NID["SERVICE_SELECTION"])[-1]
    assert sel["status"] == 0
    return sel["at"]


def test_a3_looped_file_loses_one_frame_at_the_wrap():
    """the recording (20 frames behind 1500 samples of lead-in, cut in the middle of the MSC of its last frame) played three
    times over: at each wrap the frame phase jumps.  Paced input, period of 8 frames as the reference's host asks for."""
    NID = _nid()
    iq, fib_tx, _ = ob.tx_generate(seed=41, eid=0x1234, n_frames=20, subch=[[0, 0, 3, 64]], delay=1500, snr_db=25.0, cfo_hz=-420.0)
    one = (iq.astype(np.float32) - 128.0)[:2 * (1500 + 19 * TF + 120000)]                  # the cut falls into frame 19's MSC
    host = _host(np.tile(one, 3), pace=PACE)
    try:
        host.tune(period_log2=3)
        host.wait_for(lambda e: e["nid"] == NID["PERIODIC"] and e["at"] >= 3 * (len(one) // 2) - 3 * TF, timeout=120)
        with host.lock:
            ev = list(host.events)
    finally:
        host.close()
    per = [e for e in ev if e["nid"] == NID["PERIODIC"] and e["len"]]
    first_lock = next(i for i, e in enumerate(per) if e["level"] == 3 and e["fib_err"] == 0)
    errs = [e["fib_err"] for e in per[first_lock:]]
    # two wraps inside the part looked at: each costs exactly one frame of FIBs, every other period is clean
    assert sorted(x for x in errs if x) == [12, 12], errs
    sync = [e["level"] for e in ev if e["nid"] == NID["SYNC_STATUS"]]
    i3 = sync.index(3)
    assert sync[i3:].count(0) == 2 and sync[-1] == 3, sync            # brief SYNC 0, re-lock, twice
    # and the re-lock is quick: level 3 again within three frames of input after each loss
    lost = [e["at"] for e in ev if e["nid"] == NID["SYNC_STATUS"] and e["level"] == 0 and e["at"] > 3 * TF]
    back = [min(e["at"] for e in ev if e["nid"] == NID["SYNC_STATUS"] and e["level"] == 3 and e["at"] > t) for t in lost]
    assert all(b - t <= 3 * TF + 16384 for t, b in zip(lost, back)), (lost, back)


def test_a4_one_dabplus_service_counters_units_and_start():
    NID = _nid()
    n_frames = 60
    x, aus_tx, _ = _dabplus_signal(n_frames, 25.0)
    host = _host(x, gate_at=8 * TF, pace=PACE)
    try:
        host.tune(period_log2=3)
        at_sel = _select(host)
        host.open_gate()
        host.wait_for(lambda e: e["nid"] == NID["PERIODIC"] and e["at"] >= 50 * TF, timeout=120)
        with host.lock:
            ev, audio, audio_at = list(host.events), list(host.audio), list(host.audio_at)
    finally:
        host.close()
    tx = [a.tobytes() for a in aus_tx]
    inside = [a for a, at in zip(audio, audio_at) if at <= (n_frames - 1) * TF]        # before the recording runs out
    assert len(inside) >= 90
    # every access unit: primary decoder, ASCTy 63, header 0x70 (dac 48 kHz, SBR, stereo: dabsdr.h:47-60), 289 / 290 bytes, bit-exact
    assert all(a[0] == 0 and a[1] == 63 and a[2] == 0x70 for a in inside)
    assert {len(a[3]) for a in inside} == {289, 290}
    first = tx.index(inside[0][3])
    assert first % 3 == 0 and [a[3] for a in inside] == tx[first:first + len(inside)]
    # Start of the audio.  The selection was made with the input held at frame 8 (CIF 32 of the recording, lead-in aside).  The
    # reference starts to fill its time de-interleaver on selection and delivered super frame 7 after it (App. A.4); here the
    # whole MSC is de-interleaved all the time, so the first super frame that BEGINS after the selection is delivered:
    # (logical frame r is complete with CIF r + 15 of the transmission: the 16-CIF time interleaver)
    sf_done = max(0, (at_sel - 2500) * 4 // TF - 15) // 5                               # super frames that had ended by then
    assert at_sel >= 8 * TF and sf_done >= 3
    assert sf_done <= first // 3 <= sf_done + 2, (first // 3, sf_done)                  # the reference: 7 after the selection
    # periodic counters (8 frames = 32 CIFs = 6.4 super frames of 3 units): 15..21 good units per period, no bad one
    per = [e for e in ev if e["nid"] == NID["PERIODIC"] and e["len"] and e["at"] > at_sel + 16 * TF and e["at"] <= (n_frames - 1) * TF]
    assert len(per) >= 3
    assert all(15 <= e["crc_ok"] <= 21 and e["crc_err"] == 0 and e["fib_err"] == 0 and e["rs_unc"] == 0 for e in per), per
    assert all(abs(e["audio_bytes"] - 289.34 * e["crc_ok"]) < 3 for e in per)


def test_a5_nine_db_no_fib_error_no_au_crc_error():
    NID = _nid()
    n_frames = 70
    x, aus_tx, _ = _dabplus_signal(n_frames, 9.0, seed=11)
    host = _host(x, gate_at=8 * TF)                                   # un-paced from the gate on: eight frames per step
    try:
        host.tune(period_log2=3)
        at_sel = _select(host)
        host.open_gate()
        host.wait_for(lambda e: e["nid"] == NID["PERIODIC"] and e["at"] >= (n_frames + 16) * TF, timeout=120)
        with host.lock:
            ev, audio = list(host.events), list(host.audio)
    finally:
        host.close()
    per = [e for e in ev if e["nid"] == NID["PERIODIC"] and e["len"] and e["level"] == 3]
    assert len(per) >= 6
    assert all(e["fib_err"] == 0 and e["crc_err"] == 0 for e in per), per
    assert sum(e["crc_ok"] for e in per) >= 100
    # reference: snr 7.6 dB reported for 9 dB in; this estimator reads the truth within a dB
    snr = np.array([e["snr10"] for e in per]) / 10.0
    assert abs(float(np.median(snr)) - 9.0) < 1.0 and abs(float(np.median(snr)) - 7.6) < 2.5
    good = [a for a in audio if not (a[2] & 0x80)]
    tx = [a.tobytes() for a in aus_tx]
    first = tx.index(good[0][3])
    n = min(len(good), len(tx) - first)
    assert n >= 100 and [a[3] for a in good[:n]] == tx[first:first + n]


@pytest.mark.parametrize("snr,ref_reading", [(2.0, 4.1), (5.0, 5.4), (8.0, 6.7), (30.0, 28.5)])
def test_a6_snr_reading(snr, ref_reading):
    """The reference's snr10 is compressed at the bottom end (A.6: 4.1 .. 6.7 dB reported for 2 .. 8 dB in; 5.4 is the value its
    two end points give for 5 dB).  This library estimates (PRS energy - noise) / noise with the noise level taken from the null
    symbol's spectrum: within 1 dB of the truth from 2 to 8 dB, and like the reference it tops out below 30 dB on u8 samples
    (quantisation).  So the two agree within 1.5 dB from 5 dB up and differ by about 2 dB at 2 dB, where the reference reads high."""
    NID = _nid()
    iq, _, _ = ob.tx_generate(seed=23, eid=0x1234, n_frames=30, subch=[], delay=2200, snr_db=snr)
    host = _host(iq.astype(np.float32) - 128.0, gate_at=29 * TF)
    try:
        host.tune(period_log2=0)
        host.wait_for(lambda e: e["nid"] == NID["PERIODIC"] and e["at"] >= 26 * TF, timeout=60)
        with host.lock:
            per = [e for e in host.events if e["nid"] == NID["PERIODIC"] and e["len"] and e["level"] >= 1 and 4 * TF < e["at"]]
    finally:
        host.close()
    assert len(per) >= 12
    got = float(np.median([e["snr10"] for e in per])) / 10.0
    if snr <= 8.0:
        assert abs(got - snr) < 1.0, got
    else:
        assert 26.0 <= got <= 31.0, got
    if snr >= 5.0:
        assert abs(got - ref_reading) < 1.5, (got, ref_reading)
    else:
        assert got < ref_reading and ref_reading - got < 3.0, (got, ref_reading)


def test_silent_first_frame_does_not_fix_the_gain():
    """the reference's getSamples hands out zeros while its FIFO is flushed (inputdevice.cpp:80-85).  A frame of
    zeros followed by SDR-style floats in +-0.3 must still reach FIC sync: the gain is chosen on the first frame with a signal."""
    NID = _nid()
    iq, _, _ = ob.tx_generate(seed=29, eid=0x1234, n_frames=12, subch=[], delay=900, snr_db=25.0)
    x = (iq.astype(np.float32) - 128.0) * np.float32(0.3 / 128.0)
    sig = np.concatenate([np.zeros(2 * (TF + 5000), dtype=np.float32), x])
    host = _host(sig, gate_at=(TF + 5000) // 1 + 11 * TF)
    try:
        host.tune(period_log2=0)
        host.wait_for(lambda e: e["nid"] == NID["SYNC_STATUS"] and e.get("level") == 3, timeout=30)
        per = host.wait_for(lambda e: e["nid"] == NID["PERIODIC"] and e["len"] and e["level"] == 3 and e["at"] > 8 * TF, timeout=30)[-1]
        assert per["fib_err"] == 0
    finally:
        host.close()
