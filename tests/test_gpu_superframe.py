"""k_superframe (DAB+ audio super frames on the GPU) against oracle/dab_plus.c, through the C ABI: the whole chain
IQ -> ... -> Viterbi -> fire code / RS(120,110) / AU CRC, with byte errors planted in the transmitted super frames."""
import numpy as np
import pytest

import abracadabra_amd as aa
from oracle import binding as ob

pytestmark = pytest.mark.gpu


def _payload(kbps_list, n_frames, seeds, damage):
    """MSC payload rows [4 n_frames, sum(3 kbps)] with DAB+ super frames in every sub-channel; damage(sf, k) edits them"""
    cols, aus_all = [], []
    for k, (kbps, seed) in enumerate(zip(kbps_list, seeds)):
        n_sf = (4 * n_frames) // 5 + 1
        rows, aus = ob.superframes(kbps, n_sf, seed=seed, dac_rate=k % 2, sbr=(k // 2) % 2)
        sf = rows.reshape(n_sf, -1).copy()
        damage(sf, k)
        lead = k % 5                                           # every sub-channel starts at another phase of the 5-frame cycle
        rows = np.concatenate([np.zeros((lead, 3 * kbps), np.uint8), sf.reshape(-1, 3 * kbps)])[:4 * n_frames]
        cols.append(rows); aus_all.append(aus)
    return np.concatenate(cols, axis=1), aus_all


def test_superframes_match_oracle_over_several_steps(gpu_ctx_factory):
    kbps_list = [64, 32, 96, 8, 48]
    sub, cu = [], 0
    for kbps in kbps_list:                                     # EEP 3-A: 6 CU per 8 kbit/s
        sub.append([cu, 0, 3, kbps]); cu += 6 * kbps // 8
    F, steps = 3, 6
    n_frames = F * steps + 2
    rng = np.random.default_rng(77)

    def damage(sf, k):
        s = kbps_list[k] // 8
        for j in range(s):                                     # correctable errors in super frame 1
            for pos in rng.choice(120, 1 + (j % 5), replace=False):
                sf[1, j + pos * s] ^= rng.integers(1, 256)
        for pos in 20 + rng.choice(100, 7, replace=False):     # one uncorrectable code word in super frame 3 (clear of the header bytes)
            sf[3, (k % s) + pos * s] ^= 0x3C
        sf[5, 2] ^= 0x40                                       # header byte of super frame 5 flipped (1 error: repaired)

    payload, _ = _payload(kbps_list, n_frames, [10, 11, 12, 13, 14], damage)
    iq, _, msc_tx = ob.tx_generate(seed=900, n_frames=n_frames, subch=sub, delay=1200, snr_db=30.0, cfo_hz=500.0, payload=payload)
    assert np.array_equal(msc_tx.reshape(payload.shape), payload)
    ctx = gpu_ctx_factory(n_streams=1, fmt=0, ring_frames=n_frames + 2, max_frames=F)
    ctx.set_subchannels(0, sub)
    ctx.set_dabplus(0, 0b11111)
    ctx.push(0, iq)
    decs = [ob.SuperframeDecoder(k) for k in kbps_list]
    total = 0
    for _ in range(steps):
        ctx.process(F)
        gm, gv = ctx.msc(0)
        off = 0
        for k, kbps in enumerate(kbps_list):
            frames = gm[:, :, off:off + 3 * kbps][gv == 1]     # the valid CIFs of this step, in order
            off += 3 * kbps
            orecs, odata = decs[k].push(frames)
            grecs, gdata = ctx.superframes(0, k, kbps)
            assert len(grecs) == len(orecs)
            assert grecs.tobytes() == orecs.tobytes()
            assert np.array_equal(gdata, odata)
            st = ctx.superframe_stats(0, k)
            assert st == {key: decs[k].stats()[key] for key in st}
            total += len(grecs)
    assert total >= 5 * (4 * (F * steps - 4) // 5 - 1)
    st = ctx.superframe_stats(0, 0)
    assert st["rs_corrected"] >= 8 and st["rs_uncorrectable"] == 1 and st["au_crc_err"] >= 1 and st["sync_loss"] == 0


def test_dabplus_mask_validation(gpu_ctx_factory):
    ctx = gpu_ctx_factory(n_streams=1, fmt=0, ring_frames=8, max_frames=2)
    ctx.set_subchannels(0, [[0, 0, 3, 64], [48, 2, 22, 0]])
    with pytest.raises(aa.DabxError):
        ctx.set_dabplus(0, 0b100)                              # no third sub-channel
    ctx.set_dabplus(0, 0b01)
    ctx.set_subchannels(0, [[0, 0, 3, 64]])                    # a new layout clears the flags
    iq, _, _ = ob.tx_generate(seed=1, n_frames=4, subch=[[0, 0, 3, 64]], snr_db=30.0)
    ctx.push(0, iq)
    ctx.process(2)
    with pytest.raises(aa.DabxError):
        ctx.superframes(0, 0, 64)


def test_superframes_through_a_signal_gap_s16_and_ring_wrap(gpu_ctx_factory):
    """s16 input, a ring shorter than the signal (it wraps), a burst of noise in the middle: the receiver loses lock, the
    super frame stage sees garbage frames, loses and regains synchronisation — every record still equals the oracle's."""
    kbps_list = [48, 64]
    sub = [[0, 0, 3, 48], [36, 1, 4, 64]]                      # EEP 3-A and 4-B
    rng = np.random.default_rng(5)

    def tx(seed, n_frames, delay, cfo):
        payload = np.concatenate([ob.superframes(k, 4 * n_frames // 5 + 1, seed=seed + i)[0][:4 * n_frames] for i, k in enumerate(kbps_list)], axis=1)
        iq, _, _ = ob.tx_generate(seed=seed, n_frames=n_frames, subch=sub, delay=delay, snr_db=24.0, cfo_hz=cfo, fmt=1, rms=3000.0, payload=payload)
        return iq

    a, b = tx(40, 10, 700, 1500.0), tx(50, 12, 4321, -2500.0)
    gap = rng.integers(-300, 300, 2 * 4 * ob.TF).astype(np.int16)
    iq = np.concatenate([a, gap, b])
    F = 2
    ctx = gpu_ctx_factory(n_streams=1, fmt=1, ring_frames=8, max_frames=F)
    ctx.set_subchannels(0, sub)
    ctx.set_dabplus(0, 0b11)
    orc = ob.Stream(fmt=1, subch=sub, ring_len=ctx.ring_samples, ti_slots=64)
    decs = [ob.SuperframeDecoder(k) for k in kbps_list]
    pos, n_total, nrec, locked = 0, len(iq) // 2, 0, []
    chunk = F * ob.TF
    first = (F + 1) * ob.TF + 4096
    ctx.push(0, iq[:2 * first]); orc.push(iq[:2 * first]); pos = first
    while True:
        if ctx.frames_available() < F:
            n = min(chunk, n_total - pos)
            if n <= 0:
                break
            ctx.push(0, iq[2 * pos:2 * (pos + n)]); orc.push(iq[2 * pos:2 * (pos + n)]); pos += n
            continue
        ctx.process(F)
        o = orc.process(F)
        assert o["rc"] in (0, F)
        locked.append(ctx.state(0)["locked"])
        gm, gv = ctx.msc(0)
        assert np.array_equal(gv, o["msc_valid"]) and np.array_equal(gm[gv == 1], o["msc"][o["msc_valid"] == 1])
        off = 0
        for k, kbps in enumerate(kbps_list):
            frames = gm[:, :, off:off + 3 * kbps][gv == 1]
            off += 3 * kbps
            orecs, odata = decs[k].push(frames)
            grecs, gdata = ctx.superframes(0, k, kbps)
            assert grecs.tobytes() == orecs.tobytes() and np.array_equal(gdata, odata)
            nrec += len(grecs)
    assert 0 in locked and locked[-1] == 1
    st = ctx.superframe_stats(0, 1)
    assert st == {key: decs[1].stats()[key] for key in st} and st["sync_loss"] >= 1 and st["superframes"] >= 6 and nrec >= 12
