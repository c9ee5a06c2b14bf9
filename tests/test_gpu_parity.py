"""GPU parity: the HIP path, called through the C ABI, against the CPU oracle (bit-exact)."""
import numpy as np
import pytest

import abracadabra_amd as aa
from oracle import binding as ob

pytestmark = pytest.mark.gpu


def test_fft_bit_exact(gpu_ctx_factory):
    ctx = gpu_ctx_factory(n_streams=1, max_frames=1, ring_frames=4)
    rng = np.random.default_rng(0)
    x = (rng.integers(-32768, 32768, (16, 2048)) + 1j * rng.integers(-32768, 32768, (16, 2048))).astype(np.complex64)
    g = ctx.fft2048(x)
    o = np.stack([ob.fft(v) for v in x]).astype(np.complex64)
    assert np.array_equal(g.view(np.uint32), o.view(np.uint32))       # bit for bit
    ref = np.fft.fft(x.astype(np.complex128), axis=1)
    assert np.abs(g - ref).max() <= 2e-6 * np.abs(ref).max()           # and an fp32-accurate DFT


@pytest.mark.parametrize("kind,prof", [(0, (0, 3, 64)), (1, (0, 3, 64)), (1, (0, 1, 8)), (1, (0, 2, 8)), (1, (0, 2, 40)),
                                       (1, (0, 4, 72)), (1, (1, 1, 32)), (1, (1, 2, 64)), (1, (1, 3, 32)), (1, (1, 4, 96)),
                                       (1, (0, 3, 192)), (1, (2, 0, 0)), (1, (2, 13, 0)), (1, (2, 38, 0)), (1, (2, 63, 0))])
def test_viterbi_bit_exact(gpu_ctx_factory, kind, prof):
    ctx = gpu_ctx_factory(n_streams=1, max_frames=1, ring_frames=4)
    n_coded = 2304 if kind == 0 else ob.any_profile(*prof).n_coded
    rng = np.random.default_rng(hash(prof) & 0xFFFF)
    soft = rng.integers(-31, 32, (9, n_coded)).astype(np.int8)        # the soft-bit contract: |x| <= 31
    soft[1] = 0                                   # all ties
    soft[2] = rng.integers(-1, 2, n_coded)        # many ties
    soft[3] = 31                                  # saturated, all-zero codeword
    g = ctx.viterbi(soft, kind, *prof)
    o = np.stack([ob.decode_linear(s, kind, *prof) for s in soft])
    assert np.array_equal(g, o)
    soft[4, 17] = 32                              # outside the soft-bit contract: refused, not decoded wrongly
    with pytest.raises(aa.DabxError):
        ctx.viterbi(soft, kind, *prof)


def _run_pair(ctx, streams, steps, frames, subs):
    """feed identical input to the GPU context and to one oracle per stream; compare every tap"""
    oracles = []
    for s, iq in enumerate(streams):
        ctx.set_subchannels(s, subs[s])
        ctx.push(s, iq)
        o = ob.Stream(fmt=ctx.fmt, subch=subs[s], ring_len=ctx.ring_samples, ti_slots=64)
        o.push(iq)
        oracles.append(o)
    results = []
    for _ in range(steps):
        ctx.process(frames)
        for s, orc in enumerate(oracles):
            o = orc.process(frames)
            assert o["rc"] in (0, frames)
            assert np.array_equal(ctx.sync(s), o["sync"]), f"sync records, stream {s}"
            gf, gok = ctx.fib(s)
            assert np.array_equal(gok, o["fib_ok"])
            if o["rc"]:
                assert np.array_equal(ctx.fic_soft(s), o["fic_soft"])
                assert np.array_equal(ctx.msc_soft(s), o["msc_soft"])
                assert np.array_equal(gf, o["fib"])
            gm, gv = ctx.msc(s)
            assert np.array_equal(gv, o["msc_valid"])
            assert np.array_equal(gm[gv == 1], o["msc"][o["msc_valid"] == 1])
            st = ctx.state(s)
            so = orc.state()
            assert (st["pos"], st["inc"], st["locked"], st["cif"], st["bad"], st["slope"]) == (so["pos"], so["inc"], so["locked"], so["cif"], so["bad"], so["slope"])
            results.append((s, gf, gok, gm, gv))
    return results


def test_full_chain_u8_matches_oracle_and_transmitter(gpu_ctx_factory):
    subs = [ob.subch_layout(18, 64), [[0, 0, 3, 64], [48, 1, 4, 32], [100, 0, 1, 8], [200, 0, 2, 32], [300, 0, 3, 192],
                                      [500, 2, 22, 0], [600, 2, 5, 0]]]                      # EEP A/B and two UEP sub-channels
    streams, truth = [], []
    for s in range(2):
        iq, fib, msc = ob.tx_generate(seed=20 + s, eid=0x2000 + s, n_frames=10, subch=subs[s], delay=1000 + 77777 * s,
                                      snr_db=12.0 + 10 * s, cfo_hz=-2345.0 + 5000.0 * s)
        streams.append(iq); truth.append((fib, msc))
    ctx = gpu_ctx_factory(n_streams=2, fmt=0, ring_frames=16, max_frames=4)
    res = _run_pair(ctx, streams, steps=2, frames=4, subs=subs)
    for i, (s, gf, gok, gm, gv) in enumerate(res):
        step = i // 2
        fib, msc = truth[s]
        assert gok.all() and np.array_equal(gf, fib[4 * step:4 * step + 4])
        for f in range(4):
            for c in range(4):
                if gv[f, c]:
                    assert np.array_equal(gm[f, c], msc[4 * (4 * step + f) + c - 15])


def test_full_chain_s16_and_long_subchannel(gpu_ctx_factory):
    # one 1152 kbit/s EEP 3-A sub-channel fills all 864 CU: 27654 trellis steps, decisions in HBM scratch
    sub = [[0, 0, 3, 1152]]
    iq, fib, msc = ob.tx_generate(seed=31, n_frames=7, subch=sub, delay=4321, fmt=1, snr_db=18.0, cfo_hz=901.0, rms=2500.0)
    ctx = gpu_ctx_factory(n_streams=1, fmt=1, ring_frames=16, max_frames=3)
    res = _run_pair(ctx, [iq], steps=2, frames=3, subs=[sub])
    s, gf, gok, gm, gv = res[-1]
    assert gok.all() and gv[-1, -1] == 1
    assert np.array_equal(gm[2, 3], msc[4 * 5 + 3 - 15])


def test_noise_stream_next_to_good_stream(gpu_ctx_factory):
    sub = ob.subch_layout(2, 64)
    iq, fib, _ = ob.tx_generate(seed=40, n_frames=5, subch=sub, delay=100, snr_db=25.0)
    rng = np.random.default_rng(2)
    noise = rng.integers(100, 156, iq.size, dtype=np.uint8)
    ctx = gpu_ctx_factory(n_streams=2, fmt=0, ring_frames=8, max_frames=2)
    res = _run_pair(ctx, [noise, iq], steps=1, frames=2, subs=[sub, sub])
    assert not res[0][2].any() and res[1][2].all()
    assert ctx.state(0)["locked"] == 0 and ctx.state(1)["locked"] == 1


def test_underrun_and_argument_errors(gpu_ctx_factory):
    ctx = gpu_ctx_factory(n_streams=1, fmt=0, ring_frames=8, max_frames=2)
    with pytest.raises(aa.DabxError):
        ctx.process(1)                       # nothing pushed
    with pytest.raises(aa.DabxError):
        ctx.process(3)                       # more than max_frames
    with pytest.raises(aa.DabxError):
        ctx.set_subchannels(0, [[0, 0, 3, 1160]])     # does not fit 864 CU
    with pytest.raises(aa.DabxError):
        ctx.set_subchannels(0, [[0, 0, 5, 64]])       # no such protection level
    assert ctx.frames_available() == 0


def test_periodic_ring_round_trip_at_scale(gpu_ctx_factory):
    # size-independent property at bench size per stream: every decoded FIB / logical frame of a
    # looped signal must be one of the transmitted ones, for many steps
    sub = ob.subch_layout(18, 64)
    S, P, F = 8, 8, 4
    ctx = gpu_ctx_factory(n_streams=S, fmt=0, ring_frames=P, max_frames=F)
    tx = []
    rng = np.random.default_rng(9)
    for s in range(S):
        iq, fib, msc = ob.tx_generate(seed=50 + s, n_frames=P, subch=sub, loop=1, snr_db=15.0, cfo_hz=float(rng.uniform(-3000, 3000)))
        iq = np.roll(iq.reshape(-1, 2), int(rng.integers(0, ob.TF)), axis=0).reshape(-1)
        ctx.set_subchannels(s, sub); ctx.push(s, iq); ctx.set_write_pos(s, 1 << 62)
        tx.append(({f.tobytes() for f in fib}, {m.tobytes() for m in msc}))
    for step in range(6):
        ctx.process(F)
        if step < 4:
            continue
        ok, bad = ctx.fib_counts()
        assert bad == 0 and ok == S * F * 12
        for s in range(S):
            gf, _ = ctx.fib(s)
            gm, gv = ctx.msc(s)
            assert gv.all()
            assert all(f.tobytes() in tx[s][0] for f in gf)
            assert all(m.tobytes() in tx[s][1] for m in gm.reshape(F * 4, -1))


def test_mixed_profiles_in_one_launch_match_oracle(gpu_ctx_factory):
    # one Viterbi launch holding codewords of four lengths (FIC, 18 x EEP 3-A, a UEP table entry, EEP A and B)
    subs = [ob.subch_layout(18, 64), [[0, 2, 22, 0], [100, 0, 3, 64], [200, 1, 4, 32]]]
    streams, truth = [], []
    for s in range(2):
        iq, fib, msc = ob.tx_generate(seed=60 + s, eid=0x2100 + s, n_frames=10, subch=subs[s], delay=500 + 3333 * s,
                                      snr_db=14.0, cfo_hz=1000.0 - 3000.0 * s)
        streams.append(iq); truth.append((fib, msc))
    ctx = gpu_ctx_factory(n_streams=2, fmt=0, ring_frames=16, max_frames=4)
    res = _run_pair(ctx, streams, steps=2, frames=4, subs=subs)
    for i, (s, gf, gok, gm, gv) in enumerate(res):
        assert gok.all() and np.array_equal(gf, truth[s][0][4 * (i // 2):4 * (i // 2) + 4])


def test_signal_spectrum_matches_oracle(gpu_ctx_factory):
    sub = ob.subch_layout(2, 64)
    iq, _, _ = ob.tx_generate(seed=70, n_frames=4, subch=sub, delay=1500, snr_db=20.0, cfo_hz=2500.0)
    ctx = gpu_ctx_factory(n_streams=1, fmt=0, ring_frames=8, max_frames=2)
    ctx.enable_spectrum(True)
    ctx.push(0, iq)
    ctx.process(2)
    orc = ob.Stream(ring_len=8 * ob.TF)
    orc.push(iq)
    orc.process(2)
    g, o = ctx.spectrum(0), orc.spectrum()
    assert np.array_equal(g.view(np.uint32), o.view(np.uint32))              # bit for bit
    band = np.r_[1:769, 2048 - 768:2048]
    assert g[band].mean() > 50 * g[800:1248].mean()                          # 1536 carriers stand out of the guard band


def test_pinned_overlapped_ingest_matches_oracle(gpu_ctx_factory):
    """Streaming through the host boundary the way a file reader would: page-locked staging, one strided
    dabx_push_all per step queued while the previous step decodes (dabx_process_async) — same output as the
    oracle fed with the same samples."""
    S, F, steps = 3, 2, 4
    sub = ob.subch_layout(3, 64)
    n_total = (F * steps + 2) * ob.TF
    ctx = gpu_ctx_factory(n_streams=S, fmt=0, ring_frames=2 * F + 4, max_frames=F)
    stage = ctx.alloc_pinned(S * n_total * 2)
    oracles = []
    for s in range(S):
        iq, _, _ = ob.tx_generate(seed=300 + s, n_frames=F * steps + 2, subch=sub, delay=900 * s, snr_db=18.0, cfo_hz=700.0 * s)
        stage[s * n_total * 2:(s + 1) * n_total * 2] = iq[:n_total * 2]
        ctx.set_subchannels(s, sub)
        o = ob.Stream(fmt=0, subch=sub, ring_len=n_total + ob.TF, ti_slots=64)
        o.push(iq[:n_total * 2])
        oracles.append(o)
    pos = (F + 1) * ob.TF + 4096
    ctx.push_all(stage.ctypes.data, n_total * 2, pos, kind=2)
    for k in range(steps):
        ctx.process_async(F)
        n = min(F * ob.TF, n_total - pos)
        if n > 0:                                   # travels while the step above decodes
            ctx.push_all(stage.ctypes.data + pos * 2, n_total * 2, n, kind=2)
            pos += n
        ctx.wait()
        for s, orc in enumerate(oracles):
            o = orc.process(F)
            assert o["rc"] == F
            gf, gok = ctx.fib(s)
            assert np.array_equal(gok, o["fib_ok"]) and np.array_equal(gf, o["fib"])
            gm, gv = ctx.msc(s)
            assert np.array_equal(gv, o["msc_valid"]) and np.array_equal(gm[gv == 1], o["msc"][o["msc_valid"] == 1])
    assert ctx.fib_counts()[1] == 0
    ctx.free_pinned(stage)


def test_push_all_out_of_lock_step_and_argument_checks(gpu_ctx_factory):
    """dabx_push_all falls back to one copy per stream when the streams do not stand at the same write position; the
    result is the same as pushing each stream on its own"""
    sub = ob.subch_layout(1, 64)
    sigs = [ob.tx_generate(seed=400 + s, n_frames=5, subch=sub, delay=300 * s, snr_db=20.0)[0] for s in range(2)]
    n = min(len(x) for x in sigs) // 2
    sigs = [x[:2 * n] for x in sigs]
    ctx = gpu_ctx_factory(n_streams=2, fmt=0, ring_frames=8, max_frames=2)
    stage = ctx.alloc_pinned(2 * n * 2)
    for s in range(2):
        stage[s * n * 2:(s + 1) * n * 2] = sigs[s]
        ctx.set_subchannels(s, sub)
    ctx.push(1, sigs[1][:2 * 1000])                                     # stream 1 runs 1000 samples ahead
    ctx.push_all(stage.ctypes.data, n * 2, 1000, kind=2)                # ... both advance by 1000: out of lock step -> per-stream copies
    ctx.push_pinned(0, stage.ctypes.data + 2 * 1000, n - 1000)
    ctx.push_pinned(1, stage.ctypes.data + n * 2 + 2 * 2000, n - 2000)
    # stream 1 now holds samples [0,1000) + [0,1000) again + [2000,n): a corrupted stream by construction; stream 0 is intact
    ctx.process(2)
    orc = ob.Stream(subch=sub, ring_len=ctx.ring_samples)
    orc.push(sigs[0])
    o = orc.process(2)
    gf, gok = ctx.fib(0)
    assert gok.all() and np.array_equal(gf, o["fib"]) and np.array_equal(ctx.sync(0), o["sync"])
    with pytest.raises(aa.DabxError):
        ctx.push_all(stage.ctypes.data, n * 2, 10 * ob.TF, kind=2)      # overrun is refused before anything is copied
    with pytest.raises(aa.DabxError):
        ctx.push_all(stage.ctypes.data, n * 2, 16, kind=7)
    ctx.free_pinned(stage)


def test_null_spectrum_matches_oracle_and_carries_tii(gpu_ctx_factory):
    import ctypes as C
    sub = ob.subch_layout(2, 64)
    iq, _, _ = ob.tx_generate(seed=71, n_frames=4, subch=sub, delay=800, snr_db=15.0, cfo_hz=-1800.0, tii=(52, 7))
    ctx = gpu_ctx_factory(n_streams=1, fmt=0, ring_frames=8, max_frames=2)
    ctx.enable_spectrum(3)
    ctx.push(0, iq)
    ctx.process(2)
    orc = ob.Stream(ring_len=8 * ob.TF)
    orc.push(iq)
    orc.process(2)
    g, o = ctx.null_spectrum(0), orc.null_spectrum()
    assert np.array_equal(g.view(np.uint32), o.view(np.uint32))              # bit for bit
    assert np.array_equal(ctx.spectrum(0).view(np.uint32), orc.spectrum().view(np.uint32))
    ids = np.zeros(48, dtype=np.uint8)
    ctx.L.dabsdr_amd_tii_detect.argtypes = [C.c_void_p, C.c_float, C.c_void_p, C.c_int]
    n = ctx.L.dabsdr_amd_tii_detect(g.ctypes.data, 4.0, ids.ctypes.data, 24)
    assert n == 1 and tuple(ids[:2]) == (52, 7)


def test_config2_64_streams_fft_demap_bit_exact(gpu_ctx_factory):
    """BASELINE configs[1]: 64 synthetic Mode-I IQ streams batched through sync + 2048-FFT + DQPSK
    demap; every soft bit equals the oracle's, and at 25 dB the hard decisions of the FIC symbols
    re-encode to the transmitted FIBs (checked through the decoded FIBs)."""
    S = 64
    ctx = gpu_ctx_factory(n_streams=S, fmt=0, ring_frames=8, max_frames=1)
    rng = np.random.default_rng(64)
    sub = ob.subch_layout(3, 64)
    bad = 0
    for s in range(S):
        iq, fib, _ = ob.tx_generate(seed=300 + s, eid=0x5000 + s, n_frames=3, subch=sub, delay=int(rng.integers(0, 150000)), snr_db=25.0,
                                    cfo_hz=float(rng.uniform(-5000, 5000)))
        ctx.set_subchannels(s, sub)
        ctx.push(s, iq)
        ctx.truth = getattr(ctx, "truth", {})
        ctx.truth[s] = (iq, fib)
    ctx.process(1)
    for s in range(S):
        iq, fib = ctx.truth[s]
        orc = ob.Stream(subch=sub, ring_len=ctx.ring_samples)
        orc.push(iq)
        o = orc.process(1)
        assert np.array_equal(ctx.sync(s), o["sync"])
        assert np.array_equal(ctx.fic_soft(s), o["fic_soft"]) and np.array_equal(ctx.msc_soft(s), o["msc_soft"])
        gf, gok = ctx.fib(s)
        bad += int((~gok.astype(bool)).sum())
        assert np.array_equal(gf, fib[:1])
    assert bad == 0


def test_lock_loss_and_reacquisition_matches_oracle(gpu_ctx_factory):
    from test_oracle_chain import _gap_signal
    sub, iq, fib_a, fib_b = _gap_signal()
    ctx = gpu_ctx_factory(n_streams=1, fmt=0, ring_frames=32, max_frames=2)
    ctx.set_subchannels(0, sub)
    ctx.push(0, iq)
    orc = ob.Stream(subch=sub, ring_len=ctx.ring_samples)
    orc.push(iq)
    locked = []
    for step in range(9):
        if ctx.frames_available() < 2:
            break
        ctx.process(2)
        o = orc.process(2)
        assert o["rc"] in (0, 2)
        assert np.array_equal(ctx.sync(0), o["sync"]), step
        gf, gok = ctx.fib(0)
        assert np.array_equal(gok, o["fib_ok"])
        if o["rc"]:
            assert np.array_equal(gf, o["fib"])
        st, so = ctx.state(0), orc.state()
        assert (st["pos"], st["inc"], st["locked"], st["cif"], st["bad"], st["slope"]) == (so["pos"], so["inc"], so["locked"], so["cif"], so["bad"], so["slope"])
        locked.append(st["locked"])
    assert locked[0] == 1 and 0 in locked[1:] and locked[-1] == 1


def test_silence_after_lock_takes_the_requeue_path_and_matches_the_oracle(gpu_ctx_factory):
    """A recording that goes silent (u8 value 128 = zero: the flushed FIFO of the reference's input, inputdevice.cpp:80-85)
    while the receiver is locked: the soft bits of those frames are all zero, every candidate of every trellis step ties, the
    survivors never merge — the codewords are decoded a second time by k_viterbi_requeue with their decisions spilled to
    device memory (the kernel's fallback; no scratch is held for codewords that merge).  Results equal the CPU checker's,
    whose textbook Viterbi keeps the own path on a tie."""
    sub = ob.subch_layout(4, 64) + [[192, 0, 3, 192]]                    # a long codeword too: 4614 steps
    iq, fib, _ = ob.tx_generate(seed=77, n_frames=6, subch=sub, delay=5000, snr_db=22.0)
    sig = np.concatenate([iq, np.full(2 * 8 * ob.TF, 128, dtype=np.uint8)])
    ctx = gpu_ctx_factory(n_streams=2, fmt=0, ring_frames=16, max_frames=2)
    orc = ob.Stream(subch=sub, ring_len=ctx.ring_samples)
    for s in range(2):
        ctx.set_subchannels(s, sub)
        ctx.push(s, sig)
    orc.push(sig)
    total0 = ctx.requeue_total()
    steps = 0
    while ctx.frames_available() >= 2 and steps < 6:
        ctx.process(2)
        o = orc.process(2)
        assert o["rc"] in (0, 2)
        for s in range(2):
            assert np.array_equal(ctx.sync(s), o["sync"]), steps
            gf, gok = ctx.fib(s)
            gm, gv = ctx.msc(s)
            assert np.array_equal(gok, o["fib_ok"]) and np.array_equal(gv, o["msc_valid"]), steps
            if o["rc"]:
                assert np.array_equal(gf, o["fib"]), steps
                assert np.array_equal(gm[gv.astype(bool)], o["msc"][o["msc_valid"].astype(bool)]), steps
        steps += 1
    assert steps >= 4
    assert ctx.requeue_total() - total0 >= 8, "the silent frames' FIC codewords (all ties) must have gone through the requeue kernel"
