"""The sample-rate converters in front of the ring: 4096 -> 2048 kHz half-band decimator and
the transposed Farrow resampler.  This is the one row whose reference arithmetic exists as SOURCE
(src/input/inputdevicesrc.{h,cpp}); oracle/dab_src.c restates it line by line.

CPU: the restatement against an independent numpy evaluation of the same filters (double precision) and against
signal-processing properties.  GPU: kernels equal the restatement bit for bit (same float operations, same order; the
reference's own builds may contract multiply-adds into FMAs, a relative difference of ~1e-7), streaming state included;
a 4.096 Msps and a 2.4 Msps ensemble decode with every FIB CRC good."""
import numpy as np
import pytest
from scipy import signal

from oracle import binding as ob

DS2 = [0.000223158782894952853123604619156594708, -0.00070774549637065342286290636764078954, 0.001735782601167994458266075064045708132,
       -0.003619832275410410967614316390950079949, 0.006788741778432844271862212082169207861, -0.01183550169261274320753329902800032869,
       0.019680477383812611941182879604639310855, -0.032073581325677551212560700832909788005, 0.053382280107447499517547839786857366562,
       -0.099631117404426483563639749263529665768, 0.316099577146216947909351802081800997257, 0.5]       # inputdevicesrc.h:105-109


def ds2_taps():
    h = np.zeros(43)
    for c in range(11):
        h[2 * c] = h[42 - 2 * c] = DS2[c]
    h[21] = DS2[11]
    return h


def test_ds2_is_the_43_tap_halfband_decimator():
    rng = np.random.default_rng(0)
    x = (rng.standard_normal(4000) + 1j * rng.standard_normal(4000)) * 1000
    iq = np.stack([x.real, x.imag], axis=1).astype(np.float32).reshape(-1)
    r = ob.Resampler(4096e3)
    y = np.concatenate([r.process(iq[:2 * 1234]), r.process(iq[2 * 1234:])])            # two calls: the delay line carries over
    yc = y[0::2] + 1j * y[1::2]
    xf = iq[0::2].astype(np.float64) + 1j * iq[1::2].astype(np.float64)
    want = np.convolve(xf, ds2_taps())[0:4000:2]                                        # y[n] = sum_k h[k] x[2n - k]
    assert len(yc) == 2000 and np.abs(yc - want).max() < 2e-3                           # float32 accumulation of ~1000-sized values
    h = ds2_taps()
    assert abs(h.sum() - 1.0) < 1e-3                                                    # unity gain at DC
    w, H = signal.freqz(h, worN=4096)
    assert np.abs(H[w > 0.625 * np.pi]).max() < 10 ** (-55 / 20)                        # the band that aliases onto the ensemble (beyond pi - 0.375 pi): < -55 dB
    assert np.abs(np.abs(H[w < 0.38 * np.pi]) - 1).max() < 1e-3                         # the 1.536 MHz ensemble (|f| < 0.768 MHz of 4.096) is flat


@pytest.mark.parametrize("rate", [2400e3, 3072e3, 2048e3 * 1.0001, 6000e3])
def test_farrow_resamples_a_tone(rate):
    n = 60000
    f = 300e3
    t = np.arange(n) / rate
    x = 8000 * np.exp(2j * np.pi * f * t)
    iq = np.stack([x.real, x.imag], axis=1).astype(np.float32).reshape(-1)
    r = ob.Resampler(rate)
    y = np.concatenate([r.process(iq[:2 * 777]), r.process(iq[2 * 777:2 * 40001]), r.process(iq[2 * 40001:])])
    yc = y[0::2] + 1j * y[1::2]
    assert abs(len(yc) - n * 2048e3 / rate) <= 1
    seg = yc[200:-200]
    spec = np.abs(np.fft.fft(seg * np.hanning(len(seg))))
    k = int(np.argmax(spec))
    fk = np.fft.fftfreq(len(seg), 1 / 2048e3)[k]
    assert abs(fk - f) < 2 * 2048e3 / len(seg)                                          # the tone sits where it should at 2.048 Msps
    assert 0.9 * 8000 < np.abs(seg).mean() < 1.1 * 8000                                 # gain ~1
    assert np.abs(np.abs(seg) - np.abs(seg).mean()).max() < 0.02 * 8000                 # and clean (images far down)


def test_level_detector_follows_the_reference_recursion():
    """inputdevicesrc.cpp:167-173: fast attack (50 us), slow release (50 ms) on |x|^2"""
    r = ob.Resampler(4096e3)
    burst = np.zeros(2 * 20000, dtype=np.float32)
    burst[0::2] = 100.0
    r.process(burst)
    assert 0.98 * 1e4 < r.level() <= 1e4 * 1.0001                                       # 10 000 samples at 2.048 MHz = 4.9 ms >> 50 us
    r.process(np.zeros(2 * 20000, dtype=np.float32))
    assert 0.85 * 1e4 < r.level() < 0.95 * 1e4                                          # 4.9 ms of silence: exp(-4.9/50) = 0.906


@pytest.mark.parametrize("rate", [2400000.0, 2500000.0, 3072000.0, 4000000.0, 2048001.0, 4095999.0, 2304000.0])
def test_farrow_schedule_is_exact_integer_arithmetic_up_to_4096_khz(rate):
    """The reference's schedule `mu -= R; if (mu < 0) { dump; mu += 1 }` (inputdevicesrc.cpp:241-245) in binary32, against the
    closed form the GPU threads use for R >= 0.5 (dabx_resample.hip: farrow_mu / farrow_seg): mu and R are multiples of 2^-24
    below 1, every difference and sum the recursion forms is one too and fits a float, so it is arithmetic modulo 2^24."""
    R = np.float32(2048e3 / float(np.float32(rate)))
    assert 0.5 <= R < 1.0
    Ri = int(float(R) * 2 ** 24)
    assert Ri == float(R) * 2 ** 24
    n = 40000
    for m0 in (0.0, 0.25, 0.7312345):
        M0 = int(round(m0 * 2 ** 24))
        m = np.float32(M0 / 2 ** 24)
        mu, seg = np.empty(n, np.float32), [0]
        for k in range(n):                                         # the reference's recursion
            m = np.float32(m - R)
            if m < 0:
                m = np.float32(m + np.float32(1.0))
                seg.append(k)
            mu[k] = m
        k = np.arange(1, n + 1, dtype=np.int64)
        U = M0 - k * Ri
        assert np.array_equal(mu, (np.mod(U, 2 ** 24).astype(np.float64) / 2 ** 24).astype(np.float32))
        n_done = 0 if n * Ri <= M0 else (n * Ri - M0 + 2 ** 24 - 1) >> 24
        assert n_done == len(seg) - 1
        j = np.arange(1, n_done + 1, dtype=np.int64)
        assert seg[1:] == [int(v) for v in (M0 + (j - 1) * 2 ** 24) // Ri]


# ------------------------------------------------------------------------------------------------------------- GPU
def _chunks(n, sizes):
    out, a = [], 0
    i = 0
    while a < n:
        b = min(n, a + sizes[i % len(sizes)])
        out.append((a, b)); a = b; i += 1
    return out


@pytest.mark.gpu
@pytest.mark.parametrize("rate,dtype", [(4096e3, np.int16), (4096e3, np.float32), (2400e3, np.int16), (3000e3, np.float32), (2048e3, np.float32),
                                        (6000e3, np.int16), (10e6, np.float32), (2048001.0, np.int16), (4095999.0, np.float32)])
def test_gpu_resampler_equals_the_restatement(gpu_ctx_factory, rate, dtype):
    rng = np.random.default_rng(7)
    n = 150000
    x = rng.integers(-20000, 20000, 2 * n).astype(np.int16)
    gain = 1.0 if dtype == np.int16 else 0.5
    ctx = gpu_ctx_factory(n_streams=2, fmt=1, ring_frames=4, max_frames=1)
    orc = ob.Resampler(rate)
    sizes = [2, 40, 1000, 30002, 77778] if rate == 4096e3 else [1, 3, 999, 30001, 77777]
    want, total = [], 0
    for a, b in _chunks(n, sizes):
        part = x[2 * a:2 * b].astype(dtype)
        got_n = ctx.push_resampled(1, part, rate, gain)
        y = orc.process(part.astype(np.float32))
        assert got_n == y.size // 2
        want.append(ob.to_s16(y, gain)); total += got_n
    want = np.concatenate(want)
    got = ctx.read_ring(1, 0, total)
    assert np.array_equal(got, want)
    assert np.abs(want.astype(np.int32)).max() > 1000                                   # not a trivial comparison


@pytest.mark.gpu
@pytest.mark.parametrize("rate,up,down", [(4096e3, 2, 1), (2400e3, 75, 64)])
def test_gpu_decodes_an_ensemble_recorded_at_another_rate(gpu_ctx_factory, rate, up, down):
    sub = ob.subch_layout(3, 64)
    nf = 7
    iq, fib, msc = ob.tx_generate(seed=33, n_frames=nf, subch=sub, delay=2500, fmt=1, snr_db=25.0, cfo_hz=1100.0, rms=2000.0)
    x = iq[0::2].astype(np.float64) + 1j * iq[1::2].astype(np.float64)
    xr = signal.resample_poly(x, up, down)                                              # what an SDR at `rate` would have delivered
    dev = np.stack([xr.real, xr.imag], axis=1).reshape(-1)
    dev = np.clip(np.rint(dev), -32768, 32767).astype(np.int16)
    ctx = gpu_ctx_factory(n_streams=1, fmt=1, ring_frames=nf + 2, max_frames=2)
    ctx.set_subchannels(0, sub)
    blk = 2 * 262144
    for a in range(0, dev.size, blk):
        ctx.push_resampled(0, dev[a:a + blk], rate)
    done = 0
    while done + 2 <= nf - 2:
        ctx.process(2)
        gf, gok = ctx.fib(0)
        assert gok.all() and np.array_equal(gf, fib[done:done + 2])
        done += 2
    assert ctx.state(0)["locked"] == 1


@pytest.mark.gpu
def test_gpu_resampled_stream_through_a_small_ring(gpu_ctx_factory):
    """4.096 Msps input streamed through a ring of four frames: the ring wraps three times, the windows that start near
    its end read the mirrored head the resampler kernel maintains (DABX_RING_MIRROR)"""
    sub = ob.subch_layout(2, 64)
    nf = 13
    iq, fib, _ = ob.tx_generate(seed=34, n_frames=nf, subch=sub, delay=900, fmt=1, snr_db=25.0, cfo_hz=-700.0, rms=2000.0)
    x = iq[0::2].astype(np.float64) + 1j * iq[1::2].astype(np.float64)
    xr = signal.resample_poly(x, 2, 1)
    dev = np.clip(np.rint(np.stack([xr.real, xr.imag], axis=1).reshape(-1)), -32768, 32767).astype(np.int16)
    ctx = gpu_ctx_factory(n_streams=1, fmt=1, ring_frames=4, max_frames=1)
    ctx.set_subchannels(0, sub)
    orc = ob.Resampler(4096e3)
    o = ob.Stream(fmt=1, subch=sub, ring_len=4 * ob.TF)
    pos, done = 0, 0
    chunk = 2 * 2 * 98304                                         # half a frame of output per push
    while pos < dev.size:
        part = dev[pos:pos + chunk]
        pos += chunk
        n = ctx.push_resampled(0, part, 4096e3)
        o.push(ob.to_s16(orc.process(part.astype(np.float32))))
        assert n == part.size // 4
        while ctx.frames_available() >= 1:
            ctx.process(1)
            r = o.process(1)
            gf, gok = ctx.fib(0)
            assert np.array_equal(ctx.sync(0), r["sync"]) and np.array_equal(ctx.fic_soft(0), r["fic_soft"])
            assert np.array_equal(gf, r["fib"]) and np.array_equal(gok, r["fib_ok"])
            assert gok.all() and np.array_equal(gf[0], fib[done])
            done += 1
    assert done >= nf - 2


@pytest.mark.gpu
@pytest.mark.parametrize("rate", [4096e3, 2400e3, 6000e3])
def test_gpu_resampler_asynchronous_pushes_from_pinned_memory(gpu_ctx_factory, rate):
    """dabx_push_resampled_from(DABX_SRC_PINNED): staging copies and converter kernels queued on the copy stream, no
    synchronisation per push; the ring holds what the restatement computes, bit for bit"""
    rng = np.random.default_rng(11)
    n = 400000
    x = rng.integers(-15000, 15000, 2 * n).astype(np.int16)
    ctx = gpu_ctx_factory(n_streams=1, fmt=1, ring_frames=4, max_frames=1)
    pinned = ctx.alloc_pinned(x.nbytes)
    buf = pinned.view(np.int16)
    buf[:] = x
    orc = ob.Resampler(rate)
    total, want = 0, []
    step = 2 * 65536
    for a in range(0, 2 * n, step):
        total += ctx.push_resampled_from(0, buf[a:a + step], rate, 1.0, kind=2)
        want.append(ob.to_s16(orc.process(x[a:a + step].astype(np.float32))))
    ctx.flush_copies()
    got = ctx.read_ring(0, 0, total)
    assert np.array_equal(got, np.concatenate(want)) and total > 100000
    ctx.free_pinned(pinned)


@pytest.mark.gpu
@pytest.mark.parametrize("rate,dtype", [(4096e3, np.int16), (2400e3, np.float32), (6000e3, np.int16), (2048e3, np.float32)])
def test_gpu_signal_level_equals_the_restatement(gpu_ctx_factory, rate, dtype):
    """dabx_enable_level: the reference's level detector (fast attack, slow release on |x|^2 of the input samples; a serial binary32
    recursion) run by one wave per push, bit-exact against oracle/dab_src.c over chunked pushes — a burst, silence, a weaker burst"""
    rng = np.random.default_rng(5)
    n = 60000
    env = np.concatenate([np.full(n // 3, 9000.0), np.zeros(n // 3), np.full(n - 2 * (n // 3), 1500.0)])
    x = (rng.normal(0, 1, 2 * n) * np.repeat(env, 2)).astype(np.float32)
    gain = 1.0
    if dtype == np.int16:
        x = np.clip(np.rint(x), -32768, 32767).astype(np.int16)
    else:
        x, gain = x / np.float32(32768.0), 8192.0
    ctx = gpu_ctx_factory(n_streams=2, fmt=1, ring_frames=4, max_frames=1)
    ctx.enable_level(1, True)
    orc = ob.Resampler(rate)
    sizes = [2, 40, 1000, 30002] if rate == 4096e3 else [1, 3, 999, 30001]
    seen = []
    for a, b in _chunks(n, sizes):
        part = x[2 * a:2 * b]
        ctx.push_resampled(1, part, rate, gain)
        orc.process(part.astype(np.float32))
        seen.append((ctx.level(1), orc.level()))
    assert all(g == o for g, o in seen), [s for s in seen if s[0] != s[1]][:3]
    assert seen[-1][1] > 0 and max(o for _, o in seen) > 1.1 * seen[-1][1]      # it rose with the burst and has been falling since
    assert ctx.level(0) == 0.0                                    # the other stream never asked
