"""Quick GPU bring-up script (not a test): prints stage-by-stage parity against the oracle."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import abracadabra_amd as aa
from oracle import binding as ob

rng = np.random.default_rng(0)
ctx = aa.Context(n_streams=2, fmt=0, ring_frames=16, max_frames=4)
# FFT
x = (rng.integers(-128, 128, (3, 2048)) + 1j * rng.integers(-128, 128, (3, 2048))).astype(np.complex64)
g = ctx.fft2048(x)
o = np.stack([ob.fft(v) for v in x]).astype(np.complex64)
print("fft bit-exact:", np.array_equal(g.view(np.uint32), o.view(np.uint32)), "max abs diff", np.abs(g - o).max())
# Viterbi linear
for kind, prof, ncoded in ((0, (0, 3, 64), 2304), (1, (0, 3, 64), 3072), (1, (0, 1, 8), 768), (1, (1, 4, 32), 960)):
    soft = rng.integers(-31, 32, (5, ncoded)).astype(np.int8)
    gv = ctx.viterbi(soft, kind, *prof)
    ov = np.stack([ob.decode_linear(s, kind, *prof) for s in soft])
    print("viterbi", kind, prof, "exact:", np.array_equal(gv, ov), (gv != ov).sum())
# full chain
sub = ob.subch_layout(18, 64)
outs = []
for s in range(2):
    iq, fib, msc = ob.tx_generate(seed=10 + s, eid=0x1000 + s, n_frames=10, subch=sub, delay=3000 * (s + 1), snr_db=15.0 + 10 * s, cfo_hz=-2345.0 + 4000 * s)
    ctx.set_subchannels(s, sub)
    ctx.push(s, iq)
    orc = ob.Stream(subch=sub); orc.push(iq)
    outs.append((orc, fib, msc))
for step in range(2):
    t = time.time(); ctx.process(4); dt = time.time() - t
    for s in range(2):
        orc, fib, msc = outs[s]
        o = orc.process(4)
        gs = ctx.sync(s)
        print(f"step {step} stream {s} ({dt*1e3:.1f} ms): sync equal", np.array_equal(gs, o["sync"]))
        if not np.array_equal(gs, o["sync"]):
            print(" gpu", gs); print(" cpu", o["sync"])
        print("   fic_soft equal", np.array_equal(ctx.fic_soft(s), o["fic_soft"]), " msc_soft equal", np.array_equal(ctx.msc_soft(s), o["msc_soft"]))
        gf, gok = ctx.fib(s)
        print("   fib equal", np.array_equal(gf, o["fib"]), "fib_ok", gok.sum(), o["fib_ok"].sum(), "tx match", np.array_equal(gf, fib[4 * step:4 * step + 4]))
        gm, gv = ctx.msc(s)
        print("   msc valid equal", np.array_equal(gv, o["msc_valid"]), "msc equal (valid)", np.array_equal(gm[gv == 1], o["msc"][o["msc_valid"] == 1]))
        print("   state", ctx.state(s), orc.state())
