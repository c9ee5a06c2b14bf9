"""Randomised GPU-vs-oracle soak (run by hand through gpurun, not collected by pytest):
    python tests/gpu_soak.py [n_scenarios] [seed]
Every scenario draws sub-channel layouts (EEP A/B, UEP, now and then one sub-channel that fills the multiplex), formats,
SNRs down to where the Viterbi decoder and the RS decoder have real work, channel impairments (sampling-clock offset, a
second path, DC offset), a blanked stretch of samples (erased soft bits: survivors that do not merge, lock loss), DAB+
payloads with planted byte errors, odd step sizes — and compares every output of the HIP path with the oracle bit for bit."""
import sys
import time

import numpy as np

sys.path.insert(0, __file__.rsplit("/tests/", 1)[0])
import abracadabra_amd as aa                     # noqa: E402
from oracle import binding as ob                 # noqa: E402


def layout(rng):
    if rng.random() < 0.1:                       # one long codeword: up to 27 654 trellis steps
        while True:
            cand = [0, 0, int(rng.integers(1, 5)), int(rng.choice([384, 576, 768, 1152]))]
            try:
                if ob.any_profile(cand[1], cand[2], cand[3]).n_cu <= 864:
                    return [cand]
            except ValueError:
                pass
    subs, cu = [], 0
    for _ in range(int(rng.integers(1, 7))):
        kind = rng.integers(0, 3)
        if kind == 2:
            idx = int(rng.integers(0, 64))
            n_cu = ob.any_profile(2, idx, 0).n_cu
            entry = [cu, 2, idx, 0]
        else:
            kbps = int(rng.choice([8, 16, 32, 48, 64, 96, 128] if kind == 0 else [32, 64, 96]))
            level = int(rng.integers(1, 5))
            n_cu = ob.any_profile(int(kind), level, kbps).n_cu
            entry = [cu, int(kind), level, kbps]
        if cu + n_cu > 864:
            break
        subs.append(entry); cu += n_cu + int(rng.integers(0, 20))
    return subs or [[0, 0, 3, 64]]


def scenario(k, rng):
    fmt = int(rng.integers(0, 2))
    S = int(rng.integers(1, 4))
    F = int(rng.integers(1, 5))
    steps = int(rng.integers(3, 7))
    n_frames = F * steps + 2
    ctx = aa.Context(n_streams=S, fmt=fmt, ring_frames=n_frames + 2, max_frames=F, device=0)
    oracles, decs, subs_all, kb_all = [], [], [], []
    for s in range(S):
        subs = layout(rng)
        prof = [ob.any_profile(x[1], x[2], x[3]) for x in subs]
        kbps = [p.n_in // 24 for p in prof]
        plus = [kb % 8 == 0 and 8 <= kb <= 192 and rng.random() < 0.7 for kb in kbps]
        cols = []
        for i, kb in enumerate(kbps):
            if plus[i]:
                n_sf = 4 * n_frames // 5 + 1
                sf = ob.superframes(kb, n_sf, seed=int(rng.integers(1 << 30)), dac_rate=int(rng.integers(0, 2)), sbr=int(rng.integers(0, 2)))[0].reshape(n_sf, -1).copy()
                for _ in range(int(rng.integers(0, 40))):                       # byte errors before channel coding
                    sf[rng.integers(0, n_sf), rng.integers(0, sf.shape[1])] ^= rng.integers(1, 256)
                cols.append(np.concatenate([np.zeros((int(rng.integers(0, 5)), 3 * kb), np.uint8), sf.reshape(-1, 3 * kb)])[:4 * n_frames])
            else:
                cols.append(rng.integers(0, 256, (4 * n_frames, 3 * kb), dtype=np.uint8))
        payload = np.concatenate(cols, axis=1)
        snr = float(rng.choice([3.0, 5.0, 7.0, 9.0, 12.0, 20.0]))
        imp = {}
        if rng.random() < 0.4:
            imp["sco_ppm"] = float(rng.uniform(-150.0, 150.0))
        if rng.random() < 0.4:
            imp["echo"] = (int(rng.integers(10, 450)), float(rng.uniform(-2.0, 12.0)), float(rng.uniform(0.0, 6.28)))
        if rng.random() < 0.4:
            imp["dc"] = (float(rng.uniform(-8, 8)), float(rng.uniform(-8, 8))) if fmt == 0 else (float(rng.uniform(-500, 500)), float(rng.uniform(-500, 500)))
        iq, _, _ = ob.tx_generate(seed=int(rng.integers(1 << 30)), n_frames=n_frames, subch=subs, delay=int(rng.integers(0, 150000)),
                                  snr_db=snr, cfo_hz=float(rng.uniform(-3500, 3500)), fmt=fmt, rms=28.0 if fmt == 0 else 3000.0, payload=payload, **imp)
        if rng.random() < 0.3:                   # a blanked stretch: up to 1.5 frames of silence somewhere
            a = int(rng.integers(0, iq.size // 2 - 1000))
            b = min(iq.size // 2, a + int(rng.integers(500, 300000)))
            iq[2 * a:2 * b] = 128 if fmt == 0 else 0
        ctx.set_subchannels(s, subs)
        mask = sum(1 << i for i, p in enumerate(plus) if p)
        if mask:
            ctx.set_dabplus(s, mask)
        ctx.push(s, iq)
        o = ob.Stream(fmt=fmt, subch=subs, ring_len=ctx.ring_samples, ti_slots=64)
        o.push(iq)
        oracles.append(o); subs_all.append((subs, plus)); kb_all.append(kbps)
        decs.append([ob.SuperframeDecoder(kb) if p else None for kb, p in zip(kbps, plus)])
    n_sf_total = 0
    for step in range(steps):
        try:
            ctx.process(F)
        except aa.DabxError as e:                # a drifting clock or a re-acquisition can use the recording up early: both sides must say so
            assert "not enough samples" in str(e), (k, step, str(e))
            assert any(orc.process(F)["rc"] == -1 for orc in oracles), (k, step, "underrun on the GPU only")
            break
        for s, orc in enumerate(oracles):
            o = orc.process(F)
            assert o["rc"] in (0, F), (k, "rc")
            assert np.array_equal(ctx.sync(s), o["sync"]), (k, step, s, "sync")
            assert ctx.state(s)["slope"] == orc.state()["slope"] and ctx.state(s)["pos"] == orc.state()["pos"], (k, step, s, "state")
            gf, gok = ctx.fib(s)
            assert np.array_equal(gok, o["fib_ok"]), (k, step, s, "fib_ok")
            if o["rc"]:
                assert np.array_equal(ctx.fic_soft(s), o["fic_soft"]), (k, step, s, "fic_soft")
                assert np.array_equal(ctx.msc_soft(s), o["msc_soft"]), (k, step, s, "msc_soft")
                assert np.array_equal(gf, o["fib"]), (k, step, s, "fib")
            gm, gv = ctx.msc(s)
            assert np.array_equal(gv, o["msc_valid"]) and np.array_equal(gm[gv == 1], o["msc"][o["msc_valid"] == 1]), (k, step, s, "msc")
            off = 0
            for i, kb in enumerate(kb_all[s]):
                if decs[s][i] is not None:
                    frames = gm[:, :, off:off + 3 * kb][gv == 1]
                    orecs, odata = decs[s][i].push(frames)
                    grecs, gdata = ctx.superframes(s, i, kb)
                    assert grecs.tobytes() == orecs.tobytes() and np.array_equal(gdata, odata), (k, step, s, i, "superframes")
                    st = ctx.superframe_stats(s, i)
                    assert st == {key: decs[s][i].stats()[key] for key in st}, (k, step, s, i, "sf stats")
                    n_sf_total += len(grecs)
                off += 3 * kb
    ctx.close()
    return S, F, steps, n_sf_total


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = np.random.default_rng(seed)
    t0 = time.time()
    for k in range(n):
        info = scenario(k, rng)
        print(f"scenario {k}: streams/frames/steps/superframes = {info}  ok  ({time.time() - t0:.0f} s)", flush=True)
    print(f"soak ok: {n} scenarios, seed {seed}")


if __name__ == "__main__":
    main()
