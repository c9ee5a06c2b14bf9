"""Raw-file front end: the `.uff` XML header the reference's recorder writes
(inputdevicerecorder.cpp:195-263) is recognised, headerless `.raw` files pass through."""
import numpy as np
import pytest

import abracadabra_amd as aa
from oracle import binding as ob


def uff_header(fmt, n_values, freq_khz=225648):
    bits, cont = (8, "uint8") if fmt == 0 else (16, "int16")
    xml = ('<?xml version="1.0" encoding="utf-8"?>\n<SDR>\n <Recorder Name="AbracaDABra" Version="3.0.0"/>\n'
           ' <Device Name="rawfile" Model="test"/>\n <Time Value="2026-10-04 00:00:00" Unit="UTC"/>\n'
           f' <Sample>\n  <Samplerate Value="2048000" Unit="Hz"/>\n  <Channels Bits="{bits}" Container="{cont}" Ordering="LSB">\n'
           '   <Channel Value="I"/>\n   <Channel Value="Q"/>\n  </Channels>\n </Sample>\n'
           f' <Datablocks>\n  <Datablock Number="1" Count="{n_values}" Unit="Channel" Offset="2048">\n'
           f'   <Frequency Value="{freq_khz}" Unit="kHz"/>\n   <Modulation Value="DAB"/>\n  </Datablock>\n </Datablocks>\n</SDR>\n')
    raw = xml.encode()
    return raw + bytes(2048 - len(raw))


def test_probe_uff_and_raw():
    h = aa.rawfile_probe(uff_header(1, 1000) + bytes(100))
    assert h == dict(has_header=True, fmt=1, data_offset=2048, channel_count=1000, samplerate=2048000, frequency_khz=225648)
    assert aa.rawfile_probe(uff_header(0, 77))["fmt"] == 0
    noise = np.random.default_rng(0).integers(1, 256, 4096, dtype=np.uint8).tobytes()      # headerless u8 IQ: no zero byte, no <SDR>
    assert aa.rawfile_probe(noise) == dict(has_header=False, fmt=-1, data_offset=0, channel_count=0, samplerate=0, frequency_khz=0)
    assert not aa.rawfile_probe(b"<SDR" + bytes(10))["has_header"]                          # header without a usable container


@pytest.mark.gpu
def test_decode_from_uff_file(tmp_path, gpu_ctx_factory):
    sub = ob.subch_layout(1, 64)
    iq, fib, _ = ob.tx_generate(seed=91, n_frames=4, subch=sub, delay=700, fmt=1, snr_db=25.0, rms=2000.0)
    path = tmp_path / "rec.uff"
    path.write_bytes(uff_header(1, iq.size) + iq.tobytes())
    blob = path.read_bytes()
    info = aa.rawfile_probe(blob)
    assert info["has_header"] and info["fmt"] == 1
    samples = np.frombuffer(blob, dtype=np.int16, offset=info["data_offset"], count=info["channel_count"])
    ctx = gpu_ctx_factory(n_streams=1, fmt=info["fmt"], ring_frames=8, max_frames=2)
    ctx.push(0, samples)
    ctx.process(2)
    got, ok = ctx.fib(0)
    assert ok.all() and np.array_equal(got, fib[:2])


def _oracle_run_like_the_example(files, fmt, frames_per_step=4):
    """the step sequence of examples/decode_rawfiles.c on the CPU checker, all files in lock step: the first read is one step
    + one frame + 4096 samples, then one step's worth per read; every read is followed by a decode of the frames that are
    complete in EVERY file.  Returns per file (good FIBs, bad FIBs, final state)."""
    streams = [ob.Stream(fmt=fmt, subch=[], ring_len=(2 * frames_per_step + 4) * ob.TF) for _ in files]
    n_total = min(x.size // 2 for x in files)
    want, done = frames_per_step * ob.TF + ob.TF + 4096, 0
    good, bad = [0] * len(files), [0] * len(files)

    def available(o):
        st = o.state()
        extra = 0 if (st["locked"] and st["bad"] == 0) else 1
        nf = 0
        while nf < frames_per_step and st["pos"] + (nf + 1 + extra) * ob.TF + 4096 <= st["wr"]:
            nf += 1
        return nf

    while done < n_total:
        got = min(want, n_total - done)
        for o, x in zip(streams, files):
            o.push(x[2 * done:2 * (done + got)])
        done += got
        nf = min(available(o) for o in streams)
        if nf:
            for k, o in enumerate(streams):
                r = o.process(nf, want_soft=False)
                good[k] += int(r["fib_ok"].sum())
                bad[k] += int((r["fib_ok"] == 0).sum())
        elif got < want:
            break
        want = frames_per_step * ob.TF
    out = [(good[k], bad[k], o.state()) for k, o in enumerate(streams)]
    for o in streams:
        o.close()
    return out


@pytest.mark.gpu
@pytest.mark.parametrize("fmt", [0, 1])
def test_example_program_decodes_two_recordings(tmp_path, fmt):
    """examples/decode_rawfiles.c, built and RUN: two generated `.uff` recordings side by side on the
    GPU through the batch C ABI from plain C; per file the FIB counts, carrier offset, CIF count and ensemble identifier equal
    the CPU checker's run of the same reads."""
    import os
    import re
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "decode_rawfiles")
    subprocess.check_call(["gcc", "-std=c11", "-O2", "-I", os.path.join(root, "include"), "-o", exe, os.path.join(root, "examples", "decode_rawfiles.c"),
                           "-L", os.path.join(root, "abracadabra_amd"), "-l:libdabsdr_amd.so", "-Wl,-rpath," + os.path.join(root, "abracadabra_amd")])
    files, expect = [], []
    for k, (cfo, delay, eid) in enumerate(((1234.0, 900, 0x1111), (-2750.0, 140000, 0x2222))):
        iq, _, _ = ob.tx_generate(seed=300 + k, eid=eid, n_frames=14, subch=[], delay=delay, fmt=fmt, snr_db=22.0, cfo_hz=cfo,
                                  rms=28.0 if fmt == 0 else 3000.0)
        path = tmp_path / f"rec{k}.uff"
        path.write_bytes(uff_header(fmt, iq.size) + iq.tobytes())
        files.append(str(path))
        expect.append((cfo, eid, iq))
    out = subprocess.run([exe] + files, capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, (out.stdout, out.stderr)
    lines = [l for l in out.stdout.splitlines() if l.startswith("  ")]
    assert len(lines) == 2, out.stdout
    checker = _oracle_run_like_the_example([e[2] for e in expect], fmt)
    for line, (cfo, eid, iq), (good, bad, st) in zip(lines, expect, checker):
        m = re.search(r": (locked|no DAB signal), carrier offset (-?[0-9.]+) Hz, (\d+) CIFs, FIBs (\d+) good (\d+) bad, EId ([0-9A-F]{4})", line)
        assert m, line
        assert m.group(1) == "locked" and st["locked"] == 1
        assert abs(float(m.group(2)) - st["inc"] * 2048000.0 / 4294967296.0) < 0.06 and abs(float(m.group(2)) - cfo) < 2.0
        assert int(m.group(3)) == st["cif"] and int(m.group(4)) == good and int(m.group(5)) == bad and good >= 100 and bad == 0
        assert int(m.group(6), 16) == eid


@pytest.mark.gpu
def test_config2_one_raw_file_ensemble_full_chain(tmp_path, gpu_ctx_factory):
    """BASELINE configs[2] to the letter: ONE raw-file ensemble (the survey's shape: u8, 60 frames = 23.6 MB, 30 dB), full chain
    (sync + FFT + de-interleave + FIC Viterbi): (a) through the reference's 24-function API with the raw-file input convention
    (float(u8 - 128), rawfileinput.cpp:692): FIC lock, the transmitted ensemble, no FIB error in any period; (b) the same file
    through the batch ABI: every FIB of every frame equals the CPU checker's and the transmitted one, CRC flags included."""
    from legacy_host import NID, LegacyHost
    n_frames = 60
    iq, fib_tx, _ = ob.tx_generate(seed=2024, eid=0x1234, n_frames=n_frames, subch=[[0, 0, 3, 64]], delay=3210, snr_db=30.0, cfo_hz=1750.0)
    path = tmp_path / "ensemble.raw"
    path.write_bytes(iq.tobytes())
    raw = np.fromfile(path, dtype=np.uint8)
    assert aa.rawfile_probe(raw[:4096].tobytes())["has_header"] is False and raw.size == 2 * (3210 + n_frames * ob.TF)
    # (a) legacy API
    host = LegacyHost(raw.astype(np.float32) - 128.0, gate_at=(n_frames - 1) * ob.TF)
    try:
        host.tune(period_log2=3)
        host.wait_for(lambda e: e["nid"] == NID["SYNC_STATUS"] and e.get("level") == 3)
        host.wait_for(lambda e: e["nid"] == NID["PERIODIC"] and e["len"] and e["at"] >= (n_frames - 10) * ob.TF, timeout=60)
        host.L.dabsdrRequest_GetEnsemble(host.handle)
        ens = host.wait_for(lambda e: e["nid"] == NID["ENSEMBLE_INFO"] and e["status"] == 0)[-1]
        with host.lock:
            per = [e for e in host.events if e["nid"] == NID["PERIODIC"] and e["len"]]
    finally:
        host.close()
    assert ens["ueid"] & 0xFFFF == 0x1234
    locked = [e for e in per if e["level"] == 3]
    assert len(locked) >= 5 and all(e["fib_err"] == 0 for e in locked[1:]), per
    assert all(abs(e["foff"] / 10.0 - 1750.0) < 1.0 for e in locked)
    # (b) batch ABI, FIB by FIB
    ctx = gpu_ctx_factory(n_streams=1, fmt=0, ring_frames=n_frames + 2, max_frames=4)
    ctx.push(0, raw)
    o = ob.Stream(fmt=0, subch=[], ring_len=(n_frames + 2) * ob.TF)
    o.push(raw)
    f0 = 0
    while f0 + 4 + 1 <= n_frames:
        ctx.process(4)
        r = o.process(4, want_soft=False)
        gf, gok = ctx.fib(0)
        assert np.array_equal(gok, r["fib_ok"]) and np.array_equal(gf, r["fib"]) and np.array_equal(ctx.sync(0), r["sync"])
        assert gok.all() and np.array_equal(gf, fib_tx[f0:f0 + 4]), f"frame {f0}"
        f0 += 4
    assert f0 >= 56
