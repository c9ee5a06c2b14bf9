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
