"""X-PAD of DAB+ access units -> dynamic label segments and MOT data groups (csrc/pad.hpp), fed with PAD fields built
here bit by bit from ETSI EN 300 401 §7.4 (F-PAD, contents indicators, reversed byte order) and TS 102 563 §5.4
(data_stream_element).  CPU only; the end-to-end path is in tests/test_gpu_legacy_api.py."""
import ctypes as C

import numpy as np

import abracadabra_amd as aa

LEN = [4, 6, 8, 12, 16, 24, 32, 48]


def crc16(data):
    c = 0xFFFF
    for b in data:
        c ^= b << 8
        for _ in range(8):
            c = ((c << 1) ^ 0x1021) & 0xFFFF if c & 0x8000 else (c << 1) & 0xFFFF
    return ~c & 0xFFFF


def with_crc(b):
    c = crc16(b)
    return bytes(b) + bytes([c >> 8, c & 0xFF])


def dl_groups(text, toggle, charset=0):
    """dynamic label data groups (prefix + characters + CRC) of a message"""
    segs = [text[i:i + 16] for i in range(0, len(text), 16)]
    out = []
    for k, sg in enumerate(segs):
        b0 = (toggle << 7) | ((k == 0) << 6) | ((k == len(segs) - 1) << 5) | (len(sg) - 1)
        b1 = (charset << 4) if k == 0 else (k << 4)
        out.append(with_crc(bytes([b0, b1]) + sg.encode("latin-1")))
    return out


def xpad_var(subfields, ci=True):
    """variable-size X-PAD + F-PAD from [(application type, bytes)]; every subfield is padded to the next allowed size"""
    x = bytearray()
    if ci:
        for app, d in subfields:
            x.append((next(i for i, n in enumerate(LEN) if n >= len(d)) << 5) | app)
        if len(subfields) < 4:
            x.append(0)
    for app, d in subfields:
        n = next(n for n in LEN if n >= len(d)) if ci else len(d)
        x += bytes(d) + bytes(n - len(d))
    return bytes(reversed(x)) + bytes([0x20, 0x02 if ci else 0x00])


def xpad_short(app, d):
    x = (bytes([app]) + bytes(d)) if app is not None else bytes(d)
    assert len(x) == 4
    return bytes(reversed(x)) + bytes([0x10, 0x02 if app is not None else 0x00])


def au(pad, filler=b"\x21\x00\x49\x90"):
    assert 2 <= len(pad) < 255
    return bytes([0x80 | 0x00, len(pad)]) + pad + filler          # id_syn_ele 4 (DSE), tag 0, no byte alignment needed


def decode(aus):
    L = aa.load_library()
    L.dabsdr_amd_pad_decode.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
    blob = np.frombuffer(b"".join(bytes([len(a) & 0xFF, len(a) >> 8]) + a for a in aus), dtype=np.uint8).copy()
    out = np.zeros(1 << 16, dtype=np.uint8)
    stats = np.zeros(5, dtype=np.uint32)
    n = L.dabsdr_amd_pad_decode(blob.ctypes.data, blob.size, out.ctypes.data, out.size, stats.ctypes.data)
    assert n >= 0
    recs, pos = [], 0
    while pos < n:
        ln = int(out[pos + 2]) | (int(out[pos + 3]) << 8)
        recs.append((chr(out[pos]), int(out[pos + 1]), bytes(out[pos + 4:pos + 4 + ln])))
        pos += 4 + ln
    return recs, dict(zip(("pads", "dl_ok", "dl_crc_err", "dg_ok", "dg_crc_err"), stats.tolist()))


def spread(group, first_app, chunk):
    """a data group cut into X-PAD subfields of `chunk` bytes: start application type, then its continuation type"""
    parts = [group[i:i + chunk] for i in range(0, len(group), chunk)]
    return [(first_app if i == 0 else first_app + 1, p) for i, p in enumerate(parts)]


def test_dynamic_label_over_variable_xpad():
    text = "Now playing: GRAFT FM - bit exact on 256 CUs"
    groups = dl_groups(text, toggle=1)
    aus = []
    for g in groups:
        for sub in spread(g, 2, 8):
            aus.append(au(xpad_var([sub])))
    aus.insert(3, au(bytes([0x00, 0x00])))                         # an access unit without X-PAD in between
    recs, st = decode(aus)
    assert st["dl_ok"] == len(groups) == 3 and st["dl_crc_err"] == 0
    assert [r[2] for r in recs] == [g[:-2] for g in groups]        # prefix + characters, CRC stripped (dldecoder.cpp:82-190)
    label = b"".join(r[2][2:] for r in recs).decode("latin-1")
    assert label == text and recs[0][2][0] & 0x40 and recs[-1][2][0] & 0x20


def test_two_applications_in_one_xpad_and_command():
    clear = with_crc(bytes([0x80 | 0x10 | 0x01, 0x00]))            # C flag, command 0001: clear display
    g = dl_groups("SHORT", toggle=0)[0]
    recs, st = decode([au(xpad_var([(2, clear), (2, g[:6])])), au(xpad_var([(3, g[6:])]))])
    assert [r[2] for r in recs] == [clear[:-2], g[:-2]] and st["dl_ok"] == 2


def test_short_xpad_with_and_without_contents_indicator():
    g = dl_groups("ABCDEFG", toggle=0)[0]                           # 2 + 7 + 2 = 11 bytes: 3 + 4 + 4
    aus = [au(xpad_short(2, g[0:3])), au(xpad_short(None, g[3:7])), au(xpad_short(None, g[7:11]))]
    recs, st = decode(aus)
    assert [r[2] for r in recs] == [g[:-2]] and st["dl_ok"] == 1


def test_damaged_segment_is_dropped_and_the_next_one_still_arrives():
    g1, g2 = dl_groups("0123456789abcdef0123", toggle=1)
    bad = bytearray(g1)
    bad[5] ^= 0x01
    aus = [au(xpad_var([s])) for s in spread(bytes(bad), 2, 8)] + [au(xpad_var([s])) for s in spread(g2, 2, 8)]
    recs, st = decode(aus)
    assert st["dl_crc_err"] == 1 and [r[2] for r in recs] == [g2[:-2]]


def test_mot_data_group_with_length_indicator():
    body = bytes([0x40 | 0x04, 0x00]) + bytes(range(61))           # MSC data group header: CRC flag, type 4 (MOT body) ...
    group = with_crc(body)                                          # 65 bytes
    dgli = with_crc(bytes([len(group) >> 8, len(group) & 0xFF]))
    subs = spread(group, 12, 24)
    aus = [au(xpad_var([(1, dgli), subs[0]]))] + [au(xpad_var([s])) for s in subs[1:]]
    recs, st = decode(aus)
    assert st["dg_ok"] == 1 and recs == [("G", 12, group)]
    # without the length indicator the group cannot be delimited: nothing is delivered
    recs, st = decode([au(xpad_var([s])) for s in subs])
    assert recs == [] and st["dg_ok"] == 0


def test_garbage_does_not_crash():
    rng = np.random.default_rng(1)
    aus = [bytes(rng.integers(0, 256, int(n), dtype=np.uint8)) for n in rng.integers(1, 400, 300)]
    aus += [bytes([0x80, 250]) + bytes(5), bytes([0x80, 255]), bytes([0x80, 255, 10]) + bytes(40)]
    decode(aus)


# ---- MPEG Layer II (DAB audio): the PAD sits at the end of the audio frame, before and after the ScF-CRC

def mp2_frame(kbps, pad, mono=False, lsf=False):
    """a syntactically plausible Layer II frame of 3*kbps (48 kHz) or 6*kbps (24 kHz) bytes ending in X-PAD | ScF-CRC | F-PAD"""
    br = {False: [0, 32, 48, 56, 64, 80, 96, 112, 128, 160, 192, 224, 256, 320, 384], True: [0, 8, 16, 24, 32, 40, 48, 56, 64, 80, 96, 112, 128, 144, 160]}[lsf]
    hdr = bytes([0xFF, 0xF0 | ((0 if lsf else 1) << 3) | (2 << 1) | 1, (br.index(kbps) << 4) | (1 << 2), (3 if mono else 0) << 6])
    n = (6 if lsf else 3) * kbps
    scf = 4 if kbps // (1 if mono else 2) >= 56 else 2
    rng = np.random.default_rng(kbps)
    body = bytes(rng.integers(0, 256, n - 4 - len(pad) - scf, dtype=np.uint8))
    xpad, fpad = pad[:-2], pad[-2:]
    return hdr + body + xpad + bytes(scf) + fpad


def decode_mp2(frames):
    L = aa.load_library()
    L.dabsdr_amd_pad_decode_mp2.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
    blob = np.frombuffer(b"".join(bytes([len(a) & 0xFF, len(a) >> 8]) + a for a in frames), dtype=np.uint8).copy()
    out = np.zeros(1 << 16, dtype=np.uint8)
    stats = np.zeros(5, dtype=np.uint32)
    n = L.dabsdr_amd_pad_decode_mp2(blob.ctypes.data, blob.size, out.ctypes.data, out.size, stats.ctypes.data)
    assert n >= 0
    recs, pos = [], 0
    while pos < n:
        ln = int(out[pos + 2]) | (int(out[pos + 3]) << 8)
        recs.append(bytes(out[pos + 4:pos + 4 + ln]))
        pos += 4 + ln
    return recs, stats.tolist()


def test_dynamic_label_from_mpeg_layer2_frames():
    text = "DAB classic: Layer II with a dynamic label"
    for kbps, mono, lsf in ((128, False, False), (64, True, False), (48, False, True), (192, False, False)):
        groups = dl_groups(text, toggle=0)
        frames = [mp2_frame(kbps, xpad_var([sub]), mono, lsf) for g in groups for sub in spread(g, 2, 12)]
        frames.insert(2, mp2_frame(kbps, bytes([0x00, 0x00]), mono, lsf))     # no X-PAD in this frame
        recs, st = decode_mp2(frames)
        assert recs == [g[:-2] for g in groups], (kbps, mono, lsf)
        assert b"".join(r[2:] for r in recs).decode("latin-1") == text


def test_mp2_drc_comes_from_the_fpad_of_the_audio_frame():
    """EN 300 401 §7.4.1: F-PAD type 00 with byte L indicator 0001 carries six bits of DRC data (0.25 dB steps) in byte L; the
    reference passes it on as header.mp2DRC and the host scales the NEXT frame by 10^(DRC / 80) (audiodecoder.cpp:285-294, 326).
    48 kHz frames: one logical frame each; 24 kHz (LSF) frames: two, the F-PAD ends the second."""
    L = aa.load_library()
    L.dabsdr_amd_mp2_drc.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]

    def drc_of(frames, fb):
        blob = np.frombuffer(b"".join(frames), dtype=np.uint8).copy()
        n = blob.size // fb
        out = np.zeros(n, dtype=np.uint8)
        assert L.dabsdr_amd_mp2_drc(blob.ctypes.data, n, fb, out.ctypes.data) == n
        return out.tolist()

    def fpad(drc=None, xind=0):
        return bytes([(xind << 4) | (1 if drc is not None else 0), ((drc or 0) << 2)])

    # 48 kHz, 128 kbit/s: DRC 0, 10 (2.5 dB), none, 63 (15.75 dB), an F-PAD of another type
    fr = [mp2_frame(128, fpad(0)), mp2_frame(128, fpad(10)), mp2_frame(128, fpad(None)), mp2_frame(128, fpad(63)),
          mp2_frame(128, bytes([0x80 | 0x01, 40 << 2]))]
    assert drc_of(fr, 384) == [0, 10, 0, 63, 0]
    # 24 kHz, 64 kbit/s: the audio frame spans two logical frames of 192 bytes; only its end carries the F-PAD
    fr = [mp2_frame(64, fpad(7), lsf=True), mp2_frame(64, fpad(None), lsf=True), mp2_frame(64, fpad(33), lsf=True)]
    assert drc_of(fr, 192) == [0, 7, 0, 0, 0, 33]
    # a frame that is not the start of an audio frame (lost sync): nothing
    assert drc_of([bytes(384)], 384) == [0]
