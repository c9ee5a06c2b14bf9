"""Packet mode of the MSC -> data groups (csrc/packet.hpp), fed with packets built here from ETSI EN 300 401 §5.3.2
(header bit layout, CRC) — CPU only."""
import ctypes as C

import numpy as np

import abracadabra_amd as aa
from tests.test_pad import crc16


def packets(group, addr, size=24, ci0=0):
    """cut a data group into packets of `size` bytes for packet address addr"""
    room = size - 5
    parts = [group[i:i + room] for i in range(0, len(group), room)]
    out = []
    for k, d in enumerate(parts):
        fl = (2 if k == 0 else 0) | (1 if k == len(parts) - 1 else 0)
        hdr = bytes([((size // 24 - 1) << 6) | (((ci0 + k) & 3) << 4) | (fl << 2) | (addr >> 8), addr & 0xFF, len(d)])
        body = hdr + d + bytes(room - len(d))
        c = crc16(body)
        out.append(body + bytes([c >> 8, c & 0xFF]))
    return out


def padding(size=24):
    body = bytes([(size // 24 - 1) << 6, 0, 0]) + bytes(size - 5)
    c = crc16(body)
    return body + bytes([c >> 8, c & 0xFF])


def frames_of(pkts, frame_bytes):
    """pack packets into logical frames, padding packets where nothing fits"""
    frames, cur = [], b""
    for p in pkts:
        if len(cur) + len(p) > frame_bytes:
            while len(cur) < frame_bytes:
                cur += padding(24)
            frames.append(cur); cur = b""
        cur += p
    while len(cur) < frame_bytes:
        cur += padding(24)
    frames.append(cur)
    return np.frombuffer(b"".join(frames), dtype=np.uint8).reshape(-1, frame_bytes).copy()


def decode(frames, address=-1):
    L = aa.load_library()
    L.dabsdr_amd_packet_decode.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
    out = np.zeros(1 << 16, dtype=np.uint8)
    stats = np.zeros(4, dtype=np.uint32)
    frames = np.ascontiguousarray(frames)
    n = L.dabsdr_amd_packet_decode(frames.ctypes.data, frames.shape[0], frames.shape[1], address, out.ctypes.data, out.size, stats.ctypes.data)
    assert n >= 0
    recs, pos = [], 0
    while pos < n:
        addr = int(out[pos]) | (int(out[pos + 1]) << 8)
        ln = int(out[pos + 2]) | (int(out[pos + 3]) << 8)
        recs.append((addr, bytes(out[pos + 4:pos + 4 + ln])))
        pos += 4 + ln
    return recs, dict(zip(("packets", "crc_err", "groups", "dropped"), stats.tolist()))


def test_data_groups_of_two_interleaved_addresses():
    rng = np.random.default_rng(1)
    g1 = bytes(rng.integers(0, 256, 150, dtype=np.uint8))
    g2 = bytes(rng.integers(0, 256, 61, dtype=np.uint8))
    g3 = bytes(rng.integers(0, 256, 19, dtype=np.uint8))                 # fits one packet: first and last at once
    a, b = packets(g1, 777, 48), packets(g2, 5, 24)
    mixed = [a[0], b[0], a[1], b[1], a[2], b[2], a[3], b[3]] + packets(g3, 777, 24, ci0=len(a))
    frames = frames_of(mixed, 96)                                       # a 32 kbit/s sub-channel
    recs, st = decode(frames)
    assert recs == [(777, g1), (5, g2), (777, g3)] and st["crc_err"] == 0 and st["groups"] == 3
    assert decode(frames, address=777)[0] == [(777, g1), (777, g3)]    # what FIG 0/3 selects for a component


def test_damaged_and_missing_packets_drop_only_their_group():
    rng = np.random.default_rng(2)
    g1, g2, g3 = (bytes(rng.integers(0, 256, n, dtype=np.uint8)) for n in (80, 80, 80))
    p1, p2, p3 = packets(g1, 9), packets(g2, 9, ci0=1), packets(g3, 9, ci0=2)
    bad = bytearray(p1[2]); bad[7] ^= 0x10; p1[2] = bytes(bad)           # CRC error inside group 1
    del p2[1]                                                            # a packet of group 2 lost: continuity index jumps
    recs, st = decode(frames_of(p1 + p2 + p3, 72))
    assert recs == [(9, g3)] and st["crc_err"] == 1 and st["dropped"] == 2


def test_packet_sizes_and_garbage():
    rng = np.random.default_rng(3)
    for size in (24, 48, 72, 96):
        g = bytes(rng.integers(0, 256, 300, dtype=np.uint8))
        assert decode(frames_of(packets(g, 1000, size), 96 * 3))[0] == [(1000, g)]
    junk = rng.integers(0, 256, (50, 120), dtype=np.uint8)
    recs, st = decode(junk)
    assert recs == [] and st["groups"] == 0


# ---- FEC frames (EN 300 401 §5.3.5): RS(204,188) over 94 units of packets, parity in nine FEC packets of address 1022 ----

def _gf():
    exp, log = [0] * 512, [0] * 256
    x = 1
    for i in range(255):
        exp[i], log[x] = x, i
        x <<= 1
        if x & 0x100:
            x ^= 0x11D
    for i in range(255, 512):
        exp[i] = exp[i - 255]
    return exp, log


GF_EXP, GF_LOG = _gf()


def gf_mul(a, b):
    return GF_EXP[GF_LOG[a] + GF_LOG[b]] if a and b else 0


def rs_generator(nroots=16):
    g = [1]                                                   # highest power first; g(x) = prod (x + 2^i), i = 0..15
    for i in range(nroots):
        g = [a ^ b for a, b in zip(g + [0], [0] + [gf_mul(c, GF_EXP[i]) for c in g])]
    return g


RS_GEN = rs_generator()


def rs_parity(data):
    """remainder of data(x) * x^16 by g(x): the 16 parity bytes of the (shortened) systematic code"""
    rem = [0] * 16
    for d in data:
        fb = d ^ rem[0]
        rem = [rem[i + 1] ^ gf_mul(fb, RS_GEN[i + 1]) for i in range(15)] + [gf_mul(fb, RS_GEN[16])]
    return rem


def rs_syndromes(cw):
    out = []
    for i in range(16):
        s = 0
        for c in cw:
            s = gf_mul(s, GF_EXP[i]) ^ c
        out.append(s)
    return out


def fec_frame(pkts):
    """94 units of packets (padded with padding packets) + the nine FEC packets; returns a list of 24-byte units' worth of packets"""
    body = b"".join(pkts)
    assert len(body) <= 94 * 24 and len(body) % 24 == 0
    body += padding(24) * ((94 * 24 - len(body)) // 24)
    table = np.frombuffer(body, dtype=np.uint8).reshape(188, 12)          # filled column by column: byte n -> row n % 12, column n // 12
    par = np.array([rs_parity(table[:, r].tolist()) for r in range(12)], dtype=np.uint8)     # [row][16]
    rs_bytes = par.T.reshape(-1).tobytes() + bytes(6)                    # RS Data Table read column by column, 6 bytes of padding
    fec = [bytes([(k << 2) | (1022 >> 8), 1022 & 0xFF]) + rs_bytes[22 * k:22 * k + 22] for k in range(9)]
    return body, fec


def fec_stream(groups, addr, size=24, per_frame=20):
    """data groups -> packets -> FEC frames -> byte stream; a few groups per FEC frame"""
    stream, ci = b"", 0
    for i in range(0, len(groups), per_frame):
        pk = []
        for g in groups[i:i + per_frame]:
            p = packets(g, addr, size, ci0=ci)
            ci += len(p)
            pk += p
        body, fec = fec_frame(pk)
        stream += body + b"".join(fec)
    return stream


def to_frames(stream, frame_bytes):
    stream = bytes(stream)
    stream += padding(24) * ((-len(stream)) % frame_bytes // 24)
    return np.frombuffer(stream, dtype=np.uint8).reshape(-1, frame_bytes).copy()


def decode_fec(frames, address=-1):
    L = aa.load_library()
    L.dabsdr_amd_packet_decode_fec.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
    out = np.zeros(1 << 18, dtype=np.uint8)
    stats = np.zeros(7, dtype=np.uint32)
    frames = np.ascontiguousarray(frames)
    n = L.dabsdr_amd_packet_decode_fec(frames.ctypes.data, frames.shape[0], frames.shape[1], address, out.ctypes.data, out.size, stats.ctypes.data)
    assert n >= 0
    recs, pos = [], 0
    while pos < n:
        addr = int(out[pos]) | (int(out[pos + 1]) << 8)
        ln = int(out[pos + 2]) | (int(out[pos + 3]) << 8)
        recs.append((addr, bytes(out[pos + 4:pos + 4 + ln])))
        pos += 4 + ln
    return recs, dict(zip(("packets", "crc_err", "groups", "dropped", "fec_frames", "fec_corrected", "fec_failed_rows"), stats.tolist()))


def test_rs_204_188_encoder_is_the_standards_code():
    """the test's own encoder: generator roots 2^0..2^15 over 0x11D — every codeword has zero syndromes, the generator is monic of
    degree 16 with g(1) = 0, and the code is the shortened (255,239): 51 leading zeros change nothing"""
    rng = np.random.default_rng(4)
    assert len(RS_GEN) == 17 and RS_GEN[0] == 1
    acc = 0
    for c in RS_GEN:
        acc ^= c
    assert acc == 0                                                      # x = 1 = 2^0 is a root
    d = rng.integers(0, 256, 188).tolist()
    cw = d + rs_parity(d)
    assert rs_syndromes(cw) == [0] * 16
    assert rs_parity([0] * 51 + d) == rs_parity(d)


def test_fec_frames_correct_what_the_packet_crc_would_drop():
    rng = np.random.default_rng(5)
    groups = [bytes(rng.integers(0, 256, int(rng.integers(20, 90)), dtype=np.uint8)) for _ in range(100)]
    stream = bytearray(fec_stream(groups, 300))
    n_frames_fec = len(stream) // (103 * 24)
    assert n_frames_fec == 5
    clean = to_frames(stream, 96)      # 32 kbit/s: FEC frames and logical frames not aligned
    recs0, st0 = decode_fec(clean)
    assert [r[1] for r in recs0] == groups and st0["crc_err"] == 0 and st0["fec_frames"] == 4     # the first frame is what locks the decoder
    # up to 8 byte errors in every row of every FEC frame after the first (rows = bytes r, r+12, ... of tables and parity)
    planted = 0
    for f in range(1, n_frames_fec):
        base = f * 103 * 24
        for r in range(12):
            cols = rng.choice(204, size=int(rng.integers(1, 9)), replace=False)
            for c in cols:
                off = base + c * 12 + r if c < 188 else base + 94 * 24 + ((c - 188) * 12 + r) // 22 * 24 + 2 + ((c - 188) * 12 + r) % 22
                stream[off] ^= int(rng.integers(1, 256))
                planted += 1
    noisy = to_frames(stream, 96)
    recs, st = decode_fec(noisy)
    assert [r[1] for r in recs] == groups
    assert st["fec_corrected"] == planted and st["fec_failed_rows"] == 0 and st["crc_err"] == 0 and st["dropped"] == 0
    # the same stream without the outer code loses the damaged packets
    recs_plain, st_plain = decode(noisy)
    assert st_plain["crc_err"] > 20 and len(recs_plain) < len(groups)


def test_fec_rows_beyond_eight_errors_fall_back_to_the_packet_crc():
    rng = np.random.default_rng(6)
    groups = [bytes(rng.integers(0, 256, 60, dtype=np.uint8)) for _ in range(60)]
    stream = bytearray(fec_stream(groups, 41))
    base = 1 * 103 * 24
    for c in rng.choice(188, size=12, replace=False):                    # row 3 of the second FEC frame: 12 errors
        stream[base + int(c) * 12 + 3] ^= 0x5A
    recs, st = decode_fec(to_frames(stream, 72))
    assert st["fec_failed_rows"] == 1 and st["fec_corrected"] == 0
    got = [r[1] for r in recs]
    assert 0 < len(groups) - len(got) <= 12 and all(g in groups for g in got)
    assert st["crc_err"] == len(groups) - len(got) or st["crc_err"] + st["dropped"] >= len(groups) - len(got)


def test_fec_lock_is_lost_and_found_again():
    rng = np.random.default_rng(7)
    groups = [bytes(rng.integers(0, 256, 50, dtype=np.uint8)) for _ in range(120)]
    stream = fec_stream(groups, 77)                                      # six FEC frames
    cut = stream[:2 * 103 * 24 + 24 * 31] + stream[3 * 103 * 24 + 24 * 40:]     # a stretch is missing: the frame structure slips by 9 + 103 units
    frames = to_frames(cut, 96)
    recs, st = decode_fec(frames)
    got = [r[1] for r in recs]
    assert all(g in groups for g in got) and len(set(got)) == len(got)   # nothing invented, nothing delivered twice
    assert got[:20] == groups[:20] and got[-20:] == groups[-20:]         # before the gap and after re-locking
    assert st["fec_frames"] == 2 and st["fec_failed_rows"] == 0      # frames 1 and 5; frame 4 is what re-locks the decoder
