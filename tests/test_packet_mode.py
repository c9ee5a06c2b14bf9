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
