"""FIG database: the FIGs beyond the basic set (0/3, 0/5, 0/8, 0/13, 0/14, 0/17, 0/18, 0/19), fed as hand-built
FIBs laid out bit by bit from ETSI EN 300 401 §6.3, §8.1 (independent of oracle/dab_tx.c).  CPU only."""
import ctypes as C

import numpy as np

import abracadabra_amd as aa


def crc16(data):
    c = 0xFFFF
    for b in data:
        c ^= b << 8
        for _ in range(8):
            c = ((c << 1) ^ 0x1021) & 0xFFFF if c & 0x8000 else (c << 1) & 0xFFFF
    return ~c & 0xFFFF


def fib(*figs):
    body = b"".join(figs)
    assert len(body) <= 30
    body = body + (b"\xff" if len(body) < 30 else b"") + bytes(max(0, 29 - len(body)))
    c = crc16(body)
    return body + bytes([c >> 8, c & 0xFF])


def fig0(ext, payload, pd=0):
    return bytes([len(payload) + 1, (pd << 5) | ext]) + bytes(payload)


def dump(fibs):
    L = aa.load_library()
    L.dabsdr_amd_fig_dump.argtypes = [C.c_void_p, C.c_int, C.c_char_p, C.c_int]
    buf = C.create_string_buffer(16384)
    flat = np.frombuffer(b"".join(fibs), dtype=np.uint8).copy()
    assert L.dabsdr_amd_fig_dump(flat.ctypes.data, len(fibs), buf, 16384) > 0
    return buf.value.decode()


def test_extended_figs():
    sid, sid2 = 0x1A01, 0x1A02
    f_sub = fig0(1, [0x00, 0x00, 0x80 | (0 << 4) | (2 << 2), 48,            # SubCh 0: start 0, EEP 3-A, 48 CU
                     (5 << 2) | 0, 48, 0x80 | (0 << 4) | (2 << 2), 24])     # SubCh 5: start 48, EEP 3-A, 24 CU (packet data)
    f_srv = fig0(2, [sid >> 8, sid & 0xFF, 0x02,                            # two components
                     0x3F, (0 << 2) | 0x02,                                 # TMId 0, DAB+ audio, SubCh 0, primary
                     0xC0 | (0x123 >> 6), ((0x123 & 0x3F) << 2) | 0x00,     # TMId 3, SCId 0x123, secondary
                     sid2 >> 8, sid2 & 0xFF, 0x01, 0x3F, (0 << 2) | 0x02])
    f_pkt = fig0(3, [0x12, 0x30, 0x00 | 60, (5 << 2) | (777 >> 8), 777 & 0xFF])      # SCId 0x123: DG used, DSCTy 60, SubCh 5, address 777
    f_lang = fig0(5, [0x00 | 0, 0x09, 0x80 | 0x01, 0x23, 0x08])                      # SubCh 0 English (0x09); SCId 0x123 German (0x08)
    f_glob = fig0(8, [sid >> 8, sid & 0xFF, 0x00 | 0, 0x00 | 0,                      # SCIdS 0 <-> SubCh 0
                      sid >> 8, sid & 0xFF, 0x00 | 3, 0x80 | 0x01, 0x23])            # SCIdS 3 <-> SCId 0x123 (long form)
    f_app = fig0(13, [sid >> 8, sid & 0xFF, (0 << 4) | 1, 0x002 >> 3, ((0x002 & 7) << 5) | 2, 0x0C, 0x3C,   # SLS over X-PAD
                      sid >> 8, sid & 0xFF, (3 << 4) | 1, 0x007 >> 3, ((0x007 & 7) << 5) | 0])              # SPI on the packet component
    f_fec = fig0(14, [(5 << 2) | 1])
    f_pty = fig0(17, [sid >> 8, sid & 0xFF, 0x00, 10, sid2 >> 8, sid2 & 0xFF, 0x80, 3])
    f_asu = fig0(18, [sid >> 8, sid & 0xFF, 0x00, 0x03, 2, 1, 7])                    # alarm + traffic, clusters 1 and 7
    f_asw = fig0(19, [7, 0x00, 0x02, 0x80 | 0])                                      # cluster 7: traffic announcement on SubCh 0, new
    fibs = [fib(f_sub, f_srv), fib(f_pkt, f_lang), fib(f_glob, f_app[:9]), fib(f_app[:2] + f_app[9:]), fib(f_fec, f_pty, f_asu, f_asw)]
    # FIG 0/13 split by hand: both halves need their own header
    fibs[2] = fib(f_glob, fig0(13, list(f_app[2:9])))
    fibs[3] = fib(fig0(13, list(f_app[9:])))
    text = dump(fibs)
    assert "service sid=1A01 label='' ncomp=2 [tmid=0 ty=63 subch=0 ps=1] [tmid=3 ty=0 subch=-1 ps=0]" in text
    assert "pty=10 dyn=0 asu=0003 clusters=1,7," in text
    assert "pty=3 dyn=1" in text
    assert "comp scids=0 scid=-1 apps=002:0C3C," in text
    assert "comp scids=3 scid=291 apps=007:," in text
    assert "language subch=0 code=9" in text and "language scid=291 code=8" in text
    assert "packet scid=291 subch=5 dscty=60 addr=777 dg=1" in text
    assert "fec subch=5 scheme=1" in text
    assert "switching cluster=7 flags=0002 subch=0 new=1" in text


def test_malformed_figs_are_skipped():
    """truncated entries and lengths that run past the FIB must not be read past the end"""
    bad = [fib(bytes([0x1F, 0x0D]) + bytes(28)),                       # FIG 0/13 claiming 31 bytes
           fib(fig0(13, [0x1A, 0x01, 0x0F])),                          # 15 user applications announced, none present
           fib(fig0(3, [0x12, 0x31, 0x3C, 0x14])),                     # FIG 0/3 cut short
           fib(fig0(19, [7, 0, 2])),                                   # FIG 0/19 cut short
           fib(fig0(18, [0x1A, 0x01, 0, 3, 31]))]                      # 31 clusters announced
    text = dump(bad)
    assert "packet" not in text and "switching" not in text


def test_multiplex_reconfiguration_is_applied_at_its_cif_count():
    """EN 300 401 §6.5: the next configuration travels with C/N = 1 and must not touch the current one until the CIF count
    announced by FIG 0/0 (change flags + occurrence change) is reached; labels and other service information survive."""
    sid = 0x1A01

    def fig00(cif, change=0, occ=None):
        b = [0x10, 0xAB, (change << 6) | (cif // 250), cif % 250] + ([occ] if occ is not None else [])
        return fig0(0, b)

    def mci(cn, start, size):
        sub = bytes([len([0] * 4) + 1, (cn << 7) | 1, 0x00 | (start >> 8), start & 0xFF, 0x80 | (2 << 2) | (size >> 8), size & 0xFF])   # SubCh 0, EEP 3-A
        srv = bytes([6, (cn << 7) | 2, sid >> 8, sid & 0xFF, 0x01, 0x3F, 0x02])
        return sub, srv

    label = bytes([0x20 | 21, 0x01, sid >> 8, sid & 0xFF]) + b"STAYS THE SAME  " + bytes([0xFF, 0x00])
    cur = mci(0, 0, 48)
    nxt = mci(1, 100, 72)                                            # the sub-channel moves and grows: 64 -> 96 kbit/s
    before = [fib(fig00(100), *cur), fib(label)]
    during = [fib(fig00(120, change=1, occ=130), *cur), fib(*nxt)]
    at = [fib(fig00(130, change=1, occ=130), *mci(0, 100, 72))]
    text = dump(before + during)
    assert "subch id=0 start=0 size=48" in text and "reconfiguration pending=1 next=1 applied=0" in text
    text = dump(before + during + at)
    assert "subch id=0 start=100 size=72 opt=0 level=3 kbps=96" in text and "label='STAYS THE SAME  '" in text
    assert "reconfiguration pending=1 next=0 applied=1" in text
    # a receiver that missed the FIB of the change itself applies it when the announcement is withdrawn
    text = dump(before + during + [fib(fig00(140), *mci(0, 100, 72))])
    assert "start=100 size=72" in text and "applied=1" in text
