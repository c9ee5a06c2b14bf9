"""Known answers of ETSI EN 300 401 against the oracle's constants (CPU).

The reference holds no golden vectors for this path ("parity
unpinned"), so the oracle is pinned by the standard's own published values."""
import ctypes as C

import numpy as np

from oracle import binding as ob


def test_crc16_check_value():
    # CRC-16/GENIBUS (poly 1021, init FFFF, xorout FFFF) check value for "123456789"
    data = np.frombuffer(b"123456789", dtype=np.uint8).copy()
    assert ob.lib().dab_crc16(data.ctypes.data, 9) == 0xD64E


def test_prbs_first_bits():
    bits = np.zeros(32, dtype=np.uint8)
    ob.lib().dab_prbs.argtypes = [C.c_void_p, C.c_int]
    ob.lib().dab_prbs(bits.ctypes.data, 32)
    # EN 300 401 §10: first 16 bits of the energy dispersal sequence
    assert "".join(map(str, bits[:16])) == "0000011110111110"


def test_frequency_interleaver():
    k = np.zeros(1536, dtype=np.int16)
    ob.lib().dab_freq_interleaver.argtypes = [C.c_void_p]
    ob.lib().dab_freq_interleaver(k.ctypes.data)
    # PI(1..4) = 511, 1010, 1353, 1716 -> k = -513, -14, 329, 692 (EN 300 401 §14.6.1 example)
    assert list(k[:4]) == [-513, -14, 329, 692]
    assert sorted(k.tolist()) == [c for c in range(-768, 769) if c != 0]


def test_puncturing_vectors():
    L = ob.lib()
    L.dab_punct_vector.argtypes = [C.c_int, C.c_void_p]
    rows = {  # EN 300 401 table 29, literal rows
        1: "11001000100010001000100010001000", 8: "1100" * 8, 9: "1110" + "1100" * 7,
        13: "11101110111011001110110011101100", 16: "1110" * 8, 24: "1111" * 8,
        23: "1111" * 7 + "1110",
    }
    for pi in range(1, 25):
        v = np.zeros(32, dtype=np.uint8)
        L.dab_punct_vector(pi, v.ctypes.data)
        assert v.sum() == 8 + pi
        if pi in rows:
            assert "".join(map(str, v)) == rows[pi], pi


def test_mother_code_impulse_response():
    # a single 1 followed by zeros reads out the generator polynomials 133,171,145,133 (octal)
    L = ob.lib()
    L.dab_conv_encode.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    inp = np.zeros(1, dtype=np.uint8); inp[0] = 1
    out = np.zeros(4 * 7, dtype=np.uint8)
    L.dab_conv_encode(inp.ctypes.data, 1, out.ctypes.data)
    gens = [int("".join(map(str, out[g::4])), 2) for g in range(4)]
    assert gens == [0o133, 0o171, 0o145, 0o133]


def test_protection_profiles():
    fic_bits = 21 * 4 * 24 + 3 * 4 * 23 + 12
    assert fic_bits == 2304
    cases = {(0, 3, 64): 48, (0, 1, 8): 12, (0, 2, 8): 8, (0, 2, 32): 32, (0, 4, 192): 96, (1, 1, 32): 27, (1, 4, 32): 15,
             (1, 3, 64): 36, (0, 3, 1152): 864}
    for (opt, lvl, kbps), cu in cases.items():
        p = ob.eep_profile(opt, lvl, kbps)
        assert p.n_cu == cu and p.n_coded == 64 * cu and p.n_in == kbps * 24


def test_prs_is_unit_qpsk_on_1536_carriers():
    q = np.zeros(2048, dtype=np.int8)
    ob.lib().dab_prs_quadrants.argtypes = [C.c_void_p]
    ob.lib().dab_prs_quadrants(q.ctypes.data)
    assert (q >= 0).sum() == 1536 and q[0] == -1 and (q[769:1280] == -1).all()
    # EN 300 401 table 39/41: k = -768 uses h0[0] + n = 0 + 1; k = 1 uses h0[0] + 3
    assert q[(-768) & 2047] == 1 and q[1] == 3


def test_cordic_angles():
    L = ob.lib()
    rng = np.random.default_rng(5)
    for _ in range(200):
        x, y = (int(v) for v in rng.integers(-10**9, 10**9, 2))
        if x == 0 and y == 0:
            continue
        a = L.orx_cordic(y, x) / 2**32 * 2 * np.pi
        assert abs(np.angle(np.exp(1j * (a - np.arctan2(y, x))))) < 1e-6


def test_fft_against_numpy():
    rng = np.random.default_rng(0)
    x = (rng.integers(-128, 128, 2048) + 1j * rng.integers(-128, 128, 2048)).astype(np.complex64)
    X = ob.fft(x)
    ref = np.fft.fft(x.astype(np.complex128))
    assert np.abs(X - ref).max() <= 2e-6 * np.abs(ref).max()       # fp32 tolerance: 2e-6 of full scale


def test_uep_table_is_self_consistent():
    # EN 300 401 table 8 sub-channel sizes per (bit rate, protection level 5..1)
    sizes = {32: [16, 21, 24, 29, 35], 48: [24, 29, 35, 42, 52], 56: [29, 35, 42, 52], 64: [32, 42, 48, 58, 70],
             80: [40, 52, 58, 70, 84], 96: [48, 58, 70, 84, 104], 112: [58, 70, 84, 104], 128: [64, 84, 96, 116, 140],
             160: [80, 104, 116, 140, 168], 192: [96, 116, 140, 168, 208], 224: [116, 140, 168, 208, 232],
             256: [128, 168, 192, 232, 280], 320: {5: 160, 4: 208, 2: 280}, 384: {5: 192, 3: 280, 1: 416}}
    L = ob.lib()
    L.dab_profile_uep.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    seen = 0
    for idx in range(64):
        p, kbps, lvl = ob.Profile(), C.c_int(), C.c_int()
        assert L.dab_profile_uep(idx, C.byref(p), C.byref(kbps), C.byref(lvl)) == 0
        table = sizes[kbps.value]
        size = table[lvl.value] if isinstance(table, dict) else table[5 - lvl.value]
        assert p.n_cu == size and p.n_in == 24 * kbps.value        # 24 ms of audio per CIF
        seen += 1
    assert seen == 64
