"""Numpy model of the rotating lane<->state mapping used by k_viterbi
(abracadabra_amd/csrc/dabx_kernels.hip: viterbi_wave).  It mirrors the kernel's
index arithmetic one to one so the mapping, the branch-sign tables and the
coordinate-space traceback can be checked against the oracle on the CPU."""
import numpy as np

XV = [1, 2, 7, 8, 16, 32]
PM_INIT = -1000000
GEN = [0o133 & 63, 0o171 & 63, 0o145 & 63, 0o133 & 63]


def lane_coord(lane, k):
    b2 = (lane >> 2) & 1
    if k == 0:
        return (lane & 1) ^ b2
    if k == 1:
        return ((lane >> 1) & 1) ^ b2
    return (lane >> k) & 1


def conv_out0(state):
    o = 0
    for g in GEN:
        o = (o << 1) | (bin(state & g).count("1") & 1)
    return o


def sig_tables():
    sig = np.zeros((6, 64, 4), dtype=np.int64)
    for ph in range(6):
        for lane in range(64):
            st = 0
            for i in range(6):
                st |= lane_coord(lane, (i + ph) % 6) << i
            u = st & 1
            o = conv_out0(st)
            for j in range(4):
                neg = ((o >> (3 - j)) & 1) ^ u
                sig[ph, lane, j] = -1 if neg else 1
    return sig


def decode(x4):
    """x4: [nsteps, 4] int soft values (0 = punctured, |x| <= 31), nsteps a multiple of 6.  Returns decoded bits [nsteps].

    Mirrors viterbi_wave().  Path metrics are scaled by 128; the low seven bits of a lane's value Q carry a field that
    starts every group of six steps at 63.  One number per step and lane comes from the matrix core: with the soft values
    as (2 x0, 2 x1, 2 x2, 2 x3) — twice a value fits a byte because |x| <= 31 —, the lane's signs as (+-64, +-64, +-64, +-64)
    and the constant 1 << ph as the accumulator input, X = 128 M + (1 << ph).
    The kept candidate is Q + X, the one sent to the butterfly partner Q - X: the field of a survivor goes up by 1 << ph
    where it was kept and down by 1 << ph where it was received, so after the six steps it has gone from 63 to
    2 * (sum of the kept steps' 1 << ph): bits 1..6 are the keep/receive tags of ITS survivor path, they travelled with
    the path through the max.  The field never leaves 0..126 (63 - (2^ph - 1) >= 2^ph on the way down, 63 + 63 on the way
    up), so it never touches the metric, and a metric tie keeps the own path: the kept candidate's field is larger by at
    least 2.  At the end of a group the field is stored as a byte of the lane's decision word (four groups = 24 steps per word)
    and set back to 63.  The traceback walks six steps per look-up: position ^= ~tags."""
    nsteps = len(x4)
    assert nsteps % 6 == 0 and np.abs(x4).max(initial=0) <= 31
    sig = sig_tables()
    lanes = np.arange(64)
    coordA = lanes ^ (((lanes >> 2) & 1) * 3)
    Q = np.full(64, PM_INIT * 128, dtype=np.int64)
    Q[0] = 0
    Q += 63
    G = nsteps // 6
    nwords = (G + 3) // 4
    dec = np.zeros((nwords, 64), dtype=np.uint64)
    for w in range(nwords):
        ng = min(4, G - 4 * w)
        bits = np.zeros(64, dtype=np.uint64)
        for gi in range(ng):
            for ph in range(6):
                t = (4 * w + gi) * 6 + ph
                x = x4[t].astype(np.int64)
                a = 2 * x                                                   # MFMA A row (int8 each): twice the four soft values
                assert np.abs(a).max() <= 127
                b = 64 * sig[ph]                                            # MFMA B column of every lane: its four branch signs
                X = b @ a + (1 << ph)                                       # C operand: the tag, an inline constant
                keep = Q + X
                send = Q - X
                recv = send[lanes ^ XV[ph]]
                Q = np.maximum(keep, recv)
                assert np.abs(Q).max() < 2 ** 31 and ((Q & 127) <= 126).all()
            field = Q & 127
            assert (field & 1 == 0).all()
            bits |= (Q & 255).astype(np.uint64) << np.uint64(8 * gi)       # ds_write_b8: byte gi of the lane's decision word (bit 7: metric)
            Q = (Q & ~127) | 63                                            # v_and_or_b32
        dec[w, coordA] = bits                               # group gi of the word: its tags at bits 1 + 8 gi .. 6 + 8 gi
    out = np.zeros(nsteps, dtype=np.uint8)
    A = 0
    for w in range(nwords - 1, -1, -1):
        ng = min(4, G - 4 * w)
        for gi in range(ng - 1, -1, -1):
            h = (int(dec[w, A]) >> (1 + 8 * gi)) & 63
            for q in range(6):
                out[(4 * w + gi) * 6 + q] = (A >> q) & 1
            A ^= (~h) & 63
    return out
