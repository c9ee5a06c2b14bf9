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
    """x4: [nsteps, 4] int soft values (0 = punctured), nsteps a multiple of 6.  Returns decoded bits [nsteps].

    Mirrors viterbi_wave(): path metrics are scaled by 64 and the low six bits of a metric carry the keep/receive
    tags of the last (up to) six steps of ITS survivor path (bit ph = 1: kept at the step of phase ph), so the tags
    travel with the path through the max.  At the end of a group of six steps the six tags of every lane go into
    the lane's decision word (five groups = 30 steps per 32-bit word, newest on top) and are cleared.  A tie keeps
    the own path: the kept candidate has its tag bit set, the received one does not, all higher tag bits are zero.
    The traceback then walks six steps per look-up: position ^= ~tags."""
    nsteps = len(x4)
    assert nsteps % 6 == 0
    sig = sig_tables()
    lanes = np.arange(64)
    coordA = lanes ^ (((lanes >> 2) & 1) * 3)
    pm = np.full(64, PM_INIT * 64, dtype=np.int64)
    pm[0] = 0
    G = nsteps // 6
    nwords = (G + 4) // 5
    dec = np.zeros((nwords, 64), dtype=np.uint64)
    for w in range(nwords):
        ng = min(5, G - 5 * w)
        bits = np.zeros(64, dtype=np.uint64)
        for gi in range(ng):
            for ph in range(6):
                t = (5 * w + gi) * 6 + ph
                m = 64 * (sig[ph] @ x4[t].astype(np.int64))
                keep = (pm | (1 << ph)) + m
                send = pm - m
                recv = send[lanes ^ XV[ph]]
                pm = np.maximum(keep, recv)
                assert np.abs(pm).max() < 2 ** 31
            bits = (((pm & 63).astype(np.uint64)) << np.uint64(26)) | (bits >> np.uint64(6))     # v_alignbit_b32 bits, pm, bits, 6
            pm = pm & ~63
        bits >>= np.uint64(6 * (5 - ng))
        dec[w, coordA] = bits
    out = np.zeros(nsteps, dtype=np.uint8)
    A = 0
    for w in range(nwords - 1, -1, -1):
        ng = min(5, G - 5 * w)
        for gi in range(ng - 1, -1, -1):
            h = (int(dec[w, A]) >> (2 + 6 * gi)) & 63
            for q in range(6):
                out[(5 * w + gi) * 6 + q] = (A >> q) & 1
            A ^= (~h) & 63
    return out
