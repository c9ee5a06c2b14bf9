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
    """x4: [nsteps, 4] int soft values (0 = punctured, |x| <= 63), nsteps a multiple of 6.  Returns decoded bits [nsteps].

    Mirrors viterbi_wave(): path metrics are scaled by 64 and the low six bits of a metric carry the keep/receive
    tags of the last (up to) six steps of ITS survivor path (bit ph = 1: kept at the step of phase ph), so the tags
    travel with the path through the max.  Both candidates come straight from the matrix core: with the soft values
    packed as (x0 + x3, x1, x2, 1) — generators 0 and 3 of the DAB mother code are the same polynomial, and the sum
    fits a byte because |x| <= 63 — and the lane's signs as (+-64, +-64, +-64, 1 << ph), the keep row gives
    64 M + tag and the row with a 0 in the last column gives 64 M, which is subtracted.  At the end of a
    group of six steps the six tags go into the lane's decision word (four groups = 24 steps per word) and are
    cleared.  A tie keeps the own path: the kept candidate has its tag bit set, the received one does not, all
    higher tag bits are zero.  The traceback walks six steps per look-up: position ^= ~tags."""
    nsteps = len(x4)
    assert nsteps % 6 == 0 and np.abs(x4).max(initial=0) <= 63
    sig = sig_tables()
    assert np.array_equal(sig[:, :, 0], sig[:, :, 3])                  # x0 and x3: the same generator (133 octal)
    lanes = np.arange(64)
    coordA = lanes ^ (((lanes >> 2) & 1) * 3)
    pm = np.full(64, PM_INIT * 64, dtype=np.int64)
    pm[0] = 0
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
                a_keep = np.array([x[0] + x[3], x[1], x[2], 1])        # MFMA A rows (int8 each)
                a_send = np.array([x[0] + x[3], x[1], x[2], 0])
                assert np.abs(a_keep).max() <= 127
                b = np.concatenate([64 * sig[ph][:, :3], np.full((64, 1), 1 << ph)], axis=1)   # MFMA B column of every lane
                keep = pm + b @ a_keep
                send = pm - b @ a_send
                recv = send[lanes ^ XV[ph]]
                pm = np.maximum(keep, recv)
                assert np.abs(pm).max() < 2 ** 31
            bits = (((pm & 63).astype(np.uint64)) << np.uint64(26)) | (bits >> np.uint64(6))     # v_alignbit_b32 bits, pm, bits, 6
            pm = pm & ~63
        bits >>= np.uint64(30 - 6 * ng)
        dec[w, coordA] = bits
    out = np.zeros(nsteps, dtype=np.uint8)
    A = 0
    for w in range(nwords - 1, -1, -1):
        ng = min(4, G - 4 * w)
        for gi in range(ng - 1, -1, -1):
            h = (int(dec[w, A]) >> (2 + 6 * gi)) & 63
            for q in range(6):
                out[(4 * w + gi) * 6 + q] = (A >> q) & 1
            A ^= (~h) & 63
    return out
