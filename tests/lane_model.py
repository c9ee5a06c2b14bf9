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
    """x4: [nsteps, 4] int soft values (0 = punctured). Returns decoded bits [nsteps]."""
    nsteps = len(x4)
    sig = sig_tables()
    lanes = np.arange(64)
    coordA = lanes ^ (((lanes >> 2) & 1) * 3)
    pm = np.full(64, PM_INIT, dtype=np.int64)
    pm[0] = 0
    nhb = (nsteps + 31) >> 5
    dec = np.zeros((nhb, 64), dtype=np.uint64)
    for hb in range(nhb):
        cnt = min(32, nsteps - hb * 32)
        bits = np.zeros(64, dtype=np.uint64)
        for j in range(cnt):
            t = hb * 32 + j
            ph = t % 6
            m = sig[ph] @ x4[t].astype(np.int64)
            keep = pm + m
            send = pm - m
            recv = send[lanes ^ XV[ph]]
            d = (recv > keep).astype(np.uint64)
            bits = ((bits << np.uint64(1)) | d) & np.uint64(0xFFFFFFFF)
            pm = np.maximum(keep, recv)
        word = (bits << np.uint64(32 - cnt)) & np.uint64(0xFFFFFFFF)
        dec[hb, coordA] = word
    out = np.zeros(nsteps, dtype=np.uint8)
    A = 0
    for hb in range(nhb - 1, -1, -1):
        cnt = min(32, nsteps - hb * 32)
        ph = (hb * 32 + cnt - 1) % 6
        for j in range(cnt - 1, -1, -1):
            w = int(dec[hb, A])
            d = (w >> (31 - j)) & 1
            out[hb * 32 + j] = (A >> ph) & 1
            A ^= d << ph
            ph = 5 if ph == 0 else ph - 1
    return out
