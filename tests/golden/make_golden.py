#!/usr/bin/env python3
"""Regenerates tests/golden/*.npz from the oracle (run from the repo root).

These are regression vectors of OUR oracle (inputs + expected outputs), not reference parity:
the reference holds no vectors for this path and its binary is not run.
Each fixture stores the transmitter parameters, a SHA-256 of the generated IQ (to detect
generator drift), the transmitted FIBs / payload and the oracle's decoded outputs.  One
fixture also carries two frames of u8 IQ so that the decode is pinned without the generator.
"""
import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import binding as ob  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))

CASES = {
    "u8_18x48cu_snr15": dict(seed=101, eid=0x4001, n_frames=6, subch=ob.subch_layout(18, 64), delay=4321, snr_db=15.0, cfo_hz=-1875.5, fmt=0),
    "s16_mixed_profiles": dict(seed=102, eid=0x4002, n_frames=6, subch=[[0, 0, 3, 64], [48, 1, 4, 32], [100, 0, 1, 8], [200, 0, 2, 32], [300, 0, 4, 72]],
                               delay=150000, snr_db=12.0, cfo_hz=6100.25, fmt=1, rms=3000.0),
    # a non-ideal channel: receiver clock 60 ppm slow, second path 150 samples late at -5 dB, DC offset
    "u8_impaired_channel": dict(seed=104, eid=0x4004, n_frames=8, subch=ob.subch_layout(3, 64), delay=7000, snr_db=15.0, cfo_hz=2250.0, fmt=0,
                                sco_ppm=-60.0, echo=(150, 5.0, 2.0), dc=(4.0, -3.0)),
}


def make(name, p, keep_iq_frames=0):
    iq, fib, msc = ob.tx_generate(**p)
    orc = ob.Stream(fmt=p["fmt"], subch=p["subch"], ring_len=16 * ob.TF)
    orc.push(iq)
    o = orc.process(p["n_frames"] - 2)
    out = dict(params=np.array(repr(p)), iq_sha256=np.array(hashlib.sha256(iq.tobytes()).hexdigest()),
               subch=np.array(p["subch"], dtype=np.int32), fmt=np.array(p["fmt"]), n_proc=np.array(p["n_frames"] - 2),
               tx_fib=fib, tx_msc=msc, sync=o["sync"], fib=o["fib"], fib_ok=o["fib_ok"], msc=o["msc"], msc_valid=o["msc_valid"],
               fic_soft_sha256=np.array(hashlib.sha256(o["fic_soft"].tobytes()).hexdigest()),
               msc_soft_sha256=np.array(hashlib.sha256(o["msc_soft"].tobytes()).hexdigest()))
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(name, "fib_ok", int(o["fib_ok"].sum()), "of", o["fib_ok"].size)


def make_with_iq():
    # two noiseless frames incl. IQ (u8): decode pinned without the generator
    p = dict(seed=103, eid=0x4003, n_frames=3, subch=ob.subch_layout(1, 64), delay=2000, snr_db=100.0, cfo_hz=500.0, fmt=0)
    iq, fib, msc = ob.tx_generate(**p)
    n = 2 * ob.TF + 2000 + 8192
    orc = ob.Stream(fmt=0, subch=p["subch"], ring_len=16 * ob.TF)
    orc.push(iq)
    o = orc.process(1)
    np.savez_compressed(os.path.join(HERE, "u8_iq_1frame.npz"), iq=iq[:2 * n], subch=np.array(p["subch"], dtype=np.int32),
                        sync=o["sync"], fib=o["fib"], fib_ok=o["fib_ok"], fic_soft=o["fic_soft"], tx_fib=fib[:1])
    print("u8_iq_1frame", o["fib_ok"].sum())


if __name__ == "__main__":
    for k, v in CASES.items():
        make(k, dict(v))
    make_with_iq()
