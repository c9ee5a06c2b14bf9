#!/usr/bin/env python3
"""Throughput of the drop-in 24-function path with ONE ensemble and an un-paced host (tools/legacy_rate.c): the shape of
BASELINE.json configs[0]/[2] — what the reference's own library does at 188-195 x real time FIC-only and 105-110 x with one
48-CU DAB+ service on one CPU thread.  bench.py calls run() for its `legacy_single_stream` side-leg;
as a script it prints both legs.

The oracle is used here for what the rules allow: synthesising the test signal (a sample-continuous periodic recording:
20 frames = 80 CIFs = 16 DAB+ super frames, so the loop is a valid endless transmission)."""
import json
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

SID = 0x1A01


def build_tool(out_dir):
    exe = os.path.join(out_dir, "legacy_rate")
    lib_dir = os.path.join(ROOT, "abracadabra_amd")
    subprocess.check_call(["gcc", "-O2", "-Wall", "-o", exe, os.path.join(ROOT, "tools", "legacy_rate.c"),
                           "-L" + lib_dir, "-l:libdabsdr_amd.so", "-Wl,-rpath," + lib_dir])
    return exe


def make_signal(path, snr_db=25.0, seed=5, big=False):
    import numpy as np
    from oracle import binding as ob
    if big:                                                 # 416 CU, UEP table index 63 (384 kbit/s, level 1), MPEG Layer II (ASCTy 0):
        sub, rows = [[0, 2, 63, 0]], None                   # codewords of 9 222 trellis steps, the longest audio sub-channel there is
    else:
        sub = [[0, 0, 3, 64]]                               # 48 CU, EEP 3-A, 64 kbit/s, DAB+ (FIG 0/2 ASCTy 63)
        rows, _ = ob.superframes(64, 16, seed=seed)         # 80 logical frames
    iq, _, _ = ob.tx_generate(seed=seed, eid=0x1234, n_frames=20, subch=sub, loop=1, snr_db=snr_db, payload=rows)
    np.asarray(iq, dtype=np.uint8).tofile(path)


def run(frames=3000, timeout=120, long_codewords=False):
    """Both legs; returns {"fic_only": {...}, "one_service_48cu": {...}} (the tool's JSON lines).
    long_codewords: a third leg with one 416-CU MPEG Layer II service (9 222 trellis steps per codeword): is the
    one-wave-per-codeword Viterbi decoder the limit of a single ensemble?"""
    out = {}
    with tempfile.TemporaryDirectory() as d:
        exe = build_tool(d)
        sig = os.path.join(d, "periodic.u8")
        make_signal(sig)
        legs = [("fic_only", [], sig, "none (FIC only)"), ("one_service_48cu", [hex(SID)], sig, "one 48-CU DAB+ service (64 kbit/s, EEP 3-A)")]
        if long_codewords:
            big = os.path.join(d, "periodic_big.u8")
            make_signal(big, big=True)
            legs.append(("one_service_416cu_mp2", [hex(SID)], big, "one 416-CU MPEG Layer II service (384 kbit/s, UEP 1)"))
        for name, extra, sig, what in legs:
            p = subprocess.run([exe, sig, str(frames)] + extra, capture_output=True, text=True, timeout=timeout)
            line = (p.stdout.strip().splitlines() or ["{}"])[-1]
            try:
                out[name] = json.loads(line)
            except ValueError:
                out[name] = {"error": "unparsable", "stdout": p.stdout[-300:], "stderr": p.stderr[-300:]}
            out[name]["rc"] = p.returncode
            out[name]["service"] = what
    return out


if __name__ == "__main__":
    print(json.dumps(run(int(sys.argv[1]) if len(sys.argv) > 1 else 3000, long_codewords="--long" in sys.argv), indent=1))
