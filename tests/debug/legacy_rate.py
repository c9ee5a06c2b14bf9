"""how fast does the legacy 24-function path run one ensemble when the host's input callback does not pace it?"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import binding as ob
from legacy_host import NID, LegacyHost
sub = [[0, 0, 3, 64]]
P = 12
iq, _, _ = ob.tx_generate(seed=5, eid=0x1234, n_frames=P, subch=sub, loop=1, snr_db=25.0)
if len(sys.argv) > 1:                      # write the periodic signal for the C host (legacy_rate.c) and stop
    iq.tofile(sys.argv[1]); sys.exit(0)
N = 50                                     # periods
sig = np.tile(iq.astype(np.float32) - 128.0, N)
host = LegacyHost(sig)
host.tune(periodic=0)
host.wait_for(lambda e: e["nid"] == NID["SYNC_STATUS"] and e.get("level") == 3)
t0 = time.time(); p0 = host.pos
host.wait_for(lambda e: e["nid"] == NID["PERIODIC"] and e["at"] >= (N * P - 4) * 196608, timeout=300)
dt = time.time() - t0
frames = (host.pos - p0) / 2 / 196608
print(f"legacy path: {frames:.0f} frames in {dt:.2f} s = {frames * 0.096 / dt:.0f} x real-time, {dt / frames * 1e3:.3f} ms per frame")
host.close()
