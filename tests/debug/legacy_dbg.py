import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import binding as ob
from legacy_host import NID, LegacyHost
sub = [[0, 0, 3, 64]]
iq, fib_tx, _ = ob.tx_generate(seed=91, eid=0x1234, n_frames=14, subch=sub, delay=2000, snr_db=30.0, cfo_hz=2300.0)
host = LegacyHost((iq.astype(np.float32) - 128.0))
host.tune(periodic=0)
time.sleep(1.0)
for e in host.events[:400]:
    if e["nid"] in (NID["PERIODIC"], NID["SYNC_STATUS"]):
        print(e)
host.close()
