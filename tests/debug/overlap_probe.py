"""probe: does running two half-size contexts on their own HIP streams (kernels of one overlapping the other's) beat one context?"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import abracadabra_amd as aa
from oracle import binding as ob
from concurrent.futures import ThreadPoolExecutor
S, F, P = 256, 8, 12
sub = ob.subch_layout(18, 64)
def mk(s):
    iq, _, _ = ob.tx_generate(seed=5000 + s, eid=0x1000 + s, n_frames=P, subch=sub, loop=1, snr_db=20.0, cfo_hz=100.0 * (s % 30))
    return iq
with ThreadPoolExecutor(16) as ex:
    sig = list(ex.map(mk, range(S)))
def run(nctx, stagger):
    per = S // nctx
    ctxs = [aa.Context(n_streams=per, fmt=0, ring_frames=P, max_frames=F) for _ in range(nctx)]
    for c, ctx in enumerate(ctxs):
        for s in range(per):
            ctx.set_subchannels(s, sub); ctx.push(s, sig[c * per + s]); ctx.set_write_pos(s, 1 << 62)
    def step():
        for ctx in ctxs: ctx.process_async(F)
        for ctx in ctxs: ctx.wait()
    for _ in range(4): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): step()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
    bad = sum(c.fib_counts()[1] for c in ctxs)
    for c in ctxs: c.close()
    return dt * 1e3, bad
for n in (1, 2, 4):
    print(n, "contexts:", run(n, 0))
