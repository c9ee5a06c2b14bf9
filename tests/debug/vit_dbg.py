"""bring-up: k_viterbi_linear vs oracle on a few characteristic inputs (run on the GPU box)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import abracadabra_amd as aa
from oracle import binding as ob

ctx = aa.Context(n_streams=1, max_frames=1, ring_frames=4)
rng = np.random.default_rng(1)
for kind, prof in ((0, (0, 3, 64)), (1, (0, 3, 64)), (1, (0, 1, 8)), (1, (0, 3, 1152))):
    n_coded = 2304 if kind == 0 else ob.any_profile(*prof).n_coded
    soft = rng.integers(-31, 32, (6, n_coded)).astype(np.int8)
    soft[0] = 31
    soft[1] = 0
    soft[2] = rng.integers(-1, 2, n_coded)
    # a clean codeword: encode random bits is not available here; use strong random soft (still a valid test vs oracle)
    g = ctx.viterbi(soft, kind, *prof)
    o = np.stack([ob.decode_linear(s, kind, *prof) for s in soft])
    for r, name in enumerate(("sat", "zero", "small", "rand", "rand", "rand")):
        nb = (g[r] != o[r]).sum()
        first = int(np.flatnonzero(g[r] != o[r])[0]) if nb else -1
        print(kind, prof, name, "mismatching bytes", nb, "of", g.shape[1], "first", first, "gpu", g[r][:6], "cpu", o[r][:6])
