"""The rotating lane<->state map of k_viterbi, modelled in numpy, equals the oracle's
textbook Viterbi (same decisions, same tie rule) — checked on the CPU."""
import numpy as np

import lane_model
from oracle import binding as ob


def _oracle_bits(x4):
    bits = np.zeros(len(x4), dtype=np.uint8)
    ob.lib().orx_viterbi(np.ascontiguousarray(x4).ctypes.data, len(x4), bits.ctypes.data)
    return bits


def test_basis_is_a_basis():
    seen = set()
    for lane in range(64):
        seen.add(tuple(lane_model.lane_coord(lane, k) for k in range(6)))
        rebuilt = 0
        for k in range(6):
            if lane_model.lane_coord(lane, k):
                rebuilt ^= lane_model.XV[k]
        assert rebuilt == lane
    assert len(seen) == 64


def test_model_matches_oracle_on_random_and_tied_inputs():
    rng = np.random.default_rng(3)
    for n, mode in [(774, "rand"), (36, "rand"), (168, "half"), (198, "zero"), (96, "small"), (1542, "rand"), (1542, "sat")]:
        x4 = rng.integers(-31, 32, size=(n, 4)).astype(np.int8)
        if mode == "half":
            x4[:, 2:] = 0
        if mode == "zero":
            x4[:] = 0
        if mode == "sat":
            x4[:] = 31
        if mode == "small":
            x4 = rng.integers(-1, 2, size=(n, 4)).astype(np.int8)     # many ties
        assert np.array_equal(lane_model.decode(x4), _oracle_bits(x4)), (n, mode)
