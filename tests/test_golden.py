"""Committed regression vectors (tests/golden/, made by tests/golden/make_golden.py).

CPU: the oracle reproduces them (catches drift of generator or oracle).
GPU: the HIP path reproduces them through the C ABI without the oracle in the loop."""
import ast
import hashlib
import os

import numpy as np
import pytest

from oracle import binding as ob

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CASES = ["u8_18x48cu_snr15", "s16_mixed_profiles", "u8_impaired_channel"]


def _load(name):
    return np.load(os.path.join(GOLD, name + ".npz"), allow_pickle=False)


def _regen(g):
    p = ast.literal_eval(str(g["params"]))
    iq, fib, msc = ob.tx_generate(**p)
    assert hashlib.sha256(iq.tobytes()).hexdigest() == str(g["iq_sha256"]), "synthetic transmitter drifted"
    return p, iq


@pytest.mark.parametrize("name", CASES)
def test_oracle_reproduces_golden(name):
    g = _load(name)
    p, iq = _regen(g)
    orc = ob.Stream(fmt=p["fmt"], subch=p["subch"], ring_len=16 * ob.TF)
    orc.push(iq)
    o = orc.process(int(g["n_proc"]))
    assert np.array_equal(o["sync"], g["sync"]) and np.array_equal(o["fib"], g["fib"]) and np.array_equal(o["msc"], g["msc"])
    assert hashlib.sha256(o["fic_soft"].tobytes()).hexdigest() == str(g["fic_soft_sha256"])
    assert hashlib.sha256(o["msc_soft"].tobytes()).hexdigest() == str(g["msc_soft_sha256"])
    # and the decode equals what was transmitted
    assert g["fib_ok"].all() and np.array_equal(g["fib"], g["tx_fib"][:int(g["n_proc"])])


def test_oracle_decodes_committed_iq():
    g = _load("u8_iq_1frame")
    orc = ob.Stream(fmt=0, subch=g["subch"].tolist(), ring_len=16 * ob.TF)
    orc.push(g["iq"])
    o = orc.process(1)
    assert np.array_equal(o["sync"], g["sync"]) and np.array_equal(o["fic_soft"], g["fic_soft"])
    assert np.array_equal(o["fib"], g["fib"]) and o["fib_ok"].all() and np.array_equal(o["fib"], g["tx_fib"])


@pytest.mark.gpu
@pytest.mark.parametrize("name", CASES)
def test_gpu_reproduces_golden(gpu_ctx_factory, name):
    g = _load(name)
    p, iq = _regen(g)
    n = int(g["n_proc"])
    ctx = gpu_ctx_factory(n_streams=1, fmt=p["fmt"], ring_frames=16, max_frames=n)
    ctx.set_subchannels(0, p["subch"])
    ctx.push(0, iq)
    ctx.process(n)
    fib, ok = ctx.fib(0)
    msc, valid = ctx.msc(0)
    assert np.array_equal(ctx.sync(0), g["sync"])
    assert np.array_equal(fib, g["fib"]) and np.array_equal(ok, g["fib_ok"])
    assert np.array_equal(valid, g["msc_valid"]) and np.array_equal(msc[valid == 1], g["msc"][g["msc_valid"] == 1])
    assert hashlib.sha256(ctx.fic_soft(0).tobytes()).hexdigest() == str(g["fic_soft_sha256"])
    assert hashlib.sha256(ctx.msc_soft(0).tobytes()).hexdigest() == str(g["msc_soft_sha256"])


@pytest.mark.gpu
def test_gpu_decodes_committed_iq(gpu_ctx_factory):
    g = _load("u8_iq_1frame")
    ctx = gpu_ctx_factory(n_streams=1, fmt=0, ring_frames=16, max_frames=1)
    ctx.set_subchannels(0, g["subch"].tolist())
    ctx.push(0, g["iq"])
    ctx.process(1)
    fib, ok = ctx.fib(0)
    assert np.array_equal(ctx.sync(0), g["sync"]) and np.array_equal(ctx.fic_soft(0), g["fic_soft"])
    assert np.array_equal(fib, g["fib"]) and ok.all()
