"""BASELINE.json configs at their full sizes on the GPU (the parity tests proper use smaller cases).

configs[3]: 256 concurrent synthetic ensembles, full FIC + all 864 CU of the MSC.  Checked through
size-independent properties on EVERY stream (all FIB CRCs good; every decoded FIB and every valid logical
frame is one of the transmitted ones) and bit for bit against the CPU oracle (sync records, every soft
bit, FIBs, MSC bytes, tracking state) on 8 sampled streams.
configs[4] rehearsal: bench.py's own N > 1 path with two ranks sharing the one GPU of the test box.
"""
import json
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest

from oracle import binding as ob

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_config3_256_ensembles_full_msc(gpu_ctx_factory):
    S, P, F, STEPS = 256, 12, 8, 3
    sub = ob.subch_layout(18, 64)
    sampled = list(range(5, S, S // 8))[:8]

    def make(s):
        rng = np.random.default_rng(300 + s)
        iq, fib, msc = ob.tx_generate(seed=7000 + s, eid=0x1000 + s, n_frames=P, subch=sub, loop=1, snr_db=20.0,
                                      cfo_hz=float(rng.uniform(-3000.0, 3000.0)))
        iq = np.roll(iq.reshape(-1, 2), int(rng.integers(0, ob.TF)), axis=0).reshape(-1)
        return iq, {f.tobytes() for f in fib}, {m.tobytes() for m in msc}

    with ThreadPoolExecutor(max_workers=min(16, len(os.sched_getaffinity(0)))) as ex:
        tx = list(ex.map(make, range(S)))
    ctx = gpu_ctx_factory(n_streams=S, fmt=0, ring_frames=P, max_frames=F)
    oracles = {}
    for s, (iq, _, _) in enumerate(tx):
        ctx.set_subchannels(s, sub)
        ctx.push(s, iq)
        ctx.set_write_pos(s, 1 << 62)
        if s in sampled:
            o = ob.Stream(subch=sub, ring_len=P * ob.TF, ti_slots=64)
            o.push(iq)
            o.set_write_pos(1 << 62)
            oracles[s] = o
    for step in range(STEPS):
        ctx.process(F)
        ok, bad = ctx.fib_counts()
        assert bad == 0 and ok == S * F * 12, f"step {step}: {bad} FIB CRC failures"
        for s in range(S):
            gf, gok = ctx.fib(s)
            gm, gv = ctx.msc(s)
            assert all(f.tobytes() in tx[s][1] for f in gf), f"stream {s}: decoded FIBs were not transmitted"
            if step >= 1:
                assert gv.all()                                   # 32 CIFs in: the time de-interleaver is full
            for f in range(F):
                for c in range(4):
                    if gv[f, c]:
                        assert gm[f, c].tobytes() in tx[s][2], f"stream {s} step {step} frame {f} CIF {c}: MSC bytes were not transmitted"
            if s in oracles:
                o = oracles[s].process(F)
                assert o["rc"] == F
                assert np.array_equal(ctx.sync(s), o["sync"]), f"sync records, stream {s}"
                assert np.array_equal(ctx.fic_soft(s), o["fic_soft"]) and np.array_equal(ctx.msc_soft(s), o["msc_soft"]), f"soft bits, stream {s}"
                assert np.array_equal(gf, o["fib"]) and np.array_equal(gok, o["fib_ok"])
                assert np.array_equal(gv, o["msc_valid"]) and np.array_equal(gm[gv == 1], o["msc"][o["msc_valid"] == 1])
                st, so = ctx.state(s), oracles[s].state()
                assert (st["pos"], st["inc"], st["locked"], st["cif"], st["bad"], st["slope"]) == (so["pos"], so["inc"], so["locked"], so["cif"], so["bad"], so["slope"])


def _bench(extra, timeout=900):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + extra, capture_output=True, text=True, timeout=timeout, cwd=ROOT)
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-3000:])
    return json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])


def test_bench_two_ranks_rehearsal_on_one_gpu():
    """configs[4] rehearsal: `bench.py --gpus 2` launches two ranks itself; both are put on GPU 0 (this box has one)
    and talk over gloo; the line must report n_gpus = 2 and the work of both ranks"""
    d = _bench(["--gpus", "2", "--force-device", "0", "--backend", "gloo", "--streams", "8", "--steps", "2", "--warmup", "3",
                "--no-cpu-baseline", "--no-pcie"])
    assert d["n_gpus"] == 2 and d["fib_crc_bad"] == 0 and d["payload_mismatch"] == 0
    assert d["fib_crc_ok"] == 2 * 8 * 8 * 12 and d["payload_checked"] == 2 * 8 * (8 + 32)
    assert d["config"]["streams_per_gpu"] == 8 and d["scaling"] == "weak"


def test_bench_two_ranks_over_rccl_on_one_gpu_is_refused_by_rccl():
    """the N > 1 RCCL path with two REAL ranks needs two GPUs: RCCL (like NCCL) refuses a communicator whose ranks share a
    device.  What happens on this one-GPU box is recorded here so that nobody has to guess: either the run works (a runtime
    that allows it) and must then be correct, or it fails at communicator set-up — never with a wrong result."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--force-device", "0", "--backend", "nccl",
                          "--streams", "8", "--steps", "2", "--warmup", "2", "--no-cpu-baseline", "--no-pcie", "--no-legacy"],
                         capture_output=True, text=True, timeout=600, cwd=ROOT)
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    if out.returncode == 0:
        d = json.loads(lines[-1])
        assert d["n_gpus"] == 2 and d["collective"] == "nccl" and d["fib_crc_bad"] == 0 and d["ranks_ok"] == 2
    else:
        assert not lines, "a failed run must not print a result line"
        text = (out.stderr + out.stdout).lower()
        assert "duplicate gpu" in text or "invalid usage" in text or "nccl" in text, text[-2000:]


def test_bench_collectives_run_over_rccl_on_this_gpu():
    """one rank, but with the process group and the collectives of the N > 1 path (BENCH_FORCE_DIST): init_process_group("nccl",
    device_id), barrier and all_reduce of device tensors execute over RCCL on the test box's GPU"""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), BENCH_FORCE_DIST="1",
               HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--streams", "8", "--steps", "2", "--warmup", "3",
                          "--no-cpu-baseline", "--no-pcie", "--no-legacy"], capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-3000:])
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert d["collective"] == "nccl" and d["n_gpus"] == 1 and d["fib_crc_bad"] == 0 and d["fib_crc_ok"] == 8 * 8 * 12


def test_bench_line_contract_single_gpu():
    d = _bench(["--streams", "16", "--steps", "3", "--warmup", "3", "--cpu-seconds", "1", "--cpu-threads", "2", "--no-pcie"])
    assert d["n_gpus"] == 1 and d["fib_crc_bad"] == 0 and d["payload_mismatch"] == 0
    for key in ("metric", "value", "unit", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config"):
        assert key in d
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-4
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] == 2 and c["value"] > 0 and c["fic_only"]["value"] > c["value"]
    # the drop-in 24-function library with ONE ensemble and an un-paced C host (tools/legacy_rate.c), both legs error free and
    # faster than the reference's binary on one CPU thread (188-195 x FIC-only, 105-110 x with one 48-CU service)
    g = d["legacy_single_stream"]
    assert g["ok"], g
    assert g["fic_only"]["fib_errors"] == 0 and g["fic_only"]["x_realtime"] > 195
    assert g["one_service_48cu"]["au_crc_err"] == 0 and g["one_service_48cu"]["access_units"] > 1000 and g["one_service_48cu"]["x_realtime"] > 110
