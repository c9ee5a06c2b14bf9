"""The N > 1 path of bench.py, executed for real on the CPU: `--gpus 2` makes bench.py launch two
ranks under torch.distributed.run (gloo), each rank runs bench.run_rank() — the same sharding, barrier,
timed loop, verification and all_reduce the GPU run uses — with the CPU oracle standing in for the GPU
engine (test infrastructure; the product engine is bench.GpuEngine)."""
import json
import os
import subprocess
import sys
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent("""
    import os, sys
    sys.path.insert(0, %r)
    import numpy as np
    import bench
    from oracle import binding as ob

    class OracleEngine:
        '''CPU checker engine with the interface of bench.GpuEngine'''
        uses_gpu = False

        def __init__(self, args, dev, sub):
            self.args, self.sub = args, sub
            self.rx = [ob.Stream(subch=sub, ring_len=args.period * ob.TF, ti_slots=64) for _ in range(args.streams)]
            self.last = [None] * args.streams

        def load(self, s, iq):
            self.rx[s].push(iq)
            self.rx[s].set_write_pos(1 << 62)

        def step(self):
            for s, r in enumerate(self.rx):
                self.last[s] = r.process(self.args.frames, want_soft=False)
                assert self.last[s]["rc"] in (0, self.args.frames)
            return [0.0] * 5

        def fib_counts(self):
            ok = sum(int(o["fib_ok"].sum()) for o in self.last)
            return ok, self.args.streams * self.args.frames * 12 - ok

        def fib(self, s):
            return self.last[s]["fib"], self.last[s]["fib_ok"]

        def msc(self, s):
            return self.last[s]["msc"], self.last[s]["msc_valid"]

        def close(self):
            pass

    sys.exit(bench.main(engine_factory=OracleEngine, script=os.path.abspath(__file__)))
""") % ROOT

ARGS = ["--backend", "gloo", "--streams", "2", "--frames", "1", "--period", "4", "--nsub", "1", "--steps", "2", "--warmup", "17",
        "--no-cpu-baseline", "--no-pcie"]


def run_worker(tmp_path, extra, env=None):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    return subprocess.run([sys.executable, str(script)] + extra + ARGS, capture_output=True, text=True, timeout=600, cwd=ROOT,
                          env=dict(os.environ, **(env or {})))


def test_gpus_flag_launches_that_many_ranks(tmp_path):
    out = run_worker(tmp_path, ["--gpus", "2"])
    assert out.returncode == 0, out.stderr[-3000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak"
    assert d["fib_crc_bad"] == 0 and d["fib_crc_ok"] == 2 * 2 * 12            # 2 ranks x 2 streams x 1 frame x 12 FIBs
    # 17 warm-up + 2 timed steps of one frame: the time de-interleaver is full, every CIF of the last step is checked
    assert d["payload_mismatch"] == 0 and d["payload_checked"] == 2 * 2 * (1 + 4)
    # whole-job aggregate: 4 ensembles x 1 frame x 2 steps over the max-over-ranks time
    assert abs(d["value"] - 4 * 1 * 2 * 0.096 / (d["ms_per_step"] * 2e-3)) < 0.02 * d["value"] + 0.2
    assert d["x_realtime_per_gpu"] == pytest.approx(d["value"] / 2, rel=1e-3, abs=0.1)


def test_single_rank_same_code(tmp_path):
    out = run_worker(tmp_path, ["--gpus", "1"])
    assert out.returncode == 0, out.stderr[-3000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert d["n_gpus"] == 1 and d["fib_crc_bad"] == 0 and d["payload_mismatch"] == 0 and d["payload_checked"] == 2 * 5


def test_flag_and_launcher_must_agree(tmp_path):
    out = run_worker(tmp_path, ["--gpus", "4"], env={"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert out.returncode != 0 and "disagree" in out.stderr


def test_launch_command_shape():
    sys.path.insert(0, ROOT)
    import bench
    cmd = bench.launch_cmd(8, 29511, "/x/bench.py", ["--gpus", "8", "--steps", "5"])
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and "--nproc-per-node=8" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[-5:] == ["/x/bench.py", "--gpus", "8", "--steps", "5"]
    assert bench.stream_ids(3, 256) == list(range(768, 1024))


def test_bench_without_gpu_fails_loudly():
    """the product engine has no CPU path: on a box without a GPU bench.py must refuse, not fall back"""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--streams", "1", "--steps", "1"], capture_output=True, text=True,
                         timeout=300, cwd=ROOT)
    assert out.returncode != 0 and "needs a GPU" in (out.stderr + out.stdout)
