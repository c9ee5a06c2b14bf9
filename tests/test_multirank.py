"""N > 1 path of bench.py on the CPU: two gloo ranks shard the streams, no data-path
collective, one all_reduce for the counters and the max-over-ranks time."""
import os
import socket
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent("""
    import os, sys, json
    sys.path.insert(0, %r)
    import numpy as np, torch, torch.distributed as dist
    from oracle import binding as ob
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    n_streams_total = 4
    mine = [s for s in range(n_streams_total) if s %% world == rank]          # stream -> rank = s mod world
    sub = ob.subch_layout(1, 64)
    ok = 0
    for s in mine:
        iq, fib, _ = ob.tx_generate(seed=900 + s, eid=0x3000 + s, n_frames=3, subch=sub, snr_db=25.0)
        o = ob.Stream(subch=sub); o.push(iq)
        r = o.process(1)
        ok += int(r["fib_ok"].sum()); assert np.array_equal(r["fib"], fib[:1])
    t = torch.tensor([float(rank + 1), float(ok), float(len(mine))], dtype=torch.float64)
    tmax = t.clone(); dist.all_reduce(tmax, op=dist.ReduceOp.MAX); dist.all_reduce(t, op=dist.ReduceOp.SUM)
    if rank == 0:
        print(json.dumps({"max_time": float(tmax[0]), "fib_ok": int(t[1]), "streams": int(t[2])}))
    dist.destroy_process_group()
""") % ROOT


def test_two_rank_sharding_and_reduction(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                          "--master-addr", "127.0.0.1", "--master-port", str(port), str(script)],
                         capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    import json
    d = json.loads(line)
    assert d == {"max_time": 2.0, "fib_ok": 48, "streams": 4}
