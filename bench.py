#!/usr/bin/env python3
"""bench.py — DAB Mode-I ensembles decoded x real-time per GPU (BASELINE.json metric).

One "step" = one pass of the hot path (sync -> 2048-FFT + DQPSK demap -> de-interleave ->
Viterbi FIC + all 864 CU of the MSC -> FIB CRC) over one batch of synthetic ensembles that
is already resident in HBM: `--streams` independent Mode-I ensembles (default 256, BASELINE
configs[3]) x `--frames` transmission frames each.  Every stream is a seeded, periodic
(loopable) synthetic raw-IQ signal with its own carrier offset, timing offset and AWGN.

    python bench.py --gpus N --steps K --warmup W

For N > 1 the driver launches one rank per GPU with torch.distributed.run; ensembles are
sharded one-per-stream across ranks with no data-path collective (weak scaling); RCCL
carries only the max-over-ranks time and the FIB counters.
"""
import argparse
import json
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

FRAME_S = 0.096
TF = 196608
SYMS_PER_FRAME = 76
# algorithmic HBM bytes per ensemble-frame (SURVEY.md §8d, u8 input, int8 soft bits)
BYTES_CHAIN = 393216 + 230400 + 230400 + 14208          # IQ read + soft write + soft read + decoded bytes
BYTES_VITERBI = 230400 + 14208                          # dominant kernel: soft-bit read + decoded bytes
BYTES_DEMOD = 387904 + 230400                           # k_demod: IQ of the 76 data/reference symbols + soft-bit write
ACS_PER_FRAME = (4 * 774 + 4 * 18 * 1542) * 64          # trellis steps x 64 states, 18 x 48 CU EEP 3-A
HBM_PEAK_GBS = 8000.0                                   # MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s
VALU_LANE_OPS_PEAK = 256 * 4 * 32 * 2.4e9               # 256 CU x 4 SIMD-32 x 2.4 GHz


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--streams", type=int, default=256, help="ensembles per GPU")
    ap.add_argument("--frames", type=int, default=8, help="transmission frames per stream per step")
    ap.add_argument("--period", type=int, default=12, help="period of the synthetic signal in frames (multiple of 4)")
    ap.add_argument("--snr", type=float, default=20.0)
    ap.add_argument("--nsub", type=int, default=18, help="48-CU EEP 3-A sub-channels per ensemble (18 = all 864 CU)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-pcie", action="store_true", help="skip the host-fed (PCIe-inclusive) side measurement")
    ap.add_argument("--dabplus", action="store_true",
                    help="side measurement: every sub-channel carries DAB+ super frames and k_superframe decodes them "
                         "(period 20 frames = 16 super frames per sub-channel; not the headline workload)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL); gloo only for rehearsals")
    ap.add_argument("--force-device", type=int, default=-1, help="rehearsal only: put every rank on this GPU")
    ap.add_argument("--cpu-frames", type=int, default=208)
    return ap.parse_args()


def make_stream(args, rank, s, sub):
    from oracle import binding as ob          # synthetic transmitter lives with the test infrastructure
    gid = rank * args.streams + s
    rng = np.random.default_rng(1000 + gid)
    payload = None
    if args.dabplus:                           # 4 P logical frames = whole super frames, so the periodic signal stays in sync
        assert (4 * args.period) % 5 == 0
        payload = np.concatenate([ob.superframes(sc[3], 4 * args.period // 5, seed=7 * gid + k)[0] for k, sc in enumerate(sub)], axis=1)
    iq, fib, msc = ob.tx_generate(seed=5000 + gid, eid=0x1000 + (gid & 0xFFF), n_frames=args.period, subch=sub, delay=0,
                                  loop=1, fmt=0, snr_db=args.snr, cfo_hz=float(rng.uniform(-3000.0, 3000.0)), rms=28.0, payload=payload)
    shift = int(rng.integers(0, TF))           # arbitrary start position inside the frame
    iq = np.roll(iq.reshape(-1, 2), shift, axis=0).reshape(-1)
    return iq, fib, msc, shift


def measured_traffic(S, F, kernel="k_viterbi"):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes
    (profiles/<tag>_traffic.json, written by tools/summarize_profiles.py from the same bench
    command under rocprofv3); None when no profile matches this workload."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_traffic.json")))
    for f in reversed(files):
        try:
            t = json.load(open(f))
        except Exception:
            continue
        if t.get("workload") == {"streams": S, "frames_per_step": F}:
            for k, v in t["kernels"].items():
                if kernel in k:
                    return v["hbm_bytes_per_launch"], os.path.basename(f)
    return None, None


def cpu_baseline(args, sub):
    """Oracle (scalar CPU port of the same chain) timed on one host core over a bounded sample."""
    from oracle import binding as ob
    n = args.cpu_frames
    iq, _, _ = ob.tx_generate(seed=99, n_frames=n + 2, subch=sub, delay=777, snr_db=args.snr, cfo_hz=1234.0)
    orc = ob.Stream(subch=sub, ring_len=(n + 4) * TF, ti_slots=64)
    orc.push(iq)
    orc.process(4, want_soft=False)            # acquisition outside the timed sample
    done, t0 = 0, time.perf_counter()
    while done + 4 <= n - 4:
        if orc.process(4, want_soft=False)["rc"] != 4:
            break
        done += 4
    dt = time.perf_counter() - t0
    return {"value": round(done * FRAME_S / dt, 4), "unit": "x real-time (ensemble-seconds per second)", "cores": 1,
            "kind": "port", "sample": f"1 ensemble x {done} frames, full FIC + 18 x 48 CU MSC, oracle/dab_rx.c, {dt:.1f} s"}


def main():
    args = parse_args()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    import torch
    import torch.distributed as dist
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP library has no CPU path")
    dev = args.force_device if args.force_device >= 0 else local_rank
    torch.cuda.set_device(dev)
    if world > 1:
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev))
        else:
            dist.init_process_group(args.backend)

    import abracadabra_amd as aa
    from oracle import binding as ob
    sub = ob.subch_layout(args.nsub, 64)        # default: all 864 CU = 18 x 48 CU, EEP 3-A, 64 kbit/s
    if args.dabplus:
        args.period = 20
    S, F, P = args.streams, args.frames, args.period
    assert P % 4 == 0 and P >= F + 2

    t_gen = time.perf_counter()
    with ThreadPoolExecutor(max_workers=min(16, os.cpu_count() or 4)) as ex:
        streams = list(ex.map(lambda s: make_stream(args, rank, s, sub), range(S)))
    t_gen = time.perf_counter() - t_gen

    ctx = aa.Context(n_streams=S, fmt=0, ring_frames=P, max_frames=F, device=dev)
    for s, (iq, _, _, _) in enumerate(streams):
        ctx.set_subchannels(s, sub)
        ctx.push(s, iq)                          # fills the ring exactly once: the signal is periodic
        ctx.set_write_pos(s, 1 << 62)            # resident periodic ring: never underruns
        if args.dabplus:
            ctx.set_dabplus(s, (1 << len(sub)) - 1)
    ctx.enable_timing(True)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        ctx.process(F)
    barrier()
    t0 = time.perf_counter()
    phase_ms = np.zeros(5)
    for _ in range(args.steps):
        ctx.process(F)
        phase_ms += np.array(ctx.last_timing())
    barrier()
    elapsed = time.perf_counter() - t0
    ok, bad = ctx.fib_counts()

    # ---- correctness of what was just timed (outside the timed region): FIB CRCs and,
    # for a few streams, decoded FIBs / MSC bytes against the transmitted payload
    mism = 0
    checked = 0
    for s in range(0, S, max(1, S // 8)):
        _, fib_tx, msc_tx, _ = streams[s]
        gf, gok = ctx.fib(s)
        gm, gv = ctx.msc(s)
        fib_set = {fib_tx[f].tobytes() for f in range(P)}
        for f in range(F):
            checked += 1
            mism += (gf[f].tobytes() not in fib_set)
        msc_set = {msc_tx[r].tobytes() for r in range(4 * P)}
        for f in range(F):
            for c in range(4):
                if gv[f, c]:
                    checked += 1
                    mism += (gm[f, c].tobytes() not in msc_set)

    dabplus = None
    if args.dabplus:                             # what k_superframe made of the sub-channels of a few streams
        tot = {}
        for s in range(0, S, max(1, S // 8)):
            for k in range(len(sub)):
                for key, v in ctx.superframe_stats(s, k).items():
                    tot[key] = tot.get(key, 0) + v
        dabplus = {"stats_of_sampled_subchannels": tot, "post_viterbi_ms_per_step": round(float(phase_ms[3]) / args.steps, 3),
                   "superframes_per_step": S * len(sub) * 4 * F // 5}

    # ---- the same step fed through the host boundary, reported beside `value`, never as `value` (the inputs of
    # the timed region above are resident): (a) dabx_push from pageable numpy arrays, synchronous; (b) from
    # page-locked buffers the "file reader" has already filled, queued on the copy stream while the previous
    # step decodes (DABX_SRC_PINNED + dabx_process_async)
    pcie = None
    if world == 1 and not args.no_pcie:
        ctx.close()
        ctx = None
        per = P * TF
        k_steps = 5
        first = (F + 1) * TF + 4096

        def run(pinned):
            c2 = aa.Context(n_streams=S, fmt=0, ring_frames=2 * F + 4, max_frames=F, device=dev)   # room for the step in flight + the next
            if pinned:
                stage = c2.alloc_pinned(S * per * 2)
                for s_, (iq, _, _, _) in enumerate(streams):
                    stage[s_ * per * 2:(s_ + 1) * per * 2] = iq
            pos = [0] * S

            def push(s_, n):                  # next n samples of the periodic stream s_
                a = pos[s_] % per
                for a0, n0 in ((a, min(n, per - a)), (0, n - min(n, per - a))):
                    if n0 <= 0:
                        continue
                    if pinned:
                        c2.push_pinned(s_, stage.ctypes.data + (s_ * per + a0) * 2, n0)
                    else:
                        c2.push(s_, streams[s_][0][2 * a0:2 * (a0 + n0)])
                pos[s_] += n

            def push_step(n):                 # every stream advances by n samples
                if not pinned:
                    for s_ in range(S):
                        push(s_, n)
                    return
                a = pos[0] % per              # lock step: one strided copy (two when the period wraps)
                for a0, n0 in ((a, min(n, per - a)), (0, n - min(n, per - a))):
                    if n0 > 0:
                        c2.push_all(stage.ctypes.data + a0 * 2, per * 2, n0, kind=2)
                for s_ in range(S):
                    pos[s_] += n

            for s_ in range(S):
                c2.set_subchannels(s_, sub)
            push_step(first)
            c2.process(F)                     # acquisition, untimed
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            push_step(F * TF)
            for k in range(k_steps):
                c2.process_async(F)
                if k + 1 < k_steps:
                    push_step(F * TF)         # the next step's samples travel while this one decodes
                c2.wait()
            torch.cuda.synchronize()
            dt = time.perf_counter() - t1
            _, bad2 = c2.fib_counts()
            if pinned:
                c2.free_pinned(stage)
            c2.close()
            return {"value": round(S * F * k_steps * FRAME_S / dt, 1), "ms_per_step": round(dt / k_steps * 1e3, 2), "fib_crc_bad": bad2}

        pcie = {"unit": "x real-time", "host_bytes_per_step": S * F * TF * 2,
                "pageable_synchronous": run(False), "pinned_overlapped": run(True)}

    t = torch.tensor([elapsed, float(ok), float(bad), float(mism)], dtype=torch.float64,
                     device="cuda" if args.backend == "nccl" else "cpu")
    if world > 1:
        tmax = t.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        elapsed = float(tmax[0])
        ok, bad, mism = int(t[1]), int(t[2]), int(t[3])

    if rank == 0:
        frames_total = world * S * F * args.steps
        value = frames_total * FRAME_S / elapsed
        vit_ms = phase_ms[2] / args.steps
        achieved = S * F * BYTES_VITERBI / (vit_ms * 1e-3) / 1e9
        acs_rate = S * F * (4 * 774 + 4 * args.nsub * 1542) * 64 / (vit_ms * 1e-3)
        traffic, traffic_src = measured_traffic(S, F) if args.nsub == 18 else (None, None)
        out = {
            "metric": "DAB Mode-I ensembles decoded x real-time per GPU (2048-FFT + de-interleave + Viterbi, full FIC+MSC)",
            "value": round(value, 1), "unit": "x real-time", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32 FFT / int8 soft bits / int32 path metrics", "data": "synthetic",
            "config": {"workload": f"configs[3]: {S} concurrent synthetic Mode-I ensembles per GPU, full FIC + MSC ({args.nsub} x 48 CU EEP 3-A) Viterbi",
                       "streams_per_gpu": S, "frames_per_step": F, "snr_db": args.snr, "sample_format": "u8 IQ 2.048 Msps",
                       "parallelism": f"{world} x independent streams, no data-path collective"},
            "msym_per_s": round(value / FRAME_S * SYMS_PER_FRAME / 1e6, 3),
            "x_realtime_per_gpu": round(value / world, 1),
            "fib_crc_ok": ok, "fib_crc_bad": bad, "payload_checked": checked * world if world > 1 else checked, "payload_mismatch": mism,
            "kernel_ms_per_step": {"sync": round(phase_ms[0] / args.steps, 3), "fft_demap": round(phase_ms[1] / args.steps, 3),
                                   "viterbi": round(vit_ms, 3), "crc_state": round(phase_ms[3] / args.steps, 3),
                                   "all": round(phase_ms[4] / args.steps, 3)},
            "roofline": {"kernel": "k_viterbi", "bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                         "traffic_source": traffic_src, "algorithmic_bytes_per_launch": S * F * BYTES_VITERBI,
                         "note": "VALU-issue-bound kernel (DESIGN.md §7): HBM fraction is small by construction (SURVEY.md §0.8). "
                                 "traffic = algorithmic bytes + the decision scratch (64 decision bits per trellis step, "
                                 "written once and read once by the traceback: 2 x 8 B x steps)",
                         "decision_scratch_bytes_per_launch": 2 * 8 * S * F * (4 * 774 + 4 * args.nsub * 1542),
                         "acs_per_s": round(acs_rate, 0),
                         "chain_algorithmic_GBps": round(value / world / FRAME_S * BYTES_CHAIN / 1e9, 2)},
            "other_kernels": {"k_demod": {"bound": "hbm", "algorithmic_bytes_per_launch": S * F * BYTES_DEMOD,
                                          "achieved": round(S * F * BYTES_DEMOD / (phase_ms[1] / args.steps * 1e-3) / 1e9, 1), "unit": "GB/s",
                                          "frac": round(S * F * BYTES_DEMOD / (phase_ms[1] / args.steps * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}},
            "setup_s": {"synthesis": round(t_gen, 1)},
        }
        if args.dabplus:
            out["dabplus"] = dabplus
        if pcie is not None:
            out["pcie_inclusive"] = pcie
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args, sub)
        print(json.dumps(out), flush=True)
    if ctx is not None:
        ctx.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
