#!/usr/bin/env python3
"""Turns the rocprofv3 output of tools_profile.sh (gpurun_out/prof_<tag>_*) into the committed
summaries under profiles/: <tag>_kernel_stats.csv, <tag>_rocprofv3.json and <tag>_traffic.json.

HBM traffic follows MI355X_MICROARCH.md §HBM: FETCH_SIZE and WRITE_SIZE come from separate --pmc
passes, unit KiB; on gfx950 FETCH_SIZE counts 128-byte requests as 64 bytes, so reads are doubled.
Calibration for our access pattern: k_viterbi must read the 471.9 MB of soft bits the previous
kernel wrote (far larger than the 32 MiB of L2); raw FETCH_SIZE reads 0.51x that, i.e. the same 1/2.
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main(tag, streams=256, frames=8, timed=40):
    src = os.path.join(ROOT, "gpurun_out")
    dst = os.path.join(ROOT, "profiles")
    os.makedirs(dst, exist_ok=True)
    ks = glob.glob(os.path.join(src, f"prof_{tag}_trace", "*", "*_kernel_stats.csv"))[0]
    shutil.copy(ks, os.path.join(dst, f"{tag}_kernel_stats.csv"))
    out = {"kernel_stats": list(csv.DictReader(open(ks))), "pmc": {}}
    # the same run, timed region only: the average over each kernel's LAST `timed` launches (bench.py's K timed steps follow its
    # W warm-up steps, during which the GPU still ramps its clocks; the *_kernel_stats.csv average includes those)
    kt = glob.glob(os.path.join(src, f"prof_{tag}_trace", "*", "*_kernel_trace.csv"))
    if kt:
        per = collections.defaultdict(list)
        for r in csv.DictReader(open(kt[0])):
            per[r["Kernel_Name"]].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
        out["timed_region"] = {"launches": timed, "average_ns": {k: round(sum(d for _, d in sorted(v)[-timed:]) / len(sorted(v)[-timed:]), 1)
                                                               for k, v in per.items() if k.startswith(("k_", "void k_"))}}
    for part in ("fetch", "write", "sq", "sq2"):
        f = glob.glob(os.path.join(src, f"prof_{tag}_{part}", "*", "*_counter_collection.csv"))[0]
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, d in agg.items():
            out["pmc"].setdefault(k, {}).update(
                {c: {"mean": sum(v) / len(v), "max": max(v), "dispatches": len(v)} for c, v in d.items()})
    json.dump(out, open(os.path.join(dst, f"{tag}_rocprofv3.json"), "w"), indent=1)
    traffic = {"command": "rocprofv3 --pmc FETCH_SIZE | --pmc WRITE_SIZE | --pmc SQ_... (separate passes, tools/profile.sh) -- python3 bench.py --steps 3 --warmup 3 --no-cpu-baseline --no-pcie --no-legacy",
               "workload": {"streams": streams, "frames_per_step": frames}, "kernels": {}}
    for k, d in out["pmc"].items():
        if "FETCH_SIZE" in d and "WRITE_SIZE" in d and ("k_" in k):
            fetch, write = d["FETCH_SIZE"]["max"], d["WRITE_SIZE"]["max"]        # steady-state dispatch
            traffic["kernels"][k] = {"fetch_size_kib_raw": fetch, "write_size_kib": write,
                                     "hbm_bytes_per_launch": int((2 * fetch + write) * 1024)}
            for c, key in (("SQ_INSTS_VALU", "insts_valu_per_launch"), ("SQ_INSTS_SALU", "insts_salu_per_launch"),
                           ("SQ_INSTS_LDS", "insts_lds_per_launch"), ("SQ_LDS_BANK_CONFLICT", "lds_bank_conflict_cycles"),
                           ("SQ_LDS_IDX_ACTIVE", "lds_idx_active_cycles"),
                           # in units of four cycles, summed over the waves (MI355X_MICROARCH.md: WAIT_ANY + WAIT_INST_ANY + ACTIVE_INST_ANY ~ WAVE_CYCLES)
                           ("SQ_WAVE_CYCLES", "wave_quadcycles"), ("SQ_ACTIVE_INST_VALU", "active_inst_valu_quadcycles"),
                           ("SQ_ACTIVE_INST_LDS", "active_inst_lds_quadcycles"), ("SQ_ACTIVE_INST_ANY", "active_inst_any_quadcycles"),
                           ("SQ_WAIT_ANY", "wait_any_quadcycles"), ("SQ_WAIT_INST_ANY", "wait_inst_any_quadcycles"), ("SQ_WAVES", "waves")):
                if c in d:
                    traffic["kernels"][k][key] = int(d[c]["max"])
    json.dump(traffic, open(os.path.join(dst, f"{tag}_traffic.json"), "w"), indent=1)
    for k, v in traffic["kernels"].items():
        print(k[:40], v)


if __name__ == "__main__":
    main(sys.argv[1])
