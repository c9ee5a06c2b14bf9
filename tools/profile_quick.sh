#!/bin/bash
# Quick look at one variant: kernel trace + two SQ counter passes.  Usage: bash tools/profile_quick.sh <tag>
set -o pipefail
TAG=${1:-q}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $ROOT/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-pcie --no-legacy"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_${TAG}_trace -- $BENCH > $OUT/prof_${TAG}_trace.log 2>&1 || exit 1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_WAVES SQ_INSTS_VMEM --output-format csv -d $OUT/prof_${TAG}_sq -- $BENCH > $OUT/prof_${TAG}_sq.log 2>&1 || exit 1
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU --output-format csv -d $OUT/prof_${TAG}_sq2 -- $BENCH > $OUT/prof_${TAG}_sq2.log 2>&1 || exit 1
python3 - <<PY
import csv, glob, collections
for kind in ("trace", "sq", "sq2"):
    for f in glob.glob("$OUT/prof_${TAG}_%s/**/*" % kind, recursive=True):
        if f.endswith("kernel_stats.csv"):
            for r in csv.DictReader(open(f)):
                if float(r["Percentage"]) > 0.5: print(r["Name"][:40], r["Calls"], r["AverageNs"], r["Percentage"])
        if f.endswith("counter_collection.csv"):
            acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
            for r in csv.DictReader(open(f)):
                k = r["Kernel_Name"][:24]
                acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
            for k in acc:
                if "viterbi" in k or "demod" in k:
                    print(k, {c: round(v / n[(k, c)]) for c, v in acc[k].items()})
PY
