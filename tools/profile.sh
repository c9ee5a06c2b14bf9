#!/bin/bash
# Profiling recipe (run on the GPU box through gpurun): kernel trace + separate PMC passes.
# Usage: bash tools/profile.sh <tag>     -> writes summaries under gpurun_out/prof_<tag>_*
set -o pipefail
TAG=${1:-r01}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $ROOT/bench.py --steps 3 --warmup 3 --no-cpu-baseline --no-pcie --no-legacy"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_${TAG}_trace -- python3 $ROOT/bench.py --no-cpu-baseline --no-pcie --no-legacy > $OUT/prof_${TAG}_trace.log 2>&1 || exit 1      # bench.py's default K and W
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/prof_${TAG}_fetch -- $BENCH > $OUT/prof_${TAG}_fetch.log 2>&1 || exit 1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/prof_${TAG}_write -- $BENCH > $OUT/prof_${TAG}_write.log 2>&1 || exit 1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES --output-format csv -d $OUT/prof_${TAG}_sq -- $BENCH > $OUT/prof_${TAG}_sq.log 2>&1 || exit 1
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM SQ_WAIT_INST_LDS --output-format csv -d $OUT/prof_${TAG}_sq2 -- $BENCH > $OUT/prof_${TAG}_sq2.log 2>&1 || exit 1
find $OUT -name "*.csv" -path "*prof_${TAG}*" | head -40
