#!/bin/bash
# Per-kernel time of the timed steps of one bench command (rocprofv3 --kernel-trace), on the GPU box through gpurun:
#   tools/quick_trace.sh <tag> [bench.py flags...]   ->  gpurun_out/trace/<tag>_step_breakdown.txt (+ _kernel_stats_timed.csv)
set -e
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/trace
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_$TAG -- python3 $ROOT/bench.py --no-cpu-baseline --reference-iters 0 --no-other-configs "$@" > $OUT/${TAG}_bench.json 2> $OUT/${TAG}.log
python3 $ROOT/tools/trace_breakdown.py $OUT/stats_$TAG 200 $OUT/${TAG}_kernel_stats_timed.csv > $OUT/${TAG}_step_breakdown.txt
rm -rf $OUT/stats_$TAG
cat $OUT/${TAG}_step_breakdown.txt
