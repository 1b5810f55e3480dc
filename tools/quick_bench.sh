#!/bin/bash
# Developer aid (GPU box, via gpurun): the three bench lines without the CPU leg + a kernel trace of the default one.
# usage: tools/quick_bench.sh <tag>      outputs under gpurun_out/<tag>/
T=${1:-qb}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$T
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python3 $ROOT/bench.py --no-cpu-baseline --no-other-configs > $OUT/c2.json 2> $OUT/c2.err || exit 1
python3 $ROOT/bench.py --no-cpu-baseline --config c3 > $OUT/c3.json 2> $OUT/c3.err || exit 1
python3 $ROOT/bench.py --no-cpu-baseline --config ref544 > $OUT/ref544.json 2> $OUT/ref544.err || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $ROOT/bench.py --no-cpu-baseline --reference-iters 0 --no-other-configs > $OUT/stats_bench.json 2> $OUT/stats.log || exit 1
python3 $ROOT/tools/trace_breakdown.py $OUT/stats 200 $OUT/kernel_stats_timed.csv > $OUT/step_breakdown.txt
rm -rf $OUT/stats
python3 - $OUT <<'PY'
import json, sys
o = sys.argv[1]
for f in ("c2", "c3", "ref544"):
    d = json.loads(open("%s/%s.json" % (o, f)).read().strip().splitlines()[-1])
    rr = d.get("reference_run") or {}
    print(f, "value %.0f ms/step %.4f" % (d["value"], d["ms_per_step"]), {k: round(v, 4) for k, v in d["breakdown_ms_per_step"].items() if isinstance(v, (int, float))},
          "refrun %.0f in %.2fs" % (rr.get("value", 0), rr.get("seconds", 0)), "nonconv", d.get("tracker_nonconverged_steps"))
PY
cat $OUT/step_breakdown.txt
