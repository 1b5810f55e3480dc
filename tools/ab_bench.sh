#!/bin/bash
# Developer aid (GPU box): the default bench line under several environment variants on ONE box (box-to-box spread is +-3 %).
# usage: tools/ab_bench.sh <tag> [config] VAR1=1 VAR2=1 ...     ("NONE=1" = the default build)
T=$1; shift
CFG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$T
mkdir -p $OUT
for v in "$@"; do
  n=${v%%=*}
  env $v python3 $ROOT/bench.py $AB_EXTRA --config $CFG --no-cpu-baseline --reference-iters 0 --no-other-configs > $OUT/ab_$n.json 2> $OUT/ab_$n.err || { echo "$n failed"; tail -5 $OUT/ab_$n.err; continue; }
  python3 - $OUT/ab_$n.json $n <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("%-28s value %.0f ms/step %.4f" % (sys.argv[2], d["value"], d["ms_per_step"]), {k: round(v, 4) for k, v in d["breakdown_ms_per_step"].items() if isinstance(v, (int, float))})
PY
done
