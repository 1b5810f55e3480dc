#!/bin/bash
# Developer aid (GPU box): start-up transient (first 100 steps) and the c3 bench line under environment variants on one box.
T=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$T
mkdir -p $OUT
for v in "$@"; do
  n=${v%%=*}
  for cfg in c2 c3; do
    env $v timeout -k 10 300 python3 $ROOT/tools/transient_profile.py 100 $cfg > $OUT/tr_${cfg}_$n.out 2> $OUT/tr_${cfg}_$n.err
    echo "$n $cfg: $(grep 't= 100' $OUT/tr_${cfg}_$n.out) $(grep nonconv $OUT/tr_${cfg}_$n.out)"
  done
done
