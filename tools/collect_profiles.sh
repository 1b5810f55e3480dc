#!/bin/bash
# Regenerates the evidence kept under profiles/ (run on the GPU box through gpurun, from the repo root):
#   1. rocprofv3 --kernel-trace --stats of the default bench command (per-kernel time)
#   2. two separate --pmc passes (FETCH_SIZE, WRITE_SIZE) over a short bench run, SYRK launches only
#   3. the default bench line with its cpu_baseline leg
# Outputs land in gpurun_out/profiles_new/; copy what is wanted into profiles/.
set -e
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/profiles_new
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $ROOT/bench.py --no-cpu-baseline > $OUT/stats_bench.json 2> $OUT/stats.log
cp $OUT/stats/*/*_kernel_stats.csv $OUT/r1_bench_kernel_stats.csv
python3 $ROOT/tools/trace_breakdown.py $OUT/stats 200 $OUT/r1_bench_kernel_stats_timed.csv > $OUT/r1_bench_step_breakdown.txt
rm -rf $OUT/stats
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc_$c -- python3 $ROOT/bench.py --no-cpu-baseline --steps 20 > $OUT/pmc_$c.json 2> $OUT/pmc_$c.log
  python3 - $OUT/pmc_$c $c $OUT <<'PY'
import csv, glob, sys
d, c, out = sys.argv[1:4]
f = glob.glob(d + "/*/*counter_collection.csv")[0]
rows = [r for r in csv.DictReader(open(f)) if "syrk_rda" in r["Kernel_Name"] and r["Counter_Name"] == c]
rows = rows[-20:]
with open("%s/r1_pmc_%s_syrk.csv" % (out, c.lower()), "w") as o:
    o.write("Dispatch_Id,Kernel_Name,Counter_Name,Counter_Value,Start_Timestamp,End_Timestamp\n")
    for r in rows:
        o.write("%s,syrk_rda_kernel,%s,%s,%s,%s\n" % (r["Dispatch_Id"], c, r["Counter_Value"], r.get("Start_Timestamp", ""), r.get("End_Timestamp", "")))
print(c, "mean per launch:", sum(float(r["Counter_Value"]) for r in rows) / max(len(rows), 1), "over", len(rows))
PY
  rm -rf $OUT/pmc_$c
done
cd $ROOT && python3 bench.py > $OUT/r1_bench_default.json 2> $OUT/bench_default.log
tail -c 600 $OUT/r1_bench_default.json
