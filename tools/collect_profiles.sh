#!/bin/bash
# Regenerates the evidence kept under profiles/ (run on the GPU box through gpurun, from the repo root):
#   1. rocprofv3 --kernel-trace --stats of the default bench command (per-kernel time)
#   2. two separate --pmc passes (FETCH_SIZE, WRITE_SIZE) over a short bench run, SYRK launches only
#   3. the default bench line with its cpu_baseline leg
# Outputs land in gpurun_out/profiles_new/; copy what is wanted into profiles/.
set -e
R=${DLCO_ROUND:-r4}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/profiles_new
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $ROOT/bench.py --no-cpu-baseline --reference-iters 0 --no-other-configs > $OUT/stats_bench.json 2> $OUT/stats.log
cp $OUT/stats/*/*_kernel_stats.csv $OUT/${R}_bench_kernel_stats.csv
python3 $ROOT/tools/trace_breakdown.py $OUT/stats 200 $OUT/${R}_bench_kernel_stats_timed.csv > $OUT/${R}_bench_step_breakdown.txt
rm -rf $OUT/stats
# the same trace for the rank ~128 workload (BASELINE configs[2])
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_c3 -- python3 $ROOT/bench.py --config c3 --no-cpu-baseline --reference-iters 0 --no-other-configs > $OUT/stats_c3_bench.json 2> $OUT/stats_c3.log
python3 $ROOT/tools/trace_breakdown.py $OUT/stats_c3 200 $OUT/${R}_bench_c3_kernel_stats_timed.csv > $OUT/${R}_bench_c3_step_breakdown.txt
rm -rf $OUT/stats_c3
# the reference's own shape (500000 x 544): kernel names of the fused path at a width that is not a tile multiple
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_ref544 -- python3 $ROOT/bench.py --config ref544 --no-cpu-baseline --reference-iters 0 > $OUT/stats_ref544_bench.json 2> $OUT/stats_ref544.log
cp $OUT/stats_ref544/*/*_kernel_stats.csv $OUT/${R}_bench_ref544_kernel_stats.csv
python3 $ROOT/tools/trace_breakdown.py $OUT/stats_ref544 200 $OUT/${R}_bench_ref544_kernel_stats_timed.csv > $OUT/${R}_bench_ref544_step_breakdown.txt
rm -rf $OUT/stats_ref544
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc_$c -- python3 $ROOT/bench.py --no-cpu-baseline --reference-iters 0 --no-other-configs --steps 20 > $OUT/pmc_$c.json 2> $OUT/pmc_$c.log
  DLCO_ROUND=$R python3 - $OUT/pmc_$c $c $OUT <<'PY'
import csv, glob, sys
d, c, out = sys.argv[1:4]
f = glob.glob(d + "/*/*counter_collection.csv")[0]
allrows = list(csv.DictReader(open(f)))
rows = [r for r in allrows if "syrk_planes_kernel" in r["Kernel_Name"] and r["Counter_Name"] == c][-20:]
pre = [r for r in allrows if "syrk_split_rows_kernel" in r["Kernel_Name"] and r["Counter_Name"] == c][-20:]
import os
with open("%s/%s_pmc_%s_syrk.csv" % (out, os.environ.get("DLCO_ROUND", "r2"), c.lower()), "w") as o:
    o.write("Dispatch_Id,Kernel_Name,Counter_Name,Counter_Value,Start_Timestamp,End_Timestamp\n")
    for r in rows:
        o.write("%s,syrk_planes_kernel,%s,%s,%s,%s\n" % (r["Dispatch_Id"], c, r["Counter_Value"], r.get("Start_Timestamp", ""), r.get("End_Timestamp", "")))
    for r in pre:
        o.write("%s,syrk_split_rows_kernel,%s,%s,%s,%s\n" % (r["Dispatch_Id"], c, r["Counter_Value"], r.get("Start_Timestamp", ""), r.get("End_Timestamp", "")))
# per launch of the gradient = the tile kernel + the row split in front of it
mean = sum(float(r["Counter_Value"]) for r in rows) / max(len(rows), 1) + sum(float(r["Counter_Value"]) for r in pre) / max(len(pre), 1)
open("%s/pmc_mean_%s_split.txt" % (out, c), "w").write("%r %d\n" % (sum(float(r["Counter_Value"]) for r in pre) / max(len(pre), 1), len(pre)))
print(c, "mean per launch:", mean, "over", len(rows))
open("%s/pmc_mean_%s.txt" % (out, c), "w").write("%r %d\n" % (mean, len(rows)))
# the tracker's product kernels on the packed dual average (two-way filter pass, three-way Rayleigh-Ritz pass)
for tag, key in (("sym2", "skinny_sym_kernel<3, 2,"), ("sym3", "skinny_sym_kernel<3, 3,")):
    rs = [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if key in r["Kernel_Name"] and r["Counter_Name"] == c][-40:]
    open("%s/pmc_mean_%s_%s.txt" % (out, c, tag), "w").write("%r %d\n" % (sum(rs) / max(len(rs), 1), len(rs)))
PY
  rm -rf $OUT/pmc_$c
done
DLCO_ROUND=$R python3 - $OUT <<'PY'
import json, os, sys
out = sys.argv[1]
R = os.environ.get("DLCO_ROUND", "r2")
f, nf = open(out + "/pmc_mean_FETCH_SIZE.txt").read().split()
w, nw = open(out + "/pmc_mean_WRITE_SIZE.txt").read().split()
f, w = float(f), float(w)
bench = [json.loads(l) for l in open(out + "/pmc_FETCH_SIZE.json") if l.startswith("{")][-1]
rf = bench["roofline"]
K = rf.get("mean_active_rows_per_launch")
F = 8192
js = {"command": "rocprofv3 --pmc <COUNTER> --kernel-trace --output-format csv -- python3 bench.py --no-cpu-baseline --steps 20 (separate passes for FETCH_SIZE and WRITE_SIZE; tools/collect_profiles.sh)",
      "kernel": rf["kernel"],
      "note": "values are per launch of the gradient (syrk_planes_kernel + the syrk_split_rows_kernel in front of it), mean of the last %s launches in the run (mean K ~ %s rows); FETCH_SIZE/WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE tallies 128-B requests at 64 B (MI355X guide, HBM section): reads are doubled" % (nf, K),
      "row_split_share": {"FETCH_SIZE_KiB": float(open(out + "/pmc_mean_FETCH_SIZE_split.txt").read().split()[0]), "WRITE_SIZE_KiB": float(open(out + "/pmc_mean_WRITE_SIZE_split.txt").read().split()[0])},
      "FETCH_SIZE_KiB_per_launch": f, "WRITE_SIZE_KiB_per_launch": w,
      "hbm_bytes_per_launch_raw": (f + w) * 1024.0, "hbm_bytes_per_launch_corrected": (2 * f + w) * 1024.0,
      "algorithmic_bytes_per_launch": 8 * F * F + 4 * (K or 0) * F,
      "packed_layout_bytes_per_launch": 8 * (F // 128) * (F // 128 + 1) // 2 * 128 * 128 + 4 * (K or 0) * F,
      "layout_note": "round 3: the single-rank trainer keeps dfAvg as its packed upper 128 x 128 tiles (2080 tiles, 136 MB): the SYRK reads and "
                     "writes each tile once and stores no mirror, so its own bytes are packed_layout_bytes_per_launch; algorithmic_bytes_per_launch "
                     "is SURVEY 8(d)'s dense figure (8 F^2 + 4 K F)"}
prod = {}
for tag, what in (("sym2", "two-way split filter pass, 96 rows"), ("sym3", "three-way split Rayleigh-Ritz pass, 96 rows")):
    try:
        ff, _ = open(out + "/pmc_mean_FETCH_SIZE_%s.txt" % tag).read().split()
        ww, n = open(out + "/pmc_mean_WRITE_SIZE_%s.txt" % tag).read().split()
        prod[tag] = {"kernel": "skinny_sym_kernel (%s)" % what, "FETCH_SIZE_KiB_per_launch": float(ff), "WRITE_SIZE_KiB_per_launch": float(ww),
                     "hbm_bytes_per_launch_corrected": (2 * float(ff) + float(ww)) * 1024.0, "launches_sampled": int(n),
                     "packed_matrix_bytes": (F // 128) * (F // 128 + 1) // 2 * 128 * 128 * 4, "full_matrix_bytes": 4 * F * F}
    except Exception as e:
        prod[tag] = {"error": str(e)}
js["tracker_product_on_packed_tiles"] = prod
json.dump(js, open("%s/%s_pmc_syrk.json" % (out, R), "w"), indent=1)
print(json.dumps(js))
PY
cd $ROOT && python3 bench.py > $OUT/${R}_bench_default.json 2> $OUT/bench_default.log
python3 bench.py --config c3 --cpu-steps 2 > $OUT/${R}_bench_c3.json 2> $OUT/bench_c3.log
python3 bench.py --config ref544 > $OUT/${R}_bench_ref544.json 2> $OUT/bench_ref544.log
python3 bench.py --bf16 --no-cpu-baseline > $OUT/${R}_bench_bf16.json 2> $OUT/bench_bf16.log
tail -c 600 $OUT/${R}_bench_default.json
