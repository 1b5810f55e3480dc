#!/bin/bash
# SQ counters (MFMA busy cycles, wave wait buckets) of the hot kernels, one --pmc pass:
# run on the GPU box through gpurun from the repo root; writes gpurun_out/profiles_new/r1_pmc_sq.csv
set -e
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/profiles_new
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY \
  --kernel-trace --output-format csv -d $OUT/pmc_sq -- python3 $ROOT/bench.py --no-cpu-baseline --steps 20 > $OUT/pmc_sq.json 2> $OUT/pmc_sq.log
python3 - $OUT/pmc_sq $OUT <<'PY'
import csv, glob, sys, collections
d, out = sys.argv[1:3]
f = glob.glob(d + "/*/*counter_collection.csv")[0]
rows = list(csv.DictReader(open(f)))
want = {"syrk_rda": "syrk_rda_kernel", "skinny_bf16x2_kernel<3, 2>": "skinny_bf16x2_kernel<3,2>", "skinny_bf16x2_kernel<3, 3>": "skinny_bf16x2_kernel<3,3>"}
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    for key, name in want.items():
        if key in r["Kernel_Name"]:
            agg[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(out + "/r1_pmc_sq.csv", "w") as o:
    o.write("kernel,counter,mean_per_launch,launches_sampled\n")
    for name, cs in agg.items():
        for c, v in sorted(cs.items()):
            v = v[-20:]
            o.write('"%s",%s,%.1f,%d\n' % (name, c, sum(v) / len(v), len(v)))
print(open(out + "/r1_pmc_sq.csv").read())
PY
rm -rf $OUT/pmc_sq
