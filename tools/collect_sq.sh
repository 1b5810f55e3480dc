#!/bin/bash
# SQ counters (MFMA busy cycles, wave wait buckets) of the hot kernels, one --pmc pass:
# run on the GPU box through gpurun from the repo root; writes gpurun_out/profiles_new/${R}_pmc_sq.{csv,json}
set -e
R=${DLCO_ROUND:-r4}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/profiles_new
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY \
  --kernel-trace --output-format csv -d $OUT/pmc_sq -- python3 $ROOT/bench.py --no-cpu-baseline --reference-iters 0 --no-other-configs --steps 20 > $OUT/pmc_sq.json 2> $OUT/pmc_sq.log
DLCO_ROUND=$R python3 - $OUT/pmc_sq $OUT <<'PY'
import csv, glob, json, os, sys, collections
d, out = sys.argv[1:3]
R = os.environ.get("DLCO_ROUND", "r3")
f = glob.glob(d + "/*/*counter_collection.csv")[0]
rows = list(csv.DictReader(open(f)))
want = {"syrk_planes_kernel": "syrk_planes_kernel", "syrk_split_rows_kernel": "syrk_split_rows_kernel", "jacobi_mw_kernel": "jacobi_mw_kernel", "skinny_sym_kernel<3, 2,": "skinny_sym_kernel<3,2>", "skinny_sym_kernel<3, 3,": "skinny_sym_kernel<3,3>",
        "jacobi_blk_kernel": "jacobi_blk_kernel", "chol_inv2_kernel": "chol_inv2_kernel", "project_rows_kernel": "project_rows_kernel"}
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    for key, name in want.items():
        if key in r["Kernel_Name"]:
            agg[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
js = {"command": "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY "
                 "--kernel-trace -- python3 bench.py --no-cpu-baseline --steps 20 (tools/collect_sq.sh); means over the last 20 launches of each kernel",
      "note": "GRBM_GUI_ACTIVE is summed over the 8 XCDs (elapsed shader cycles = value / 8); SQ_VALU_MFMA_BUSY_CYCLES is summed over the 1024 SIMDs; "
              "mfma_util = MFMA_BUSY / (elapsed * 1024). SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_ANY count quad-cycles per wave (MI355X guide, PMC section).",
      "kernels": {}}
with open("%s/%s_pmc_sq.csv" % (out, R), "w") as o:
    o.write("kernel,counter,mean_per_launch,launches_sampled\n")
    for name, cs in agg.items():
        m = {}
        for c, v in sorted(cs.items()):
            v = v[-20:]
            m[c] = sum(v) / len(v)
            o.write('"%s",%s,%.1f,%d\n' % (name, c, m[c], len(v)))
        k = {"counters": m}
        if m.get("GRBM_GUI_ACTIVE"):
            el = m["GRBM_GUI_ACTIVE"] / 8.0
            k["elapsed_cycles"] = el
            if "SQ_VALU_MFMA_BUSY_CYCLES" in m:
                k["mfma_util"] = m["SQ_VALU_MFMA_BUSY_CYCLES"] / (el * 1024.0)
        if m.get("SQ_WAVE_CYCLES"):
            wc = m["SQ_WAVE_CYCLES"]
            k["wave_cycles_share"] = {"wait_any(s_waitcnt/barrier)": m.get("SQ_WAIT_ANY", 0) / wc, "wait_inst_any(issue stall)": m.get("SQ_WAIT_INST_ANY", 0) / wc,
                                      "active_inst_any": m.get("SQ_ACTIVE_INST_ANY", 0) / wc}
        js["kernels"][name] = k
json.dump(js, open("%s/%s_pmc_sq.json" % (out, R), "w"), indent=1)
print(open("%s/%s_pmc_sq.csv" % (out, R)).read())
PY
rm -rf $OUT/pmc_sq
