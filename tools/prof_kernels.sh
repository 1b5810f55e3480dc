#!/bin/bash
# Developer profile of the tracker's single kernels (run on the GPU box through gpurun from the repo root):
# kernel trace + a few PMC passes of tools/kern_time; output under gpurun_out/$1.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/${1:-kprof}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
MODE=${2:-prod}
timeout -k 10 150 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $ROOT/tools/kern_time $MODE > $OUT/trace.log 2>&1
timeout -k 10 150 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- $ROOT/tools/kern_time $MODE > $OUT/pmc_fetch.log 2>&1
timeout -k 10 150 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum --kernel-trace --output-format csv -d $OUT/pmc_tcc -- $ROOT/tools/kern_time $MODE > $OUT/pmc_tcc.log 2>&1
timeout -k 10 150 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d $OUT/pmc_sq -- $ROOT/tools/kern_time $MODE > $OUT/pmc_sq.log 2>&1
timeout -k 10 150 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_lds -- $ROOT/tools/kern_time $MODE > $OUT/pmc_lds.log 2>&1
python3 - $OUT <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for f in glob.glob(out + "/trace/*/*kernel_stats.csv"):
    print(open(f).read()[:3000])
for d in ("pmc_fetch", "pmc_tcc", "pmc_sq", "pmc_lds"):
    fs = glob.glob(out + "/" + d + "/*/*counter_collection.csv")
    if not fs:
        print(d, "no output", open(out + "/" + d + ".log").read()[-600:]); continue
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(fs[0])):
        agg[r["Kernel_Name"][:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in agg.items():
        print(d, k, {c: round(sum(v) / len(v), 1) for c, v in cs.items()}, "n=%d" % len(next(iter(cs.values()))))
PY
