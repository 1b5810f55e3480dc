mkdir -p gpurun_out/r3
O=$(pwd)/gpurun_out/r3
R=$(pwd)
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
  n=$(echo $c | cut -d' ' -f1)
  timeout -k 10 200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_$n -- $R/tools/kern_time prod > $O/pmc_$n.txt 2>&1 || { tail -5 $O/pmc_$n.txt; exit 1; }
  python3 - $O/pmc_$n <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/*/*counter_collection.csv")[0]
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"]
    if "skinny" in k:
        acc[(k.split("(")[0][-40:], r["Counter_Name"])].append(float(r["Counter_Value"]))
for k, v in sorted(acc.items()):
    print(k, "n=%d mean=%.1f min=%.1f max=%.1f" % (len(v), sum(v)/len(v), min(v), max(v)))
PY
  rm -rf $O/pmc_$n
done
