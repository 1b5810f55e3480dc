O=$(pwd)/gpurun_out/r3
R=$(pwd)
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for cfg in c2 c3; do
rocprofv3 --kernel-trace --output-format csv -d $O/trace_$cfg -- python3 $R/bench.py --config $cfg --no-cpu-baseline --reference-iters 0 > $O/trace_$cfg.json 2> $O/trace_$cfg.log || exit 1
python3 $R/tools/trace_breakdown.py $O/trace_$cfg 200 > $O/step_breakdown_$cfg.txt
# idle gaps in the timed region: which kernel the GPU waited in front of
python3 - $O/trace_$cfg <<'PY' >> $O/step_breakdown_$cfg.txt
import csv, glob, sys, collections, re
f = glob.glob(sys.argv[1] + '/*/*_kernel_trace.csv')[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
syrk = [i for i, r in enumerate(rows) if 'syrk_rda' in r['Kernel_Name']]
i0 = syrk[-200]
gaps = collections.defaultdict(lambda: [0, 0])
prev_end = int(rows[i0]['End_Timestamp'])
for r in rows[i0 + 1:]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    m = re.search(r'(\w+_kernel\d*(<[^>]*>)?|__amd_\w+)', r['Kernel_Name'])
    k = m.group(1) if m else r['Kernel_Name'][:40]
    g = max(0, s - prev_end)
    gaps[k][0] += 1; gaps[k][1] += g
    prev_end = max(prev_end, e)
    if 'emit_w' in r['Kernel_Name'] and r is rows[-1]: break
tot = sum(v[1] for v in gaps.values())
print("idle in front of a kernel, timed region: total %.3f ms/step" % (tot / 200 / 1e6))
for k, v in sorted(gaps.items(), key=lambda kv: -kv[1][1])[:14]:
    print("  before %-34s n/step %5.2f  gap us/launch %6.2f  ms/step %.4f" % (k[:34], v[0] / 200, v[1] / v[0] / 1e3, v[1] / 200 / 1e6))
PY
rm -rf $O/trace_$cfg
done
cat $O/step_breakdown_c2.txt; cat $O/step_breakdown_c3.txt
