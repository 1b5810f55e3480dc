import csv, collections, glob, re, sys
f = glob.glob(sys.argv[1] + '/*/*_kernel_trace.csv')[0]
nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
rows = list(csv.DictReader(open(f)))
syrk = [r for r in rows if 'syrk_split_rows' in r['Kernel_Name'] or 'syrk_round_rows' in r['Kernel_Name']]      # first kernel of a step's gradient
start = int(syrk[-nsteps]['Start_Timestamp'])
# the region ends with the last step's residual / publish kernel (which also emits W; `emit_w_kernel` only on the
# host-ordered path); what follows (the LogStep block that bench.py runs after timing) is not part of it
emit = [r for r in rows if 'emit_w' in r['Kernel_Name'] or 'residual_publish' in r['Kernel_Name']]
t1 = int(emit[-1]['End_Timestamp']) if emit else max(int(r['End_Timestamp']) for r in rows)
rows = [r for r in rows if int(r['Start_Timestamp']) <= t1]
agg = collections.defaultdict(lambda: [0, 0])
for r in rows:
    if int(r['Start_Timestamp']) >= start:
        m = re.search(r'(\w+_kernel\d*(<[^>]*>)?|__amd_\w+)', r['Kernel_Name'])
        k = m.group(1) if m else r['Kernel_Name'][:40]
        agg[k][0] += 1; agg[k][1] += int(r['End_Timestamp']) - int(r['Start_Timestamp'])
tot = sum(v[1] for v in agg.values())
print("wall ms/step %.3f  gpu busy ms/step %.3f  launches/step %.1f" % ((t1 - start) / nsteps / 1e6, tot / nsteps / 1e6, sum(v[0] for v in agg.values()) / nsteps))
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:20]:
    print("%-36s calls/step %6.2f  us/call %8.1f  ms/step %.3f" % (k[:36], v[0] / nsteps, v[1] / v[0] / 1e3, v[1] / nsteps / 1e6))
# optional third argument: CSV of the same timed region in the layout of rocprofv3's kernel stats
if len(sys.argv) > 3:
    full = collections.defaultdict(lambda: [0, 0, 10**18, 0])
    for r in rows:
        if int(r['Start_Timestamp']) >= start:
            d = int(r['End_Timestamp']) - int(r['Start_Timestamp'])
            v = full[r['Kernel_Name']]
            v[0] += 1; v[1] += d; v[2] = min(v[2], d); v[3] = max(v[3], d)
    tot_ns = sum(v[1] for v in full.values())
    with open(sys.argv[3], 'w') as o:
        o.write('"Name","Calls","TotalDurationNs","AverageNs","Percentage","MinNs","MaxNs"\n')
        for k, v in sorted(full.items(), key=lambda kv: -kv[1][1]):
            o.write('"%s",%d,%d,%.3f,%.2f,%d,%d\n' % (k, v[0], v[1], v[1] / v[0], 100.0 * v[1] / tot_ns, v[2], v[3]))

