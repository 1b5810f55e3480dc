mkdir -p gpurun_out/r3
O=$(pwd)/gpurun_out/r3
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_full_width_gpu.py -m gpu -q -s -p no:cacheprovider -k "not pair_index_golden" > $O/gputest7.log 2>&1; echo rc=$? >> $O/gputest7.log; tail -4 $O/gputest7.log; grep "err_A\|max_err" $O/gputest7.log
for cfg in c2 c3; do
timeout -k 10 200 python bench.py --no-cpu-baseline --config $cfg > $O/bench_${cfg}_b.json 2> $O/bench_${cfg}_b.err || exit 1
python3 -c "
import json
d=json.loads(open('$O/bench_${cfg}_b.json').read().strip().split('\n')[-1])
print('$cfg', round(d['value']), round(d['ms_per_step'],4), json.dumps(d['breakdown_ms_per_step']), d['roofline']['frac'], d['roofline']['avg_launch_ms'])
rr=d.get('reference_run')
if rr: print('  reference_run', round(rr['value']), round(rr['seconds'],3), [ (w['t'], round(w['Ttime'],3), round(w['Vtime'],4), w['rank']) for w in rr['windows']])
"
done
