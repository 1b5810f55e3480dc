mkdir -p gpurun_out/r3
O=$(pwd)/gpurun_out/r3
for v in a zc; do
  [ $v = zc ] && export DLCO_IDS_ZEROCOPY=1
  timeout -k 10 200 python bench.py --no-cpu-baseline --reference-iters 0 > $O/bench_c2_$v.json 2> $O/bench_c2_$v.err || { tail -5 $O/bench_c2_$v.err; exit 1; }
  python3 -c "
import json
d=json.loads(open('$O/bench_c2_$v.json').read().strip().split('\n')[-1])
print('$v', round(d['value']), round(d['ms_per_step'],4), json.dumps(d['breakdown_ms_per_step']), d['roofline']['mean_active_rows_per_launch'])
"
done
