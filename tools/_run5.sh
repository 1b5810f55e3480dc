mkdir -p gpurun_out/r3
O=$(pwd)/gpurun_out/r3
timeout -k 10 200 ./tools/kern_time prod > $O/kern_prod.txt 2>&1 || { cat $O/kern_prod.txt; exit 1; }
cat $O/kern_prod.txt
timeout -k 10 400 python -m pytest tests/test_gpu_parity.py -m gpu -q -s -p no:cacheprovider -k "symmetric_product or tracker_block or tracker_product" > $O/gputest4.log 2>&1; echo rc=$? >> $O/gputest4.log
tail -5 $O/gputest4.log
timeout -k 10 200 python bench.py --no-cpu-baseline --reference-iters 0 > $O/bench_c2_sym.json 2> $O/bench_c2_sym.err || { tail -5 $O/bench_c2_sym.err; exit 1; }
timeout -k 10 200 python bench.py --no-cpu-baseline --reference-iters 0 --config c3 > $O/bench_c3_sym.json 2> $O/bench_c3_sym.err || exit 1
DLCO_NO_PACKED=1 timeout -k 10 200 python bench.py --no-cpu-baseline --reference-iters 0 > $O/bench_c2_nopack.json 2> $O/bench_c2_nopack.err || exit 1
for f in bench_c2_sym bench_c3_sym bench_c2_nopack; do python3 -c "
import json
d=json.loads(open('$O/$f.json').read().strip().split('\n')[-1])
print('$f', round(d['value']), round(d['ms_per_step'],4), json.dumps(d['breakdown_ms_per_step']), d['config']['state_after_run'])
"; done
