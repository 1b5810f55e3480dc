mkdir -p gpurun_out/r3
O=$(pwd)/gpurun_out/r3
timeout -k 10 150 ./tools/kern_time small > $O/kern_small2.txt 2>&1 || { tail -5 $O/kern_small2.txt; exit 1; }
grep chol $O/kern_small2.txt
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_full_width_gpu.py -m gpu -q -s -p no:cacheprovider -k "psd_project or teacher_forced or config2 or end_to_end or config0 or tracker_block" > $O/gputest10.log 2>&1; echo rc=$? >> $O/gputest10.log; tail -4 $O/gputest10.log; grep "err_A =" $O/gputest10.log
for cfg in c2 c3; do
timeout -k 10 200 python bench.py --no-cpu-baseline --config $cfg --reference-iters 0 > $O/bench_${cfg}_c.json 2> $O/bench_${cfg}_c.err || exit 1
python3 -c "
import json
d=json.loads(open('$O/bench_${cfg}_c.json').read().strip().split('\n')[-1])
print('$cfg', round(d['value']), round(d['ms_per_step'],4), json.dumps(d['breakdown_ms_per_step']), d['roofline']['frac'], d['roofline']['avg_launch_ms'], d['roofline']['tracker_nonconverged_steps'])
"
done
