mkdir -p gpurun_out/r3
O=$(pwd)/gpurun_out/r3
export DLCO_SYRK_SPLIT3=1
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -s -p no:cacheprovider -k "grad_rda or symmetric_product or teacher_forced" > $O/gputest8.log 2>&1; echo rc=$? >> $O/gputest8.log; tail -4 $O/gputest8.log
timeout -k 10 200 python bench.py --no-cpu-baseline --reference-iters 0 > $O/bench_c2_s3.json 2> $O/bench_c2_s3.err || exit 1
python3 -c "
import json
d=json.loads(open('$O/bench_c2_s3.json').read().strip().split('\n')[-1])
print('split3', round(d['value']), round(d['ms_per_step'],4), json.dumps(d['breakdown_ms_per_step']), d['config']['state_after_run'])
"
