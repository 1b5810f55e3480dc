mkdir -p gpurun_out/r3
O=gpurun_out/r3
timeout -k 10 120 ./tools/kern_time small > $O/kern_small_blk.txt 2>&1 || exit 1
DLCO_JACOBI_SEAT=1 timeout -k 10 120 ./tools/kern_time small > $O/kern_small_seat.txt 2>&1 || exit 1
DLCO_EIG_TOL=5e-5 timeout -k 10 400 python -m pytest tests/test_gpu_parity.py -m gpu -q -s -p no:cacheprovider -k "psd_project or teacher_forced or tracker_block or zz_report or project_sqdist or validation_and_stats or pair_mode" > $O/gputest2.log 2>&1; echo rc=$? >> $O/gputest2.log
timeout -k 10 200 python bench.py --no-cpu-baseline > $O/bench_c2_a.json 2> $O/bench_c2_a.err || exit 1
timeout -k 10 200 python bench.py --no-cpu-baseline --eig-tol 5e-5 --reference-iters 0 > $O/bench_c2_tol5e5.json 2> $O/bench_c2_tol5e5.err || exit 1
timeout -k 10 200 python bench.py --no-cpu-baseline --config c3 --reference-iters 0 > $O/bench_c3_a.json 2> $O/bench_c3_a.err || exit 1
timeout -k 10 200 python bench.py --no-cpu-baseline --config c3 --eig-tol 5e-5 --reference-iters 0 > $O/bench_c3_tol5e5.json 2> $O/bench_c3_tol5e5.err || exit 1
timeout -k 10 120 python tools/logstep_time.py > $O/logstep.txt 2>&1
tail -3 $O/gputest2.log
