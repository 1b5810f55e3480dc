mkdir -p gpurun_out/r3
O=$(pwd)/gpurun_out/r3
timeout -k 10 150 ./tools/kern_time small big > $O/kern_big_mw.txt 2>&1 || { tail -5 $O/kern_big_mw.txt; exit 1; }
grep jacobi $O/kern_big_mw.txt
DLCO_JACOBI_NO_MW=1 timeout -k 10 150 ./tools/kern_time small big > $O/kern_big_old.txt 2>&1 || exit 1
grep jacobi $O/kern_big_old.txt | tail -4
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -s -p no:cacheprovider -k "tracker_block or psd_project or config0 or end_to_end" > $O/gputest6.log 2>&1; echo rc=$? >> $O/gputest6.log; tail -4 $O/gputest6.log
timeout -k 10 120 python tools/transient_profile.py 100 > $O/transient_mw.txt 2>&1 || exit 1
cat $O/transient_mw.txt
