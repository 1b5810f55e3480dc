mkdir -p gpurun_out/r3
O=$(pwd)/gpurun_out/r3
R=$(pwd)
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -s -p no:cacheprovider -k "psd_project or teacher_forced or tracker_block or zz_report" > $O/gputest3.log 2>&1; echo rc=$? >> $O/gputest3.log
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_transient -- python3 $R/tools/transient_profile.py 100 > $O/transient.txt 2> $O/transient.err || exit 1
cp $O/prof_transient/*/*_kernel_stats.csv $O/transient_kernel_stats.csv
rm -rf $O/prof_transient
cd $R
tail -3 $O/gputest3.log; cat $O/transient.txt
