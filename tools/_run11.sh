mkdir -p gpurun_out/r3
O=$(pwd)/gpurun_out/r3
R=$(pwd)
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_transient -- python3 $R/tools/transient_profile.py 100 > $O/transient2.txt 2> $O/transient2.err || exit 1
cp $O/prof_transient/*/*_kernel_stats.csv $O/transient_kernel_stats2.csv
rm -rf $O/prof_transient
