R=$GRAFT_REPO_ROOT; o=$R/gpurun_out; mkdir -p $o
python $R/bench.py > $o/r05l_headline.json 2> $o/r05l_headline.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $o/prof_r05l_headline -- python3 $R/bench.py --no-cpu-baseline --no-extras > $o/r05l_prof_headline.log 2>&1
for f in $(find $o/prof_r05l_headline -name "*kernel_stats.csv"); do cp $f $o/r05l_headline_kernel_stats.csv; done; rm -rf $o/prof_r05l_headline
head -3 $o/r05l_headline_kernel_stats.csv
