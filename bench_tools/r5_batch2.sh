#!/bin/bash
# usage (repo root): bench_tools/r5_batch2.sh <tag>
export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out/$1; mkdir -p $O
timeout 300 bench_tools/ubench_tower_forms > $O/r05_ubench_tower_forms.txt 2>&1; echo "ubench rc=$?" >> $O/rc.txt
for lg in 20 21 22 23; do timeout 300 python3 bench_tools/sweep_msm_pipe.py $lg 1 > $O/sweep$lg.txt 2>&1; done
cd /tmp
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O/trace_pipe -o t -- python3 $R/bench_tools/trace_msm_pipe.py 24 0 > $O/trace_pipe.log 2>&1; echo "trace rc=$?" >> $O/rc.txt
cd $R
python3 bench_tools/trace_vec_timeline.py $O/trace_pipe > $O/r05_msm_pipe_timeline.txt 2>&1
find $O -name '*.csv' -size +2M -delete; find $O -name '*.db' -delete
cat $O/rc.txt; cat $O/r05_ubench_tower_forms.txt; tail -30 $O/r05_msm_pipe_timeline.txt
