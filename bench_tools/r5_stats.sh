#!/bin/bash
# Round-5 quick batch (one MI355X): MSM parity tests, kernel statistics of the headline step under rocprofv3, host-pointer sweep.
# usage (repo root): bench_tools/r5_stats.sh <tag>
export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out/$1; mkdir -p $O
timeout 900 python3 -m pytest tests/test_gpu_msm_pipe.py tests/test_gpu_parity.py -x -q -m gpu -k "msm or chunked or skewed or heavy or headline or config2 or same_and" > $O/pytest_msm.txt 2>&1; echo "pytest rc=$?" >> $O/rc.txt
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o b -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras --kem-log2n 0 > $O/bench_under_rocprofv3.json 2> $O/stats.err; echo "stats rc=$?" >> $O/rc.txt
cd $R
find $O -name '*kernel_trace.csv' -size +2M -delete; find $O -name '*.db' -delete
head -30 $O/stats/*/*kernel_stats.csv 2>/dev/null || find $O/stats -name '*stats*' | head
tail -3 $O/pytest_msm.txt; cat $O/rc.txt
