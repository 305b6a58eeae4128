#!/bin/bash
# Round-5 measurement batch (one MI355X). Usage (repo root): bench_tools/measure_r05.sh <tag> [full]
#   the -m gpu suite; the default bench line; kernel statistics of the headline step under rocprofv3; PMC FETCH_SIZE / WRITE_SIZE of every MSM
#   kernel (separate passes); SQ instruction counters of the pairing and FK23 kernels; with `full`: FETCH / WRITE of the FK23 + pairing kernels.
export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out/$1; mkdir -p $O
timeout 1500 python3 -m pytest tests -x -q -m gpu > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?" >> $O/rc.txt
python3 $R/bench.py > $O/r05_bench_default.json 2> $O/bench_default.err; echo "bench rc=$?" >> $O/rc.txt
BENCH1="python3 $R/bench.py --log2n 24 --steps 1 --warmup 0 --no-cpu-baseline --no-extras --kem-log2n 0"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o b -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras --kem-log2n 0 > $O/r05_bench_under_rocprofv3.json 2> $O/stats.err; echo "stats rc=$?" >> $O/rc.txt
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_f -o f -- $BENCH1 > $O/pmc_f.log 2>&1; echo "pmc_f rc=$?" >> $O/rc.txt
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_w -o w -- $BENCH1 > $O/pmc_w.log 2>&1; echo "pmc_w rc=$?" >> $O/rc.txt
cd $R
python3 bench_tools/collect_pmc_traffic.py $O/pmc_f $O/pmc_w $O/r05_msm_2p24_hbm_traffic_pmc.json > $O/collect.log 2>&1
cp $O/stats/b_kernel_stats.csv $O/r05_bench_default_kernel_stats.csv 2>/dev/null
bench_tools/collect_pmc_sq_r05.sh $1 >> $O/collect.log 2>&1
if [ "$2" == "full" ]; then bench_tools/collect_pmc_fk_pairing.sh $1 >> $O/collect.log 2>&1; fi
find $O -name '*kernel_trace.csv' -size +2M -delete; find $O -name '*counter_collection.csv' -size +1M -delete; find $O -name '*.db' -delete
tail -3 $O/pytest_gpu.txt; cat $O/rc.txt; tail -5 $O/collect.log
