export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out/r4a; mkdir -p $O
timeout 900 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "msm" > $O/pytest_msm.log 2>&1; echo "pytest_msm rc=$?" >> $O/rc.txt
tail -5 $O/pytest_msm.log
timeout 600 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras > $O/bench.log 2>&1; echo "bench rc=$?" >> $O/rc.txt
tail -c 3000 $O/bench.log
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o b -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras --kem-log2n 0 > $O/stats.log 2>&1; echo "stats rc=$?" >> $O/rc.txt
cd $R
find $O -name '*kernel_trace.csv' -size +2M -delete; find $O -name '*.db' -delete
head -30 $(find $O/stats -name '*kernel_stats.csv' | head -1) | cut -c1-60,200-400
cat $O/rc.txt
