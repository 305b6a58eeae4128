#!/bin/bash
# The measurement batch behind profiles/ (one MI355X): GPU test suite, the default bench line with the kernel statistics of the same
# command, the PMC traffic passes of the bucket kernel, the KEM benches, the pairing kernel's instruction counters, single-call
# latencies, G2 MSM, FK23 (un-sharded and per rank), Laconic OT at 2^20. Usage (from the repo root): bench_tools/measure_all.sh <tag>
export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out/$1; mkdir -p $O
timeout 1700 python3 -m pytest tests -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?" > $O/rc.txt; tail -3 $O/pytest.log
python3 bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" >> $O/rc.txt
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o b -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/stats.log 2>&1; echo "stats rc=$?" >> $O/rc.txt
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_f -o f -- python3 $R/bench.py --log2n 24 --steps 1 --warmup 0 --no-cpu-baseline --no-extras --kem-log2n 0 > $O/pmc_f.log 2>&1; echo "pmc_f rc=$?" >> $O/rc.txt
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_w -o w -- python3 $R/bench.py --log2n 24 --steps 1 --warmup 0 --no-cpu-baseline --no-extras --kem-log2n 0 > $O/pmc_w.log 2>&1; echo "pmc_w rc=$?" >> $O/rc.txt
rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --kernel-trace --output-format csv -d $O/pmc -o p -- python3 $R/bench_tools/profile_pairing_split.py > $O/pmc.log 2>&1; echo "pmc_pair rc=$?" >> $O/rc.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $O/split -o s -- python3 $R/bench_tools/profile_pairing_split.py > $O/split.log 2>&1; echo "split rc=$?" >> $O/rc.txt
cd $R
python3 bench_tools/collect_pmc_traffic.py $O/pmc_f $O/pmc_w $O/r03_msm_2p24_hbm_traffic_pmc.json > $O/collect.log 2>&1
cp $O/r03_msm_2p24_hbm_traffic_pmc.json profiles/ 2>/dev/null
python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_after_restamp.json 2>> $O/bench.err
python3 bench_kem.py > $O/kem16.json 2>&1
python3 bench_kem.py --log2n 20 > $O/kem20.json 2>&1
python3 bench_tools/bench_small_calls.py > $O/small_calls.txt 2>&1
python3 bench_tools/bench_msm_g2.py 16 20 > $O/g2.txt 2>&1
python3 bench_tools/bench_fk.py 16 20 21 > $O/fk.txt 2>&1
python3 bench_tools/bench_fk_shard.py 16 20 21 > $O/fk_shard.txt 2>&1
python3 laconic_ot.py --log2n 20 > $O/laconic20.json 2>&1
cat $O/rc.txt
tail -c 300 $O/bench.json; echo
tail -1 $O/laconic20.json | cut -c1-400
