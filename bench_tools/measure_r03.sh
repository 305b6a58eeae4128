#!/bin/bash
# Round-3 measurement batch (one MI355X). Usage (repo root): bench_tools/measure_r03.sh <tag> [quick]
#   kernel statistics of the default bench line, the PMC traffic passes of the bucket kernel (FETCH_SIZE / WRITE_SIZE, separate passes),
#   the L2-side counters that account for its over-fetch (TCC hits / misses / fabric read requests by size, TCP->TCC requests),
#   FK23 kernel statistics at d = 2^21, the pairing kernel's SQ counters.
export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out/$1; mkdir -p $O
BENCH1="python3 $R/bench.py --log2n 24 --steps 1 --warmup 0 --no-cpu-baseline --no-extras --kem-log2n 0"
cd /tmp
rocprofv3 -L > $O/counters_list.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o b -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras > $O/stats.log 2>&1; echo "stats rc=$?" >> $O/rc.txt
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_f -o f -- $BENCH1 > $O/pmc_f.log 2>&1; echo "pmc_f rc=$?" >> $O/rc.txt
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_w -o w -- $BENCH1 > $O/pmc_w.log 2>&1; echo "pmc_w rc=$?" >> $O/rc.txt
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum --kernel-trace --output-format csv -d $O/pmc_t1 -o t -- $BENCH1 > $O/pmc_t1.log 2>&1; echo "pmc_t1 rc=$?" >> $O/rc.txt
rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_BUBBLE_sum --kernel-trace --output-format csv -d $O/pmc_t2 -o t -- $BENCH1 > $O/pmc_t2.log 2>&1; echo "pmc_t2 rc=$?" >> $O/rc.txt
rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_NC_READ_REQ_sum --kernel-trace --output-format csv -d $O/pmc_t3 -o t -- $BENCH1 > $O/pmc_t3.log 2>&1; echo "pmc_t3 rc=$?" >> $O/rc.txt
rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $O/pmc_sq -o q -- $BENCH1 > $O/pmc_sq.log 2>&1; echo "pmc_sq rc=$?" >> $O/rc.txt
if [ "$2" != "quick" ]; then
rocprofv3 --kernel-trace --stats --output-format csv -d $O/fk -o fk -- python3 $R/bench_tools/bench_fk.py 21 > $O/fk.log 2>&1; echo "fk rc=$?" >> $O/rc.txt
rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS --kernel-trace --output-format csv -d $O/pmc -o p -- python3 $R/bench_tools/profile_pairing_split.py > $O/pmc.log 2>&1; echo "pmc_pair rc=$?" >> $O/rc.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $O/split -o s -- python3 $R/bench_tools/profile_pairing_split.py > $O/split.log 2>&1; echo "split rc=$?" >> $O/rc.txt
fi
cd $R
python3 bench_tools/collect_pmc_traffic.py $O/pmc_f $O/pmc_w $O/r03_msm_2p24_hbm_traffic_pmc.json > $O/collect.log 2>&1
# keep the outputs small enough to travel back: the traces are large, the per-kernel rows of the counter files are what matters
for d in pmc_t1 pmc_t2 pmc_t3 pmc_sq pmc_f pmc_w pmc; do
  for f in $(find $O/$d -name '*counter_collection.csv' 2>/dev/null); do
    python3 - "$f" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if any(k in r["Kernel_Name"] for k in ("accumulate", "part_", "pairing", "k_miller", "k_final"))]
if rows:
    w = csv.DictWriter(open(sys.argv[1] + ".sel.csv", "w"), fieldnames=list(rows[0].keys())); w.writeheader(); w.writerows(rows)
PY
  done
done
find $O -name '*kernel_trace.csv' -size +2M -delete; find $O -name '*counter_collection.csv' -size +2M -delete; find $O -name '*.db' -delete
cat $O/rc.txt; ls $O; du -sh $O
