#!/bin/bash
# Round-4 measurement batch (one MI355X). Usage (repo root): bench_tools/measure_r04.sh <tag>
#   default bench line; kernel statistics of the same command under rocprofv3; PMC FETCH_SIZE / WRITE_SIZE of every MSM kernel (separate passes);
#   SQ counters of the two sort passes; small-call latencies.
export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out/$1; mkdir -p $O
BENCH1="python3 $R/bench.py --log2n 24 --steps 1 --warmup 0 --no-cpu-baseline --no-extras --kem-log2n 0"
python3 $R/bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?" >> $O/rc.txt
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o b -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras > $O/bench_under_rocprofv3.json 2> $O/stats.err; echo "stats rc=$?" >> $O/rc.txt
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_f -o f -- $BENCH1 > $O/pmc_f.log 2>&1; echo "pmc_f rc=$?" >> $O/rc.txt
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_w -o w -- $BENCH1 > $O/pmc_w.log 2>&1; echo "pmc_w rc=$?" >> $O/rc.txt
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $O/pmc_sq -o q -- $BENCH1 > $O/pmc_sq.log 2>&1; echo "pmc_sq rc=$?" >> $O/rc.txt
cd $R
python3 bench_tools/collect_pmc_traffic.py $O/pmc_f $O/pmc_w $O/r04_msm_2p24_hbm_traffic_pmc.json > $O/collect.log 2>&1
python3 - $O <<'PY'
import csv, sys, glob, collections, json
O = sys.argv[1]
agg = collections.OrderedDict()
for f in glob.glob(O + "/pmc_sq/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if any(k in r["Kernel_Name"] for k in ("tile_sort", "chunk_sort", "cell_prefix", "accumulate", "k_msm_reduce")):
            agg.setdefault((r["Kernel_Name"].split("(")[0][-40:], r["Counter_Name"]), []).append(float(r["Counter_Value"]))
with open(O + "/r04_sort_sq_counters.txt", "w") as out:
    for (k, c), v in agg.items():
        out.write("%-42s %-20s launches=%d mean=%.5g\n" % (k, c, len(v), sum(v) / len(v)))
PY
python3 bench_tools/bench_small_calls.py > $O/r04_small_calls.txt 2>&1
find $O -name '*kernel_trace.csv' -size +2M -delete; find $O -name '*counter_collection.csv' -size +1M -delete; find $O -name '*.db' -delete
cat $O/rc.txt; ls $O
