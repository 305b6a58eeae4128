#!/bin/bash
# SQ counters of the bucket-sort kernels (two passes of <= 8 counters), one 2^24 MSM per pass. Usage: bench_tools/r4_pmc_sort.sh <tag>
export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out/$1; mkdir -p $O
BENCH1="python3 $R/bench.py --log2n 24 --steps 1 --warmup 0 --no-cpu-baseline --no-extras --kem-log2n 0"
cd /tmp
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $O/pmc1 -o q -- $BENCH1 > $O/pmc1.log 2>&1; echo "pmc1 rc=$?" >> $O/rc.txt
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_BRANCH SQ_ACTIVE_INST_SCA --kernel-trace --output-format csv -d $O/pmc2 -o q -- $BENCH1 > $O/pmc2.log 2>&1; echo "pmc2 rc=$?" >> $O/rc.txt
cd $R
for d in pmc1 pmc2; do
  for f in $(find $O/$d -name '*counter_collection.csv' 2>/dev/null); do
    python3 - "$f" <<'PY'
import csv, sys, collections
rows = [r for r in csv.DictReader(open(sys.argv[1])) if any(k in r["Kernel_Name"] for k in ("tile_sort", "chunk_sort", "cell_prefix", "accumulate", "k_msm_reduce"))]
agg = collections.OrderedDict()
for r in rows:
    key = (r["Kernel_Name"].split("(")[0][-40:], r["Counter_Name"])
    agg.setdefault(key, []).append(float(r["Counter_Value"]))
for (k, c), v in agg.items():
    print("%-42s %-24s n=%d mean=%.4g" % (k, c, len(v), sum(v) / len(v)))
PY
  done
done | tee $O/summary.txt
find $O -name '*kernel_trace.csv' -size +2M -delete; find $O -name '*counter_collection.csv' -size +1M -delete; find $O -name '*.db' -delete
cat $O/rc.txt
