#!/bin/bash
# Where the bucket kernel's cycles go: SQ busy / wait / instruction-fetch counters of one 2^24 step. Usage: bench_tools/pmc_accumulate_stalls.sh <tag>
export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out/$1; mkdir -p $O
BENCH1="python3 $R/bench.py --log2n 24 --steps 1 --warmup 0 --no-cpu-baseline --no-extras --kem-log2n 0"
cd /tmp
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_BUSY_CYCLES SQ_INSTS_BRANCH SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $O/p1 -o a -- $BENCH1 > $O/p1.log 2>&1; echo "p1 rc=$?" >> $O/rc.txt
rocprofv3 --pmc SQ_IFETCH SQ_IFETCH_LEVEL SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_THREAD_CYCLES_VALU SQ_LEVEL_WAVES --kernel-trace --output-format csv -d $O/p2 -o a -- $BENCH1 > $O/p2.log 2>&1; echo "p2 rc=$?" >> $O/rc.txt
cd $R
python3 - $O <<'PY'
import csv, glob, sys
O = sys.argv[1]
for p in ("p1", "p2"):
    d = {}
    for f in glob.glob("%s/%s/**/*counter_collection.csv" % (O, p), recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_msm_accumulate_g1_u29" in r["Kernel_Name"]:
                d.setdefault(int(r["Dispatch_Id"]), {})[r["Counter_Name"]] = float(r["Counter_Value"])
    if d:
        v = d[max(d)]
        for k in sorted(v): print("%-26s %.6g" % (k, v[k]))
PY
find $O -name '*.csv' -size +1M -delete; find $O -name '*.db' -delete; cat $O/rc.txt
