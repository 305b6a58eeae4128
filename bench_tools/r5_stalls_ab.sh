#!/bin/bash
# SQ busy / wait counters of the bucket kernel, library of HEAD (ab_old/) against the tree, one box. usage: bench_tools/r5_stalls_ab.sh <tag>
export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out/$1; mkdir -p $O
cd /tmp
for side in old new; do
  D=$R; [ $side = old ] && D=$R/ab_old
  BENCH1="python3 $D/bench.py --log2n 24 --steps 1 --warmup 0 --no-cpu-baseline --no-extras --kem-log2n 0 --g2-log2n 0"
  rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_BUSY_CYCLES SQ_INSTS_BRANCH SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $O/${side}_p1 -o a -- $BENCH1 > $O/${side}_p1.log 2>&1; echo "$side p1 rc=$?" >> $O/rc.txt
  rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM_RD SQ_THREAD_CYCLES_VALU SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_IFETCH --kernel-trace --output-format csv -d $O/${side}_p2 -o a -- $BENCH1 > $O/${side}_p2.log 2>&1; echo "$side p2 rc=$?" >> $O/rc.txt
done
cd $R
python3 - $O <<'PY'
import csv, glob, sys
O = sys.argv[1]
for side in ("old", "new"):
  print("==", side)
  for p in ("p1", "p2"):
    d = {}; dur = {}
    for f in glob.glob("%s/%s_%s/**/*counter_collection.csv" % (O, side, p), recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_msm_accumulate_g1_u29" in r["Kernel_Name"]:
                d.setdefault(int(r["Dispatch_Id"]), {})[r["Counter_Name"]] = float(r["Counter_Value"])
    for f in glob.glob("%s/%s_%s/**/*kernel_trace.csv" % (O, side, p), recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_msm_accumulate_g1_u29" in r["Kernel_Name"]:
                dur[int(r["Dispatch_Id"])] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    if d:
        v = d[max(d)]
        print("  duration ms", dur.get(max(d)))
        for k in sorted(v): print("  %-26s %.6g" % (k, v[k]))
PY
find $O -name '*.csv' -size +1M -delete; find $O -name '*.db' -delete; cat $O/rc.txt
