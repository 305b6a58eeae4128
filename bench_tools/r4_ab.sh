#!/bin/bash
# A/B on ONE box: the round-3 library (ab_old/, built from f82ce1b) against the tree, kernel statistics of the same bench command.
export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out/$1; mkdir -p $O
cd /tmp
for side in old new old new; do
  D=$R; [ $side = old ] && D=$R/ab_old
  n=$(ls $O | grep -c "^${side}_stats")
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${side}_stats$n -o b -- python3 $D/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras --kem-log2n 0 > $O/${side}_$n.log 2>&1; echo "$side$n rc=$?" >> $O/rc.txt
done
cd $R
find $O -name '*kernel_trace.csv' -delete; find $O -name '*.db' -delete
python3 - $O <<'PY'
import csv, sys, glob, os
O = sys.argv[1]
for d in sorted(glob.glob(O + "/*_stats*")):
    f = glob.glob(d + "/**/*kernel_stats.csv", recursive=True)
    if not f: continue
    rows = list(csv.DictReader(open(f[0])))
    tot = 0.0
    print("==", os.path.basename(d))
    for r in rows:
        nm = r["Name"].split("(")[0][-44:]
        if int(r["Calls"]) >= 12 or "build_tables" in nm or "mul_batch" in nm:
            per = float(r["AverageNs"]) / 1e3
            print("  %-46s %4s %10.1f us" % (nm, r["Calls"], per))
            if int(r["Calls"]) >= 12 and "accumulate" not in nm: tot += per * int(r["Calls"]) / 12
    print("  non-accumulate per MSM: %.1f us" % tot)
PY
cat $O/rc.txt
