#!/bin/bash
# kernel statistics of the tree's bench under a few tuning environments, one box. Usage: bench_tools/r4_new.sh <tag> "ENV=.. ENV=.." ...
export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out/$1; mkdir -p $O; shift
cd /tmp
i=0
for envs in "$@"; do
  ( export $envs; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/s$i -o b -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras --kem-log2n 0 > $O/s$i.log 2>&1; echo "s$i [$envs] rc=$?" >> $O/rc.txt )
  i=$((i+1))
done
cd $R
find $O -name '*kernel_trace.csv' -delete; find $O -name '*.db' -delete
python3 - $O <<'PY'
import csv, sys, glob, os
O = sys.argv[1]
for d in sorted(glob.glob(O + "/s[0-9]*")):
    if not os.path.isdir(d): continue
    f = glob.glob(d + "/**/*kernel_stats.csv", recursive=True)
    if not f: continue
    rows = list(csv.DictReader(open(f[0])))
    tot = 0.0; acc = 0.0
    print("==", os.path.basename(d))
    for r in rows:
        nm = r["Name"].split("(")[0][-44:]
        if int(r["Calls"]) >= 12:
            per = float(r["AverageNs"]) / 1e3 * int(r["Calls"]) / 12
            if per > 20: print("  %-46s %4s %10.1f us" % (nm, r["Calls"], per))
            if "accumulate" in nm: acc += per
            else: tot += per
    print("  accumulate %.1f  non-accumulate %.1f  sum %.1f us" % (acc, tot, acc + tot))
PY
cat $O/rc.txt
