#!/bin/bash
# A/B on ONE box: the library of HEAD (ab_old/: `git archive HEAD ... | tar -x -C ab_old`, built there) against the tree -- the MSM parity tests first,
# then kernel statistics of the same bench command, old / new alternating. usage (repo root): bench_tools/r5_ab.sh <tag>
export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out/$1; mkdir -p $O
timeout 1500 python3 -m pytest tests/test_gpu_msm_pipe.py tests/test_gpu_baseline_sizes.py tests/test_gpu_parity.py -x -q -m gpu -k "msm or commit or open or config" > $O/tests.log 2>&1; echo "tests rc=$?" > $O/rc.txt
tail -3 $O/tests.log
cd /tmp
for side in old new old new; do
  D=$R; [ $side = old ] && D=$R/ab_old
  n=$(ls $O | grep -c "^${side}_stats")
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${side}_stats$n -o b -- python3 $D/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --kem-log2n 0 --g2-log2n 0 > $O/${side}_$n.log 2>&1; echo "$side$n rc=$?" >> $O/rc.txt
done
cd $R
find $O -name '*kernel_trace.csv' -delete; find $O -name '*.db' -delete
python3 - $O <<'PY'
import csv, sys, glob, os, json
O = sys.argv[1]
for d in sorted(glob.glob(O + "/*_stats*")):
    f = glob.glob(d + "/**/*kernel_stats.csv", recursive=True)
    if not f: continue
    rows = list(csv.DictReader(open(f[0])))
    print("==", os.path.basename(d))
    for r in rows:
        nm = r["Name"].split("(")[0][-44:]
        if "accumulate" in nm or "chunk_sort" in nm or "tile_sort" in nm or "msm_reduce" in nm:
            print("  %-46s %4s %10.1f us" % (nm, r["Calls"], float(r["AverageNs"]) / 1e3))
for f in sorted(glob.glob(O + "/*_[0-9].log")):
    for l in open(f):
        if l.startswith("{"):
            j = json.loads(l); print(os.path.basename(f), "ms_per_step", j["ms_per_step"], "value", j["value"], "check", j.get("checks"))
PY
cat $O/rc.txt
