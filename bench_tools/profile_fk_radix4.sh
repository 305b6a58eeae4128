export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out/r3r4; mkdir -p $O
cd /tmp
for v in 1 0; do
  export KEAKI_FK_RADIX4=$v
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/s$v -o s -- python3 $R/bench_tools/fk_calls.py 21 3 > $O/s$v.log 2>&1
done
cd $R
python3 - <<'PY'
import csv, glob
for v in (1, 0):
    f = glob.glob("gpurun_out/r3r4/s%d/**/*kernel_stats.csv" % v, recursive=True)[0]
    print("radix4 =", v)
    for r in list(csv.DictReader(open(f)))[:7]:
        print("  %-62s calls %4s avg %9.1f us tot %8.1f ms" % (r["Name"][:62], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
find $O -name '*kernel_trace.csv' -delete
