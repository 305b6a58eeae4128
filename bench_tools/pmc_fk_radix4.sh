export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out/r3r5; mkdir -p $O
cd /tmp
for v in 1 0; do
  export KEAKI_FK_RADIX4=$v
  rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU --kernel-trace --output-format csv -d $O/p$v -o p -- python3 $R/bench_tools/fk_calls.py 21 1 > $O/p$v.log 2>&1
done
cd $R
python3 - <<'PY'
import csv, glob, collections
for v in (1, 0):
    f = glob.glob("gpurun_out/r3r5/p%d/**/*counter_collection.csv" % v, recursive=True)[0]
    per = collections.defaultdict(lambda: collections.defaultdict(float)); names = {}
    for r in csv.DictReader(open(f)):
        if "stage" in r["Kernel_Name"]:
            per[r["Dispatch_Id"]][r["Counter_Name"]] += float(r["Counter_Value"]); names[r["Dispatch_Id"]] = r["Kernel_Name"].split("(")[0][-44:]
    print("radix4 =", v)
    for d in sorted(per, key=int):
        a = per[d]
        print("  %5s %-44s waves %7d valu/wave %8d vmem_rd %5d vmem_wr %5d lds %5d salu %6d" % (d, names[d], a["SQ_WAVES"], a["SQ_INSTS_VALU"] / a["SQ_WAVES"], a["SQ_INSTS_VMEM_RD"] / a["SQ_WAVES"], a["SQ_INSTS_VMEM_WR"] / a["SQ_WAVES"], a["SQ_INSTS_LDS"] / a["SQ_WAVES"], a["SQ_INSTS_SALU"] / a["SQ_WAVES"]))
PY
find $O -name '*.csv' -size +1M -delete
