export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out/r2g; mkdir -p $O; cd /tmp
python3 $R/bench_kem.py > $O/kem16.json 2>&1; tail -c 400 $O/kem16.json | head -c 400; echo
rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --kernel-trace --output-format csv -d $O/pmc -o p -- python3 $R/bench_tools/profile_pairing_split.py > $O/pmc.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/split -o s -- python3 $R/bench_tools/profile_pairing_split.py > $O/split.log 2>&1
python3 - <<PY
import csv, collections
for r in list(csv.DictReader(open("$O/split/s_kernel_stats.csv")))[:4]: print(r["Name"][:50], r["Calls"], r["AverageNs"])
agg=collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open("$O/pmc/p_counter_collection.csv")):
    agg[r["Kernel_Name"][:40]][r["Counter_Name"]]+=float(r["Counter_Value"])
for k,v in agg.items():
    if "pairing" in k: print(k, {c: round(x) for c,x in v.items()}, "VALU/wave", round(v["SQ_INSTS_VALU"]/max(v["SQ_WAVES"],1)), "VMEM/wave", round((v["SQ_INSTS_VMEM_RD"]+v["SQ_INSTS_VMEM_WR"])/max(v["SQ_WAVES"],1)))
import csv
tr=list(csv.DictReader(open("$O/split/s_kernel_trace.csv")))
for r in tr:
    if "k_pairing" in r["Kernel_Name"]: print("k_pairing", (int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e6, "ms", "scratch", r.get("Scratch_Size"), "vgpr", r.get("VGPR_Count"))
PY
