#!/bin/bash
# Round-6 measurement batch (one MI355X). Usage (repo root): bench_tools/measure_r06.sh <tag>
#   the default bench line; kernel statistics of the headline step under rocprofv3 (--kernel-trace --stats only); PMC FETCH_SIZE / WRITE_SIZE / SQ
#   counters of every MSM kernel (separate --pmc passes with --kernel-trace only, the program directly after `--`). The pairing / FK23 kernels did
#   not change this round: their r05 counter files carry the hash of the loaded library and stay valid.
export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out/$1; mkdir -p $O
python3 $R/bench.py > $O/r06_bench_default.json 2> $O/bench_default.err; echo "bench rc=$?" >> $O/rc.txt
BENCH1="python3 $R/bench.py --log2n 24 --steps 1 --warmup 0 --no-cpu-baseline --no-extras --kem-log2n 0"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o b -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras --kem-log2n 0 > $O/r06_bench_under_rocprofv3.json 2> $O/stats.err; echo "stats rc=$?" >> $O/rc.txt
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_f -o f -- $BENCH1 > $O/pmc_f.log 2>&1; echo "pmc_f rc=$?" >> $O/rc.txt
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_w -o w -- $BENCH1 > $O/pmc_w.log 2>&1; echo "pmc_w rc=$?" >> $O/rc.txt
rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS --kernel-trace --output-format csv -d $O/pmc_sq -o q -- $BENCH1 > $O/pmc_sq.log 2>&1; echo "pmc_sq rc=$?" >> $O/rc.txt
cd $R
python3 - $O <<'PY'
import csv, glob, json, os, sys, ctypes
O = sys.argv[1]
sys.path.insert(0, os.getcwd())
from bench_tools.srchash import built_hash
lib = ctypes.CDLL(os.path.join(os.getcwd(), "keaki_amd", "libkeaki_hip.so")); lib.keaki_hip_version.restype = ctypes.c_char_p
last = {}
for f in glob.glob(O + "/pmc_sq/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        last.setdefault(k, {}).setdefault(int(r["Dispatch_Id"]), {})[r["Counter_Name"]] = float(r["Counter_Value"])
out = {"source": "rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS --kernel-trace of `bench.py --log2n 24 --steps 1 --warmup 0 --no-cpu-baseline "
                 "--no-extras --kem-log2n 0`: the LAST launch of every MSM kernel (the timed step); MI355X", "library": lib.keaki_hip_version().decode(),
       "kernel_source_sha256": built_hash(lib, "msm"), "log2n": 24, "precompute": True, "kernels": {}}
for k, d in last.items():
    if any(x in k for x in ("k_msm", "k_tile_sort", "k_chunk_sort", "k_cnt", "k_cell", "k_bin")):
        v = d[max(d)]
        out["kernels"][k[-60:]] = {c.lower(): v[c] for c in v}
        if "k_msm_accumulate_g1_u29" in k:
            out["accumulate_valu"] = v.get("SQ_INSTS_VALU"); out["accumulate_waves"] = v.get("SQ_WAVES")
json.dump(out, open(O + "/r06_msm_sq_insts.json", "w"), indent=1)
print("bucket kernel: %.4g VALU wave-instructions, %.0f waves -> %.1f per wave-addition" % (out.get("accumulate_valu", 0), out.get("accumulate_waves", 0), out.get("accumulate_valu", 0) / (2**24 * 12 / 64)))
PY
python3 bench_tools/collect_pmc_traffic.py $O/pmc_f $O/pmc_w $O/r06_msm_2p24_hbm_traffic_pmc.json > $O/collect.log 2>&1
cp $O/stats/b_kernel_stats.csv $O/r06_bench_default_kernel_stats.csv 2>/dev/null
find $O -name '*kernel_trace.csv' -size +2M -delete; find $O -name '*counter_collection.csv' -size +1M -delete; find $O -name '*.db' -delete
cat $O/rc.txt; tail -12 $O/collect.log
