#!/bin/bash
# A/B of the table-row loads of the G1 bucket kernel: plain vs non-temporal (KEAKI_ACC_NT, read at ctx creation). Time from the bench line,
# fabric reads from FETCH_SIZE / TCC counters. Usage: bench_tools/ab_acc_nt.sh <tag>
export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out/$1; mkdir -p $O
B="python3 $R/bench.py --log2n 24 --steps 5 --warmup 2 --no-cpu-baseline --no-extras --kem-log2n 0"
B1="python3 $R/bench.py --log2n 24 --steps 1 --warmup 0 --no-cpu-baseline --no-extras --kem-log2n 0"
for nt in 0 1; do
  export KEAKI_ACC_NT=$nt
  $B > $O/bench_nt$nt.json 2> $O/bench_nt$nt.err
  (cd /tmp; rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/f$nt -o f -- $B1 > $O/f$nt.log 2>&1)
  (cd /tmp; rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum --kernel-trace --output-format csv -d $O/t$nt -o t -- $B1 > $O/t$nt.log 2>&1)
  (cd /tmp; rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum --kernel-trace --output-format csv -d $O/p$nt -o p -- $B1 > $O/p$nt.log 2>&1)
done
python3 - $O <<'PY'
import csv, glob, json, sys
O = sys.argv[1]
for nt in (0, 1):
    j = json.loads(open("%s/bench_nt%d.json" % (O, nt)).read().strip().split("\n")[-1])
    out = {"nt": nt, "ms_per_step": j["ms_per_step"], "kernel_ms": j["roofline"]["kernel_ms"], "kernel_ms_min": j["roofline"]["kernel_ms_min"]}
    for d in ("f", "t", "p"):
        for f in glob.glob("%s/%s%d/**/*counter_collection.csv" % (O, d, nt), recursive=True):
            for r in csv.DictReader(open(f)):
                if "accumulate" in r["Kernel_Name"]:
                    out[r["Counter_Name"]] = float(r["Counter_Value"])        # last launch wins = the 2^24 step
    print(json.dumps(out))
PY
find $O -name '*.csv' -size +1M -delete; find $O -name '*.db' -delete
