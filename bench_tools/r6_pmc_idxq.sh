#!/bin/bash
# Fabric / L1->L2 read requests of the shipped bucket kernel with the index stream by quads (acc_idxq = 1, round 6) and one entry per load (0): product library.
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp KEAKI_PRODUCT_LIB=1
OUT=gpurun_out/r06/pmc_idxq
mkdir -p $OUT
for q in 1 0; do
  for pass in "TCC_EA0_RDREQ_sum TCP_TCC_READ_REQ_sum" "TCC_HIT_sum TCC_MISS_sum" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVES"; do
    tag=idxq_${q}_$(echo $pass | tr ' ' '+')
    timeout 600 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $OUT/$tag -- python3 bench_tools/r6_pmc_bucket_mask.py 0 acc_idxq:$q > $OUT/$tag.log 2>&1
    echo "pass $tag rc=$?"
    python3 - "$OUT/$tag" <<'PY'
import csv, glob, sys
best = {}
for path in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(path)):
        if "k_msm_accumulate" in row["Kernel_Name"]:
            best[(row["Counter_Name"], int(row["Dispatch_Id"]))] = float(row["Counter_Value"])
last = max(d for _, d in best) if best else None
rows = 12 * (1 << 24)
for (c, d), v in sorted(best.items()):
    if d == last:
        print("   k_msm_accumulate_g1_u29 %-24s %.4e  = %.3f per table row" % (c, v, v / rows))
PY
  done
done
