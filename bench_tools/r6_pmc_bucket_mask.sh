#!/bin/bash
# Fabric (TCC_EA0_RDREQ) and L1->L2 (TCP_TCC_READ_REQ) read requests of the bucket kernel with its table rows anywhere / confined to 2 MiB (L2-resident:
# what is left is the index stream) -- bench_tools/r6_pmc_bucket_mask.py under rocprofv3, the program directly after `--`.
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
OUT=gpurun_out/r06/pmc_bucket_mask
mkdir -p $OUT
for mask in 0 0x7fff; do
  for pass in "TCC_EA0_RDREQ_sum TCP_TCC_READ_REQ_sum" "TCC_HIT_sum TCC_MISS_sum"; do
    tag=mask_${mask}_$(echo $pass | tr ' ' '+')
    timeout 600 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $OUT/$tag -- python3 bench_tools/r6_pmc_bucket_mask.py $mask "$@" > $OUT/$tag.log 2>&1
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
