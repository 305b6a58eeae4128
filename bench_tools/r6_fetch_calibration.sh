#!/bin/bash
# FETCH_SIZE (and the request counters behind it) on known byte counts in this tree's access shapes: bench_tools/ubench_fetch_calib.hip under rocprofv3,
# one counter set per pass, the program directly after `--`. Output: gpurun_out/r06/fetch_calib/<pass>/..., folded by bench_tools/fold_fetch_calibration.py.
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
OUT=gpurun_out/r06/fetch_calib
mkdir -p $OUT
./bench_tools/ubench_fetch_calib > $OUT/plain_run.txt 2>&1
for pass in "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCP_TCC_READ_REQ_sum TCC_REQ_sum" "TCC_HIT_sum TCC_MISS_sum" "TCC_BUBBLE_sum TCC_EA0_RDREQ_DRAM_sum"; do
  tag=$(echo $pass | tr ' ' '+')
  timeout 300 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $OUT/$tag -- ./bench_tools/ubench_fetch_calib > $OUT/$tag.log 2>&1
  echo "pass $tag rc=$?"
done
python3 bench_tools/fold_fetch_calibration.py $OUT
