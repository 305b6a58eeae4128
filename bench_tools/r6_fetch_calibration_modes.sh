cd /root/repo; export TMPDIR=/tmp
for mode in 1 2; do
  O=gpurun_out/r06/fetch_calib_mode$mode; mkdir -p $O
  ./bench_tools/ubench_fetch_calib 26 $mode > $O/plain_run.txt 2>&1
  for pass in "FETCH_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCP_TCC_READ_REQ_sum TCC_REQ_sum"; do
    tag=$(echo $pass | tr ' ' '+')
    timeout 300 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $O/$tag -- ./bench_tools/ubench_fetch_calib 26 $mode > $O/$tag.log 2>&1
  done
  echo "== mode $mode"; python3 bench_tools/fold_fetch_calibration.py $O | cut -c1-230
done
