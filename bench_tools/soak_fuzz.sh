mkdir -p gpurun_out/r3soak2
fail=0
for s in 101 102 103 104 105 106; do
  timeout 300 python3 tests/fuzz/fuzz_msm.py 60 $s > gpurun_out/r3soak2/msm_$s.log 2>&1 || { echo "msm seed $s FAILED"; fail=1; }
  timeout 300 python3 tests/fuzz/fuzz_kem.py 40 $s > gpurun_out/r3soak2/kem_$s.log 2>&1 || { echo "kem seed $s FAILED"; fail=1; }
  timeout 300 python3 tests/fuzz/fuzz_mixed.py 40 $s > gpurun_out/r3soak2/mixed_$s.log 2>&1 || { echo "mixed seed $s FAILED"; fail=1; }
  timeout 300 python3 tests/fuzz/fuzz_group.py 40 $s $(( (s % 4) + 2 )) > gpurun_out/r3soak2/group_$s.log 2>&1 || { echo "group seed $s FAILED"; fail=1; }
done
tail -n 1 gpurun_out/r3soak2/*.log | grep -v "^$" | tail -60
echo "soak fail=$fail"
