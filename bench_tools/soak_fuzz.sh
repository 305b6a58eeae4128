#!/bin/bash
# longer runs of the four fuzzers over several seeds. usage (repo root): bench_tools/soak_fuzz.sh <tag> [seed ...]
O=gpurun_out/${1:-soak}; mkdir -p $O
shift
seeds=${@:-501 502 503 504 505 506}
fail=0
for s in $seeds; do
  timeout 400 python3 tests/fuzz/fuzz_msm.py 60 $s > $O/msm_$s.log 2>&1 || { echo "msm seed $s FAILED"; fail=1; }
  timeout 300 python3 tests/fuzz/fuzz_kem.py 40 $s > $O/kem_$s.log 2>&1 || { echo "kem seed $s FAILED"; fail=1; }
  timeout 300 python3 tests/fuzz/fuzz_mixed.py 40 $s > $O/mixed_$s.log 2>&1 || { echo "mixed seed $s FAILED"; fail=1; }
  timeout 300 python3 tests/fuzz/fuzz_group.py 40 $s $(( (s % 4) + 2 )) > $O/group_$s.log 2>&1 || { echo "group seed $s FAILED"; fail=1; }
done
tail -q -n 1 $O/*.log | sort | uniq -c
echo "soak fail=$fail"
