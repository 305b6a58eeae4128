#!/bin/bash
# usage (repo root): bench_tools/r5_batch3.sh <tag>
export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out/$1; mkdir -p $O
timeout 900 python3 -m pytest tests/test_gpu_msm_pipe.py tests/test_gpu_parity.py tests/test_gpu_keaki_api.py tests/test_gpu_group.py -x -q -m gpu > $O/pytest.txt 2>&1; echo "pytest rc=$?" >> $O/rc.txt
for lg in 20 22 24; do timeout 300 python3 bench_tools/bench_open.py --log2n $lg > $O/open$lg.txt 2>&1; done
timeout 300 bench_tools/ubench_tower_forms > $O/r05_ubench_tower_forms.txt 2>&1
tail -4 $O/pytest.txt; cat $O/rc.txt; cat $O/open*.txt; grep -E "^cut" $O/r05_ubench_tower_forms.txt
