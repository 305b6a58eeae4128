#!/bin/bash
# Round 5's final build (ab_old/: `git archive cc18092 | tar -x -C ab_old`, `make` in ab_old/keaki_amd/csrc and host) against the tree on ONE box, alternating:
# the headline step only (20 steps, 5 warm-up, no extras). usage (repo root): bench_tools/r6_ab_round.sh
R=$PWD
for side in old new old new old new; do
  D=$R; [ $side = old ] && D=$R/ab_old
  python3 $D/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extras --kem-log2n 0 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$side  ms_per_step %.3f  bucket kernel %.3f ms (min %.3f)  value %.4e' % (j['ms_per_step'], j['roofline']['kernel_ms'], j['roofline']['kernel_ms_min'], j['value']))"
done
